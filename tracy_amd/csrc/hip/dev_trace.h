// dev_trace.h -- device functions of the per-sample path-tracing hot path (gfx950).
//
// Reference behaviour restated for the GPU (citations into carcass82/tracy):
//   fastrand                    src/random.h:71-77
//   Camera::GetRayFrom          src/camera.h:28-35, Ray ctor src/ray.h:15-19
//   collision::RayTriangle      src/collision.h:33-74   (back-face cull, EPS edge rejection, t ties)
//   ComputeIntersection (tail)  cpu_details.cpp:169-182 (attribute interpolation, normal NOT normalised)
//   Material::Scatter + getters src/material.h:125-268
//   Texture::GetPixel           src/texture.h:50-57
//   sky lookup + RR             cpu_trace.cpp:146-166
// The traversal itself (stack walk over this module's own BVH2 records, bvh_builder.h) has no
// counterpart in the reference: it replaces accel::IntersectsWithTree (src/kdtree.h:382-429).
//
// Arithmetic contract (DESIGN.md "Decreed arithmetic"): this file is compiled with
// -ffp-contract=off; every fused multiply-add is an explicit __builtin_fmaf and appears only in
// dot_f/cross_f (RayTriangle) and in the slab distances of the conservative box tests. Division and
// sqrt are IEEE (hipcc default); the only approximate reciprocal (v_rcp_f32) feeds the box tests.
#pragma once
#include "dev_types.h"

#define DEVFN __device__ __forceinline__
#define TRH_SCALAR_AS __attribute__((address_space(4)))   // constant address space: a load from a wave-uniform address becomes s_load

namespace dev {

typedef uint32_t U32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t U32x16 __attribute__((ext_vector_type(16)));
typedef float F32x4 __attribute__((ext_vector_type(4)));

constexpr float PI = 3.1415926535897932f;
constexpr float EPS = 1.e-8f;

struct V3 { float x, y, z; };
typedef _Float16 H2 __attribute__((ext_vector_type(2)));
DEVFN V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEVFN V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEVFN V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEVFN V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEVFN V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
DEVFN V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
DEVFN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEVFN V3 cross(V3 a, V3 b) { return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
DEVFN V3 normalize(V3 v) { return v * (1.f / __builtin_sqrtf(dot(v, v))); }
DEVFN float lerp(float x, float y, float a) { return x * (1.f - a) + y * a; }
DEVFN V3 lerp(V3 x, V3 y, float a) { return x * (1.f - a) + y * a; }
DEVFN V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.f; }
DEVFN V3 refract(V3 I, V3 N, float eta) {
    const float d = dot(N, I);
    const float k = 1.f - eta * eta * (1.f - d * d);
    if (k >= 0.f) return eta * I - (eta * d + __builtin_sqrtf(k)) * N;
    return mk(0.f, 0.f, 0.f);
}
// fused forms, RayTriangle only
DEVFN float dot_f(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
DEVFN V3 cross_f(V3 a, V3 b) {
    return mk(__builtin_fmaf(a.y, b.z, -(b.y * a.z)), __builtin_fmaf(a.z, b.x, -(b.z * a.x)), __builtin_fmaf(a.x, b.y, -(b.x * a.y)));
}

// ------------------------------------------------------------------ RNG (src/random.h:71-77)
DEVFN float fastrand(uint32_t& ctx) {
    const uint32_t state = ctx;
    ctx = ctx * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    // x / float(UINT32_MAX): float(UINT32_MAX) rounds to 2^32, so this is an exact scaling
    return (float)((word >> 22u) ^ word) * 2.3283064365386963e-10f;
}
DEVFN uint32_t pcg_hash(uint32_t v) {
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// counter-based per-path state; seed_hash = pcg_hash(seed32) is computed once on the host
DEVFN uint32_t path_rng_state(uint32_t pixel_index, uint32_t sample_index, uint32_t seed_hash) {
    return pcg_hash(pixel_index ^ pcg_hash(sample_index ^ seed_hash));
}

struct Counters {
    uint32_t rays, box_tests, tri_tests, hits, tex_fetches;
    uint32_t w_inner, w_leaf, w_shade, w_gen, l_gen;   // instrumented kernel only
    unsigned long long cyc[5];                         // instrumented kernel only: wave cycles per phase (see DevCounters)
    unsigned long long t_start;                        // instrumented kernel only: wave start on the device-wide clock
    uint32_t idle_leaf, idle_wait, idle_none;          // instrumented kernel only, wave-level (lane 0 counts): see DevCounters
};
// true for exactly one of the currently active lanes (wave-level event counting)
DEVFN bool first_active_lane() {
    const unsigned long long m = __ballot(true);
    return (uint32_t)(__ffsll((long long)m) - 1) == (threadIdx.x & 63u);
}

struct Hit {
    float t, u, v;
    int prim;        // leaf-order primitive index, -1 = miss
    uint32_t id;     // packed reference id (mesh << 24) | triangle
};

// ------------------------------------------------------------------ camera (src/camera.h:28-35)
DEVFN void camera_ray(const DevScene& sc, float s, float t, V3& o, V3& d) {
    // mat4 * vec4(nx, ny, 1, 1) as (m0*x + m1*y) + (m2*z + m3*w): with z = w = 1 the second half is m2 + m3 exactly,
    // the same for every ray. The host adds it once (sc.cam_k, same fp32 add); left to the compiler it is hoisted out
    // of the persistent loop into four VGPRs that the megakernel then spills (profiles/r02_spill_diet.md).
    const float nx = s * 2.f - 1.f, ny = t * 2.f - 1.f;
    float p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a0 = sc.ivp[0 * 4 + i] * nx + sc.ivp[1 * 4 + i] * ny;
        p[i] = a0 + sc.cam_k[i];
    }
    o = mk(sc.eye[0], sc.eye[1], sc.eye[2]);
    // opaque to the optimiser from here on: a wave-uniform origin lets it hoist (eye - frame_c) of trav_begin out of
    // the persistent loop, three more long-lived VGPRs for one v_sub each
    asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z));
    d = normalize(mk(p[0] / p[3], p[1] / p[3], p[2] / p[3]) - o);
}

// ------------------------------------------------------------------ candidate acceptance
// Reference rule (collision.h:66): reject t < EPS || t > RayT, so an equal t REPLACES the stored
// hit: the last-tested triangle wins. The reference tests meshes in mesh-id order (one TLAS leaf
// for <= 16 objects, cpu_details.cpp:128-142) and, before it looks at a later mesh, prunes it with
// RayAABB(mesh AABB, minT) whose second clause is the STRICT tmin < minT (collision.h:130).
// This BVH visits triangles in no particular order, so:
//   * same mesh, exact tie            -> the larger triangle id wins (documented approximation);
//   * different meshes, |dt| <= 1e-5 t -> replay the reference's pairwise decision: the later mesh b
//     beats the earlier mesh a iff t_b <= t_a and slab_tmin(AABB_b) < t_a. This is what decides
//     coincident coplanar faces of two objects (data/scenes/random.scn: light box on a wall).
DEVFN bool accept_candidate(const DevScene& sc, V3 o, V3 d, float t, uint32_t id, const Hit& hit) {
    const uint32_t mc = id >> 24, mh = hit.id >> 24;
    // the common outcome without branches (bitwise operators: short-circuit forms become a chain of exec-mask regions);
    // only a candidate of ANOTHER mesh within 1e-5 t of the hit enters the replay of the reference's pairwise rule below
    const bool plain = (mc == mh) | !(fabsf(t - hit.t) <= 1e-5f * hit.t);
    const bool nearer = !(t < EPS) & ((t < hit.t) | ((t == hit.t) & (id > hit.id)));
    if (plain) return nearer;
    if (t < EPS) return false;
    const bool c_later = mc > mh;
    const float t_a = c_later ? hit.t : t;
    const float t_b = c_later ? t : hit.t;
    const uint32_t mb = c_later ? mc : mh;
    const float4 bmn = sc.mesh_aabb[mb * 2 + 0], bmx = sc.mesh_aabb[mb * 2 + 1];
    // Ray ctor (src/ray.h:18). This branch is rare; the empty asm pins the three IEEE divisions here -- without it the
    // compiler hoists them (d is loop-invariant) to the head of every traversal phase: 33 instructions for all lanes
    // and three live registers through the hot loop.
    float dx = d.x, dy = d.y, dz = d.z;
    asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dz));
    const V3 inv = mk(1.f / dx, 1.f / dy, 1.f / dz);
    const float x0 = (bmn.x - o.x) * inv.x, x1 = (bmx.x - o.x) * inv.x;
    const float y0 = (bmn.y - o.y) * inv.y, y1 = (bmx.y - o.y) * inv.y;
    const float z0 = (bmn.z - o.z) * inv.z, z1 = (bmx.z - o.z) * inv.z;
    const float tx = (x1 < x0) ? x1 : x0, ty = (y1 < y0) ? y1 : y0, tz = (z1 < z0) ? z1 : z0;   // glm::min
    const float tyz = (ty < tz) ? tz : ty;                                                       // glm::max
    const float tmin_b = (tx < tyz) ? tyz : tx;
    const bool b_wins = (t_b <= t_a) && (tmin_b < t_a);
    return c_later ? b_wins : !b_wins;
}

// ------------------------------------------------------------------ nearest hit
// Resumable stack traversal of the BVH2 records (32-byte fp16 form, or the 64-byte fp32 fallback). The state lives in registers (Trav) plus
// this lane's LDS stack column (row i at stack[i * TRH_BLOCK]); a kernel may run trav_inner /
// trav_leaf for a while, do something else (shade other lanes, refill) and come back.
// Row 0 of every column holds the sentinel -1 ("nothing left"), written once by the kernel: popping an empty
// stack needs no special case, and the stack pointer is kept as the LDS address of the top row, so a step
// reads the top at offset 0 and pushes at the immediate offset of one row -- no address arithmetic.
struct Trav {
    V3 inv;      // rcp(direction) (src/ray.h:18) expressed in the BVH frame: 1 / (d * frame_s)
    V3 oi;       // (origin - frame_c) * rcp(direction): slab distance = fma(bound', inv, -oi)
    uint32_t selx, sely, selz;   // v_perm_b32 selectors: swap the (min | max << 16) halves of an axis word when d < 0
    int node;    // >= 0: inner-node record; < -1: leaf link ~((first << 4) | count); -1: nothing to traverse (finished, or no ray)
    int* top;    // this lane's top stack row (row 0 = the sentinel)
    Hit hit;
};
DEVFN bool trav_active(const Trav& tv) { return tv.node != -1; }

DEVFN void trav_begin(const DevScene& sc, Trav& tv, V3 o, V3 d, int* __restrict__ stack) {
    // Node bounds live in the frame x' = (x - c) * s with s a power of two (c = 0, s = 1 for fp32 records), so
    // t = (x - o) / d = x' * (1 / (d s)) - (o - c) * (1 / d): one fma per plane, as before.
    // |1/d| is clamped so that the product with 1/s stays finite (only matters for s < 1, i.e. scenes
    // larger than +-32768 units): a direction change of < 1e-36, far inside the box padding.
    // v_rcp_f32 (1 ulp) instead of the IEEE division of Ray's ctor: these reciprocals only feed the
    // conservative box tests (relative slab error 1e-7 against a 1e-5 padding); Moller-Trumbore uses d.
    const float lim = sc.frame_inv_lim;
    const V3 inv = mk(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -lim, lim), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -lim, lim),
                      __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -lim, lim));
    tv.oi = mk((o.x - sc.frame_c[0]) * inv.x, (o.y - sc.frame_c[1]) * inv.y, (o.z - sc.frame_c[2]) * inv.z);
    tv.inv = inv * sc.frame_sinv;
    // fp16 records keep (min | max << 16) per axis: with the halves ordered by the sign of d the low half is
    // the plane the ray enters through and the high half the one it leaves through -- no min/max per axis
    tv.selx = (__float_as_uint(d.x) >> 31) ? 0x01000302u : 0x03020100u;
    tv.sely = (__float_as_uint(d.y) >> 31) ? 0x01000302u : 0x03020100u;
    tv.selz = (__float_as_uint(d.z) >> 31) ? 0x01000302u : 0x03020100u;
    tv.hit.t = 3.402823466e+38f;  // HitData.t = FLT_MAX (cpu_trace.cpp:119)
    tv.hit.u = tv.hit.v = 0.f;
    tv.hit.prim = -1;
    tv.hit.id = TRH_MISS_ID;
    // NaN direction (normalize(0) after total internal reflection, material.h:242-263): every RayAABB
    // comparison is false in the reference, i.e. a miss
    const bool nan = d.x != d.x;
    tv.top = stack;
    tv.node = nan ? -1 : 0;
}

// The lane index, recomputed (two v_mbcnt) instead of kept: volatile so that the compiler neither hoists it out of a
// persistent loop nor keeps the result alive across one.
DEVFN uint32_t fresh_lane_id() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// A lane's stack column from any of its stack pointers: rows are TRH_BLOCK ints apart and the column is the thread's
// index in the workgroup, so column = (pointer - base) mod TRH_BLOCK (also for the pointer one row below the sentinel).
DEVFN int* stack_column(int* lds_base, const int* top) {
    return lds_base + ((uint32_t)(top - lds_base) & (uint32_t)(TRH_BLOCK - 1));
}

DEVFN void trav_pop(Trav& tv) {
    tv.node = *tv.top;        // the sentinel -1 when nothing is left
    tv.top -= TRH_BLOCK;
}

// one inner-node record: one fetch, two slab tests. Precondition: tv.node >= 0.
template <bool STATS, bool N16>
DEVFN void trav_inner(const DevScene& sc, V3 o, Trav& tv, int* __restrict__ stack, Counters& cnt) {
    // Two record formats, chosen per scene at upload; N16 is a template parameter of every kernel that
    // traverses (a run-time branch here lets the compiler merge the two tails and lose the free conversion):
    //  * 32-byte record, fp16 bounds in the BVH frame: two 16-byte loads; the fp16 -> fp32 conversion is
    //    free (v_fma_mix_f32 reads either half of a register as an fp16 operand);
    //  * 64-byte record, fp32 bounds: four loads. Fallback for scenes whose geometry is so small against
    //    its distance from the frame centre that half floats would inflate the boxes (bvh_builder.h).
    // Slab distances in fused form, one fma per plane: fma(bound, inv, -oi). This rounds differently from
    // the reference's (bound - o) * inv (collision.h:121-122), which only matters for WHICH boxes are
    // visited, never for t/u/v: the builder pads every box by 1e-5 of its size plus 2e-6 of the scene
    // extent, far more than the difference.
    const V3 inv = tv.inv, oi = tv.oi;
    float tmin0, tmax0, tmin1, tmax1;
    int c0, c1;
    // the entry a miss of both children would pop, read together with the record: its LDS latency hides behind the
    // loads, and the step has no divergent pop branch (-2.5 % on the dragon against popping on demand)
    const int top = *tv.top;
    if (N16) {
        // 32-bit byte offset from the (wave-uniform) array base: one shift, and the loads use the SGPR-base form
        const uint4* rec = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.nodes16) + ((uint32_t)tv.node << 5));
        const uint4 qa = rec[0];
        const uint4 qb = rec[1];
        // order the halves of every axis word by the ray's direction sign: low = entry plane, high = exit plane
        const H2 hx0 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(qa.x, qa.x, tv.selx)), hy0 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(qa.y, qa.y, tv.sely));
        const H2 hx1 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(qa.z, qa.z, tv.selx)), hy1 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(qa.w, qa.w, tv.sely));
        const H2 hz0 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(qb.x, qb.x, tv.selz)), hz1 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(qb.y, qb.y, tv.selz));
        const float anx = __builtin_fmaf((float)hx0.x, inv.x, -oi.x), afx = __builtin_fmaf((float)hx0.y, inv.x, -oi.x);
        const float any = __builtin_fmaf((float)hy0.x, inv.y, -oi.y), afy = __builtin_fmaf((float)hy0.y, inv.y, -oi.y);
        const float anz = __builtin_fmaf((float)hz0.x, inv.z, -oi.z), afz = __builtin_fmaf((float)hz0.y, inv.z, -oi.z);
        const float bnx = __builtin_fmaf((float)hx1.x, inv.x, -oi.x), bfx = __builtin_fmaf((float)hx1.y, inv.x, -oi.x);
        const float bny = __builtin_fmaf((float)hy1.x, inv.y, -oi.y), bfy = __builtin_fmaf((float)hy1.y, inv.y, -oi.y);
        const float bnz = __builtin_fmaf((float)hz1.x, inv.z, -oi.z), bfz = __builtin_fmaf((float)hz1.y, inv.z, -oi.z);
        // fmaxf / fminf drop a NaN operand (0 * inf on an axis-parallel ray): that axis is ignored, conservatively
        tmin0 = fmaxf(fmaxf(anx, any), fmaxf(anz, 0.f));
        tmax0 = fminf(fminf(afx, afy), fminf(afz, tv.hit.t));
        tmin1 = fmaxf(fmaxf(bnx, bny), fmaxf(bnz, 0.f));
        tmax1 = fminf(fminf(bfx, bfy), fminf(bfz, tv.hit.t));
        c0 = (int)qb.z; c1 = (int)qb.w;
    } else {
        const float4 n0 = sc.nodes[tv.node * 4 + 0];
        const float4 n1 = sc.nodes[tv.node * 4 + 1];
        const float4 nz = sc.nodes[tv.node * 4 + 2];
        const float4 lk = sc.nodes[tv.node * 4 + 3];
        const float ax0 = __builtin_fmaf(n0.x, inv.x, -oi.x), ax1 = __builtin_fmaf(n0.y, inv.x, -oi.x);
        const float ay0 = __builtin_fmaf(n0.z, inv.y, -oi.y), ay1 = __builtin_fmaf(n0.w, inv.y, -oi.y);
        const float az0 = __builtin_fmaf(nz.x, inv.z, -oi.z), az1 = __builtin_fmaf(nz.y, inv.z, -oi.z);
        const float bx0 = __builtin_fmaf(n1.x, inv.x, -oi.x), bx1 = __builtin_fmaf(n1.y, inv.x, -oi.x);
        const float by0 = __builtin_fmaf(n1.z, inv.y, -oi.y), by1 = __builtin_fmaf(n1.w, inv.y, -oi.y);
        const float bz0 = __builtin_fmaf(nz.z, inv.z, -oi.z), bz1 = __builtin_fmaf(nz.w, inv.z, -oi.z);
        tmin0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), 0.f));
        tmax0 = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fminf(fmaxf(az0, az1), tv.hit.t));
        tmin1 = fmaxf(fmaxf(fminf(bx0, bx1), fminf(by0, by1)), fmaxf(fminf(bz0, bz1), 0.f));
        tmax1 = fminf(fminf(fmaxf(bx0, bx1), fmaxf(by0, by1)), fminf(fmaxf(bz0, bz1), tv.hit.t));
        c0 = __float_as_int(lk.x); c1 = __float_as_int(lk.y);
    }
    if (STATS) { cnt.box_tests += 2; if (first_active_lane()) cnt.w_inner += 1; }
    const bool h0 = tmin0 <= tmax0;
    const bool h1 = tmin1 <= tmax1;
    const bool swp = h1 & (!h0 | (tmin1 < tmin0));   // child 1 is the nearer (or the only) hit; bitwise: no short-circuit branch
    const int nearc = swp ? c1 : c0, farc = swp ? c0 : c1;
    tv.top[TRH_BLOCK] = farc;   // unconditional: the row above the top is free; it only counts if both were hit
    const bool any = h0 || h1;
    tv.node = any ? nearc : top;
    tv.top += (h0 && h1) ? TRH_BLOCK : (any ? 0 : -TRH_BLOCK);
}

// Moller-Trumbore on one 48-byte record (collision.h:33-74) and the candidate rule; p = leaf-order primitive index.
// The record may sit in vector registers (trav_leaf: every lane its own triangle) or in scalar registers
// (trav_packet: one triangle for the whole wave).
DEVFN void tri_test(const DevScene& sc, V3 o, V3 d, float4 r0, float4 r1, float4 r2, uint32_t p, Hit& hit) {
    const V3 v0 = mk(r0.x, r0.y, r0.z), e1 = mk(r1.x, r1.y, r1.z), e2 = mk(r2.x, r2.y, r2.z);
    const uint32_t id = __float_as_uint(r0.w);
    const V3 pvec = cross_f(d, e2);
    const V3 tvec = o - v0;
    const float det = dot_f(e1, pvec);
    const float u = dot_f(tvec, pvec);
    const V3 qvec = cross_f(tvec, e1);
    const float v = dot_f(d, qvec);
    // collision.h:44,51,59 evaluated without short-circuit branches
    const bool ok = !(det < EPS) & !(u < EPS) & !(u > det) & !(v < EPS) & !(u + v > det);
    if (ok) {
        const float rdet = 1.f / det;
        const float t = dot_f(e2, qvec) * rdet;
        // selects, not a second branch: most candidates that reach this point are accepted
        const bool a = accept_candidate(sc, o, d, t, id, hit);
        hit.t = a ? t : hit.t;
        hit.u = a ? u * rdet : hit.u;
        hit.v = a ? v * rdet : hit.v;
        hit.prim = a ? (int)p : hit.prim;
        hit.id = a ? id : hit.id;
    }
}

// ONE triangle of the current leaf, then advance inside the leaf link ((first + 1) << 4 | (count - 1)) or pop
// when the leaf is used up. Precondition: tv.node < -1.
template <bool STATS>
DEVFN void trav_leaf(const DevScene& sc, V3 o, V3 d, Trav& tv, int* __restrict__ stack, Counters& cnt) {
    const uint32_t bits = ~(uint32_t)tv.node;
    const uint32_t p = bits >> 4;
    if ((bits & 15u) != 0u) {
        // 32-bit byte offset, as in trav_inner; 48 p as shift-adds (v_mul_lo_u32 runs at quarter rate; the empty asm keeps
        // the compiler from folding the two steps back into a multiply)
        uint32_t p3 = (p << 1) + p;
        asm volatile("" : "+v"(p3));
        const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.tris) + (p3 << 4));
        const float4 r0 = rec[0];
        const float4 r1 = rec[1];
        const float4 r2 = rec[2];
        if (STATS) { cnt.tri_tests += 1; if (first_active_lane()) cnt.w_leaf += 1; }
        tri_test(sc, o, d, r0, r1, r2, p, tv.hit);
    }
    // (reading the stack top ahead, as trav_inner does, was measured here too: no gain, a leaf pops only once per leaf)
    if ((bits & 15u) > 1u) tv.node = (int)~(bits + 15u);
    else trav_pop(tv);
}

// ------------------------------------------------------------------ packet traversal (coherent rays)
// The 64 rays of a wave walk the tree TOGETHER: one wave-uniform node index, one stack for the wave (64 entries in
// the lanes of one VGPR, v_writelane / v_readlane), records and triangles fetched by SCALAR loads (one 32-byte request
// per step for the whole wave instead of 128 vector-L1 requests) and used as scalar operands of the lane's slab and
// Moller-Trumbore arithmetic. A subtree is entered when ANY lane's ray meets its box, the nearer child first by the
// majority of the lanes that meet both. Every lane tests every triangle of every leaf the wave visits; a triangle in
// a box its own ray misses cannot be hit (the box holds the triangle), so each lane ends with the same nearest hit as
// its own traversal would find (same candidate rule, accept_candidate). Meant for camera rays of one pixel or of a few
// neighbouring pixels: the union of their walks is hardly larger than each walk, and no lane waits for another.

// lane `index` of a VGPR as one element of a wave-uniform array (index and value wave-uniform)
DEVFN uint32_t write_lane(uint32_t v, uint32_t index, uint32_t value) {
    // v_writelane_b32 takes one scalar register besides M0 (constant bus): the lane select goes through M0
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0"
                 : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane((int)value)), "s"(__builtin_amdgcn_readfirstlane((int)index)) : "m0");
    return v;
}
DEVFN uint32_t read_lane(uint32_t v, uint32_t index) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)index); }

template <bool N16>
DEVFN void trav_packet(const DevScene& sc, V3 o, V3 d, bool valid, Hit& hit) {
    const float lim = sc.frame_inv_lim;
    V3 inv = mk(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -lim, lim), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -lim, lim),
                __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -lim, lim));
    const V3 oi = mk((o.x - sc.frame_c[0]) * inv.x, (o.y - sc.frame_c[1]) * inv.y, (o.z - sc.frame_c[2]) * inv.z);
    inv = inv * sc.frame_sinv;
    const uint32_t selx = (__float_as_uint(d.x) >> 31) ? 0x01000302u : 0x03020100u;
    const uint32_t sely = (__float_as_uint(d.y) >> 31) ? 0x01000302u : 0x03020100u;
    const uint32_t selz = (__float_as_uint(d.z) >> 31) ? 0x01000302u : 0x03020100u;
    hit.t = 3.402823466e+38f;
    hit.u = hit.v = 0.f;
    hit.prim = -1;
    hit.id = TRH_MISS_ID;
    const bool act = valid && !(d.x != d.x);   // NaN direction: a miss (trav_begin)
    uint32_t vstack = 0;                        // lane i = stack entry i
    uint32_t sp = 0;
    const unsigned long long actm = __builtin_amdgcn_ballot_w64(act);   // wave-uniform mask of the lanes that hold a ray
    int node = actm != 0ull ? 0 : -1;
    for (;;) {
        if (node >= 0) {
            float tmin0, tmax0, tmin1, tmax1;
            int c0, c1;
            if (N16) {
                const U32x8 q = *reinterpret_cast<const TRH_SCALAR_AS U32x8*>(reinterpret_cast<uintptr_t>(sc.nodes16) + ((size_t)(uint32_t)node << 5));
                const H2 hx0 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(q[0], q[0], selx)), hy0 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(q[1], q[1], sely));
                const H2 hx1 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(q[2], q[2], selx)), hy1 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(q[3], q[3], sely));
                const H2 hz0 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(q[4], q[4], selz)), hz1 = __builtin_bit_cast(H2, __builtin_amdgcn_perm(q[5], q[5], selz));
                const float anx = __builtin_fmaf((float)hx0.x, inv.x, -oi.x), afx = __builtin_fmaf((float)hx0.y, inv.x, -oi.x);
                const float any = __builtin_fmaf((float)hy0.x, inv.y, -oi.y), afy = __builtin_fmaf((float)hy0.y, inv.y, -oi.y);
                const float anz = __builtin_fmaf((float)hz0.x, inv.z, -oi.z), afz = __builtin_fmaf((float)hz0.y, inv.z, -oi.z);
                const float bnx = __builtin_fmaf((float)hx1.x, inv.x, -oi.x), bfx = __builtin_fmaf((float)hx1.y, inv.x, -oi.x);
                const float bny = __builtin_fmaf((float)hy1.x, inv.y, -oi.y), bfy = __builtin_fmaf((float)hy1.y, inv.y, -oi.y);
                const float bnz = __builtin_fmaf((float)hz1.x, inv.z, -oi.z), bfz = __builtin_fmaf((float)hz1.y, inv.z, -oi.z);
                tmin0 = fmaxf(fmaxf(anx, any), fmaxf(anz, 0.f));
                tmax0 = fminf(fminf(afx, afy), fminf(afz, hit.t));
                tmin1 = fmaxf(fmaxf(bnx, bny), fmaxf(bnz, 0.f));
                tmax1 = fminf(fminf(bfx, bfy), fminf(bfz, hit.t));
                c0 = (int)q[6]; c1 = (int)q[7];
            } else {
                const U32x16 q = *reinterpret_cast<const TRH_SCALAR_AS U32x16*>(reinterpret_cast<uintptr_t>(sc.nodes) + ((size_t)(uint32_t)node << 6));
                const float ax0 = __builtin_fmaf(__uint_as_float(q[0]), inv.x, -oi.x), ax1 = __builtin_fmaf(__uint_as_float(q[1]), inv.x, -oi.x);
                const float ay0 = __builtin_fmaf(__uint_as_float(q[2]), inv.y, -oi.y), ay1 = __builtin_fmaf(__uint_as_float(q[3]), inv.y, -oi.y);
                const float az0 = __builtin_fmaf(__uint_as_float(q[8]), inv.z, -oi.z), az1 = __builtin_fmaf(__uint_as_float(q[9]), inv.z, -oi.z);
                const float bx0 = __builtin_fmaf(__uint_as_float(q[4]), inv.x, -oi.x), bx1 = __builtin_fmaf(__uint_as_float(q[5]), inv.x, -oi.x);
                const float by0 = __builtin_fmaf(__uint_as_float(q[6]), inv.y, -oi.y), by1 = __builtin_fmaf(__uint_as_float(q[7]), inv.y, -oi.y);
                const float bz0 = __builtin_fmaf(__uint_as_float(q[10]), inv.z, -oi.z), bz1 = __builtin_fmaf(__uint_as_float(q[11]), inv.z, -oi.z);
                tmin0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), 0.f));
                tmax0 = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fminf(fmaxf(az0, az1), hit.t));
                tmin1 = fmaxf(fmaxf(fminf(bx0, bx1), fminf(by0, by1)), fmaxf(fminf(bz0, bz1), 0.f));
                tmax1 = fminf(fminf(fmaxf(bx0, bx1), fmaxf(by0, by1)), fminf(fmaxf(bz0, bz1), hit.t));
                c0 = (int)q[12]; c1 = (int)q[13];
            }
            // (masks, not per-lane booleans: __ballot(int) would materialise 0 / 1 in a VGPR and compare it again)
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(tmin0 <= tmax0) & actm;
            const unsigned long long b1 = __builtin_amdgcn_ballot_w64(tmin1 <= tmax1) & actm;
            if ((b0 != 0ull) & (b1 != 0ull)) {
                const unsigned long long sw = b1 & (~b0 | __builtin_amdgcn_ballot_w64(tmin1 < tmin0));   // lanes for which child 1 is the nearer (or only) hit
                const bool first1 = 2u * (uint32_t)__popcll(sw) > (uint32_t)__popcll(b0 | b1);
                vstack = write_lane(vstack, sp, (uint32_t)(first1 ? c0 : c1));
                sp += 1u;
                node = first1 ? c1 : c0;
            } else {
                node = b0 != 0ull ? c0 : (b1 != 0ull ? c1 : -1);
            }
        } else if (node < -1) {
            const uint32_t bits = ~(uint32_t)node;
            const uint32_t first = bits >> 4, count = bits & 15u;
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t p = first + k;
                const TRH_SCALAR_AS F32x4* rec = reinterpret_cast<const TRH_SCALAR_AS F32x4*>(reinterpret_cast<uintptr_t>(sc.tris) + (size_t)p * 48u);
                const F32x4 s0 = rec[0], s1 = rec[1], s2 = rec[2];
                if (act) tri_test(sc, o, d, make_float4(s0[0], s0[1], s0[2], s0[3]), make_float4(s1[0], s1[1], s1[2], s1[3]), make_float4(s2[0], s2[1], s2[2], s2[3]), p, hit);
            }
            node = -1;
        }
        if (node == -1) {   // pop
            if (sp == 0u) break;
            sp -= 1u;
            node = (int)read_lane(vstack, sp);
        }
    }
}

template <bool STATS, bool N16>
DEVFN void intersect(const DevScene& sc, V3 o, V3 d, Hit& hit, int* __restrict__ stack, Counters& cnt) {
    Trav tv;
    stack[0] = -1;   // sentinel row
    trav_begin(sc, tv, o, d, stack);
    while (trav_active(tv)) {
        while (tv.node >= 0) trav_inner<STATS, N16>(sc, o, tv, stack, cnt);
        if (trav_active(tv)) trav_leaf<STATS>(sc, o, d, tv, stack, cnt);
    }
    hit = tv.hit;
}

// ------------------------------------------------------------------ textures (src/texture.h:50-57)
template <bool STATS>
DEVFN float4 get_pixel(const DevScene& sc, uint32_t tex, float u, float v, Counters& cnt) {
    const DevTexture t = sc.textures[tex];
    if (STATS) cnt.tex_fetches += 1;
    const float fw = (float)t.width, fh = (float)t.height;
    const float fu = u - floorf(u);
    const float w1 = 1.f - v;
    const float fv = w1 - floorf(w1);
    // clamp(x, 0, n-1) then u32 conversion; a NaN index is 0 (fmaxf drops the NaN)
    const uint32_t i = (uint32_t)fminf(fmaxf(fu * fw, 0.f), fw - 1.f);
    const uint32_t j = (uint32_t)fminf(fmaxf(fv * fh, 0.f), fh - 1.f);
    return t.texels[(size_t)j * (size_t)t.width + i];
}

DEVFN float pow2(float x) { return x * x; }
DEVFN float pow5(float x) { const float x2 = x * x; return x2 * x2 * x; }
DEVFN float schlick(float c, float ref_idx) {
    const float r0 = pow2((1.f - ref_idx) / (1.f + ref_idx));
    return r0 + (1.f - r0) * pow5(1.f - c);
}

struct PathState {
    V3 o, d;            // current ray
    V3 throughput;
    V3 radiance;        // "pixel" of CpuTrace::Trace
    uint32_t rng;
    uint32_t bounce;    // iterations done so far
};

// One iteration of the bounce loop of CpuTrace::Trace (cpu_trace.cpp:113-167) AFTER the
// intersection: shading, sky, Russian roulette. Returns true if the path continues.
template <bool STATS>
DEVFN bool shade(const DevScene& sc, PathState& ps, const Hit& hit, Counters& cnt) {
    cnt.rays += 1;  // ++raycount_ (cpu_trace.cpp:115-116)
    if (STATS) { if (first_active_lane()) cnt.w_shade += 1; }
    ps.bounce += 1;
    if (hit.prim >= 0) {
        if (STATS) cnt.hits += 1;
        // ---- attribute interpolation (cpu_details.cpp:169-182)
        const float4 s0 = sc.shade[hit.prim * 4 + 0];
        const float4 s1 = sc.shade[hit.prim * 4 + 1];
        const float4 s2 = sc.shade[hit.prim * 4 + 2];
        const float4 s3 = sc.shade[hit.prim * 4 + 3];
        const V3 n0 = mk(s0.x, s0.y, s0.z), n1 = mk(s0.w, s1.x, s1.y), n2 = mk(s1.z, s1.w, s2.x);
        const float w = 1.f - hit.u - hit.v;
        const V3 point = ps.o + hit.t * ps.d;
        V3 normal = w * n0 + hit.u * n1 + hit.v * n2;
        const float tu = w * s2.y + hit.u * s2.w + hit.v * s3.y;
        const float tv = w * s2.z + hit.u * s3.x + hit.v * s3.z;
        // Scenes have a handful of materials and the lanes of a shade call mostly hit the same one (camera rays of one pixel,
        // bounce rays of a few neighbours): then the 64-byte record comes through the scalar cache -- one request for the
        // wave and a shorter wait than the third dependent vector round trip of this function
        const uint32_t mid = __float_as_uint(s3.w);
        const uint32_t mid0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mid);
        DevMaterial m;
        if (__builtin_amdgcn_ballot_w64(mid != mid0) == 0ull)
            m = __builtin_bit_cast(DevMaterial, *reinterpret_cast<const TRH_SCALAR_AS U32x16*>(reinterpret_cast<uintptr_t>(sc.materials) + (size_t)mid0 * sizeof(DevMaterial)));
        else
            m = sc.materials[mid];

        // ---- texture-or-constant (src/material.h:164-203), in Scatter's order (:217-221)
        V3 emissive = mk(m.emissive[0], m.emissive[1], m.emissive[2]);
        if (m.tex[4] != TRH_MISS_ID) { const float4 p = get_pixel<STATS>(sc, m.tex[4], tu, tv, cnt); emissive = mk(p.x, p.y, p.z); }
        float metalness = m.metalness;
        if (m.tex[3] != TRH_MISS_ID) metalness = get_pixel<STATS>(sc, m.tex[3], tu, tv, cnt).x;
        V3 basecolor = mk(m.albedo[0], m.albedo[1], m.albedo[2]);
        if (m.tex[0] != TRH_MISS_ID) { const float4 p = get_pixel<STATS>(sc, m.tex[0], tu, tv, cnt); basecolor = mk(p.x, p.y, p.z); }
        float roughness = m.roughness;
        if (m.tex[2] != TRH_MISS_ID) roughness = get_pixel<STATS>(sc, m.tex[2], tu, tv, cnt).x;
        if (m.tex[1] != TRH_MISS_ID) {
            const float4 t0 = sc.tangents[hit.prim * 3 + 0];
            const float4 t1 = sc.tangents[hit.prim * 3 + 1];
            const float4 t2 = sc.tangents[hit.prim * 3 + 2];
            const V3 tangent = w * mk(t0.x, t0.y, t0.z) + hit.u * mk(t0.w, t1.x, t1.y) + hit.v * mk(t1.z, t1.w, t2.x);
            const float4 p = get_pixel<STATS>(sc, m.tex[1], tu, tv, cnt);
            const V3 nm = mk(p.x * 2.f - 1.f, p.y * 2.f - 1.f, p.z * 2.f - 1.f);
            const V3 bitangent = cross(normal, normalize(tangent - dot(tangent, normal) * normal));
            normal = normalize(bitangent * nm.x + tangent * nm.y + normal * nm.z);
        }

        // ---- debug views (cpu_trace.cpp:127-137): the path ends here and returns one material input
        if (sc.debug_view != 0u) {
            V3 out = emissive;
            if (sc.debug_view == 1u) out = basecolor;
            else if (sc.debug_view == 2u) {
                const float* v = sc.view3;   // glm mat3 * vec3: (col0 * x + col1 * y) + col2 * z
                const V3 vn = mk((v[0] * normal.x + v[3] * normal.y) + v[6] * normal.z, (v[1] * normal.x + v[4] * normal.y) + v[7] * normal.z,
                                 (v[2] * normal.x + v[5] * normal.y) + v[8] * normal.z);
                out = 0.5f * normalize(mk(1.f + vn.x, 1.f + vn.y, 1.f + vn.z));
            } else if (sc.debug_view == 3u) out = mk(metalness, metalness, metalness);
            else if (sc.debug_view == 4u) out = mk(roughness, roughness, roughness);
            ps.radiance = out;
            return false;
        }

        // ---- Material::Scatter (src/material.h:210-268)
        const V3 raydir = ps.d;
        const float VdotN = dot(raydir, normal);
        const float z = 2.f * fastrand(ps.rng) - 1.f;
        const float phi = 2.f * PI * fastrand(ps.rng);
        const float r = __builtin_sqrtf(1.f - pow2(z));
        float sphi, cphi;
        sincosf(phi, &sphi, &cphi);
        const V3 scattered = normal + mk(r * cphi, r * sphi, z);
        const V3 reflected = reflect(raydir, normal);
        const V3 specular = lerp(reflected, scattered, roughness);
        V3 dir, attenuation;
        if (m.translucent > EPS) {  // BTDF
            const bool inside = VdotN > EPS;
            const float cosine = inside ? __builtin_sqrtf(1.f - pow2(m.ior) * (1.f - pow2(VdotN))) : -VdotN;
            const float ior = inside ? m.ior : 1.f / m.ior;
            const V3 refracted = refract(raydir, normal, ior);
            const V3 transmitted = lerp(refracted, scattered, roughness);
            const bool is_specular = fastrand(ps.rng) < schlick(cosine, ior);
            dir = is_specular ? specular : transmitted;
            attenuation = basecolor;
        } else {  // BRDF
            const V3 specularcolor = lerp(mk(.85f, .85f, .85f), basecolor, metalness);
            const float materialspecularchance = lerp(.1f, 1.f, metalness);
            const float fresnelspecularchance = lerp(materialspecularchance, 1.f, (1.f - roughness) * schlick(-VdotN, 1.f));
            const bool is_specular = fastrand(ps.rng) < fresnelspecularchance;
            dir = is_specular ? specular : scattered;
            attenuation = is_specular ? specularcolor : basecolor;
        }
        dir = normalize(dir);
        ps.o = point + 0.001f * dir;
        ps.d = dir;
        ps.radiance = ps.radiance + emissive * ps.throughput;  // cpu_trace.cpp:141
        ps.throughput = ps.throughput * attenuation;           // cpu_trace.cpp:143
    } else {
        // ---- sky (cpu_trace.cpp:146-156)
        const DevMaterial sky = __builtin_bit_cast(DevMaterial, *reinterpret_cast<const TRH_SCALAR_AS U32x16*>(reinterpret_cast<uintptr_t>(sc.materials)));   // material 0 (scene.h:21,67), through the scalar cache
        V3 emission = mk(sky.emissive[0], sky.emissive[1], sky.emissive[2]);
        if (sky.tex[4] != TRH_MISS_ID) {
            const float su = atan2f(ps.d.z, ps.d.x) / (2 * PI) + 0.5f;
            const float sv = asinf(ps.d.y) / PI + 0.5f;
            const float4 p = get_pixel<STATS>(sc, sky.tex[4], su, sv, cnt);
            emission = mk(p.x, p.y, p.z);
        }
        ps.radiance = ps.radiance + emission * ps.throughput;
        return false;
    }
    if (ps.bounce >= sc.max_bounces) return false;  // for (t < kMaxBounces)
    if (sc.flags & 1u) {  // USE_RUSSIAN_ROULETTE (cpu_trace.cpp:158-166)
        const float p = EPS + fmaxf(ps.throughput.x, fmaxf(ps.throughput.y, ps.throughput.z));
        if (fastrand(ps.rng) > p) return false;
        ps.throughput = ps.throughput * (1.f / p);
    }
    return true;
}

// start of a camera path: RNG state, jitter, primary ray (cpu_trace.cpp:59-63)
DEVFN void generate(const DevScene& sc, PathState& ps, int px, int py, uint32_t sample, uint32_t seed_hash) {
    const uint32_t pixel = (uint32_t)(py * sc.width + px);
    ps.rng = path_rng_state(pixel, sample, seed_hash);
    const float u = ((float)px + fastrand(ps.rng)) / sc.width_f;
    const float v = ((float)py + fastrand(ps.rng)) / sc.height_f;
    camera_ray(sc, u, v, ps.o, ps.d);
    ps.throughput = mk(1.f, 1.f, 1.f);
    ps.radiance = mk(0.f, 0.f, 0.f);
    ps.bounce = 0;
}

}  // namespace dev
