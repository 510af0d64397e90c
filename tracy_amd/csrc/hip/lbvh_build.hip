// lbvh_build.hip -- linear BVH built on the GPU (gfx950); see lbvh_build.h for the what and why.
//
// Stages (all on one stream; a read-back of a few bytes after the bounds, per tree level and at the end):
//   k_bounds      scene box + centroid box: wave reduction, then ordered-integer atomic min / max
//   k_morton      63-bit Morton code of every centroid (21 bits per axis inside the centroid box)
//   radix sort    hipcub::DeviceRadixSort::SortPairs (code, triangle) -- stable, so equal codes keep input order
//   clusters      (>= 4096 triangles) ranges that share their top 6..21 code bits (about two triangles each): boxes by one kernel
//   k_top_level   the UPPER levels: binned SAH over the cluster boxes, one launch per level, one workgroup per node (round 3:
//                 on the device; 12 bytes -- cluster count, biggest cluster -- are all the host reads); one seed range per cluster
//   k_level_sah   below the seeds: binned SAH again, one wave per range, one launch per generation of the work queue
//   (k_level      TRH_LBVH_SPLIT=morton, kept for A/B: the linear BVH's splits -- a range of > 2 triangles (kLeafMax) is
//                 split where the highest differing code bit flips (the LBVH split), or at its median when
//                 the codes are equal or the remaining depth budget demands it (the depth guard of the host
//                 builder: a Morton tree of a scene with one dense object in a large room is otherwise 40+
//                 levels deep); the inner node is stored at its split position, which is unique, so the
//                 layout does not depend on the order the work queue fills. A range of <= 2 triangles is a
//                 leaf: its thread computes the box and walks up; the second thread to arrive at a node
//                 merges the child boxes (acquire/release counter per node))
//   scan          exclusive sum over the created nodes = compacted record index (the root is moved to record 0)
//   k_emit        64-byte fp32 and 32-byte fp16 records (layouts in bvh_builder.h), boxes padded and rounded
//                 outward exactly as the host builder does; sub-trees of <= 2 triangles become leaf links
//   k_triangles   48-byte Moller-Trumbore records in leaf (= sorted) order
#include "lbvh_build.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "../host/half_round.h"

namespace lbvh {
namespace {

constexpr int kBlock = 256;
constexpr int kDepthAim = 27;             // see build(): the depth the guard aims for
// triangles per leaf (the link format allows bvh::kMaxLeafSize = 8). Swept with the SAH top tree in place (64 spp per launch,
// dragon / spheres / dragon 400 k): 1: 12.00 / 7.47 / 16.0 ms, 2: 11.54 / 7.18 / 13.8, 3: 11.66 / 7.28, 4: 11.91 / 7.52 / 15.5, 6: 13.02 / 7.92
#ifndef TRH_LBVH_LEAF
#define TRH_LBVH_LEAF 2
#endif
constexpr uint32_t kLeafMax = TRH_LBVH_LEAF;
// ... and with SAH splits (k_level_sah) a range of up to this many triangles stays a leaf when the host builder's rule says a leaf
// is cheaper than its best split (n C_intersect <= C_traverse + C_intersect cost / area)
#ifndef TRH_LBVH_SAH_LEAF
#define TRH_LBVH_SAH_LEAF 4
#endif
constexpr uint32_t kSahLeafMax = TRH_LBVH_SAH_LEAF;
// (C_traverse comes from the host builder at run time, bvh::cost_traverse(): one constant, one scale for trh_bvh_info.sah_cost)
constexpr float kCostIntersect = 1.0f;

struct Tri { float v0[3], v1[3], v2[3]; uint32_t id; };
static_assert(sizeof(Tri) == sizeof(bvh::InputTriangle), "Tri mirrors bvh::InputTriangle");

struct Box { float mn[3], mx[3]; };

struct Stats {
    double area_exact, area_quant, sah;
    uint32_t max_depth, max_leaf;
};

struct Frame {
    float cmn[3], cscale[3];   // Morton grid: q = (c - cmn) * cscale
    float scene_extent;        // padding term of the host builder
    float root_area_inv;
    float cost_traverse;       // bvh::cost_traverse(): the host builder's C_traverse, so that sah_cost of the two builders is on one scale
    float frame_c[3];
    double frame_s;
};

#define LB_TRY(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return 1; }     \
    } while (0)

// floats ordered as unsigned integers (atomicMin / atomicMax on floats of either sign)
__host__ __device__ inline uint32_t float_to_ordered(float f) {
    const uint32_t u = bvh::float_to_bits(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ordered_to_float(uint32_t o) {
    return bvh::bits_to_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__device__ inline Box tri_box(const Tri& t) {
    Box b;
    for (int a = 0; a < 3; ++a) {
        b.mn[a] = fminf(t.v0[a], fminf(t.v1[a], t.v2[a]));
        b.mx[a] = fmaxf(t.v0[a], fmaxf(t.v1[a], t.v2[a]));
    }
    return b;
}
__device__ inline float half_area(const Box& b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (dx < 0.f || dy < 0.f || dz < 0.f) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}

// out[0..2] min, [3..5] max of the triangle boxes; [6..8] min, [9..11] max of their centres
__global__ __launch_bounds__(kBlock) void k_bounds(const Tri* __restrict__ tris, uint32_t n, uint32_t* __restrict__ out) {
    float v[12];
    for (int a = 0; a < 3; ++a) { v[a] = FLT_MAX; v[3 + a] = -FLT_MAX; v[6 + a] = FLT_MAX; v[9 + a] = -FLT_MAX; }
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const Box b = tri_box(tris[i]);
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * (b.mn[a] + b.mx[a]);
            v[a] = fminf(v[a], b.mn[a]);
            v[3 + a] = fmaxf(v[3 + a], b.mx[a]);
            v[6 + a] = fminf(v[6 + a], c);
            v[9 + a] = fmaxf(v[9 + a], c);
        }
    }
    for (int k = 0; k < 12; ++k) {
        const bool is_min = (k % 6) < 3;
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(v[k], off, 64);
            v[k] = is_min ? fminf(v[k], o) : fmaxf(v[k], o);
        }
        if ((threadIdx.x & 63) == 0) {
            if (is_min) atomicMin(&out[k], float_to_ordered(v[k]));
            else atomicMax(&out[k], float_to_ordered(v[k]));
        }
    }
}

__device__ inline uint64_t spread3(uint32_t v) {   // 21 bits -> every third bit of 63
    uint64_t x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(kBlock) void k_morton(const Tri* __restrict__ tris, uint32_t n, Frame f, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Box b = tri_box(tris[i]);
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float c = 0.5f * (b.mn[a] + b.mx[a]);
        const float x = (c - f.cmn[a]) * f.cscale[a];
        q[a] = (uint32_t)fminf(fmaxf(x, 0.f), 2097151.f);
    }
    keys[i] = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    vals[i] = i;
}

// ---- Morton clusters (the SAH top tree of build()): a cluster = the sorted positions that share the top `shift`ed bits
__global__ __launch_bounds__(kBlock) void k_cluster_flags(const uint64_t* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0u || (keys[i] >> shift) != (keys[i - 1u] >> shift)) ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void k_cluster_starts(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ index, uint32_t n, uint32_t* __restrict__ starts) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) starts[index[i]] = i;
}
// (the number of clusters is still on the device when this is launched: flags[n - 1] + index[n - 1]; the grid covers the upper bound)
__global__ __launch_bounds__(kBlock) void k_cluster_boxes(const Tri* __restrict__ tris, const uint32_t* __restrict__ prim, const uint32_t* __restrict__ starts, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ index,
                                                          uint32_t n, Box* __restrict__ boxes, uint32_t* __restrict__ biggest) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t n_clusters = flags[n - 1u] + index[n - 1u];
    if (c >= n_clusters) return;
    const uint32_t lo = starts[c], hi = c + 1u < n_clusters ? starts[c + 1u] : n;
    if (hi - lo > __hip_atomic_load(biggest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(biggest, hi - lo);
    Box b = tri_box(tris[prim[lo]]);
    for (uint32_t p = lo + 1u; p < hi; ++p) {
        const Box t = tri_box(tris[prim[p]]);
        for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], t.mn[a]); b.mx[a] = fmaxf(b.mx[a], t.mx[a]); }
    }
    boxes[c] = b;
}

struct Item { int lo, hi, parent, side, depth; };   // a range of sorted positions, the node above it, which child it is, the depth of the record it would become

__host__ __device__ inline int levels_needed(uint32_t n) {   // record levels a median-split sub-tree of n triangles needs
    int l = 0;
    while (n > kLeafMax) { n = (n + 1) / 2; ++l; }
    return l;
}

// ---- SAH top tree over the Morton clusters, on the device (round 3; round 2 built it with one host thread after two round trips).
// The Morton curve clusters the upper levels of a scene badly (a split at the highest differing bit is a spatial-median
// split, whatever the geometry), and that is where a linear BVH loses against SAH; inside a small cluster the difference
// is nil. So the upper levels are built with binned SAH (16 bins per axis, cost = half area x triangle count)
// over the cluster boxes, level by level: one launch per level, one workgroup per node of the level. A node owns a
// range of `order` (cluster indices, Morton order to begin with); the workgroup bins it (LDS atomics on ordered
// integers), one thread sweeps the 3 x 15 candidate planes with the host builder's rule -- first minimum wins, both
// sides must still fit below max_seed_depth -- and the range is stably partitioned into the level's output array (ballot +
// prefix counts), so the result does not depend on thread timing. A node takes cluster (mid - 1)'s last sorted position
// as its index (mid = its split position in `order`: unique, and the linear build never uses a cluster's last position);
// a side of ONE cluster becomes a seed range of the linear build (k_level), a larger side an item of the next level.
struct TopItem { int begin, end, parent, side, depth; };

__global__ __launch_bounds__(kBlock) void k_iota(int* __restrict__ out, int n) {
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (i < n) out[i] = i;
}

__device__ inline int ceil_log2_u32(uint32_t k) { int l = 0; while ((1u << l) < k) ++l; return l; }

__global__ __launch_bounds__(kBlock) void k_top_level(const TopItem* __restrict__ items, const unsigned int* __restrict__ item_count, TopItem* __restrict__ next_items,
                                                      unsigned int* __restrict__ next_count, const int* __restrict__ order_in, int* __restrict__ order_out,
                                                      const Box* __restrict__ cboxes, const uint32_t* __restrict__ starts, uint32_t n_clusters, uint32_t n, int max_seed_depth,
                                                      Item* __restrict__ seeds, unsigned int* __restrict__ seed_count, Stats* __restrict__ stats,
                                                      int* __restrict__ parent, int* __restrict__ child, uint32_t* __restrict__ live, int* __restrict__ root_out) {
    constexpr int B = 16;
    if (blockIdx.x >= *item_count) return;
    const TopItem it = items[blockIdx.x];
    const int k = it.end - it.begin;   // >= 2
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ uint32_t s_cb[6];                 // centroid bounds, ordered
    __shared__ uint32_t s_bb[3][B][6];           // bin boxes, ordered
    __shared__ uint32_t s_cnt[3][B], s_num[3][B];
    __shared__ int s_axis, s_bin, s_left;
    __shared__ int s_wl[kBlock / 64];            // left elements per wave of the current chunk
    __shared__ int s_id;
    for (int i = tid; i < 3 * B * 6; i += kBlock) (&s_bb[0][0][0])[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
    for (int i = tid; i < 3 * B; i += kBlock) { (&s_cnt[0][0])[i] = 0u; (&s_num[0][0])[i] = 0u; }
    if (tid < 6) s_cb[tid] = tid < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    auto cluster_count = [&](int c) { return ((uint32_t)c + 1u < n_clusters ? starts[c + 1] : n) - starts[c]; };
    // ---- bounds of the centroids
    {
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = it.begin + tid; i < it.end; i += kBlock) {
            const Box b = cboxes[order_in[i]];
            for (int a = 0; a < 3; ++a) { const float c = 0.5f * (b.mn[a] + b.mx[a]); mn[a] = fminf(mn[a], c); mx[a] = fmaxf(mx[a], c); }
        }
        for (int a = 0; a < 3; ++a) {
            for (int off = 32; off > 0; off >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64)); }
            if (lane == 0) { atomicMin(&s_cb[a], float_to_ordered(mn[a])); atomicMax(&s_cb[3 + a], float_to_ordered(mx[a])); }
        }
    }
    __syncthreads();
    float cmn[3], ext[3], scale[3];
    for (int a = 0; a < 3; ++a) {
        cmn[a] = ordered_to_float(s_cb[a]);
        ext[a] = ordered_to_float(s_cb[3 + a]) - cmn[a];
        scale[a] = ext[a] > 0.f ? (float)B / ext[a] : 0.f;
    }
    auto bin_of = [&](const Box& b, int a) { const float c = 0.5f * (b.mn[a] + b.mx[a]); const int x = (int)((c - cmn[a]) * scale[a]); return x < 0 ? 0 : (x > B - 1 ? B - 1 : x); };
    // ---- bins
    for (int i = it.begin + tid; i < it.end; i += kBlock) {
        const int c = order_in[i];
        const Box b = cboxes[c];
        const uint32_t cnt = cluster_count(c);
        for (int a = 0; a < 3; ++a) {
            if (!(ext[a] > 0.f)) continue;
            const int x = bin_of(b, a);
            for (int d = 0; d < 3; ++d) { atomicMin(&s_bb[a][x][d], float_to_ordered(b.mn[d])); atomicMax(&s_bb[a][x][3 + d], float_to_ordered(b.mx[d])); }
            atomicAdd(&s_cnt[a][x], cnt);
            atomicAdd(&s_num[a][x], 1u);
        }
    }
    __syncthreads();
    // ---- the sweep (one thread: 45 candidate planes)
    if (tid == 0) {
        float best = FLT_MAX;
        int best_axis = -1, best_bin = -1, best_left = 0;
        for (int a = 0; a < 3; ++a) {
            if (!(ext[a] > 0.f)) continue;
            float rarea[B];
            uint32_t rcnt[B], rnum[B];
            Box acc;
            for (int x = 0; x < 3; ++x) { acc.mn[x] = FLT_MAX; acc.mx[x] = -FLT_MAX; }
            uint32_t c = 0, m = 0;
            for (int b = B - 1; b > 0; --b) {
                if (s_num[a][b]) for (int x = 0; x < 3; ++x) { acc.mn[x] = fminf(acc.mn[x], ordered_to_float(s_bb[a][b][x])); acc.mx[x] = fmaxf(acc.mx[x], ordered_to_float(s_bb[a][b][3 + x])); }
                c += s_cnt[a][b]; m += s_num[a][b];
                rarea[b] = half_area(acc); rcnt[b] = c; rnum[b] = m;
            }
            Box lacc;
            for (int x = 0; x < 3; ++x) { lacc.mn[x] = FLT_MAX; lacc.mx[x] = -FLT_MAX; }
            uint32_t lc = 0, lm = 0;
            for (int b = 0; b < B - 1; ++b) {
                if (s_num[a][b]) for (int x = 0; x < 3; ++x) { lacc.mn[x] = fminf(lacc.mn[x], ordered_to_float(s_bb[a][b][x])); lacc.mx[x] = fmaxf(lacc.mx[x], ordered_to_float(s_bb[a][b][3 + x])); }
                lc += s_cnt[a][b]; lm += s_num[a][b];
                if (lm == 0 || rnum[b + 1] == 0) continue;
                // depth guard: both sides must still fit below max_seed_depth
                if (it.depth + 1 + ceil_log2_u32(lm > rnum[b + 1] ? lm : rnum[b + 1]) > max_seed_depth) continue;
                const float cost = half_area(lacc) * (float)lc + rarea[b + 1] * (float)rcnt[b + 1];
                if (cost < best) { best = cost; best_axis = a; best_bin = b; best_left = (int)lm; }
            }
        }
        s_axis = best_axis;
        s_bin = best_bin;
        s_left = best_axis >= 0 ? best_left : k / 2;   // no plane (equal centroids, or the depth guard): the median of the range as it stands
        // this node
        const int mid = it.begin + s_left;
        const int id = (int)(((uint32_t)mid < n_clusters ? starts[mid] : n) - 1u);   // cluster (mid - 1)'s last sorted position
        s_id = id;
        parent[id] = it.parent;
        live[id] = 1u;
        if (it.parent >= 0) child[2 * it.parent + it.side] = id; else *root_out = id;
        if ((uint32_t)it.depth > __hip_atomic_load(&stats->max_depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats->max_depth, (uint32_t)it.depth);
        const int nl = s_left, nr = k - s_left;
        if (nl > 1) { const unsigned int pos = atomicAdd(next_count, 1u); next_items[pos] = TopItem{it.begin, mid, id, 0, it.depth + 1}; }
        if (nr > 1) { const unsigned int pos = atomicAdd(next_count, 1u); next_items[pos] = TopItem{mid, it.end, id, 1, it.depth + 1}; }
    }
    __syncthreads();
    // ---- stable partition into order_out; a side of one cluster becomes a seed of the linear build
    const int axis = s_axis, bin = s_bin, nl = s_left, id = s_id;
    int run_l = 0, run_r = 0;   // elements of the chunks before this one (the same in every thread)
    for (int base = it.begin; base < it.end; base += kBlock) {
        const int i = base + tid;
        const bool in = i < it.end;
        int c = 0;
        bool left = false;
        if (in) {
            c = order_in[i];
            left = axis >= 0 ? bin_of(cboxes[c], axis) <= bin : (i - it.begin) < nl;
        }
        const unsigned long long ml = __ballot(in && left);
        if (lane == 0) s_wl[wave] = (int)__popcll(ml);
        __syncthreads();
        int before_l = 0, total_l = 0;
        for (int w = 0; w < kBlock / 64; ++w) { if (w < wave) before_l += s_wl[w]; total_l += s_wl[w]; }
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        const int my_l = before_l + (int)__popcll(ml & below);   // left elements of this chunk before this thread's
        if (in) {   // (the threads of a chunk that hold an element are 0 .. chunk - 1: tid elements come before this one)
            const int pos = left ? it.begin + run_l + my_l : it.begin + nl + run_r + (tid - my_l);
            order_out[pos] = c;
            const bool alone = (left && nl == 1) || (!left && k - nl == 1);
            if (alone) {
                const uint32_t lo = starts[c], hi = ((uint32_t)c + 1u < n_clusters ? starts[c + 1] : n) - 1u;
                const unsigned int sp = atomicAdd(seed_count, 1u);
                seeds[sp] = Item{(int)lo, (int)hi, id, left ? 0 : 1, it.depth + 1};
            }
        }
        const int chunk = it.end - base < kBlock ? it.end - base : kBlock;
        run_l += total_l;
        run_r += chunk - total_l;
        __syncthreads();
    }
}

// child links: >= 0 inner node (index = its split position), < -1 leaf ~((first << 4) | count)
__global__ __launch_bounds__(kBlock) void k_level(const Item* __restrict__ items, int n_items, int depth_limit, const uint64_t* __restrict__ keys,
                                                  const Tri* __restrict__ tris, const uint32_t* __restrict__ prim, Item* __restrict__ next_items,
                                                  const unsigned int* __restrict__ item_count, unsigned int* __restrict__ next_count, Stats* __restrict__ stats, int* __restrict__ parent, int* child,
                                                  uint32_t* __restrict__ live, Box* leaf_box, Box* node_box, unsigned int* arrived, int* __restrict__ root_out) {
    const int k = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (k >= n_items || (unsigned int)k >= *item_count) return;   // n_items: the launch's upper bound; *item_count: what the level above queued
    const Item it = items[k];
    const int depth = it.depth;
    const uint32_t count = (uint32_t)(it.hi - it.lo + 1);
    if (count > kLeafMax) {
        // ---- inner node: LBVH split = last position whose code shares more than the range's common prefix with the first
        int gamma = it.lo + (int)(count / 2) - 1;   // median (equal codes, or the depth guard below)
        const uint64_t first = keys[it.lo], last = keys[it.hi];
        if (first != last) {
            const int prefix = __clzll((long long)(first ^ last));
            int lo = it.lo, hi = it.hi;             // invariant: keys[lo] shares > prefix bits with first, keys[hi] does not
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                const uint64_t x = first ^ keys[mid];
                if (x == 0 || __clzll((long long)x) > prefix) lo = mid; else hi = mid;
            }
            const uint32_t nl = (uint32_t)(lo - it.lo + 1), nr = count - nl;
            const int need = levels_needed(nl) > levels_needed(nr) ? levels_needed(nl) : levels_needed(nr);
            if (depth + need <= depth_limit) gamma = lo;
        }
        parent[gamma] = it.parent;
        live[gamma] = 1u;
        // (ranges of one generation sit at different depths once the build is seeded: every thread checks, few need the atomic)
        if ((uint32_t)depth > __hip_atomic_load(&stats->max_depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats->max_depth, (uint32_t)depth);
        if (it.parent >= 0) child[2 * it.parent + it.side] = gamma; else *root_out = gamma;
        const unsigned int pos = atomicAdd(next_count, 2u);
        next_items[pos] = Item{it.lo, gamma, gamma, 0, depth + 1};
        next_items[pos + 1] = Item{gamma + 1, it.hi, gamma, 1, depth + 1};
        return;
    }
    // ---- leaf: box of its triangles, then up the tree
    Box b = tri_box(tris[prim[it.lo]]);
    for (int p = it.lo + 1; p <= it.hi; ++p) {
        const Box t = tri_box(tris[prim[p]]);
        for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], t.mn[a]); b.mx[a] = fmaxf(b.mx[a], t.mx[a]); }
    }
    leaf_box[it.lo] = b;
    child[2 * it.parent + it.side] = ~(int)(((uint32_t)it.lo << 4) | count);
    int node = it.parent;
    while (node >= 0) {
        // release this thread's box and link, acquire the sibling's: the second thread to arrive carries on
        const unsigned int old = __hip_atomic_fetch_add(&arrived[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0u) return;
        Box u;
        for (int side = 0; side < 2; ++side) {
            const int link = child[2 * node + side];
            const Box c = link < 0 ? leaf_box[(~(uint32_t)link) >> 4] : node_box[link];
            for (int a = 0; a < 3; ++a) {
                u.mn[a] = side ? fminf(u.mn[a], c.mn[a]) : c.mn[a];
                u.mx[a] = side ? fmaxf(u.mx[a], c.mx[a]) : c.mx[a];
            }
        }
        node_box[node] = u;
        node = parent[node];
    }
}

// ---- binned-SAH splits below the top tree (round 3; the Morton splits of k_level are kept for A/B, TRH_LBVH_SPLIT=morton).
// The ranges the top tree leaves (one per Morton cluster, tens of triangles) and every range below them are split like the
// host builder splits: 16 bins per axis over the triangle centres, cost = half area x triangle count, first minimum wins;
// the depth guard of k_level applies to every candidate plane (both sides must still be finishable by medians), and a
// range without a valid plane is cut at its median position. One WAVE per range: bins in the wave's LDS block (atomics on
// ordered integers), 45 lanes evaluate the 45 planes, stable partition by ballot + prefix count from `prim_in` into
// `prim_out` (the arrays alternate between generations; a leaf copies its triangles to `prim_final`, the order the triangle
// records are emitted in). Node index = split position, as in k_level: unique, so the layout does not depend on the order
// of the work queue. Sides of <= kLeafMax triangles go to the leaf queue (k_leaves, after the last generation), larger ones
// to the next generation's.
struct WaveBins { uint32_t bb[3][16][6]; uint32_t num[3][16]; };

__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// the leaves of k_level_sah, one thread each, after the last generation (every parent link exists by then): box of the leaf's
// triangles, leaf link, then up the tree as far as this thread is the second to arrive at a node (the classic bottom-up fit).
// One launch for all leaves: the acquire / release pair per node costs an L2 write-back and invalidate on this chip, and
// a thread per leaf shares them between the 64 leaves of a wave (made by a lane or two of the wave that had split their parent,
// the leaves cost k_level_sah 48 ms for 1.6 M triangles; k_level_sah now takes 4.9 ms and this kernel 4.1: profiles/r03_lbvh.txt).
__global__ __launch_bounds__(kBlock) void k_leaves(const Item* __restrict__ leaves, const unsigned int* __restrict__ leaf_count, const Tri* __restrict__ tris,
                                                   const uint32_t* __restrict__ prim, const int* __restrict__ parent, int* child, Box* leaf_box, Box* node_box, unsigned int* arrived) {
    const unsigned int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= *leaf_count) return;
    const Item it = leaves[k];
    Box b = tri_box(tris[prim[it.lo]]);
    for (int p = it.lo + 1; p <= it.hi; ++p) {
        const Box t = tri_box(tris[prim[p]]);
        for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], t.mn[a]); b.mx[a] = fmaxf(b.mx[a], t.mx[a]); }
    }
    leaf_box[it.lo] = b;
    child[2 * it.parent + it.side] = ~(int)(((uint32_t)it.lo << 4) | (uint32_t)(it.hi - it.lo + 1));
    int node = it.parent;
    while (node >= 0) {
        // release this thread's box and link, acquire the sibling's: the second thread to arrive carries on
        const unsigned int old = __hip_atomic_fetch_add(&arrived[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0u) return;
        Box u;
        for (int sd = 0; sd < 2; ++sd) {
            const int link = child[2 * node + sd];
            const Box c = link < 0 ? leaf_box[(~(uint32_t)link) >> 4] : node_box[link];
            for (int a = 0; a < 3; ++a) {
                u.mn[a] = sd ? fminf(u.mn[a], c.mn[a]) : c.mn[a];
                u.mx[a] = sd ? fmaxf(u.mx[a], c.mx[a]) : c.mx[a];
            }
        }
        node_box[node] = u;
        node = parent[node];
    }
}

__global__ __launch_bounds__(kBlock) void k_level_sah(const Item* __restrict__ items, const unsigned int* __restrict__ item_count, Item* __restrict__ next_items,
                                                      unsigned int* __restrict__ next_count, Item* __restrict__ leaf_items, unsigned int* __restrict__ leaf_count, int depth_limit,
                                                      float cost_traverse, const Tri* __restrict__ tris, const uint32_t* prim_in, uint32_t* prim_out, uint32_t* prim_final, Stats* __restrict__ stats,
                                                      int* __restrict__ parent, int* __restrict__ child, uint32_t* __restrict__ live, int* __restrict__ root_out) {
    constexpr int B = 16;
    __shared__ WaveBins s_bins[kBlock / 64];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    WaveBins& wb = s_bins[wave];
    const unsigned int count = *item_count;
    const unsigned int total_waves = gridDim.x * (unsigned)(kBlock / 64);
    // ranges for the next generation and leaves, gathered in the wave (lane j holds the j-th; everything about a range is
    // wave-uniform) and appended to their queues with one atomic per ~60
    Item buf = Item{0, 0, 0, 0, 0}, lbuf = buf;
    int nbuf = 0, nlbuf = 0;   // wave-uniform
    auto flush = [&](Item& b, int& nb, Item* out, unsigned int* out_count) {
        if (nb == 0) return;
        unsigned int pos = 0;
        if (lane == 0) pos = atomicAdd(out_count, (unsigned int)nb);
        pos = (unsigned int)__shfl((int)pos, 0, 64);
        if (lane < nb) out[pos + (unsigned int)lane] = b;
        nb = 0;
    };
    for (unsigned int k = blockIdx.x * (unsigned)(kBlock / 64) + (unsigned)wave; k < count; k += total_waves) {
        const Item it = items[k];
        const int m = it.hi - it.lo + 1;
        if ((uint32_t)m <= kLeafMax) {   // a seed range as small as a leaf (generation 1: prim_in is the final order already)
            if (lane == nlbuf) lbuf = it;
            if (++nlbuf >= 62) flush(lbuf, nlbuf, leaf_items, leaf_count);
            continue;
        }
        // ---- bounds of the centres
        float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = it.lo + lane; i <= it.hi; i += 64) {
            const Box b = tri_box(tris[prim_in[i]]);
            for (int a = 0; a < 3; ++a) { const float c = 0.5f * (b.mn[a] + b.mx[a]); cmn[a] = fminf(cmn[a], c); cmx[a] = fmaxf(cmx[a], c); }
        }
        float ext[3], scale[3];
        for (int a = 0; a < 3; ++a) {
            for (int off = 32; off > 0; off >>= 1) { cmn[a] = fminf(cmn[a], __shfl_xor(cmn[a], off, 64)); cmx[a] = fmaxf(cmx[a], __shfl_xor(cmx[a], off, 64)); }
            ext[a] = cmx[a] - cmn[a];
            scale[a] = ext[a] > 0.f ? (float)B / ext[a] : 0.f;
        }
        auto bin_of = [&](const Box& b, int a) { const float c = 0.5f * (b.mn[a] + b.mx[a]); const int x = (int)((c - cmn[a]) * scale[a]); return x < 0 ? 0 : (x > B - 1 ? B - 1 : x); };
        // ---- bins
        wave_sync();   // (the previous range's readers are done)
        for (int i = lane; i < 3 * B * 6; i += 64) (&wb.bb[0][0][0])[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
        for (int i = lane; i < 3 * B; i += 64) (&wb.num[0][0])[i] = 0u;
        wave_sync();
        for (int i = it.lo + lane; i <= it.hi; i += 64) {
            const Box b = tri_box(tris[prim_in[i]]);
            for (int a = 0; a < 3; ++a) {
                if (!(ext[a] > 0.f)) continue;
                const int x = bin_of(b, a);
                for (int d = 0; d < 3; ++d) { atomicMin(&wb.bb[a][x][d], float_to_ordered(b.mn[d])); atomicMax(&wb.bb[a][x][3 + d], float_to_ordered(b.mx[d])); }
                atomicAdd(&wb.num[a][x], 1u);
            }
        }
        wave_sync();
        // ---- 45 lanes, 45 planes: lane = axis * 15 + bin, the plane after that bin
        float cost = FLT_MAX, my_area = 0.f;
        int my_nl = 0;
        if (lane < 3 * (B - 1)) {
            const int a = lane / (B - 1), pb = lane % (B - 1);
            if (ext[a] > 0.f) {
                Box l, r;
                for (int x = 0; x < 3; ++x) { l.mn[x] = r.mn[x] = FLT_MAX; l.mx[x] = r.mx[x] = -FLT_MAX; }
                uint32_t nl = 0, nr = 0;
                for (int b = 0; b < B; ++b) {
                    const uint32_t nb = wb.num[a][b];
                    if (!nb) continue;
                    Box& t = b <= pb ? l : r;
                    for (int x = 0; x < 3; ++x) { t.mn[x] = fminf(t.mn[x], ordered_to_float(wb.bb[a][b][x])); t.mx[x] = fmaxf(t.mx[x], ordered_to_float(wb.bb[a][b][3 + x])); }
                    if (b <= pb) nl += nb; else nr += nb;
                }
                if (nl != 0u && nr != 0u) {
                    const int need = levels_needed(nl) > levels_needed(nr) ? levels_needed(nl) : levels_needed(nr);
                    if (it.depth + need <= depth_limit) {
                        cost = half_area(l) * (float)nl + half_area(r) * (float)nr;
                        my_nl = (int)nl;
                        for (int x = 0; x < 3; ++x) { l.mn[x] = fminf(l.mn[x], r.mn[x]); l.mx[x] = fmaxf(l.mx[x], r.mx[x]); }
                        my_area = half_area(l);   // the range's own box
                    }
                }
            }
        }
        // first minimum in (axis, bin) order = lowest lane among the cheapest
        float best = cost;
        int best_lane = lane;
        for (int off = 32; off > 0; off >>= 1) {
            const float oc = __shfl_xor(best, off, 64);
            const int ol = __shfl_xor(best_lane, off, 64);
            if (oc < best || (oc == best && ol < best_lane)) { best = oc; best_lane = ol; }
        }
        const bool planar = best < FLT_MAX;
        const int axis = planar ? best_lane / (B - 1) : -1, pbin = planar ? best_lane % (B - 1) : 0;
        const int nl = planar ? __shfl(my_nl, best_lane, 64) : m / 2;   // no plane (equal centres, or the depth guard): the median position
        // the host builder's leaf rule for small ranges
        if ((uint32_t)m <= kSahLeafMax) {
            const float area = planar ? __shfl(my_area, best_lane, 64) : 0.f;
            if (!planar || kCostIntersect * (float)m <= cost_traverse + kCostIntersect * best / fmaxf(area, 1e-30f)) {
                for (int i = it.lo + lane; i <= it.hi; i += 64) prim_final[i] = prim_in[i];
                if (lane == nlbuf) lbuf = it;
                if (++nlbuf >= 62) flush(lbuf, nlbuf, leaf_items, leaf_count);
                continue;
            }
        }
        // ---- stable partition
        int run_l = 0, run_r = 0;
        for (int base = it.lo; base <= it.hi; base += 64) {
            const int i = base + lane;
            const bool in = i <= it.hi;
            uint32_t t = 0;
            bool left = false;
            if (in) {
                t = prim_in[i];
                left = planar ? bin_of(tri_box(tris[t]), axis) <= pbin : (i - it.lo) < nl;
            }
            const unsigned long long ml = __ballot(in && left);
            const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
            const int my_l = (int)__popcll(ml & below);
            if (in) {
                const int pos = left ? it.lo + run_l + my_l : it.lo + nl + run_r + (lane - my_l);
                prim_out[pos] = t;
                if ((uint32_t)(left ? nl : m - nl) <= kLeafMax) prim_final[pos] = t;   // a side that is a leaf: this is its final order
            }
            const int chunk = it.hi - base + 1 < 64 ? it.hi - base + 1 : 64;
            run_l += (int)__popcll(ml);
            run_r += chunk - (int)__popcll(ml);
        }
        // ---- the node, its sides
        const int gamma = it.lo + nl - 1;
        if (lane == 0) {
            parent[gamma] = it.parent;
            live[gamma] = 1u;
            if ((uint32_t)it.depth > __hip_atomic_load(&stats->max_depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats->max_depth, (uint32_t)it.depth);
            if (it.parent >= 0) child[2 * it.parent + it.side] = gamma; else *root_out = gamma;
        }
        // the sides: leaves to the leaf queue (k_leaves), the others to the next generation
        if ((uint32_t)nl > kLeafMax) { if (lane == nbuf) buf = Item{it.lo, gamma, gamma, 0, it.depth + 1}; ++nbuf; }
        else { if (lane == nlbuf) lbuf = Item{it.lo, gamma, gamma, 0, it.depth + 1}; ++nlbuf; }
        if ((uint32_t)(m - nl) > kLeafMax) { if (lane == nbuf) buf = Item{gamma + 1, it.hi, gamma, 1, it.depth + 1}; ++nbuf; }
        else { if (lane == nlbuf) lbuf = Item{gamma + 1, it.hi, gamma, 1, it.depth + 1}; ++nlbuf; }
        if (nbuf >= 62) flush(buf, nbuf, next_items, next_count);
        if (nlbuf >= 62) flush(lbuf, nlbuf, leaf_items, leaf_count);
    }
    flush(buf, nbuf, next_items, next_count);
    flush(lbuf, nlbuf, leaf_items, leaf_count);
}

__device__ inline void pad_box(Box& b, float scene_extent) {   // == Builder::pad (bvh_builder.cpp)
    for (int a = 0; a < 3; ++a) {
        const float m = fmaxf(fmaxf(fabsf(b.mn[a]), fabsf(b.mx[a])), b.mx[a] - b.mn[a]);
        const float d = 1e-5f * m + 2e-6f * scene_extent + 1e-7f;
        b.mn[a] -= d;
        b.mx[a] += d;
    }
}
__device__ inline uint32_t pack_axis(float mn, float mx, float c, double s) {
    return (uint32_t)bvh::half_floor(((double)mn - (double)c) * s) | ((uint32_t)bvh::half_ceil(((double)mx - (double)c) * s) << 16);
}
__device__ inline double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(kBlock) void k_emit(int n_inner, const int* __restrict__ child, const uint32_t* __restrict__ live, const uint32_t* __restrict__ new_idx,
                                                 uint32_t root_idx, const Box* __restrict__ leaf_box, const Box* __restrict__ node_box, Frame f,
                                                 float* __restrict__ nodes, uint32_t* __restrict__ nodes16, Stats* __restrict__ stats) {
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    double area_exact = 0.0, area_quant = 0.0, sah = 0.0;
    uint32_t max_leaf = 0;
    // compacted index with the root moved to record 0 (the kernels start there)
    auto record_of = [&](int node) { const uint32_t x = new_idx[node]; return x == root_idx ? 0u : (x < root_idx ? x + 1u : x); };
    if (i < n_inner && live[i]) {
        Box b[2];
        int link[2];
        sah += (double)(half_area(node_box[i]) * f.root_area_inv * f.cost_traverse);
        for (int k = 0; k < 2; ++k) {
            const int c = child[2 * i + k];
            if (c < 0) {                          // <= kLeafMax triangles, contiguous in sorted order
                const uint32_t bits = ~(uint32_t)c, count = bits & 15u;
                b[k] = leaf_box[bits >> 4];
                link[k] = c;
                sah += (double)(half_area(b[k]) * f.root_area_inv * kCostIntersect * (float)count);
                max_leaf = max_leaf > count ? max_leaf : count;
            } else {
                b[k] = node_box[c];
                link[k] = (int)record_of(c);
            }
        pad_box(b[k], f.scene_extent);
        }
        const uint32_t me = record_of(i);
        float* r = nodes + (size_t)me * 16;
        r[0] = b[0].mn[0]; r[1] = b[0].mx[0]; r[2] = b[0].mn[1]; r[3] = b[0].mx[1];
        r[4] = b[1].mn[0]; r[5] = b[1].mx[0]; r[6] = b[1].mn[1]; r[7] = b[1].mx[1];
        r[8] = b[0].mn[2]; r[9] = b[0].mx[2]; r[10] = b[1].mn[2]; r[11] = b[1].mx[2];
        r[12] = __int_as_float(link[0]);
        r[13] = __int_as_float(link[1]);
        r[14] = 0.f;
        r[15] = 0.f;
        uint32_t* q = nodes16 + (size_t)me * 8;
        q[0] = pack_axis(b[0].mn[0], b[0].mx[0], f.frame_c[0], f.frame_s);
        q[1] = pack_axis(b[0].mn[1], b[0].mx[1], f.frame_c[1], f.frame_s);
        q[2] = pack_axis(b[1].mn[0], b[1].mx[0], f.frame_c[0], f.frame_s);
        q[3] = pack_axis(b[1].mn[1], b[1].mx[1], f.frame_c[1], f.frame_s);
        q[4] = pack_axis(b[0].mn[2], b[0].mx[2], f.frame_c[2], f.frame_s);
        q[5] = pack_axis(b[1].mn[2], b[1].mx[2], f.frame_c[2], f.frame_s);
        q[6] = (uint32_t)link[0];
        q[7] = (uint32_t)link[1];
        // fp16 inflation measure, as Builder::quantise
        const int qx[2][3] = {{0, 1, 4}, {2, 3, 5}};
        for (int k = 0; k < 2; ++k) {
            area_exact += (double)(b[k].mx[0] - b[k].mn[0]) * (double)(b[k].mx[1] - b[k].mn[1]) + (double)(b[k].mx[1] - b[k].mn[1]) * (double)(b[k].mx[2] - b[k].mn[2]) +
                          (double)(b[k].mx[2] - b[k].mn[2]) * (double)(b[k].mx[0] - b[k].mn[0]);
            double e[3];
            for (int a = 0; a < 3; ++a)
                e[a] = ((double)bvh::half_to_float((uint16_t)(q[qx[k][a]] >> 16)) - (double)bvh::half_to_float((uint16_t)(q[qx[k][a]] & 0xffffu))) / f.frame_s;
            area_quant += e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
        }
    }
    area_exact = wave_sum(area_exact);
    area_quant = wave_sum(area_quant);
    sah = wave_sum(sah);
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)max_leaf, off, 64);
        max_leaf = max_leaf > o ? max_leaf : o;
    }
    if ((threadIdx.x & 63) == 0) {
        if (area_exact != 0.0) atomicAdd(&stats->area_exact, area_exact);
        if (area_quant != 0.0) atomicAdd(&stats->area_quant, area_quant);
        if (sah != 0.0) atomicAdd(&stats->sah, sah);
        if (max_leaf) atomicMax(&stats->max_leaf, max_leaf);
    }
}

__global__ __launch_bounds__(kBlock) void k_triangles(const Tri* __restrict__ tris, const uint32_t* __restrict__ vals, uint32_t n, float* __restrict__ out) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const Tri t = tris[vals[p]];
    float* r = out + (size_t)p * 12;
    for (int a = 0; a < 3; ++a) {
        r[a] = t.v0[a];
        r[4 + a] = t.v1[a] - t.v0[a];   // same subtraction RayTriangle does (collision.h:35-36)
        r[8 + a] = t.v2[a] - t.v0[a];
    }
    r[3] = __uint_as_float(t.id);
    r[7] = 0.f;
    r[11] = 0.f;
}

// One arena for every temporary (a hipMalloc costs more than most of these kernels), freed on every return path;
// the arrays that outlive the build are separate allocations.
struct Scratch {
    char* base = nullptr;
    size_t size = 0, used = 0;
    std::vector<void*> outputs;
    bool failed = false;
    ~Scratch() {
        (void)hipFree(base);
        for (void* p : outputs) (void)hipFree(p);
    }
    hipError_t reserve(size_t bytes) {
        size = bytes;
        return hipMalloc((void**)&base, bytes);
    }
    template <class T>
    hipError_t alloc(T** out, size_t count) {
        used = (used + 255) & ~(size_t)255;
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        if (used + bytes > size) { *out = nullptr; return hipErrorOutOfMemory; }
        *out = (T*)(base + used);
        used += bytes;
        return hipSuccess;
    }
    template <class T>
    hipError_t alloc_output(T** out, size_t count) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) outputs.push_back(p);
        *out = (T*)p;
        return e;
    }
    void keep(void* p) { outputs.erase(std::remove(outputs.begin(), outputs.end(), p), outputs.end()); }
};

}  // namespace

int build(const std::vector<bvh::InputTriangle>& tris, hipStream_t stream, DeviceBvh& out, std::string& err) {
    out = DeviceBvh{};
    const size_t n = tris.size();
    if (n < kMinTriangles) { err = "lbvh::build: too few triangles (use the host builder)"; return 1; }
    if (n >= (1u << 27)) { err = "lbvh::build: more than 2^27 triangles"; return 1; }
    const int ni = (int)n, n_inner = ni - 1;
    const unsigned grid_n = (unsigned)((n + kBlock - 1) / kBlock), grid_i = (unsigned)((n_inner + kBlock - 1) / kBlock);
    Scratch tmp;
    size_t sort_bytes = 0, scan_bytes = 0;
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, ni, 0, 63, stream));
    LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, n_inner, stream));
    size_t scan_bytes_n = 0;
    LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes_n, (uint32_t*)nullptr, (uint32_t*)nullptr, ni, stream));
    // triangles 40, keys 2 x 8, values 4, parent 4, child links 8, boxes 2 x 24, counters 3 x 4, work queues 2 x 16 bytes per triangle
    LB_TRY(tmp.reserve(n * 344 + sort_bytes + scan_bytes + scan_bytes_n + (5u << 20)));
    Tri* d_tris = nullptr;
    LB_TRY(tmp.alloc(&d_tris, n));
    LB_TRY(hipMemcpyAsync(d_tris, tris.data(), n * sizeof(Tri), hipMemcpyHostToDevice, stream));
    hipEvent_t e0, e1;
    LB_TRY(hipEventCreate(&e0));
    LB_TRY(hipEventCreate(&e1));
    struct EventGuard { hipEvent_t a, b; ~EventGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } guard{e0, e1};
    LB_TRY(hipEventRecord(e0, stream));

    // ---- scene box and centroid box
    uint32_t* d_ord = nullptr;
    LB_TRY(tmp.alloc(&d_ord, 12));
    uint32_t ord[12];
    for (int k = 0; k < 12; ++k) ord[k] = (k % 6) < 3 ? 0xffffffffu : 0u;
    LB_TRY(hipMemcpyAsync(d_ord, ord, sizeof(ord), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_bounds, dim3(std::min(grid_n, 2048u)), dim3(kBlock), 0, stream, d_tris, (uint32_t)n, d_ord);
    LB_TRY(hipGetLastError());
    LB_TRY(hipMemcpyAsync(ord, d_ord, sizeof(ord), hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    float b[12];
    for (int k = 0; k < 12; ++k) b[k] = ordered_to_float(ord[k]);
    Frame f{};
    float extent = 0.f;
    for (int a = 0; a < 3; ++a) {
        if (!(std::fabs(b[a]) < 1e30f) || !(std::fabs(b[3 + a]) < 1e30f)) { err = "lbvh::build: non-finite vertex"; return 1; }
        extent = std::max(extent, std::max(b[3 + a] - b[a], std::max(std::fabs(b[a]), std::fabs(b[3 + a]))));
        const float ce = b[9 + a] - b[6 + a];
        f.cmn[a] = b[6 + a];
        f.cscale[a] = ce > 0.f ? 2097152.f / ce : 0.f;
    }
    f.scene_extent = extent;
    {
        const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
        f.root_area_inv = 1.f / std::max(dx * dy + dy * dz + dz * dx, 1e-30f);
        f.cost_traverse = bvh::cost_traverse();
    }

    // ---- Morton codes, sort
    uint64_t *d_keys = nullptr, *d_keys_sorted = nullptr;
    uint32_t *d_vals = nullptr, *d_prim = nullptr;
    LB_TRY(tmp.alloc(&d_keys, n));
    LB_TRY(tmp.alloc(&d_keys_sorted, n));
    LB_TRY(tmp.alloc(&d_vals, n));
    LB_TRY(tmp.alloc_output(&d_prim, n));
    hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(kBlock), 0, stream, d_tris, (uint32_t)n, f, d_keys, d_vals);
    LB_TRY(hipGetLastError());
    unsigned char* d_sort_tmp = nullptr;
    LB_TRY(tmp.alloc(&d_sort_tmp, sort_bytes));
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, sort_bytes, d_keys, d_keys_sorted, d_vals, d_prim, ni, 0, 63, stream));

    // ---- hierarchy and boxes, one launch per level
    int *d_parent = nullptr, *d_child = nullptr, *d_root = nullptr;
    Box *d_leaf_box = nullptr, *d_node_box = nullptr;
    unsigned int* d_arrived = nullptr;
    uint32_t *d_live = nullptr, *d_new_idx = nullptr;
    Item *d_items_a = nullptr, *d_items_b = nullptr;
    LB_TRY(tmp.alloc(&d_parent, (size_t)n_inner));
    LB_TRY(tmp.alloc(&d_child, (size_t)n_inner * 2));
    LB_TRY(tmp.alloc(&d_root, 1));
    LB_TRY(tmp.alloc(&d_leaf_box, n));
    LB_TRY(tmp.alloc(&d_node_box, (size_t)n_inner));
    LB_TRY(tmp.alloc(&d_arrived, (size_t)n_inner));
    LB_TRY(tmp.alloc(&d_live, (size_t)n_inner));
    LB_TRY(tmp.alloc(&d_new_idx, (size_t)n_inner));
    LB_TRY(tmp.alloc(&d_items_a, n));
    LB_TRY(tmp.alloc(&d_items_b, n));
    LB_TRY(hipMemsetAsync(d_arrived, 0, (size_t)n_inner * sizeof(unsigned int), stream));
    LB_TRY(hipMemsetAsync(d_live, 0, (size_t)n_inner * sizeof(uint32_t), stream));
    // (the depth limit, moved up from the level loop: the seeding needs it)
    // The depth of the tree is the number of LDS stack rows the render kernels need: above 30 rows a CU holds four
    // workgroups instead of five. The guard therefore aims for kDepthAim levels as long as that leaves a median tree of
    // this many triangles ten levels of slack (dragon 100 k: depth 31 -> 27, render 14.8 -> 13.2 ms; with less slack the
    // forced median splits cost more than the fifth workgroup brings: 1.6 M triangles at 27 levels 22.4 ms against 20.5
    // at 29-32). Development override: TRH_LBVH_DEPTH.
    int depth_limit = std::min(bvh::kMaxDepth, std::max(kDepthAim, levels_needed((uint32_t)n) + 10));
    if (const char* e = getenv("TRH_LBVH_DEPTH")) depth_limit = std::max(levels_needed((uint32_t)n) + 1, std::min(bvh::kMaxDepth, atoi(e)));

    // (statistics and the per-generation queue lengths: the device-side top tree already writes to both)
    Stats* d_stats = nullptr;
    LB_TRY(tmp.alloc(&d_stats, 1));
    LB_TRY(hipMemsetAsync(d_stats, 0, sizeof(Stats), stream));
    unsigned int* d_counts = nullptr;
    LB_TRY(tmp.alloc(&d_counts, (size_t)bvh::kMaxDepth + 3));
    LB_TRY(hipMemsetAsync(d_counts, 0, ((size_t)bvh::kMaxDepth + 3) * sizeof(unsigned int), stream));
    // ---- seeds: the whole range under no parent (plain linear BVH), or one range per Morton cluster under an SAH top tree
    size_t n_seeds = 1;
    bool seeded = false;
    const char* top_env = getenv("TRH_LBVH_TOP");   // development override: 0 = plain linear BVH
    if (!(top_env && atoi(top_env) == 0) && n >= 4096) {
        // cluster = shared top `bits` Morton bits (a cube of the centroid grid). Round 2 (top tree by one host thread): about 8
        // triangles each, at most 2^15 cubes. With the top tree on the device the cubes can be as fine as two triangles each
        // (at most 2^21 cubes): the top tree is then nearly the whole tree, i.e. binned SAH over boxes of triangle pairs --
        // dragon 8.93 -> 8.64 ms per 64 spp (host SAH tree: 8.36), SAH cost 11.6 -> 10.5 (10.3), profiles/r03_lbvh.txt
        int bits = 6;
        size_t per_cluster = 2;
        if (const char* e = getenv("TRH_LBVH_CLUSTER")) per_cluster = (size_t)std::max(1, atoi(e));   // development knob
        int max_bits = 21;
        if (const char* e = getenv("TRH_LBVH_MAXBITS")) max_bits = std::min(24, std::max(6, atoi(e)));   // development knob
        while (bits < max_bits && ((size_t)1 << bits) * per_cluster < n) bits += 3;
        uint32_t *d_flags = nullptr, *d_cidx = nullptr, *d_starts = nullptr;
        unsigned char* d_scan2 = nullptr;
        Box* d_cboxes = nullptr;
        LB_TRY(tmp.alloc(&d_flags, n));
        LB_TRY(tmp.alloc(&d_cidx, n));
        LB_TRY(tmp.alloc(&d_scan2, scan_bytes_n));
        hipLaunchKernelGGL(k_cluster_flags, dim3(grid_n), dim3(kBlock), 0, stream, d_keys_sorted, (uint32_t)n, 63 - bits, d_flags);
        LB_TRY(hipGetLastError());
        LB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan2, scan_bytes_n, d_flags, d_cidx, ni, stream));
        // cluster starts and boxes for the upper bound of their number, then the one small read-back of the seeding (12 bytes):
        // how many clusters there are and how big the biggest is decide the launch sizes and the depth budget below
        const size_t cluster_cap = std::min<size_t>(n, (size_t)1 << bits);
        uint32_t* d_biggest = nullptr;
        LB_TRY(tmp.alloc(&d_starts, cluster_cap));
        LB_TRY(tmp.alloc(&d_cboxes, cluster_cap));
        LB_TRY(tmp.alloc(&d_biggest, 1));
        LB_TRY(hipMemsetAsync(d_biggest, 0, 4, stream));
        hipLaunchKernelGGL(k_cluster_starts, dim3(grid_n), dim3(kBlock), 0, stream, d_flags, d_cidx, (uint32_t)n, d_starts);
        LB_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_cluster_boxes, dim3((unsigned)((cluster_cap + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_tris, d_prim, d_starts, d_flags, d_cidx, (uint32_t)n, d_cboxes, d_biggest);
        LB_TRY(hipGetLastError());
        uint32_t last_flag = 0, last_cidx = 0, biggest = 0;
        LB_TRY(hipMemcpyAsync(&last_flag, d_flags + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        LB_TRY(hipMemcpyAsync(&last_cidx, d_cidx + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        LB_TRY(hipMemcpyAsync(&biggest, d_biggest, 4, hipMemcpyDeviceToHost, stream));
        LB_TRY(hipStreamSynchronize(stream));
        const uint32_t n_clusters = last_cidx + last_flag;
        if (n_clusters >= 2 && n_clusters <= cluster_cap) {
            // ---- the top tree on the device (k_top_level): one launch per level, no data to the host and back
            // a cluster's root record at depth D needs D + levels_needed(count) <= depth_limit to be finishable by medians
            const int max_seed_depth = depth_limit - levels_needed(biggest);
            int clog = 0;
            while ((1u << clog) < n_clusters) ++clog;
            if (biggest >= 1 && 1 + clog <= max_seed_depth) {
                TopItem *d_top_a = nullptr, *d_top_b = nullptr;
                int *d_order_a = nullptr, *d_order_b = nullptr;
                unsigned int* d_top_counts = nullptr;   // [level]: items of the level; level 1 = the root
                LB_TRY(tmp.alloc(&d_top_a, n_clusters));
                LB_TRY(tmp.alloc(&d_top_b, n_clusters));
                LB_TRY(tmp.alloc(&d_order_a, n_clusters));
                LB_TRY(tmp.alloc(&d_order_b, n_clusters));
                LB_TRY(tmp.alloc(&d_top_counts, (size_t)bvh::kMaxDepth + 3));
                LB_TRY(hipMemsetAsync(d_top_counts, 0, ((size_t)bvh::kMaxDepth + 3) * sizeof(unsigned int), stream));
                const TopItem root_top{0, (int)n_clusters, -1, 0, 1};
                const unsigned int one = 1u;
                LB_TRY(hipMemcpyAsync(d_top_a, &root_top, sizeof(root_top), hipMemcpyHostToDevice, stream));
                LB_TRY(hipMemcpyAsync(d_top_counts + 1, &one, sizeof(one), hipMemcpyHostToDevice, stream));
                hipLaunchKernelGGL(k_iota, dim3((n_clusters + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, d_order_a, (int)n_clusters);
                LB_TRY(hipGetLastError());
                for (int level = 1; level < max_seed_depth; ++level) {
                    // nodes of a level: at most 2^(level-1), at most one per two clusters
                    const size_t bound = std::min<size_t>(level > 20 ? n_clusters : ((size_t)1 << (level - 1)), (size_t)n_clusters / 2 + 1);
                    hipLaunchKernelGGL(k_top_level, dim3((unsigned)bound), dim3(kBlock), 0, stream, d_top_a, d_top_counts + level, d_top_b, d_top_counts + level + 1, d_order_a, d_order_b,
                                       d_cboxes, d_starts, n_clusters, (uint32_t)n, max_seed_depth, d_items_a, d_counts + 1, d_stats, d_parent, d_child, d_live, d_root);
                    LB_TRY(hipGetLastError());
                    std::swap(d_top_a, d_top_b);
                    std::swap(d_order_a, d_order_b);
                }
                n_seeds = n_clusters;   // every cluster ends as a seed (the guard leaves room for the medians that get it there)
                seeded = true;
            }
        }
    }
    if (!seeded) {
        const Item root_item{0, ni - 1, -1, 0, 1};
        LB_TRY(hipMemcpyAsync(d_items_a, &root_item, sizeof(Item), hipMemcpyHostToDevice, stream));
    }
    // Every level is launched with an upper bound of its queue length (2^(depth-1), at most n ranges) and reads the
    // real length from the device: no host round trip per level. The depth guard ends the tree at kMaxDepth records;
    // level kMaxDepth + 1 only holds leaves.
    const unsigned int first_count = (unsigned int)n_seeds;
    LB_TRY(hipMemcpyAsync(d_counts + 1, &first_count, sizeof(first_count), hipMemcpyHostToDevice, stream));
    const char* split_env = getenv("TRH_LBVH_SPLIT");   // development override: "morton" = the linear BVH's splits (k_level) below the top tree
    const bool sah_splits = !(split_env && std::string(split_env) == "morton");
    uint32_t *d_prim_a = nullptr, *d_prim_b = nullptr;
    Item* d_leaf_items = nullptr;   // every leaf of k_level_sah; its length is d_counts[0]
    if (sah_splits) {
        LB_TRY(tmp.alloc(&d_prim_a, n));
        LB_TRY(tmp.alloc(&d_prim_b, n));
        LB_TRY(tmp.alloc(&d_leaf_items, n));
    }
    const uint32_t* prim_in = d_prim;   // generation 1 reads the Morton order; a leaf writes its final order back into d_prim
    uint32_t* prim_out = d_prim_a;
    for (int depth = 1; depth <= bvh::kMaxDepth + 1; ++depth) {
        // (generation `depth` of the work queue: with seeds the records of one generation sit at different tree depths)
        const size_t bound = depth > 27 ? n : std::min<size_t>(n, n_seeds << (depth - 1));
        if (sah_splits) {
            // one wave per range, waves stride over the queue (its real length is read on the device)
            const size_t blocks = std::min<size_t>((bound + (kBlock / 64) - 1) / (kBlock / 64), 4096);
            hipLaunchKernelGGL(k_level_sah, dim3((unsigned)blocks), dim3(kBlock), 0, stream, d_items_a, d_counts + depth, d_items_b, d_counts + depth + 1, d_leaf_items, d_counts, depth_limit,
                               f.cost_traverse, d_tris, prim_in, prim_out, d_prim, d_stats, d_parent, d_child, d_live, d_root);
            prim_in = prim_out;
            prim_out = prim_out == d_prim_a ? d_prim_b : d_prim_a;
        } else {
            hipLaunchKernelGGL(k_level, dim3((unsigned)((bound + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_items_a, (int)bound, depth_limit, d_keys_sorted, d_tris, d_prim,
                               d_items_b, d_counts + depth, d_counts + depth + 1, d_stats, d_parent, d_child, d_live, d_leaf_box, d_node_box, d_arrived, d_root);
        }
        LB_TRY(hipGetLastError());
        std::swap(d_items_a, d_items_b);
    }
    if (sah_splits) {
        hipLaunchKernelGGL(k_leaves, dim3(grid_n), dim3(kBlock), 0, stream, d_leaf_items, d_counts, d_tris, d_prim, d_parent, d_child, d_leaf_box, d_node_box, d_arrived);
        LB_TRY(hipGetLastError());
    }
    unsigned char* d_scan_tmp = nullptr;
    LB_TRY(tmp.alloc(&d_scan_tmp, scan_bytes));
    LB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp, scan_bytes, d_live, d_new_idx, n_inner, stream));
    uint32_t last_live = 0, last_idx = 0, mid_prim = 0, root_idx = 0;
    int root_node = -1;
    LB_TRY(hipMemcpyAsync(&last_live, d_live + (n_inner - 1), 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(&last_idx, d_new_idx + (n_inner - 1), 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(&mid_prim, d_prim + n / 2, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(&root_node, d_root, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    const uint32_t node_count = last_idx + last_live;
    if (node_count == 0 || mid_prim >= n || root_node < 0 || root_node >= n_inner) { err = "lbvh::build: inconsistent hierarchy"; return 1; }
    LB_TRY(hipMemcpyAsync(&root_idx, d_new_idx + root_node, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));

    // ---- frame of the fp16 records: centre = centroid of the middle triangle in Morton order (a point inside
    // the bulk of the geometry), scale = the power of two that maps the farthest padded bound to [16384, 32768)
    double far = 0.0;
    {
        const bvh::InputTriangle& t = tris[mid_prim];
        for (int a = 0; a < 3; ++a) {
            const float lo = std::min(t.v0[a], std::min(t.v1[a], t.v2[a])), hi = std::max(t.v0[a], std::max(t.v1[a], t.v2[a]));
            f.frame_c[a] = 0.5f * (lo + hi);
            const float m = std::max(std::max(std::fabs(b[a]), std::fabs(b[3 + a])), b[3 + a] - b[a]);
            const float d = 1e-5f * m + 2e-6f * extent + 1e-7f;   // the padding of the scene box bounds every record's
            far = std::max(far, std::max(std::fabs((double)(b[a] - d) - (double)f.frame_c[a]), std::fabs((double)(b[3 + a] + d) - (double)f.frame_c[a])));
        }
    }
    int k2 = 0;
    if (far > 0.0) {
        int ex;
        (void)std::frexp(far, &ex);
        k2 = 15 - ex;
    }
    k2 = std::min(60, std::max(-60, k2));
    f.frame_s = std::ldexp(1.0, k2);

    // ---- emit
    float* d_nodes = nullptr;
    uint32_t* d_nodes16 = nullptr;
    float* d_out_tris = nullptr;
    LB_TRY(tmp.alloc_output(&d_nodes, (size_t)node_count * 16));
    LB_TRY(tmp.alloc_output(&d_nodes16, (size_t)node_count * 8));
    LB_TRY(tmp.alloc_output(&d_out_tris, n * 12));
    hipLaunchKernelGGL(k_emit, dim3(grid_i), dim3(kBlock), 0, stream, n_inner, d_child, d_live, d_new_idx, root_idx, d_leaf_box, d_node_box, f, d_nodes, d_nodes16, d_stats);
    LB_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_triangles, dim3(grid_n), dim3(kBlock), 0, stream, d_tris, d_prim, (uint32_t)n, d_out_tris);
    LB_TRY(hipGetLastError());
    Stats st{};
    unsigned int unprocessed = 0;   // ranges queued below the last level: must be none (the depth guard)
    LB_TRY(hipMemcpyAsync(&st, d_stats, sizeof(st), hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(&unprocessed, d_counts + bvh::kMaxDepth + 2, sizeof(unprocessed), hipMemcpyDeviceToHost, stream));
    LB_TRY(hipEventRecord(e1, stream));
    LB_TRY(hipStreamSynchronize(stream));
    float ms = 0.f;
    LB_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (unprocessed != 0 || st.max_depth > (uint32_t)bvh::kMaxDepth) { err = "lbvh::build: tree deeper than the traversal stack"; return 2; }

    tmp.keep(d_nodes);
    tmp.keep(d_nodes16);
    tmp.keep(d_out_tris);
    tmp.keep(d_prim);
    out.nodes = d_nodes;
    out.nodes16 = d_nodes16;
    out.tris = d_out_tris;
    out.prim_index = d_prim;
    out.node_count = node_count;
    out.tri_count = (uint32_t)n;
    out.max_depth = st.max_depth;
    out.max_leaf = st.max_leaf;
    memcpy(out.frame_center, f.frame_c, sizeof(out.frame_center));
    out.frame_scale = (float)f.frame_s;
    out.fp16_inflation = st.area_exact > 0.0 ? st.area_quant / st.area_exact : 1.0;
    out.sah_cost = st.sah;
    out.device_ms = ms;
    return 0;
}

int download(const DeviceBvh& d, hipStream_t stream, bvh::Result& out, std::string& err) {
    out = bvh::Result{};
    out.nodes.resize((size_t)d.node_count * 16);
    out.nodes16.resize((size_t)d.node_count * 8);
    out.tris.resize((size_t)d.tri_count * 12);
    out.prim_index.resize(d.tri_count);
    LB_TRY(hipMemcpyAsync(out.nodes.data(), d.nodes, out.nodes.size() * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.nodes16.data(), d.nodes16, out.nodes16.size() * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.tris.data(), d.tris, out.tris.size() * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.prim_index.data(), d.prim_index, out.prim_index.size() * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    memcpy(out.frame_center, d.frame_center, sizeof(out.frame_center));
    out.frame_scale = d.frame_scale;
    out.fp16_inflation = d.fp16_inflation;
    out.max_depth = d.max_depth;
    out.max_leaf = d.max_leaf;
    out.sah_cost = d.sah_cost;
    return 0;
}

void release(DeviceBvh& d) {
    (void)hipFree(d.nodes);
    (void)hipFree(d.nodes16);
    (void)hipFree(d.tris);
    (void)hipFree(d.prim_index);
    d = DeviceBvh{};
}

}  // namespace lbvh
