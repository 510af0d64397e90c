// lbvh_build.hip -- linear BVH built on the GPU (gfx950); see lbvh_build.h for the what and why.
//
// Stages (all on one stream; a read-back of a few bytes after the bounds, per tree level and at the end):
//   k_bounds      scene box + centroid box: wave reduction, then ordered-integer atomic min / max
//   k_morton      63-bit Morton code of every centroid (21 bits per axis inside the centroid box)
//   radix sort    hipcub::DeviceRadixSort::SortPairs (code, triangle) -- stable, so equal codes keep input order
//   clusters      (>= 4096 triangles) ranges that share their top 6..15 code bits; their boxes go to the host, which builds
//                 the UPPER levels with binned SAH over the cluster boxes (TopBuilder) and seeds one range per cluster
//   k_level       top-down over the sorted codes, one launch per tree level (<= 32): a range of > 2 triangles (kLeafMax) is
//                 split where the highest differing code bit flips (the LBVH split), or at its median when
//                 the codes are equal or the remaining depth budget demands it (the depth guard of the host
//                 builder: a Morton tree of a scene with one dense object in a large room is otherwise 40+
//                 levels deep); the inner node is stored at its split position, which is unique, so the
//                 layout does not depend on the order the work queue fills. A range of <= 2 triangles is a
//                 leaf: its thread computes the box and walks up; the second thread to arrive at a node
//                 merges the child boxes (acquire/release counter per node)
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
constexpr int kLinkUnknown = (int)0x80000000;   // k_scatter_top: a child link the linear build fills in
constexpr int kDepthAim = 27;             // see build(): the depth the guard aims for
// triangles per leaf (the link format allows bvh::kMaxLeafSize = 8). Swept with the SAH top tree in place (64 spp per launch,
// dragon / spheres / dragon 400 k): 1: 12.00 / 7.47 / 16.0 ms, 2: 11.54 / 7.18 / 13.8, 3: 11.66 / 7.28, 4: 11.91 / 7.52 / 15.5, 6: 13.02 / 7.92
#ifndef TRH_LBVH_LEAF
#define TRH_LBVH_LEAF 2
#endif
constexpr uint32_t kLeafMax = TRH_LBVH_LEAF;
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
__global__ __launch_bounds__(kBlock) void k_cluster_boxes(const Tri* __restrict__ tris, const uint32_t* __restrict__ prim, const uint32_t* __restrict__ starts, uint32_t n_clusters,
                                                          uint32_t n, Box* __restrict__ boxes) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= n_clusters) return;
    const uint32_t lo = starts[c], hi = c + 1u < n_clusters ? starts[c + 1u] : n;
    Box b = tri_box(tris[prim[lo]]);
    for (uint32_t p = lo + 1u; p < hi; ++p) {
        const Box t = tri_box(tris[prim[p]]);
        for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], t.mn[a]); b.mx[a] = fmaxf(b.mx[a], t.mx[a]); }
    }
    boxes[c] = b;
}

__global__ __launch_bounds__(kBlock) void k_scatter_top(const int* __restrict__ ids, const int* __restrict__ parents, const int* __restrict__ children, int n_top,
                                                        int* __restrict__ parent, int* __restrict__ child, uint32_t* __restrict__ live) {
    const int t = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (t >= n_top) return;
    const int id = ids[t];
    parent[id] = parents[t];
    live[id] = 1u;
    // links to cluster ranges are written by the linear build (k_level); only links between top nodes are known here
    if (children[2 * t] != kLinkUnknown) child[2 * id] = children[2 * t];
    if (children[2 * t + 1] != kLinkUnknown) child[2 * id + 1] = children[2 * t + 1];
}

struct Item { int lo, hi, parent, side, depth; };   // a range of sorted positions, the node above it, which child it is, the depth of the record it would become

__host__ __device__ inline int levels_needed(uint32_t n) {   // record levels a median-split sub-tree of n triangles needs
    int l = 0;
    while (n > kLeafMax) { n = (n + 1) / 2; ++l; }
    return l;
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

// ---- SAH top tree over the Morton clusters (host, one thread: <= 32768 clusters). The Morton curve clusters the upper
// levels of a scene badly (a split at the highest differing bit is a spatial-median split, whatever the geometry), and that
// is where a linear BVH loses against SAH; inside a small cluster the difference is nil. So the upper levels are
// built with binned SAH (16 bins, cost = half area x triangle count) over the cluster boxes, and every cluster becomes a
// seed range for the linear build below its leaf.
struct TopNode { int parent, side, child[2], depth; };   // child: >= 0 top node, < 0: ~cluster
struct Cluster { Box box; float c[3]; uint32_t lo, hi, count; };

struct TopBuilder {
    const std::vector<Cluster>& cl;
    std::vector<int> order;
    std::vector<TopNode> nodes;
    int max_seed_depth;   // deepest record a cluster's root may be

    static int ceil_log2(uint32_t k) { int l = 0; while ((1u << l) < k) ++l; return l; }

    // builds the sub-tree over order[begin, end) whose root record sits at `depth`; returns its link
    int build(int begin, int end, int depth, int parent, int side) {
        const int k = end - begin;
        if (k == 1) return ~order[(size_t)begin];
        const int me = (int)nodes.size();
        nodes.push_back(TopNode{parent, side, {0, 0}, depth});
        // bounds of the centroids, bins
        float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = begin; i < end; ++i)
            for (int a = 0; a < 3; ++a) { cmn[a] = std::min(cmn[a], cl[(size_t)order[(size_t)i]].c[a]); cmx[a] = std::max(cmx[a], cl[(size_t)order[(size_t)i]].c[a]); }
        constexpr int B = 16;
        float best = FLT_MAX;
        int best_axis = -1, best_bin = -1;
        for (int a = 0; a < 3; ++a) {
            const float ext = cmx[a] - cmn[a];
            if (!(ext > 0.f)) continue;
            Box bb[B];
            uint32_t cnt[B] = {0}, num[B] = {0};
            for (int b = 0; b < B; ++b) for (int x = 0; x < 3; ++x) { bb[b].mn[x] = FLT_MAX; bb[b].mx[x] = -FLT_MAX; }
            const float scale = (float)B / ext;
            for (int i = begin; i < end; ++i) {
                const Cluster& c = cl[(size_t)order[(size_t)i]];
                const int b = std::min(B - 1, std::max(0, (int)((c.c[a] - cmn[a]) * scale)));
                for (int x = 0; x < 3; ++x) { bb[b].mn[x] = std::min(bb[b].mn[x], c.box.mn[x]); bb[b].mx[x] = std::max(bb[b].mx[x], c.box.mx[x]); }
                cnt[b] += c.count;
                num[b] += 1u;
            }
            float rarea[B];
            uint32_t rcnt[B], rnum[B];
            Box acc;
            for (int x = 0; x < 3; ++x) { acc.mn[x] = FLT_MAX; acc.mx[x] = -FLT_MAX; }
            uint32_t c = 0, m = 0;
            auto area = [](const Box& b) { const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2]; return (dx < 0.f || dy < 0.f || dz < 0.f) ? 0.f : dx * dy + dy * dz + dz * dx; };
            for (int b = B - 1; b > 0; --b) {
                for (int x = 0; x < 3; ++x) { acc.mn[x] = std::min(acc.mn[x], bb[b].mn[x]); acc.mx[x] = std::max(acc.mx[x], bb[b].mx[x]); }
                c += cnt[b]; m += num[b];
                rarea[b] = area(acc); rcnt[b] = c; rnum[b] = m;
            }
            Box lacc;
            for (int x = 0; x < 3; ++x) { lacc.mn[x] = FLT_MAX; lacc.mx[x] = -FLT_MAX; }
            uint32_t lc = 0, lm = 0;
            for (int b = 0; b < B - 1; ++b) {
                for (int x = 0; x < 3; ++x) { lacc.mn[x] = std::min(lacc.mn[x], bb[b].mn[x]); lacc.mx[x] = std::max(lacc.mx[x], bb[b].mx[x]); }
                lc += cnt[b]; lm += num[b];
                if (lm == 0 || rnum[b + 1] == 0) continue;
                // depth guard: both sides must still fit below max_seed_depth
                if (depth + 1 + ceil_log2(std::max(lm, rnum[b + 1])) > max_seed_depth) continue;
                const float cost = area(lacc) * (float)lc + rarea[b + 1] * (float)rcnt[b + 1];
                if (cost < best) { best = cost; best_axis = a; best_bin = b; }
            }
        }
        int mid = begin + k / 2;
        if (best_axis >= 0) {
            const float scale = (float)B / (cmx[best_axis] - cmn[best_axis]);
            auto it = std::stable_partition(order.begin() + begin, order.begin() + end, [&](int ci) {
                return std::min(B - 1, std::max(0, (int)((cl[(size_t)ci].c[best_axis] - cmn[best_axis]) * scale))) <= best_bin;
            });
            mid = (int)(it - order.begin());
        } else {   // median by number of clusters along the widest axis (equal centroids, or the depth guard)
            int axis = 0;
            for (int a = 1; a < 3; ++a) if (cmx[a] - cmn[a] > cmx[axis] - cmn[axis]) axis = a;
            std::stable_sort(order.begin() + begin, order.begin() + end, [&](int x, int y) { return cl[(size_t)x].c[axis] < cl[(size_t)y].c[axis]; });
        }
        const int c0 = build(begin, mid, depth + 1, me, 0);
        const int c1 = build(mid, end, depth + 1, me, 1);
        nodes[(size_t)me].child[0] = c0;
        nodes[(size_t)me].child[1] = c1;
        return me;
    }
};

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
    LB_TRY(tmp.reserve(n * 224 + sort_bytes + scan_bytes + scan_bytes_n + (2u << 20)));
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

    // ---- seeds: the whole range under no parent (plain linear BVH), or one range per Morton cluster under an SAH top tree
    size_t n_seeds = 1;
    uint32_t top_depth = 0;
    bool seeded = false;
    const char* top_env = getenv("TRH_LBVH_TOP");   // development override: 0 = plain linear BVH
    if (!(top_env && atoi(top_env) == 0) && n >= 4096) {
        // cluster = shared top `bits` Morton bits (a cube of the centroid grid), about 8 triangles each where 2^15 cubes
        // allow it (dragon 100 k: 32 per cluster 12.4 ms per 64 spp, 8 per cluster 11.9; more cubes cost host time)
        int bits = 6;
        size_t per_cluster = 8;
        if (const char* e = getenv("TRH_LBVH_CLUSTER")) per_cluster = (size_t)std::max(1, atoi(e));   // development knob
        while (bits < 15 && ((size_t)1 << bits) * per_cluster < n) bits += 3;
        uint32_t *d_flags = nullptr, *d_cidx = nullptr, *d_starts = nullptr;
        unsigned char* d_scan2 = nullptr;
        Box* d_cboxes = nullptr;
        LB_TRY(tmp.alloc(&d_flags, n));
        LB_TRY(tmp.alloc(&d_cidx, n));
        LB_TRY(tmp.alloc(&d_scan2, scan_bytes_n));
        hipLaunchKernelGGL(k_cluster_flags, dim3(grid_n), dim3(kBlock), 0, stream, d_keys_sorted, (uint32_t)n, 63 - bits, d_flags);
        LB_TRY(hipGetLastError());
        LB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan2, scan_bytes_n, d_flags, d_cidx, ni, stream));
        uint32_t last_flag = 0, last_cidx = 0;
        LB_TRY(hipMemcpyAsync(&last_flag, d_flags + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        LB_TRY(hipMemcpyAsync(&last_cidx, d_cidx + (n - 1), 4, hipMemcpyDeviceToHost, stream));
        LB_TRY(hipStreamSynchronize(stream));
        const uint32_t n_clusters = last_cidx + last_flag;
        if (n_clusters >= 2 && n_clusters <= (1u << 15)) {
            LB_TRY(tmp.alloc(&d_starts, n_clusters));
            LB_TRY(tmp.alloc(&d_cboxes, n_clusters));
            hipLaunchKernelGGL(k_cluster_starts, dim3(grid_n), dim3(kBlock), 0, stream, d_flags, d_cidx, (uint32_t)n, d_starts);
            LB_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_cluster_boxes, dim3((n_clusters + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, d_tris, d_prim, d_starts, n_clusters, (uint32_t)n, d_cboxes);
            LB_TRY(hipGetLastError());
            std::vector<uint32_t> starts(n_clusters);
            std::vector<Box> cboxes(n_clusters);
            LB_TRY(hipMemcpyAsync(starts.data(), d_starts, (size_t)n_clusters * 4, hipMemcpyDeviceToHost, stream));
            LB_TRY(hipMemcpyAsync(cboxes.data(), d_cboxes, (size_t)n_clusters * sizeof(Box), hipMemcpyDeviceToHost, stream));
            LB_TRY(hipStreamSynchronize(stream));
            std::vector<Cluster> cl(n_clusters);
            uint32_t biggest = 0;
            for (uint32_t c = 0; c < n_clusters; ++c) {
                cl[c].box = cboxes[c];
                cl[c].lo = starts[c];
                cl[c].hi = (c + 1 < n_clusters ? starts[c + 1] : (uint32_t)n) - 1u;
                cl[c].count = cl[c].hi - cl[c].lo + 1u;
                for (int a = 0; a < 3; ++a) cl[c].c[a] = 0.5f * (cboxes[c].mn[a] + cboxes[c].mx[a]);
                biggest = std::max(biggest, cl[c].count);
            }
            // a cluster's root record at depth D needs D + levels_needed(count) <= depth_limit to be finishable by medians
            const int max_seed_depth = depth_limit - levels_needed(biggest);
            if (1 + TopBuilder::ceil_log2(n_clusters) <= max_seed_depth) {
                TopBuilder tb{cl, {}, {}, max_seed_depth};
                tb.order.resize(n_clusters);
                for (uint32_t c = 0; c < n_clusters; ++c) tb.order[c] = (int)c;
                tb.nodes.reserve(n_clusters);
                const int root_link = tb.build(0, (int)n_clusters, 1, -1, 0);
                // a top node takes the last sorted position of a cluster as its index: the linear build below uses the
                // positions [lo, hi - 1] of a range for its own nodes, so these are free (n_clusters - 1 nodes, and the
                // last cluster's last position, n - 1, is outside the node arrays and stays unused)
                auto node_id = [&](int t) { return (int)cl[(size_t)t].hi; };
                std::vector<Item> seeds(n_clusters);
                std::vector<int> h_parent(tb.nodes.size()), h_ids(tb.nodes.size());
                std::vector<int> h_child(tb.nodes.size() * 2);
                size_t ns = 0;
                for (size_t t = 0; t < tb.nodes.size(); ++t) {
                    const TopNode& tn = tb.nodes[t];
                    h_ids[t] = node_id((int)t);
                    h_parent[t] = tn.parent >= 0 ? node_id(tn.parent) : -1;
                    top_depth = std::max(top_depth, (uint32_t)tn.depth);
                    for (int sd = 0; sd < 2; ++sd) {
                        const int link = tn.child[sd];
                        if (link >= 0) h_child[2 * t + sd] = node_id(link);
                        else {
                            const Cluster& c = cl[(size_t)~link];
                            h_child[2 * t + sd] = kLinkUnknown;   // written by the range's own thread (inner node or leaf)
                            seeds[ns++] = Item{(int)c.lo, (int)c.hi, node_id((int)t), sd, tn.depth + 1};
                        }
                    }
                }
                if (root_link >= 0 && ns == n_clusters) {
                    // scatter the top nodes into the node arrays (a few thousand words: one small kernel-free path via
                    // per-array staging would need index lists; the arrays are written entry by entry from pinned-less
                    // host memory in three strided copies instead)
                    int *d_ids = nullptr, *d_hp = nullptr, *d_hc = nullptr;
                    LB_TRY(tmp.alloc(&d_ids, tb.nodes.size()));
                    LB_TRY(tmp.alloc(&d_hp, tb.nodes.size()));
                    LB_TRY(tmp.alloc(&d_hc, tb.nodes.size() * 2));
                    LB_TRY(hipMemcpyAsync(d_ids, h_ids.data(), h_ids.size() * 4, hipMemcpyHostToDevice, stream));
                    LB_TRY(hipMemcpyAsync(d_hp, h_parent.data(), h_parent.size() * 4, hipMemcpyHostToDevice, stream));
                    LB_TRY(hipMemcpyAsync(d_hc, h_child.data(), h_child.size() * 4, hipMemcpyHostToDevice, stream));
                    hipLaunchKernelGGL(k_scatter_top, dim3((unsigned)((tb.nodes.size() + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_ids, d_hp, d_hc, (int)tb.nodes.size(),
                                       d_parent, d_child, d_live);
                    LB_TRY(hipGetLastError());
                    const int root_id = node_id(root_link);
                    LB_TRY(hipMemcpyAsync(d_root, &root_id, 4, hipMemcpyHostToDevice, stream));
                    LB_TRY(hipMemcpyAsync(d_items_a, seeds.data(), ns * sizeof(Item), hipMemcpyHostToDevice, stream));
                    LB_TRY(hipStreamSynchronize(stream));   // the host vectors above go out of scope
                    n_seeds = ns;
                    seeded = true;
                }
            }
        }
    }
    if (!seeded) {
        const Item root_item{0, ni - 1, -1, 0, 1};
        LB_TRY(hipMemcpyAsync(d_items_a, &root_item, sizeof(Item), hipMemcpyHostToDevice, stream));
        top_depth = 0;
    }
    // Every level is launched with an upper bound of its queue length (2^(depth-1), at most n ranges) and reads the
    // real length from the device: no host round trip per level. The depth guard ends the tree at kMaxDepth records;
    // level kMaxDepth + 1 only holds leaves.
    Stats* d_stats = nullptr;
    LB_TRY(tmp.alloc(&d_stats, 1));
    LB_TRY(hipMemsetAsync(d_stats, 0, sizeof(Stats), stream));
    unsigned int* d_counts = nullptr;
    LB_TRY(tmp.alloc(&d_counts, (size_t)bvh::kMaxDepth + 3));
    LB_TRY(hipMemsetAsync(d_counts, 0, ((size_t)bvh::kMaxDepth + 3) * sizeof(unsigned int), stream));
    const unsigned int first_count = (unsigned int)n_seeds;
    LB_TRY(hipMemcpyAsync(d_counts + 1, &first_count, sizeof(first_count), hipMemcpyHostToDevice, stream));
    if (top_depth) LB_TRY(hipMemcpyAsync(&d_stats->max_depth, &top_depth, sizeof(top_depth), hipMemcpyHostToDevice, stream));
    for (int depth = 1; depth <= bvh::kMaxDepth + 1; ++depth) {
        // (generation `depth` of the work queue: with seeds the records of one generation sit at different tree depths)
        const size_t bound = depth > 27 ? n : std::min<size_t>(n, n_seeds << (depth - 1));
        hipLaunchKernelGGL(k_level, dim3((unsigned)((bound + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_items_a, (int)bound, depth_limit, d_keys_sorted, d_tris, d_prim,
                           d_items_b, d_counts + depth, d_counts + depth + 1, d_stats, d_parent, d_child, d_live, d_leaf_box, d_node_box, d_arrived, d_root);
        LB_TRY(hipGetLastError());
        std::swap(d_items_a, d_items_b);
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
