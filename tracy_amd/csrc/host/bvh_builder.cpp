// bvh_builder.cpp -- binned-SAH BVH2 build, see bvh_builder.h for the layout.
#include "bvh_builder.h"
#include "half_round.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <thread>

namespace bvh {
namespace {

struct Box {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float p[3]) {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); }
    }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0.f || dy < 0.f || dz < 0.f) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Ref {
    Box box;
    float c[3];
    uint32_t idx;
};

float kCostTraverse = 0.8f;   // one inner-record step (two slab tests) against one triangle test; swept on the GPU, profiles/r02_bvh_knobs.md
constexpr float kCostIntersect = 1.0f;   // one Moller-Trumbore test

int levels_needed(size_t n) {  // inner-record levels a median-split subtree of n prims needs
    int l = 0;
    while (n > (size_t)kMaxLeafSize) { n = (n + 1) / 2; ++l; }
    return l;
}

// The tree is first built into temporary records (parallel: sub-ranges of `refs` are independent, so
// sub-trees are separate tasks and the binning of big nodes is split over threads), then emitted in
// depth-first pre-order by one thread. The result does not depend on the number of threads: the
// partitions are deterministic, bin merging is exact (min / max / counts), and the final layout is fixed
// by the pre-order walk.
struct TmpNode {
    Box lb, rb;
    int c0 = 0, c1 = 0;   // >= 0: temporary record index; < 0: leaf ~((first << 4) | count), first = position in refs
};

struct Builder {
    const std::vector<InputTriangle>& in;
    std::vector<Ref> refs;
    Result& out;
    float root_area = 1.f;
    float scene_extent = 0.f;
    std::vector<TmpNode> tmp;
    std::atomic<int> tmp_next{0};
    std::atomic<int> spare_threads{0};
    std::atomic<int> tasks_spawned{0};
    static constexpr size_t kTaskMin = 8192;       // smaller sub-trees are not worth a task
    static constexpr size_t kParallelBinMin = 1u << 17;

    Builder(const std::vector<InputTriangle>& t, Result& r) : in(t), out(r) {}

    struct Bins {
        Box bb[3][kSahBins];
        uint32_t cnt[3][kSahBins];
        Bins() { memset(cnt, 0, sizeof(cnt)); }
        void merge(const Bins& o) {
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < kSahBins; ++b) { bb[a][b].grow(o.bb[a][b]); cnt[a][b] += o.cnt[a][b]; }
        }
    };

    void bin_range(size_t begin, size_t end, const Box& cb, const bool use_axis[3], Bins& bins) const {
        float lo[3], scale[3];
        for (int a = 0; a < 3; ++a) { lo[a] = cb.mn[a]; scale[a] = use_axis[a] ? kSahBins / (cb.mx[a] - cb.mn[a]) : 0.f; }
        for (size_t i = begin; i < end; ++i) {
            const Ref& r = refs[i];
            for (int a = 0; a < 3; ++a) {
                if (!use_axis[a]) continue;
                const int b = std::min(kSahBins - 1, std::max(0, (int)((r.c[a] - lo[a]) * scale[a])));
                bins.bb[a][b].grow(r.box);
                ++bins.cnt[a][b];
            }
        }
    }

    int make_leaf(size_t begin, size_t end) const { return ~(int)(((uint32_t)begin << 4) | (uint32_t)(end - begin)); }

    // returns child link; box = bounds of [begin,end)
    int build(size_t begin, size_t end, const Box& box, int depth /* depth of the record this range would become */) {
        const size_t n = end - begin;
        Box cb;
        for (size_t i = begin; i < end; ++i) cb.grow(refs[i].c);

        // --- find the best binned SAH split over the three axes
        float best_cost = FLT_MAX;
        int best_axis = -1, best_bin = -1;
        if (n > 1) {
            bool use_axis[3];
            for (int a = 0; a < 3; ++a) use_axis[a] = (cb.mx[a] - cb.mn[a]) > 0.f;
            Bins bins;
            int helpers = 0;
            if (n >= kParallelBinMin) {   // big node: split the binning over the spare threads
                helpers = spare_threads.exchange(0);
            }
            if (helpers > 0) {
                const int parts = helpers + 1;
                std::vector<Bins> part((size_t)parts);
                std::vector<std::future<void>> fut;
                for (int t = 1; t < parts; ++t)
                    fut.push_back(std::async(std::launch::async, [&, t]() { bin_range(begin + n * t / parts, begin + n * (t + 1) / parts, cb, use_axis, part[t]); }));
                bin_range(begin, begin + n / parts, cb, use_axis, part[0]);
                for (auto& f : fut) f.get();
                for (int t = 0; t < parts; ++t) bins.merge(part[t]);
                spare_threads.fetch_add(helpers);
            } else {
                bin_range(begin, end, cb, use_axis, bins);
            }
            for (int axis = 0; axis < 3; ++axis) {
                if (!use_axis[axis]) continue;
                const Box* bb = bins.bb[axis];
                const uint32_t* cnt = bins.cnt[axis];
                float right_area[kSahBins];
                uint32_t right_cnt[kSahBins];
                Box acc;
                uint32_t c = 0;
                for (int b = kSahBins - 1; b > 0; --b) {
                    acc.grow(bb[b]);
                    c += cnt[b];
                    right_area[b] = acc.half_area();
                    right_cnt[b] = c;
                }
                Box lacc;
                uint32_t lc = 0;
                for (int b = 0; b < kSahBins - 1; ++b) {
                    lacc.grow(bb[b]);
                    lc += cnt[b];
                    if (lc == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = lacc.half_area() * lc + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
        }
        const float parent_area = std::max(box.half_area(), 1e-30f);
        const float split_cost = kCostTraverse + kCostIntersect * best_cost / parent_area;
        const float leaf_cost = kCostIntersect * (float)n;
        if (n <= (size_t)kMaxLeafSize && (best_axis < 0 || leaf_cost <= split_cost || n == 1 || depth > kMaxDepth)) return make_leaf(begin, end);

        // --- partition
        size_t mid = begin;
        bool median = (best_axis < 0);
        if (!median) {
            const float lo = cb.mn[best_axis], scale = kSahBins / (cb.mx[best_axis] - cb.mn[best_axis]);
            auto it = std::partition(refs.begin() + begin, refs.begin() + end, [&](const Ref& r) {
                int b = std::min(kSahBins - 1, std::max(0, (int)((r.c[best_axis] - lo) * scale)));
                return b <= best_bin;
            });
            mid = (size_t)(it - refs.begin());
            if (mid == begin || mid == end) median = true;
            // depth guard: fall back to a median split if the SAH split could not be finished within kMaxDepth
            else if (depth + std::max(levels_needed(mid - begin), levels_needed(end - mid)) > kMaxDepth) median = true;
        }
        if (median) {
            int axis = 0;
            float e = -1.f;
            for (int a = 0; a < 3; ++a) if (cb.mx[a] - cb.mn[a] > e) { e = cb.mx[a] - cb.mn[a]; axis = a; }
            mid = begin + n / 2;
            std::nth_element(refs.begin() + begin, refs.begin() + mid, refs.begin() + end,
                             [axis](const Ref& a, const Ref& b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.idx < b.idx); });
        }

        Box lb, rb;
        for (size_t i = begin; i < mid; ++i) lb.grow(refs[i].box);
        for (size_t i = mid; i < end; ++i) rb.grow(refs[i].box);

        const int me = tmp_next.fetch_add(1);
        TmpNode& node = tmp[(size_t)me];
        node.lb = lb;
        node.rb = rb;
        // the left sub-tree becomes a task if it is big enough and a thread is spare
        bool spawned = false;
        std::future<int> left;
        if (mid - begin >= kTaskMin && end - mid >= kTaskMin) {
            int s = spare_threads.load();
            while (s > 0 && !spare_threads.compare_exchange_weak(s, s - 1)) {}
            if (s > 0) {
                spawned = true;
                tasks_spawned.fetch_add(1);
                left = std::async(std::launch::async, [this, begin, mid, lb, depth]() { return build(begin, mid, lb, depth + 1); });
            }
        }
        if (!spawned) node.c0 = build(begin, mid, lb, depth + 1);
        node.c1 = build(mid, end, rb, depth + 1);
        if (spawned) {
            node.c0 = left.get();
            spare_threads.fetch_add(1);
        }
        return me;
    }

    // depth-first pre-order emission of the final records (the layout the sequential recursion produced)
    int emit(int link, const Box& box, int depth) {
        if (link < 0) {
            const uint32_t bits = ~(uint32_t)link;
            const uint32_t count = bits & 15u;
            out.max_leaf = std::max(out.max_leaf, count);
            out.sah_cost += (double)box.half_area() / root_area * count * kCostIntersect;
            return link;
        }
        const TmpNode& t = tmp[(size_t)link];
        const int me = (int)(out.nodes.size() / 16);
        out.nodes.resize(out.nodes.size() + 16, 0.f);
        out.max_depth = std::max(out.max_depth, (uint32_t)depth);
        out.sah_cost += (double)box.half_area() / root_area * kCostTraverse;
        const int c0 = emit(t.c0, t.lb, depth + 1);
        const int c1 = emit(t.c1, t.rb, depth + 1);
        write_record(me, t.lb, t.rb, c0, c1);
        return me;
    }

    void write_triangles() {
        out.tris.resize(refs.size() * 12);
        out.prim_index.resize(refs.size());
        for (size_t i = 0; i < refs.size(); ++i) {
            const InputTriangle& t = in[refs[i].idx];
            float* rec = &out.tris[i * 12];
            for (int a = 0; a < 3; ++a) {
                rec[a] = t.v0[a];
                rec[4 + a] = t.v1[a] - t.v0[a];  // same subtraction RayTriangle does (collision.h:35-36)
                rec[8 + a] = t.v2[a] - t.v0[a];
            }
            memcpy(&rec[3], &t.packed_id, 4);
            rec[7] = 0.f;
            rec[11] = 0.f;
            out.prim_index[i] = refs[i].idx;
        }
    }

    void pad(Box& b) const {
        // conservative inflation: the slab test and Moller-Trumbore round differently
        for (int a = 0; a < 3; ++a) {
            float m = std::max(std::max(std::fabs(b.mn[a]), std::fabs(b.mx[a])), b.mx[a] - b.mn[a]);
            float d = 1e-5f * m + 2e-6f * scene_extent + 1e-7f;
            b.mn[a] -= d;
            b.mx[a] += d;
        }
    }

    void write_record(int me, Box lb, Box rb, int c0, int c1) {
        pad(lb);
        pad(rb);
        float* r = &out.nodes[(size_t)me * 16];
        r[0] = lb.mn[0]; r[1] = lb.mx[0]; r[2] = lb.mn[1]; r[3] = lb.mx[1];
        r[4] = rb.mn[0]; r[5] = rb.mx[0]; r[6] = rb.mn[1]; r[7] = rb.mx[1];
        r[8] = lb.mn[2]; r[9] = lb.mx[2]; r[10] = rb.mn[2]; r[11] = rb.mx[2];
        memcpy(&r[12], &c0, 4);
        memcpy(&r[13], &c1, 4);
        r[14] = 0.f;
        r[15] = 0.f;
    }

    // 32-byte records from the finished fp32 records (layout in bvh_builder.h)
    void quantise() {
        // frame: centre = per-axis median of the centroids, scale = the power of two that maps the
        // farthest padded bound to [16384, 32768)
        double c[3] = {0, 0, 0};
        if (!refs.empty()) {
            std::vector<float> tmp(refs.size());
            for (int a = 0; a < 3; ++a) {
                for (size_t i = 0; i < refs.size(); ++i) tmp[i] = refs[i].c[a];
                std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
                c[a] = tmp[tmp.size() / 2];
            }
        }
        double far = 0.0;
        const size_t n = out.nodes.size() / 16;
        for (size_t i = 0; i < n; ++i) {
            const float* r = &out.nodes[i * 16];
            const int axis_of[12] = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 2, 2};
            for (int k = 0; k < 12; ++k)
                if (std::fabs(r[k]) < 1e30f) far = std::max(far, std::fabs((double)r[k] - c[axis_of[k]]));
        }
        int k2 = 0;
        if (far > 0.0) {
            int ex;
            (void)std::frexp(far, &ex);       // far in [2^(ex-1), 2^ex)
            k2 = 15 - ex;                     // far * 2^k2 in [16384, 32768)
        }
        k2 = std::min(60, std::max(-60, k2));
        const double s = std::ldexp(1.0, k2);
        for (int a = 0; a < 3; ++a) out.frame_center[a] = (float)c[a];
        out.frame_scale = (float)s;
        out.nodes16.assign(n * 8, 0u);
        auto half_area = [](double dx, double dy, double dz) { return (dx < 0 || dy < 0 || dz < 0) ? 0.0 : dx * dy + dy * dz + dz * dx; };
        // a fixed number of chunks (independent of the thread count) so that the area sums do not depend on it
        constexpr size_t kChunks = 64;
        double exact_part[kChunks] = {0}, quant_part[kChunks] = {0};
        auto do_chunk = [&](size_t ch) {
            double area_exact = 0.0, area_quant = 0.0;
            for (size_t i = n * ch / kChunks; i < n * (ch + 1) / kChunks; ++i) {
                const float* r = &out.nodes[i * 16];
                uint32_t* q = &out.nodes16[i * 8];
                auto pack = [&](float mn, float mx, int axis) {
                    const double cc = (double)out.frame_center[axis];   // the float the kernel subtracts
                    const uint16_t lo = half_floor(((double)mn - cc) * s);
                    const uint16_t hi = half_ceil(((double)mx - cc) * s);
                    return (uint32_t)lo | ((uint32_t)hi << 16);
                };
                q[0] = pack(r[0], r[1], 0);  q[1] = pack(r[2], r[3], 1);
                q[2] = pack(r[4], r[5], 0);  q[3] = pack(r[6], r[7], 1);
                q[4] = pack(r[8], r[9], 2);  q[5] = pack(r[10], r[11], 2);
                memcpy(&q[6], &r[12], 4);
                memcpy(&q[7], &r[13], 4);
                // inflation measure (boxes of empty / padding leaves are skipped: |bound| >= 1e30)
                const int ex[2][6] = {{0, 1, 2, 3, 8, 9}, {4, 5, 6, 7, 10, 11}};   // minx maxx miny maxy minz maxz in r[]
                const int qx[2][3] = {{0, 1, 4}, {2, 3, 5}};                        // words holding x, y, z (min | max << 16)
                for (int c = 0; c < 2; ++c) {
                    if (std::fabs(r[ex[c][0]]) >= 1e30f) continue;
                    area_exact += half_area((double)r[ex[c][1]] - r[ex[c][0]], (double)r[ex[c][3]] - r[ex[c][2]], (double)r[ex[c][5]] - r[ex[c][4]]);
                    double d[3];
                    for (int a = 0; a < 3; ++a) d[a] = ((double)half_to_float((uint16_t)(q[qx[c][a]] >> 16)) - (double)half_to_float((uint16_t)(q[qx[c][a]] & 0xffffu))) / s;
                    area_quant += half_area(d[0], d[1], d[2]);
                }
            }
            exact_part[ch] = area_exact;
            quant_part[ch] = area_quant;
        };
        {
            int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
            if (const char* e = getenv("TRH_BUILD_THREADS")) threads = std::max(1, atoi(e));
            if (n < 65536) threads = 1;
            std::atomic<size_t> next_chunk{0};
            auto worker = [&]() { for (size_t ch; (ch = next_chunk.fetch_add(1)) < kChunks;) do_chunk(ch); };
            std::vector<std::future<void>> fut;
            for (int t = 1; t < threads; ++t) fut.push_back(std::async(std::launch::async, worker));
            worker();
            for (auto& f : fut) f.get();
        }
        double area_exact = 0.0, area_quant = 0.0;
        for (size_t ch = 0; ch < kChunks; ++ch) { area_exact += exact_part[ch]; area_quant += quant_part[ch]; }
        out.fp16_inflation = area_exact > 0.0 ? area_quant / area_exact : 1.0;
    }

    void run() {
        refs.resize(in.size());
        Box all;
        for (size_t i = 0; i < in.size(); ++i) {
            Ref& r = refs[i];
            r.idx = (uint32_t)i;
            r.box.grow(in[i].v0);
            r.box.grow(in[i].v1);
            r.box.grow(in[i].v2);
            for (int a = 0; a < 3; ++a) r.c[a] = 0.5f * (r.box.mn[a] + r.box.mx[a]);
            all.grow(r.box);
        }
        root_area = std::max(all.half_area(), 1e-30f);
        for (int a = 0; a < 3; ++a) scene_extent = std::max(scene_extent, std::max(all.mx[a] - all.mn[a], std::max(std::fabs(all.mn[a]), std::fabs(all.mx[a]))));
        if (in.empty()) scene_extent = 0.f;
        out.nodes.clear();
        out.tris.clear();
        out.prim_index.clear();
        const int empty_leaf = ~(1 << 4);  // first 1, count 0 (never -1: the kernels read a link of -1 as "nothing to traverse")
        if (in.empty()) {
            out.nodes.resize(16, 0.f);
            write_record(0, Box{}, Box{}, empty_leaf, empty_leaf);
            return;
        }
        int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char* e = getenv("TRH_BUILD_THREADS")) threads = std::max(1, atoi(e));
        spare_threads.store(threads - 1);
        tmp.resize(std::max<size_t>(in.size(), 1));
        const auto tb0 = std::chrono::steady_clock::now();
        const int root = build(0, in.size(), all, 1);
        if (getenv("TRH_VERBOSE"))
            fprintf(stderr, "bvh::build: %d threads, recursion %.1f ms, %d tasks\n", threads,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(), tasks_spawned.load());
        out.nodes.reserve((size_t)tmp_next.load() * 16 + 16);
        if (root < 0) {
            // the whole scene fits one leaf: wrap it in a root record (traversal always starts at record 0)
            out.nodes.resize(16, 0.f);
            Box none;
            none.mn[0] = none.mn[1] = none.mn[2] = 0.f;
            none.mx[0] = none.mx[1] = none.mx[2] = 0.f;
            out.max_leaf = (uint32_t)in.size();
            out.sah_cost = (double)in.size() * kCostIntersect;
            write_record(0, all, none, root, empty_leaf);
            out.max_depth = 1;
        } else {
            emit(root, all, 1);
        }
        write_triangles();
    }
};

}  // namespace

void build(const std::vector<InputTriangle>& tris, Result& out) {
    out = Result{};
    if (const char* e = getenv("TRH_SAH_CT")) { const float v = (float)atof(e); if (v > 0.f) kCostTraverse = v; }   // development knob
    Builder b(tris, out);
    const auto t0 = std::chrono::steady_clock::now();
    b.run();
    const auto t1 = std::chrono::steady_clock::now();
    b.quantise();
    const auto t2 = std::chrono::steady_clock::now();
    if (getenv("TRH_VERBOSE"))
        fprintf(stderr, "bvh::build: %zu triangles, tree %.1f ms, fp16 records %.1f ms\n", tris.size(),
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
}

namespace {

uint64_t fnv1a(const void* p, size_t bytes, uint64_t h = 1469598103934665603ull) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < bytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

struct CacheHeader {
    char magic[8];            // "TRHBVH1\0"
    uint64_t key;             // hash of the input triangles + builder constants
    uint64_t n_nodes_f32, n_nodes16, n_tris_f32, n_prims;   // element counts of the four arrays
    float frame_center[3];
    float frame_scale;
    double fp16_inflation, sah_cost;
    uint32_t max_depth, max_leaf;
    uint64_t checksum;        // hash of the four arrays
};
static_assert(sizeof(CacheHeader) == 96, "cache header layout");
constexpr char kCacheMagic[8] = {'T', 'R', 'H', 'B', 'V', 'H', '1', 0};

uint64_t cache_key(const std::vector<InputTriangle>& tris) {
    uint64_t h = fnv1a(tris.data(), tris.size() * sizeof(InputTriangle));
    const float consts[4] = {kCostTraverse, kCostIntersect, (float)kSahBins, (float)kMaxLeafSize};
    h = fnv1a(consts, sizeof(consts), h);
    const uint64_t n = tris.size();
    return fnv1a(&n, sizeof(n), h);
}
uint64_t arrays_checksum(const Result& r) {
    uint64_t h = fnv1a(r.nodes.data(), r.nodes.size() * 4);
    h = fnv1a(r.nodes16.data(), r.nodes16.size() * 4, h);
    h = fnv1a(r.tris.data(), r.tris.size() * 4, h);
    return fnv1a(r.prim_index.data(), r.prim_index.size() * 4, h);
}

bool cache_load(const std::string& path, uint64_t key, size_t n_tris, Result& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    CacheHeader h{};
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && memcmp(h.magic, kCacheMagic, 8) == 0 && h.key == key;
    // sizes must be what a tree over n_tris triangles can have, before anything is allocated from them
    ok = ok && h.n_prims == n_tris && h.n_tris_f32 == (uint64_t)n_tris * 12 && h.n_nodes_f32 % 16 == 0 && h.n_nodes16 * 2 == h.n_nodes_f32 &&
         h.n_nodes_f32 / 16 >= 1 && h.n_nodes_f32 / 16 <= std::max<uint64_t>(n_tris, 1) && h.max_depth <= (uint32_t)kMaxDepth;
    // header scalars that reach the kernels: the frame scale must be a finite positive power of two, the centre finite
    {
        int e = 0;
        ok = ok && std::isfinite(h.frame_scale) && h.frame_scale > 0.f && std::frexp(h.frame_scale, &e) == 0.5f && std::isfinite(h.frame_center[0]) &&
             std::isfinite(h.frame_center[1]) && std::isfinite(h.frame_center[2]) && std::isfinite(h.fp16_inflation) && std::isfinite(h.sah_cost);
    }
    if (ok) {
        out = Result{};
        out.nodes.resize(h.n_nodes_f32);
        out.nodes16.resize(h.n_nodes16);
        out.tris.resize(h.n_tris_f32);
        out.prim_index.resize(h.n_prims);
        ok = fread(out.nodes.data(), 4, out.nodes.size(), f) == out.nodes.size() && fread(out.nodes16.data(), 4, out.nodes16.size(), f) == out.nodes16.size() &&
             fread(out.tris.data(), 4, out.tris.size(), f) == out.tris.size() && fread(out.prim_index.data(), 4, out.prim_index.size(), f) == out.prim_index.size() &&
             fgetc(f) == EOF && arrays_checksum(out) == h.checksum;
        memcpy(out.frame_center, h.frame_center, sizeof(out.frame_center));
        out.frame_scale = h.frame_scale;
        out.fp16_inflation = h.fp16_inflation;
        out.sah_cost = h.sah_cost;
        out.max_depth = h.max_depth;
        out.max_leaf = h.max_leaf;
    }
    fclose(f);
    return ok;
}

void cache_store(const std::string& path, uint64_t key, const Result& r) {
    CacheHeader h{};
    memcpy(h.magic, kCacheMagic, 8);
    h.key = key;
    h.n_nodes_f32 = r.nodes.size();
    h.n_nodes16 = r.nodes16.size();
    h.n_tris_f32 = r.tris.size();
    h.n_prims = r.prim_index.size();
    memcpy(h.frame_center, r.frame_center, sizeof(h.frame_center));
    h.frame_scale = r.frame_scale;
    h.fp16_inflation = r.fp16_inflation;
    h.sah_cost = r.sah_cost;
    h.max_depth = r.max_depth;
    h.max_leaf = r.max_leaf;
    h.checksum = arrays_checksum(r);
    const std::string tmp = path + ".tmp" + std::to_string((unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;   // an unwritable cache directory is not an error: the tree was built
    const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(r.nodes.data(), 4, r.nodes.size(), f) == r.nodes.size() &&
                    fwrite(r.nodes16.data(), 4, r.nodes16.size(), f) == r.nodes16.size() && fwrite(r.tris.data(), 4, r.tris.size(), f) == r.tris.size() &&
                    fwrite(r.prim_index.data(), 4, r.prim_index.size(), f) == r.prim_index.size();
    const bool closed = fclose(f) == 0;
    if (!ok || !closed || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
}

}  // namespace

void build_cached(const std::vector<InputTriangle>& tris, Result& out, const std::string& dir, bool* from_cache_out) {
    if (from_cache_out) *from_cache_out = false;
    if (dir.empty()) { build(tris, out); return; }
    if (const char* e = getenv("TRH_SAH_CT")) { const float v = (float)atof(e); if (v > 0.f) kCostTraverse = v; }   // part of the key
    const uint64_t key = cache_key(tris);
    char name[40];
    snprintf(name, sizeof(name), "/%016llx.trhbvh", (unsigned long long)key);
    const std::string path = dir + name;
    if (cache_load(path, key, tris.size(), out)) {
        // the checksum guards against a damaged file, not against a wrong one: links and leaf ranges go to the GPU
        // unchecked by the kernels, so a loaded tree passes the same structural check as the tests' before it is used
        std::string why;
        if (validate(tris, out, why, nullptr) == 0) {
            if (from_cache_out) *from_cache_out = true;
            return;
        }
    }
    build(tris, out);
    cache_store(path, key, out);
}

uint64_t validate(const std::vector<InputTriangle>& tris, const Result& r, std::string& first, uint64_t* hash_out) {
    uint64_t bad = 0;
    auto fail = [&](const std::string& what) { if (bad++ == 0) first = what; };
    const size_t n_nodes = r.nodes.size() / 16, n_prims = r.prim_index.size();
    // This check is also the gate between an on-disk cache file and the GPU (build_cached), so it must be memory-safe
    // on ANY input: wrong array sizes or a permutation entry out of range end it at once -- everything below indexes
    // the arrays through those.
    if (r.nodes.size() % 16 != 0) { fail("node array size"); return bad; }
    if (r.nodes16.size() != n_nodes * 8) { fail("nodes16 size"); return bad; }
    if (r.tris.size() != n_prims * 12 || n_prims != tris.size()) { fail("triangle array size"); return bad; }
    std::vector<uint8_t> seen_prim(tris.size(), 0), seen_leaf_pos(n_prims, 0), seen_node(n_nodes, 0);
    for (uint32_t p : r.prim_index) {
        if (p >= tris.size()) { fail("prim_index out of range"); return bad; }
        if (seen_prim[p]++) fail("triangle stored twice");
    }
    for (size_t i = 0; i < tris.size(); ++i) if (!seen_prim[i]) { fail("triangle missing from the tree"); break; }
    // the frame of the fp16 records: a finite centre and a finite positive power-of-two scale (the kernels divide by it)
    {
        int e = 0;
        const bool scale_ok = std::isfinite(r.frame_scale) && r.frame_scale > 0.f && std::frexp(r.frame_scale, &e) == 0.5f;
        const bool centre_ok = std::isfinite(r.frame_center[0]) && std::isfinite(r.frame_center[1]) && std::isfinite(r.frame_center[2]);
        if (!scale_ok || !centre_ok) { fail("frame of the fp16 records is not finite / not a power of two"); return bad; }
    }
    const double s = r.frame_scale;
    struct Item { int node; int depth; };
    std::vector<Item> todo;
    if (n_nodes) todo.push_back({0, 1});
    uint32_t max_depth = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.node < 0 || (size_t)it.node >= n_nodes) { fail("child link out of range"); continue; }
        if (seen_node[(size_t)it.node]++) { fail("record reachable twice"); continue; }
        max_depth = std::max(max_depth, (uint32_t)it.depth);
        const float* f = &r.nodes[(size_t)it.node * 16];
        const uint32_t* q = &r.nodes16[(size_t)it.node * 8];
        const int fi[2][6] = {{0, 1, 2, 3, 8, 9}, {4, 5, 6, 7, 10, 11}};
        const int qi[2][3] = {{0, 1, 4}, {2, 3, 5}};
        for (int c = 0; c < 2; ++c) {
            int link;
            memcpy(&link, &f[12 + c], 4);
            int link16;
            memcpy(&link16, &q[6 + c], 4);
            if (link != link16) fail("fp16 record link differs from the fp32 record");
            if (link == -1) fail("link -1 is reserved");
            const bool sentinel = std::fabs(f[fi[c][0]]) >= 1e30f;
            if (!sentinel) {
                for (int a = 0; a < 3; ++a) {
                    // containment in the frame the kernel works in: x' = (x - c) * s with the float centre c
                    const double lo = (double)half_to_float((uint16_t)(q[qi[c][a]] & 0xffffu)), hi = (double)half_to_float((uint16_t)(q[qi[c][a]] >> 16));
                    const double flo = ((double)f[fi[c][2 * a]] - (double)r.frame_center[a]) * s, fhi = ((double)f[fi[c][2 * a + 1]] - (double)r.frame_center[a]) * s;
                    if (lo > flo || hi < fhi) fail("fp16 box does not contain the fp32 box");
                }
            }
            if (link >= 0) {
                todo.push_back({link, it.depth + 1});
            } else {
                const uint32_t bits = ~(uint32_t)link, first_p = bits >> 4, count = bits & 15u;
                if (count > (uint32_t)kMaxLeafSize) fail("leaf larger than kMaxLeafSize");
                for (uint32_t k = 0; k < count; ++k) {
                    const size_t pos = (size_t)first_p + k;
                    if (pos >= n_prims) { fail("leaf range out of bounds"); break; }
                    if (seen_leaf_pos[pos]++) fail("leaf ranges overlap");
                    const InputTriangle& t = tris[r.prim_index[pos]];
                    const float* vs[3] = {t.v0, t.v1, t.v2};
                    for (const float* v : vs)
                        for (int a = 0; a < 3; ++a)
                            if (v[a] < f[fi[c][2 * a]] || v[a] > f[fi[c][2 * a + 1]]) fail("leaf box does not contain its triangle");
                    const float* rec = &r.tris[pos * 12];
                    uint32_t id;
                    memcpy(&id, &rec[3], 4);
                    if (id != t.packed_id) fail("triangle record id mismatch");
                    // the record must BE this triangle (a stale or foreign cache file with a matching key would otherwise
                    // render other geometry without complaint): v0, v1 - v0, v2 - v0 exactly as the builders write them
                    for (int a = 0; a < 3; ++a)
                        if (rec[a] != t.v0[a] || rec[4 + a] != t.v1[a] - t.v0[a] || rec[8 + a] != t.v2[a] - t.v0[a]) { fail("triangle record geometry mismatch"); break; }
                }
            }
        }
    }
    for (size_t i = 0; i < n_prims; ++i) if (!seen_leaf_pos[i]) { fail("triangle slot not covered by a leaf"); break; }
    for (size_t i = 0; i < n_nodes; ++i) if (!seen_node[i]) { fail("record unreachable"); break; }
    if (max_depth > (uint32_t)kMaxDepth) fail("tree deeper than kMaxDepth");
    if (max_depth != r.max_depth && !tris.empty()) fail("max_depth field wrong");
    if (hash_out) {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&h](const void* p, size_t bytes) {
            const uint8_t* b = (const uint8_t*)p;
            for (size_t i = 0; i < bytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
        };
        mix(r.nodes.data(), r.nodes.size() * 4);
        mix(r.nodes16.data(), r.nodes16.size() * 4);
        mix(r.tris.data(), r.tris.size() * 4);
        *hash_out = h;
    }
    return bad;
}

float cost_traverse() {
    if (const char* e = getenv("TRH_SAH_CT")) { const float v = (float)atof(e); if (v > 0.f) return v; }   // development knob, as in build()
    return kCostTraverse;
}

}  // namespace bvh
