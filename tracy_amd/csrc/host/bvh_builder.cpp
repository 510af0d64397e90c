// bvh_builder.cpp -- binned-SAH BVH2 build, see bvh_builder.h for the layout.
#include "bvh_builder.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

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

float kCostTraverse = 0.5f;   // one inner-record step (two slab tests)
constexpr float kCostIntersect = 1.0f;   // one Moller-Trumbore test

// ---- IEEE binary16 with directed rounding (host only; g++ 11 has no _Float16 on x86)
float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 1024u), (int)e - 25);
    return sign ? -v : v;
}
uint16_t half_next_down(uint16_t h) {   // next representable value toward -inf (h finite or +inf)
    if (h == 0x0000u) return 0x8001u;
    return (h & 0x8000u) ? (uint16_t)(h + 1u) : (uint16_t)(h - 1u);
}
uint16_t half_next_up(uint16_t h) {     // toward +inf (h finite or -inf)
    if (h == 0x8000u) return 0x0001u;
    return (h & 0x8000u) ? (uint16_t)(h - 1u) : (uint16_t)(h + 1u);
}
uint16_t half_nearest(double v) {       // any rounding will do: the callers fix the direction afterwards
    if (!(v == v)) return 0x7E00u;
    const uint16_t sign = std::signbit(v) ? 0x8000u : 0u;
    double a = std::fabs(v);
    if (a >= 65520.0) return sign | 0x7C00u;
    if (a < std::ldexp(1.0, -25)) return sign;
    int ex;
    (void)std::frexp(a, &ex);           // a = f * 2^ex, f in [0.5, 1)
    int e = ex - 1 + 15;                // biased exponent of the leading bit
    if (e <= 0) {                       // subnormal: units of 2^-24
        const uint32_t q = (uint32_t)std::lround(std::ldexp(a, 24));
        return sign | (uint16_t)q;      // q == 1024 rolls over into the smallest normal, as it should
    }
    uint32_t q = (uint32_t)std::lround(std::ldexp(a, 10 - (ex - 1)));  // in [1024, 2048]
    if (q == 2048u) { q = 1024u; ++e; }
    if (e >= 31) return sign | 0x7C00u;
    return sign | (uint16_t)((uint32_t)e << 10) | (uint16_t)(q - 1024u);
}
uint16_t half_floor(double v) {         // largest half <= v (v > -65504 assumed; -inf otherwise)
    if (v >= 65504.0) return 0x7BFFu;
    if (v < -65504.0) return 0xFC00u;
    uint16_t h = half_nearest(v);
    while ((double)half_to_float(h) > v) h = half_next_down(h);
    return h;
}
uint16_t half_ceil(double v) {          // smallest half >= v
    if (v <= -65504.0) return 0xFBFFu;
    if (v > 65504.0) return 0x7C00u;
    uint16_t h = half_nearest(v);
    while ((double)half_to_float(h) < v) h = half_next_up(h);
    return h;
}

int levels_needed(size_t n) {  // inner-record levels a median-split subtree of n prims needs
    int l = 0;
    while (n > (size_t)kMaxLeafSize) { n = (n + 1) / 2; ++l; }
    return l;
}

struct Builder {
    const std::vector<InputTriangle>& in;
    std::vector<Ref> refs;
    Result& out;
    float root_area = 1.f;
    float scene_extent = 0.f;

    Builder(const std::vector<InputTriangle>& t, Result& r) : in(t), out(r) {}

    int make_leaf(size_t begin, size_t end, const Box& box) {
        const uint32_t first = (uint32_t)out.prim_index.size();
        const uint32_t count = (uint32_t)(end - begin);
        for (size_t i = begin; i < end; ++i) {
            const InputTriangle& t = in[refs[i].idx];
            float rec[12];
            for (int a = 0; a < 3; ++a) {
                rec[a] = t.v0[a];
                rec[4 + a] = t.v1[a] - t.v0[a];  // same subtraction RayTriangle does (collision.h:35-36)
                rec[8 + a] = t.v2[a] - t.v0[a];
            }
            memcpy(&rec[3], &t.packed_id, 4);
            rec[7] = 0.f;
            rec[11] = 0.f;
            out.tris.insert(out.tris.end(), rec, rec + 12);
            out.prim_index.push_back(refs[i].idx);
        }
        out.max_leaf = std::max(out.max_leaf, count);
        out.sah_cost += (double)box.half_area() / root_area * count * kCostIntersect;
        return ~(int)((first << 4) | count);
    }

    // returns child link; box = bounds of [begin,end)
    int build(size_t begin, size_t end, const Box& box, int depth /* depth of the record this range would become */) {
        const size_t n = end - begin;
        Box cb;
        for (size_t i = begin; i < end; ++i) cb.grow(refs[i].c);

        // --- find the best binned SAH split over the three axes
        float best_cost = FLT_MAX;
        int best_axis = -1, best_bin = -1;
        if (n > 1) {
            for (int axis = 0; axis < 3; ++axis) {
                const float lo = cb.mn[axis], ext = cb.mx[axis] - cb.mn[axis];
                if (!(ext > 0.f)) continue;
                Box bb[kSahBins];
                uint32_t cnt[kSahBins] = {0};
                const float scale = kSahBins / ext;
                for (size_t i = begin; i < end; ++i) {
                    int b = std::min(kSahBins - 1, std::max(0, (int)((refs[i].c[axis] - lo) * scale)));
                    bb[b].grow(refs[i].box);
                    ++cnt[b];
                }
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
        if (n <= (size_t)kMaxLeafSize && (best_axis < 0 || leaf_cost <= split_cost || n == 1 || depth > kMaxDepth)) return make_leaf(begin, end, box);

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

        const int me = (int)(out.nodes.size() / 16);
        out.nodes.resize(out.nodes.size() + 16, 0.f);
        out.max_depth = std::max(out.max_depth, (uint32_t)depth);
        out.sah_cost += (double)box.half_area() / root_area * kCostTraverse;
        const int c0 = build(begin, mid, lb, depth + 1);
        const int c1 = build(mid, end, rb, depth + 1);
        write_record(me, lb, rb, c0, c1);
        return me;
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
        double area_exact = 0.0, area_quant = 0.0;
        auto half_area = [](double dx, double dy, double dz) { return (dx < 0 || dy < 0 || dz < 0) ? 0.0 : dx * dy + dy * dz + dz * dx; };
        for (size_t i = 0; i < n; ++i) {
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
        out.tris.reserve(in.size() * 12);
        out.prim_index.reserve(in.size());
        const int empty_leaf = ~(1 << 4);  // first 1, count 0 (never -1: the kernels read a link of -1 as "nothing to traverse")
        if (in.empty()) {
            out.nodes.resize(16, 0.f);
            write_record(0, Box{}, Box{}, empty_leaf, empty_leaf);
            return;
        }
        int root = build(0, in.size(), all, 1);
        if (root < 0) {
            // the whole scene fits one leaf: wrap it in a root record (traversal always starts at record 0)
            out.nodes.resize(16, 0.f);
            Box none;
            none.mn[0] = none.mn[1] = none.mn[2] = 0.f;
            none.mx[0] = none.mx[1] = none.mx[2] = 0.f;
            write_record(0, all, none, root, empty_leaf);
            out.max_depth = 1;
        }
    }
};

}  // namespace

void build(const std::vector<InputTriangle>& tris, Result& out) {
    out = Result{};
    if (const char* e = getenv("TRH_SAH_CT")) { const float v = (float)atof(e); if (v > 0.f) kCostTraverse = v; }   // development knob
    Builder b(tris, out);
    b.run();
    b.quantise();
}

}  // namespace bvh
