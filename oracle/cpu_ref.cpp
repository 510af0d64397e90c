// cpu_ref.cpp -- CPU ORACLE for the Tracy per-sample path-tracing hot path.
//
// *** TEST INFRASTRUCTURE, NOT PRODUCT. ***  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this library. The HIP module never links or calls it.
//
// What it is: an independent C++17 restatement of the reference CPU/OpenMP render module
// (carcass82/tracy), following these reference files line by line:
//   CpuTrace::OnUpdate / Trace              src/kernels/raytracing/software/cpu_trace.cpp:33-170
//   CPUDetails::ProcessScene / ComputeIntersection / UpdateOutput
//                                           src/kernels/raytracing/software/cpu_details.cpp:26-199
//   Tri / Obj                               src/kernels/raytracing/software/cpu_details.h:28-66
//   kd-tree build / flatten / traversal     src/kdtree.h:160-429   (DISABLE_SAH, the CMake default)
//   RayTriangle / RayAABB / TriangleAABB    src/collision.h:33-202
//   Material::Scatter + getters + helpers   src/material.h:125-268
//   Texture::GetPixel                       src/texture.h:50-57
//   Camera::GetRayFrom, Ray                 src/camera.h:28-35, src/ray.h:15-19
//   fastrand (PCG)                          src/random.h:66-77
//   BBox                                    src/aabb.h
//
// PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this path
// (appveyor.yml:11-25 is compile-only) and cannot be compiled here: its math library
// ext/cclib is an empty, un-vendored submodule (.gitmodules:1-3) and glm is not installed.
// The semantics of the missing math functions are the ones the reference's interchangeable
// glm branch documents (src/common.h:100-166); see DESIGN.md "Decreed arithmetic". The only
// exact pin available is the integer PCG generator (tests/test_oracle_rng.py recomputes it
// with Python integers straight from src/random.h:71-77).
//
// Deliberate re-definition (documented): the reference shares ONE
// static u32 RNG state between all OpenMP threads (cpu_trace.cpp:42), so its output is a data
// race. Here the state of every camera path is derived from (pixel, sample, seed) and then
// advanced with the reference's exact fastrand. oref_render_serial_rng reproduces the literal
// single-stream, x-major, single-threaded behaviour for reference.
//
// Floating point: compiled with -ffp-contract=off; the ONLY fused operations are the explicit
// fmaf() calls in dot_f/cross_f used by RayTriangle (the reference is built with -Ofast
// -march=native, so where its compiler fuses is unspecified; this is the form the HIP kernel
// also uses, which makes t/u/v and therefore hit ids bit-comparable).
#include "../include/tracy_hip.h"

#include <cmath>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <vector>
#include <memory>
#include <algorithm>
#if defined(_OPENMP)
#include <omp.h>
#endif

// TEST-ONLY switch (oref_set_matchance0): the floor of materialspecularchance, 0.1 in the reference (material.h:253).
// Setting it to 0 removes the 10 % white specular lobe of non-metals -- the hypothesis DESIGN.md section 2 tests against
// the reference's doc/dragon.jpg screenshot. Never touched by any other test or by the product.
static float g_matchance_floor = .1f;

namespace {

using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;

constexpr float PI = 3.1415926535897932f;
constexpr float EPS = 1.e-8f;

struct vec2 { float x = 0.f, y = 0.f; };
struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    vec3() = default;
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float a) : x(a), y(a), z(a) {}
    float& operator[](u32 i) { return (&x)[i]; }
    float operator[](u32 i) const { return (&x)[i]; }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(const vec3& a, const vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(const vec3& a, const vec3& b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(const vec3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, const vec3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(const vec3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator+(const vec3& a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline vec3 operator-(const vec3& a, float s) { return {a.x - s, a.y - s, a.z - s}; }
inline vec3& operator+=(vec3& a, const vec3& b) { a = a + b; return a; }
inline vec3& operator*=(vec3& a, const vec3& b) { a = a * b; return a; }
inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
inline vec2 operator+(const vec2& a, const vec2& b) { return {a.x + b.x, a.y + b.y}; }
inline vec2 operator*(float s, const vec2& a) { return {s * a.x, s * a.y}; }

inline float fmax_(float a, float b) { return (a < b) ? b : a; }  // glm::max
inline float fmin_(float a, float b) { return (b < a) ? b : a; }  // glm::min
inline vec3 pmin(const vec3& a, const vec3& b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
inline vec3 pmax(const vec3& a, const vec3& b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline float rcp(float x) { return 1.f / x; }
inline vec3 rcp(const vec3& v) { return {1.f / v.x, 1.f / v.y, 1.f / v.z}; }
inline float clampf(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
inline float frac(float x) { return x - std::floor(x); }
inline float lerp(float x, float y, float a) { return x * (1.f - a) + y * a; }
inline vec3 lerp(const vec3& x, const vec3& y, float a) { return x * (1.f - a) + y * a; }

// plain (unfused) products, used everywhere except RayTriangle
inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3& a, const vec3& b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline vec3 normalize(const vec3& v) { return v * (1.f / std::sqrt(dot(v, v))); }
// GLSL reflect / refract as glm writes them
inline vec3 reflect(const vec3& I, const vec3& N) { return I - N * dot(N, I) * 2.f; }
inline vec3 refract(const vec3& I, const vec3& N, float eta) {
    const float d = dot(N, I);
    const float k = 1.f - eta * eta * (1.f - d * d);
    return (k >= 0.f) ? (eta * I - (eta * d + std::sqrt(k)) * N) : vec3(0.f);
}
// fused forms: the decreed RayTriangle arithmetic (see file header)
inline float dot_f(const vec3& a, const vec3& b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
inline vec3 cross_f(const vec3& a, const vec3& b) {
    return {std::fmaf(a.y, b.z, -(b.y * a.z)), std::fmaf(a.z, b.x, -(b.z * a.x)), std::fmaf(a.x, b.y, -(b.x * a.y))};
}

// ---------------------------------------------------------------- src/aabb.h
struct BBox {
    vec3 minbound{FLT_MAX}, maxbound{-FLT_MAX};
    BBox() = default;
    BBox(const vec3& a, const vec3& b) : minbound(a), maxbound(b) {}
    vec3 GetCenter() const { return (minbound + maxbound) * .5f; }
    vec3 GetSize() const { return maxbound - minbound; }
    bool Contains(const vec3& p) const {
        return p.x >= minbound.x && p.x <= maxbound.x && p.y >= minbound.y && p.y <= maxbound.y && p.z >= minbound.z && p.z <= maxbound.z;
    }
    void Extend(const BBox& b) { minbound = pmin(minbound, b.minbound); maxbound = pmax(maxbound, b.maxbound); }
};

// ---------------------------------------------------------------- src/ray.h:15-19
struct Ray {
    vec3 origin, direction, inv_direction;
    Ray() = default;
    Ray(const vec3& o, const vec3& d) : origin(o), direction(d), inv_direction(rcp(d)) {}
    vec3 GetPoint(float t) const { return origin + t * direction; }
};

// ---------------------------------------------------------------- src/common.h:237-247
struct HitData {
    u32 object_index = 0, triangle_index = 0;
    float t = 0.f;
    vec2 uv;
    vec3 point, normal, tangent;
    u32 material = 0;
};

// ---------------------------------------------------------------- src/collision.h:33-74
struct TriangleHitData { float RayT; vec2 TriangleUV; };

inline bool RayTriangle(const vec3& ray_origin, const vec3& ray_direction, const vec3& v0, const vec3& v1, const vec3& v2, TriangleHitData& inout_hit) {
    const vec3 v0v1 = v1 - v0;
    const vec3 v0v2 = v2 - v0;
    const vec3 pvec = cross_f(ray_direction, v0v2);
    const vec3 tvec = ray_origin - v0;
    const float det = dot_f(v0v1, pvec);
    if (det < EPS) return false;  // parallel or back facing
    const float u = dot_f(tvec, pvec);
    if (u < EPS || u > det) return false;
    const vec3 qvec = cross_f(tvec, v0v1);
    const float v = dot_f(ray_direction, qvec);
    if (v < EPS || u + v > det) return false;
    const float t = dot_f(v0v2, qvec) * rcp(det);
    if (t < EPS || t > inout_hit.RayT) return false;
    inout_hit.RayT = t;
    inout_hit.TriangleUV = vec2{u * rcp(det), v * rcp(det)};
    return true;
}

// ---------------------------------------------------------------- src/collision.h:119-131
inline bool RayAABB(const vec3& o, const vec3& inv_dir, const vec3& bmin, const vec3& bmax, float in_tmax) {
    const vec3 minbound = (bmin - o) * inv_dir;
    const vec3 maxbound = (bmax - o) * inv_dir;
    const vec3 tmin1 = pmin(minbound, maxbound);
    const vec3 tmax1 = pmax(minbound, maxbound);
    const float tmin = fmax_(tmin1.x, fmax_(tmin1.y, tmin1.z));
    const float tmax = fmin_(tmax1.x, fmin_(tmax1.y, tmax1.z));
    return (tmax >= fmax_(EPS, tmin) && tmin < in_tmax);
}

// ---------------------------------------------------------------- src/collision.h:140-202
inline bool TriangleAABB(const vec3& v0, const vec3& v1, const vec3& v2, const BBox& aabb) {
    const vec3 e0 = v1 - v0, e1 = v2 - v1, e2 = v0 - v2;
    const vec3 fe0(std::fabs(e0.x), std::fabs(e0.y), std::fabs(e0.z));
    const vec3 fe1(std::fabs(e1.x), std::fabs(e1.y), std::fabs(e1.z));
    const vec3 fe2(std::fabs(e2.x), std::fabs(e2.y), std::fabs(e2.z));
    const vec3 h = aabb.GetSize() / 2.f;
    auto AxisTester = [](float a, float b, float fa, float fb, float v0_0, float v0_1, float v1_0, float v1_1, float h0, float h1) {
        float p0 = a * v0_0 + b * v0_1;
        float p1 = a * v1_0 + b * v1_1;
        float rad = fa * h0 + fb * h1;
        return (fmin_(p0, p1) > rad || fmax_(p0, p1) < -rad);
    };
    if (AxisTester(e0.z, -e0.y, fe0.z, fe0.y, v0.y, v0.z, v2.y, v2.z, h.y, h.z) ||
        AxisTester(-e0.z, e0.x, fe0.z, fe0.x, v0.x, v0.z, v2.x, v2.z, h.x, h.z) ||
        AxisTester(e0.y, -e0.x, fe0.y, fe0.x, v1.x, v1.y, v2.x, v2.y, h.x, h.y) ||
        AxisTester(e1.z, -e1.y, fe1.z, fe1.y, v0.y, v0.z, v2.y, v2.z, h.y, h.z) ||
        AxisTester(-e1.z, e1.x, fe1.z, fe1.x, v0.x, v0.z, v2.x, v2.z, h.x, h.z) ||
        AxisTester(e1.y, -e1.x, fe1.y, fe1.x, v0.x, v0.y, v1.x, v1.y, h.x, h.y) ||
        AxisTester(e2.z, -e2.y, fe2.z, fe2.y, v0.y, v0.z, v1.y, v1.z, h.y, h.z) ||
        AxisTester(-e2.z, e2.x, fe2.z, fe2.x, v0.x, v0.z, v1.x, v1.z, h.x, h.z) ||
        AxisTester(e2.y, -e2.x, fe2.y, fe2.x, v1.x, v1.y, v2.x, v2.y, h.x, h.y))
        return false;
    vec3 trimin = pmin(v0, pmin(v1, v2));
    vec3 trimax = pmax(v0, pmax(v1, v2));
    if ((trimin.x > h.x || trimax.x < -h.x) || (trimin.y > h.y || trimax.y < -h.y) || (trimin.z > h.z || trimax.z < -h.z)) return false;
    vec3 n = cross(e0, e1);
    vec3 vmin, vmax;
    vmin.x = (n.x > .0f) ? -h.x - v0.x : h.x - v0.x;
    vmax.x = (n.x > .0f) ? h.x - v0.x : -h.x - v0.x;
    vmin.y = (n.y > .0f) ? -h.y - v0.y : h.y - v0.y;
    vmax.y = (n.y > .0f) ? h.y - v0.y : -h.y - v0.y;
    vmin.z = (n.z > .0f) ? -h.z - v0.z : h.z - v0.z;
    vmax.z = (n.z > .0f) ? h.z - v0.z : -h.z - v0.z;
    return !(dot(n, vmin) > .0f || dot(n, vmax) < .0f);
}

// ---------------------------------------------------------------- src/kdtree.h
constexpr u32 TREE_MAXDEPTH = 32, TREE_MINOBJECTS = 16, TREE_USELESS_SPLIT_THRESHOLD = 200;
constexpr u32 NO_CHILD = UINT32_MAX;

template <typename T>
struct Node {  // kdtree.h:42-79
    Node* children[2]{nullptr, nullptr};  // [Left, Right]
    BBox aabb;
    std::vector<T> elems;
    u32 depth = 0;
    explicit Node(u32 d = 0) : depth(d) {}
    ~Node() { delete children[0]; delete children[1]; }
    bool IsEmpty() const { return elems.empty(); }
    u32 GetSize() const { return (u32)elems.size(); }
};

struct FlatNode {  // kdtree.h:84-119
    BBox aabb;
    u32 first = 0, last = 0;
    u32 children[2]{NO_CHILD, NO_CHILD};  // [Left, Right]
    bool IsEmpty() const { return first == last; }
};

template <typename T>
struct FlatTree {  // kdtree.h:124-134
    std::vector<FlatNode> nodes;
    std::vector<T> elements;
    const FlatNode* GetChild(u32 idx) const { return (idx < nodes.size()) ? &nodes[idx] : nullptr; }
};

// kdtree.h:160-237, DISABLE_SAH branch
template <typename T, class Tester>
u32 SplitAndGetDuplicationPercentage(const Node<T>& cur, Node<T>& right, Node<T>& left, const Tester& ObjectBoxTester) {
    constexpr float ROUND = 1.e-4f;
    const u32 axis = cur.depth % 3;
    const float split = (cur.aabb.GetSize()[axis] / 2.f) - ROUND;
    BBox left_bbox(cur.aabb);
    left_bbox.maxbound[axis] -= split;
    left.aabb = left_bbox;
    BBox right_bbox(cur.aabb);
    right_bbox.minbound[axis] += split;
    right.aabb = right_bbox;
    for (const T& e : cur.elems) if (ObjectBoxTester(e, right.aabb)) right.elems.push_back(e);
    for (const T& e : cur.elems) if (ObjectBoxTester(e, left.aabb)) left.elems.push_back(e);
    return (right.GetSize() + left.GetSize()) * 100 / cur.GetSize();
}

// kdtree.h:243-282
template <typename T, class Tester>
void BuildTree(Node<T>* tree, const Tester& ObjectBoxTester) {
    if (tree->IsEmpty()) return;
    std::vector<Node<T>*> build_queue{tree};
    while (!build_queue.empty()) {
        Node<T>* cur = build_queue.back();
        build_queue.pop_back();
        if (cur->GetSize() > TREE_MINOBJECTS && cur->depth < TREE_MAXDEPTH) {
            cur->children[1] = new Node<T>(cur->depth + 1);
            cur->children[0] = new Node<T>(cur->depth + 1);
            if (SplitAndGetDuplicationPercentage(*cur, *cur->children[1], *cur->children[0], ObjectBoxTester) < TREE_USELESS_SPLIT_THRESHOLD) {
                build_queue.push_back(cur->children[1]);
                build_queue.push_back(cur->children[0]);
                std::vector<T>().swap(cur->elems);
            } else {
                delete cur->children[1]; cur->children[1] = nullptr;
                delete cur->children[0]; cur->children[0] = nullptr;
            }
        }
    }
}

// kdtree.h:288-354 (an element-less node is laid out as an inner node even if it has no children:
// its two flat children are then default nodes with an inverted box, first == last, no children)
template <typename T>
void FlattenTree(Node<T>& src, FlatTree<T>& out) {
    out.nodes.clear();
    out.elements.clear();
    std::vector<std::pair<u32, Node<T>*>> build_queue;
    out.nodes.push_back(FlatNode());
    build_queue.emplace_back(0u, &src);
    while (!build_queue.empty()) {
        auto cur = build_queue.back();
        build_queue.pop_back();
        if (!cur.second) continue;
        out.nodes[cur.first].aabb = cur.second->aabb;
        if (cur.second->IsEmpty()) {
            u32 right_child = (u32)out.nodes.size();
            out.nodes[cur.first].children[1] = right_child;
            out.nodes.push_back(FlatNode());
            build_queue.emplace_back(right_child, cur.second->children[1]);
            u32 left_child = (u32)out.nodes.size();
            out.nodes[cur.first].children[0] = left_child;
            out.nodes.push_back(FlatNode());
            build_queue.emplace_back(left_child, cur.second->children[0]);
        } else {
            out.nodes[cur.first].first = (u32)out.elements.size();
            out.elements.insert(out.elements.end(), cur.second->elems.begin(), cur.second->elems.end());
            out.nodes[cur.first].last = (u32)out.elements.size();
        }
    }
}

struct TraversalCounters { u64 box = 0, tri = 0; };

// kdtree.h:382-429
template <typename T, class RayTester>
bool IntersectsWithTree(const FlatTree<T>& tree, const Ray& ray, HitData& inout, const RayTester& ObjectTester, TraversalCounters* cnt) {
    const FlatNode* stack[TREE_MAXDEPTH + 1];
    i32 head = -1;
    const vec3 rayO = ray.origin, rayI = ray.inv_direction;
    float minT = inout.t;
    bool hit_something = false;
    const FlatNode* current = tree.GetChild(0);
    while (current || head != -1) {
        while (current && ((cnt ? ++cnt->box : 0), RayAABB(rayO, rayI, current->aabb.minbound, current->aabb.maxbound, minT))) {
            stack[++head] = current;
            current = tree.GetChild(current->children[0]);
        }
        if (head != -1) {
            current = stack[head--];
            if (!current->IsEmpty() && ObjectTester(tree.elements.data(), current->first, current->last, ray, inout)) {
                minT = inout.t;
                hit_something = true;
            }
        }
        current = tree.GetChild(current->children[1]);
    }
    return hit_something;
}

// ---------------------------------------------------------------- cpu_details.h:28-66
struct Tri {
    u32 packed_tri_info = 0;
    vec3 vertices[3];
    Tri() = default;
    Tri(u32 mesh_idx, u32 tri_idx, const vec3& a, const vec3& b, const vec3& c) : packed_tri_info((mesh_idx << 24) | tri_idx), vertices{a, b, c} {}
    u32 GetMeshId() const { return packed_tri_info >> 24; }
    u32 GetTriangleId() const { return packed_tri_info & 0xffffff; }
};
struct Obj { u32 object_id = 0; };

// ---------------------------------------------------------------- src/random.h:66-77
inline float fastrand(u32& ctx) {
    u32 state = ctx;
    ctx = ctx * 747796405u + 2891336453u;
    u32 word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return float((word >> 22u) ^ word) / static_cast<float>(UINT32_MAX);
}
inline u32 pcg_hash(u32 v) {  // one fastrand step as an integer hash (same constants)
    u32 state = v * 747796405u + 2891336453u;
    u32 word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// per-path RNG state (the documented re-definition of the reference's shared static state)
inline u32 path_rng_state(u32 pixel_index, u32 sample_index, u64 seed) {
    const u32 seed32 = (u32)seed ^ ((u32)(seed >> 32) * 0x9E3779B9u);
    return pcg_hash(pixel_index ^ pcg_hash(sample_index ^ pcg_hash(seed32)));
}

struct Vertex { vec3 pos, normal; vec2 uv0; vec3 tangent, bitangent; };
static_assert(sizeof(Vertex) == TRH_VERTEX_FLOATS * 4, "vertex layout");

struct Texture {
    i32 width = 0, height = 0;
    std::vector<float> pixels;  // RGBA32F
    // src/texture.h:50-57
    const float* GetPixel(const vec2& uv) const {
        float fi = clampf(frac(uv.x) * width, 0.f, width - 1.f);
        float fj = clampf(frac(1.f - uv.y) * height, 0.f, height - 1.f);
        // float -> u32 of a NaN is undefined in C++; the reference never guards it. Decree: 0.
        u32 i = (fi == fi) ? static_cast<u32>(fi) : 0u;
        u32 j = (fj == fj) ? static_cast<u32>(fj) : 0u;
        return &pixels[((size_t)j * width + i) * 4];
    }
};

}  // namespace

struct oref_scene {
    std::vector<Vertex> vertices;
    std::vector<u32> indices;
    std::vector<trh_mesh> meshes;
    std::vector<trh_material> materials;
    std::vector<Texture> textures;
    u32 width = 0, height = 0, max_bounces = 5, flags = TRH_FLAG_DEFAULT;
    vec3 eye;
    float ivp[16]{};
    // run-time form of the DEBUG_SHOW_* compile-time switches (CMakeLists.txt:23-35): 0 none, 1 base colour,
    // 2 normals, 3 metalness, 4 roughness, 5 emissive; view3 = mat3(Camera::GetView()), column-major
    u32 debug_view = 0;
    float view3[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    // accel (cpu_details.h:101-107)
    FlatTree<Obj> TLAS_tree;
    std::vector<FlatTree<Tri>> BLAS_tree;
    // render state (cpu_details.h:95-99)
    std::vector<vec3> output;
    u64 frame_counter = 0;
    u64 raycount = 0;
    u64 box_tests = 0, tri_tests = 0;
    u32 serial_rng = 0xABCDEFu;  // initrand() (src/random.h:66-69)

    const Vertex& GetVertex(u32 mesh, u32 local_index) const { return vertices[meshes[mesh].first_vertex + local_index]; }
    u32 GetIndex(u32 mesh, u32 i) const { return indices[meshes[mesh].first_index + i]; }
};

namespace {

// cpu_details.cpp:26-86
void ProcessScene(oref_scene& s) {
    auto ObjectAABBTester = [&s](const Obj& o, const BBox& box) {
        const trh_mesh& m = s.meshes[o.object_id];
        BBox mb(vec3(m.aabb_min[0], m.aabb_min[1], m.aabb_min[2]), vec3(m.aabb_max[0], m.aabb_max[1], m.aabb_max[2]));
        return box.Contains(mb.GetCenter());
    };
    auto TriangleAABBTester = [&s](const Tri& t, const BBox& box) {
        const u32 m = t.GetMeshId();
        const vec3 c = box.GetCenter();
        vec3 v0 = s.GetVertex(m, s.GetIndex(m, t.GetTriangleId() * 3 + 0)).pos - c;
        vec3 v1 = s.GetVertex(m, s.GetIndex(m, t.GetTriangleId() * 3 + 1)).pos - c;
        vec3 v2 = s.GetVertex(m, s.GetIndex(m, t.GetTriangleId() * 3 + 2)).pos - c;
        return TriangleAABB(v0, v1, v2, box);
    };
    const u32 nobj = (u32)s.meshes.size();
    s.BLAS_tree.assign(nobj, FlatTree<Tri>());
    Node<Obj> TempObjectsTree;
    BBox scene_bbox;
    for (u32 i = 0; i < nobj; ++i) {
        const trh_mesh& m = s.meshes[i];
        BBox mb(vec3(m.aabb_min[0], m.aabb_min[1], m.aabb_min[2]), vec3(m.aabb_max[0], m.aabb_max[1], m.aabb_max[2]));
        Node<Tri> TempTrianglesTree;
        const u32 ntri = m.index_count / 3;
        TempTrianglesTree.elems.reserve(ntri);
        for (u32 t = 0; t < ntri; ++t) {
            vec3 v0 = s.GetVertex(i, s.GetIndex(i, t * 3 + 0)).pos;
            vec3 v1 = s.GetVertex(i, s.GetIndex(i, t * 3 + 1)).pos;
            vec3 v2 = s.GetVertex(i, s.GetIndex(i, t * 3 + 2)).pos;
            TempTrianglesTree.elems.emplace_back(i, t, v0, v1, v2);
        }
        TempTrianglesTree.aabb = mb;
        BuildTree<Tri>(&TempTrianglesTree, TriangleAABBTester);
        FlattenTree<Tri>(TempTrianglesTree, s.BLAS_tree[i]);
        TempObjectsTree.elems.push_back(Obj{i});
        scene_bbox.Extend(mb);
    }
    TempObjectsTree.aabb = scene_bbox;
    BuildTree<Obj>(&TempObjectsTree, ObjectAABBTester);
    FlattenTree<Obj>(TempObjectsTree, s.TLAS_tree);
}

// cpu_details.cpp:88-185
bool ComputeIntersection(const oref_scene& s, const Ray& ray, HitData& data, TraversalCounters* cnt) {
    const auto TriangleRayTester = [cnt](const Tri* tris, u32 first, u32 last, const Ray& r, HitData& d) {
        bool hit_triangle = false;
        const vec3 rayO = r.origin, rayD = r.direction;
        TriangleHitData th{d.t, vec2{}};
        for (u32 idx = first; idx < last; ++idx) {
            const Tri& tri = tris[idx];
            if (cnt) ++cnt->tri;
            if (RayTriangle(rayO, rayD, tri.vertices[0], tri.vertices[1], tri.vertices[2], th)) {
                d.t = th.RayT;
                d.uv = th.TriangleUV;
                d.object_index = tri.GetMeshId();
                d.triangle_index = tri.GetTriangleId() * 3;
                hit_triangle = true;
            }
        }
        return hit_triangle;
    };
    const auto ObjectRayTester = [&s, &TriangleRayTester, cnt](const Obj* objs, u32 first, u32 last, const Ray& r, HitData& d) {
        bool hit_object = false;
        for (u32 idx = first; idx < last; ++idx) {
            u32 tree_id = objs[idx].object_id;
            if (IntersectsWithTree<Tri>(s.BLAS_tree[tree_id], r, d, TriangleRayTester, cnt)) hit_object = true;
        }
        return hit_object;
    };
    bool hit_any_mesh = IntersectsWithTree<Obj>(s.TLAS_tree, ray, data, ObjectRayTester, cnt);
    if (hit_any_mesh) {
        const u32 m = data.object_index;
        const Vertex& v0 = s.GetVertex(m, s.GetIndex(m, data.triangle_index + 0));
        const Vertex& v1 = s.GetVertex(m, s.GetIndex(m, data.triangle_index + 1));
        const Vertex& v2 = s.GetVertex(m, s.GetIndex(m, data.triangle_index + 2));
        const vec2 uv = data.uv;
        data.point = ray.GetPoint(data.t);
        data.normal = (1.f - uv.x - uv.y) * v0.normal + uv.x * v1.normal + uv.y * v2.normal;
        data.tangent = (1.f - uv.x - uv.y) * v0.tangent + uv.x * v1.tangent + uv.y * v2.tangent;
        data.uv = (1.f - uv.x - uv.y) * v0.uv0 + uv.x * v1.uv0 + uv.y * v2.uv0;
        data.material = s.meshes[m].material;
    }
    return hit_any_mesh;
}

// ---------------------------------------------------------------- src/material.h:125-203
inline float pow2(float x) { return x * x; }
inline float pow5(float x) { const float x2 = x * x; return x2 * x2 * x; }
inline float schlick(float cos, float ref_idx) {
    const float r0 = pow2((1.f - ref_idx) / (1.f + ref_idx));
    return r0 + (1.f - r0) * pow5(1.f - cos);
}
inline vec3 random_on_unit_sphere(u32& rng) {
    const float z = 2.f * fastrand(rng) - 1.f;
    const float phi = 2.f * PI * fastrand(rng);
    const float r = std::sqrt(1.f - pow2(z));
    return vec3{r * std::cos(phi), r * std::sin(phi), z};
}
inline bool HasTexture(const trh_material& m, int id) { return m.tex[id] != TRH_NO_TEXTURE; }
inline vec3 rgb(const float* p) { return vec3(p[0], p[1], p[2]); }
inline vec3 GetBaseColor(const oref_scene& s, const trh_material& m, const HitData& h) {
    return HasTexture(m, TRH_TEX_BASECOLOR) ? rgb(s.textures[m.tex[TRH_TEX_BASECOLOR]].GetPixel(h.uv)) : vec3(m.albedo[0], m.albedo[1], m.albedo[2]);
}
inline float GetRoughness(const oref_scene& s, const trh_material& m, const HitData& h) {
    return HasTexture(m, TRH_TEX_ROUGHNESS) ? s.textures[m.tex[TRH_TEX_ROUGHNESS]].GetPixel(h.uv)[0] : m.roughness;
}
inline float GetMetalness(const oref_scene& s, const trh_material& m, const HitData& h) {
    return HasTexture(m, TRH_TEX_METALNESS) ? s.textures[m.tex[TRH_TEX_METALNESS]].GetPixel(h.uv)[0] : m.metalness;
}
inline vec3 GetEmissive(const oref_scene& s, const trh_material& m, const HitData& h) {
    return HasTexture(m, TRH_TEX_EMISSIVE) ? rgb(s.textures[m.tex[TRH_TEX_EMISSIVE]].GetPixel(h.uv)) : vec3(m.emissive[0], m.emissive[1], m.emissive[2]);
}
inline vec3 GetNormal(const oref_scene& s, const trh_material& m, const HitData& h) {
    if (HasTexture(m, TRH_TEX_NORMAL)) {
        vec3 n = rgb(s.textures[m.tex[TRH_TEX_NORMAL]].GetPixel(h.uv)) * 2.f - 1.f;
        vec3 bitangent = cross(h.normal, normalize(h.tangent - dot(h.tangent, h.normal) * h.normal));
        // mat3 tbn{bitangent, tangent, normal} (columns) times n
        vec3 r = bitangent * n.x + h.tangent * n.y + h.normal * n.z;
        return normalize(r);
    }
    return h.normal;
}

// ---------------------------------------------------------------- src/material.h:210-268
void Scatter(const oref_scene& s, const trh_material& m, const Ray& ray, const HitData& hit, vec3& out_attenuation, vec3& out_emission, Ray& out_scattered, u32& rng) {
    constexpr float kRayOffset = 0.001f;
    const vec3 raydir = ray.direction;
    const vec3 emissive = GetEmissive(s, m, hit);
    const float metalness = GetMetalness(s, m, hit);
    const vec3 basecolor = GetBaseColor(s, m, hit);
    const float roughness = GetRoughness(s, m, hit);
    const vec3 normal = GetNormal(s, m, hit);
    const vec3 scatteredOrigin = hit.point;
    vec3 scatteredDirection, attenuation;
    const float VdotN = dot(raydir, normal);
    const vec3 scattered = normal + random_on_unit_sphere(rng);
    const vec3 reflected = reflect(raydir, normal);
    const vec3 specular = lerp(reflected, scattered, roughness);
    if (m.translucent > EPS) {  // BTDF
        const bool inside = VdotN > EPS;
        const float cosine = inside ? std::sqrt(1.f - pow2(m.ior) * (1.f - pow2(VdotN))) : -VdotN;
        const float ior = inside ? m.ior : rcp(m.ior);
        const vec3 refracted = refract(raydir, normal, ior);
        const vec3 transmitted = lerp(refracted, scattered, roughness);
        const bool is_specular = fastrand(rng) < schlick(cosine, ior);
        scatteredDirection = is_specular ? specular : transmitted;
        attenuation = basecolor;
    } else {  // BRDF
        const vec3 specularcolor = lerp(vec3(.85f), basecolor, metalness);
        const float materialspecularchance = lerp(g_matchance_floor, 1.f, metalness);   // .1f in the reference (material.h:253)
        const float fresnelspecularchance = lerp(materialspecularchance, 1.f, (1.f - roughness) * schlick(-VdotN, 1.f));
        const bool is_specular = fastrand(rng) < fresnelspecularchance;
        scatteredDirection = is_specular ? specular : scattered;
        attenuation = is_specular ? specularcolor : basecolor;
    }
    scatteredDirection = normalize(scatteredDirection);
    out_scattered = Ray(scatteredOrigin + kRayOffset * scatteredDirection, scatteredDirection);
    out_attenuation = attenuation;
    out_emission = emissive;
}

// ---------------------------------------------------------------- src/camera.h:28-35
Ray GetRayFrom(const oref_scene& s, float fs, float ft) {
    const float nx = fs * 2.f - 1.f, ny = ft * 2.f - 1.f, nz = 1.f * 2.f - 1.f;
    const float* m = s.ivp;  // column-major
    float p[4];
    for (int i = 0; i < 4; ++i) {  // glm mat4*vec4: (m0*x + m1*y) + (m2*z + m3*w)
        float a0 = m[0 * 4 + i] * nx + m[1 * 4 + i] * ny;
        float a1 = m[2 * 4 + i] * nz + m[3 * 4 + i] * 1.f;
        p[i] = a0 + a1;
    }
    vec3 q(p[0] / p[3], p[1] / p[3], p[2] / p[3]);
    return Ray(s.eye, normalize(q - s.eye));
}

// ---------------------------------------------------------------- cpu_trace.cpp:107-170
vec3 Trace(const oref_scene& s, Ray ray, u32& rng, u64& raycount, TraversalCounters* cnt, u32* bounces_out) {
    Ray current_ray = ray;
    vec3 throughput{1.f, 1.f, 1.f};
    vec3 pixel;  // zero
    u32 iters = 0;
    for (u32 t = 0; t < s.max_bounces; ++t) {
        ++raycount;
        ++iters;
        HitData hit;
        hit.t = FLT_MAX;
        vec3 attenuation, emission;
        if (ComputeIntersection(s, current_ray, hit, cnt)) {
            if (s.debug_view) {   // cpu_trace.cpp:127-137: the path ends at its first hit and returns one material input
                const trh_material& m = s.materials[hit.material];
                if (bounces_out) *bounces_out = iters;
                switch (s.debug_view) {
                case 1: return GetBaseColor(s, m, hit);
                case 2: {   // .5f * normalize((1.f + mat3(view) * normal))
                    const vec3 n = GetNormal(s, m, hit);
                    const float* v = s.view3;   // glm mat3 * vec3: (col0 * x + col1 * y) + col2 * z
                    const vec3 vn{(v[0] * n.x + v[3] * n.y) + v[6] * n.z, (v[1] * n.x + v[4] * n.y) + v[7] * n.z, (v[2] * n.x + v[5] * n.y) + v[8] * n.z};
                    return 0.5f * normalize(vec3{1.f + vn.x, 1.f + vn.y, 1.f + vn.z});
                }
                case 3: { const float x = GetMetalness(s, m, hit); return vec3{x, x, x}; }
                case 4: { const float x = GetRoughness(s, m, hit); return vec3{x, x, x}; }
                default: return GetEmissive(s, m, hit);
                }
            }
            Ray next;
            Scatter(s, s.materials[hit.material], current_ray, hit, attenuation, emission, next, rng);
            current_ray = next;
            pixel += emission * throughput;
            throughput *= attenuation;
        } else {
            const vec3 v = current_ray.direction;
            hit.uv = vec2{std::atan2(v.z, v.x) / (2 * PI) + 0.5f, std::asin(v.y) / PI + 0.5f};
            emission = GetEmissive(s, s.materials[0], hit);
            pixel += emission * throughput;
            break;
        }
        if (s.flags & TRH_FLAG_RUSSIAN_ROULETTE) {
            float p = EPS + fmax_(throughput.x, fmax_(throughput.y, throughput.z));
            if (fastrand(rng) > p) break;
            throughput *= rcp(p);
        }
    }
    if (bounces_out) *bounces_out = iters;
    return pixel;
}

}  // namespace

// ======================================================================================
// C ABI of the oracle (used through ctypes by tests/ and bench.py's cpu_baseline leg)
// ======================================================================================
extern "C" {

oref_scene* oref_create(const trh_scene_desc* d) {
    if (!d) return nullptr;
    auto* s = new oref_scene();
    s->vertices.resize(d->vertex_count);
    if (d->vertex_count) memcpy(static_cast<void*>(s->vertices.data()), d->vertices, (size_t)d->vertex_count * sizeof(Vertex));
    s->indices.assign(d->indices, d->indices + d->index_count);
    s->meshes.assign(d->meshes, d->meshes + d->mesh_count);
    s->materials.assign(d->materials, d->materials + d->material_count);
    s->textures.resize(d->texture_count);
    for (u32 i = 0; i < d->texture_count; ++i) {
        s->textures[i].width = (i32)d->textures[i].width;
        s->textures[i].height = (i32)d->textures[i].height;
        s->textures[i].pixels.assign(d->textures[i].rgba, d->textures[i].rgba + (size_t)d->textures[i].width * d->textures[i].height * 4);
    }
    s->width = d->width;
    s->height = d->height;
    s->max_bounces = d->max_bounces;
    s->flags = d->flags;
    s->output.assign((size_t)s->width * s->height, vec3());
    ProcessScene(*s);
    return s;
}

void oref_destroy(oref_scene* s) { delete s; }

void oref_set_camera(oref_scene* s, const float eye[3], const float inv_view_proj[16]) {
    s->eye = vec3(eye[0], eye[1], eye[2]);
    memcpy(s->ivp, inv_view_proj, sizeof(s->ivp));
}

void oref_set_matchance0(int on) { g_matchance_floor = on ? 0.f : .1f; }

void oref_set_debug_view(oref_scene* s, int view, const float view3x3[9]) {
    s->debug_view = (u32)view;
    if (view3x3) memcpy(s->view3, view3x3, sizeof(s->view3));
}

// TracyEvent::eCameraCut (cpu_trace.cpp:76-78). The reference keeps output[] and only resets the
// blend counter (frame 0 then overwrites it with blend factor 0).
void oref_reset(oref_scene* s) { s->frame_counter = 0; }

// kd-tree size report: nodes / elements of TLAS and all BLAS (duplicates included)
void oref_tree_info(const oref_scene* s, uint64_t* tlas_nodes, uint64_t* blas_nodes, uint64_t* blas_elements) {
    uint64_t n = 0, e = 0;
    for (auto& t : s->BLAS_tree) { n += t.nodes.size(); e += t.elements.size(); }
    if (tlas_nodes) *tlas_nodes = s->TLAS_tree.nodes.size();
    if (blas_nodes) *blas_nodes = n;
    if (blas_elements) *blas_elements = e;
}

// Parity probe (a): nearest hit of the un-jittered ray through every pixel centre.
void oref_primary_hits(oref_scene* s, uint32_t* ids, float* t_out, float* uv_out) {
    const i32 w = (i32)s->width, h = (i32)s->height;
#pragma omp parallel for schedule(dynamic, 4)
    for (i32 y = 0; y < h; ++y) {
        for (i32 x = 0; x < w; ++x) {
            const i32 idx = y * w + x;
            float u = (x + 0.5f) / float(w);
            float v = (y + 0.5f) / float(h);
            Ray r = GetRayFrom(*s, u, v);
            HitData hit;
            hit.t = FLT_MAX;
            bool any = ComputeIntersection(*s, r, hit, nullptr);
            if (ids) ids[idx] = any ? ((hit.object_index << 24) | (hit.triangle_index / 3)) : 0xffffffffu;
            if (t_out) t_out[idx] = any ? hit.t : FLT_MAX;
            if (uv_out) {
                if (any) {
                    // re-run the winning triangle to recover Moller-Trumbore (u, v)
                    const u32 m = hit.object_index, ti = hit.triangle_index;
                    TriangleHitData th{FLT_MAX, vec2{}};
                    RayTriangle(r.origin, r.direction, s->GetVertex(m, s->GetIndex(m, ti)).pos, s->GetVertex(m, s->GetIndex(m, ti + 1)).pos,
                                s->GetVertex(m, s->GetIndex(m, ti + 2)).pos, th);
                    uv_out[2 * idx] = th.TriangleUV.x;
                    uv_out[2 * idx + 1] = th.TriangleUV.y;
                } else {
                    uv_out[2 * idx] = uv_out[2 * idx + 1] = 0.f;
                }
            }
        }
    }
}

// Parity probe (b): radiance of one sample index for every pixel; .w = bounce iterations.
void oref_sample_radiance(oref_scene* s, uint32_t sample_index, uint64_t seed, float* rgba) {
    const i32 w = (i32)s->width, h = (i32)s->height;
#pragma omp parallel for schedule(dynamic, 4)
    for (i32 y = 0; y < h; ++y) {
        for (i32 x = 0; x < w; ++x) {
            const i32 idx = y * w + x;
            u32 rng = path_rng_state((u32)idx, sample_index, seed);
            float u = (x + fastrand(rng)) / float(w);
            float v = (y + fastrand(rng)) / float(h);
            u64 rc = 0;
            u32 b = 0;
            vec3 c = Trace(*s, GetRayFrom(*s, u, v), rng, rc, nullptr, &b);
            rgba[4 * idx + 0] = c.x; rgba[4 * idx + 1] = c.y; rgba[4 * idx + 2] = c.z; rgba[4 * idx + 3] = float(b);
        }
    }
}

// Diagnostic for the parity tests (no counterpart in the reference): the path of ONE (pixel, sample) bounce by bounce.
// Per bounce iteration 6 floats: packed hit id (mesh << 24 | triangle, as bits; 0xffffffff = miss), t, and -- from a brute-
// force pass over every triangle of the scene with a fresh RayT -- the candidate whose t is nearest to the hit's without
// being the hit itself: its packed id (bits), its t, and the number of candidates within 1e-4 (relative) of the hit's t.
// A path that differs between the module and the oracle on a bounce with such a rival is a "tie" difference (coincident or
// touching faces: collision.h:66 keeps the last-tested one, and traversal order decides); one without is something else.
uint32_t oref_trace_path(oref_scene* s, uint32_t px, uint32_t py, uint32_t sample_index, uint64_t seed, float* records, uint32_t max_records) {
    const u32 idx = py * s->width + px;
    u32 rng = path_rng_state(idx, sample_index, seed);
    const float u = (px + fastrand(rng)) / float(s->width);
    const float v = (py + fastrand(rng)) / float(s->height);
    Ray ray = GetRayFrom(*s, u, v);
    vec3 throughput{1.f, 1.f, 1.f};
    u32 iters = 0;
    for (u32 t = 0; t < s->max_bounces; ++t) {
        HitData hit;
        hit.t = FLT_MAX;
        const bool any = ComputeIntersection(*s, ray, hit, nullptr);
        float* rec = iters < max_records ? records + 6 * iters : nullptr;
        ++iters;
        if (rec) {
            const u32 hid = any ? ((hit.object_index << 24) | (hit.triangle_index / 3)) : 0xffffffffu;
            u32 rival = 0xffffffffu, close = 0;
            float rival_t = 0.f, best = FLT_MAX;
            if (any) {
                for (u32 m = 0; m < (u32)s->meshes.size(); ++m)
                    for (u32 ti = 0; ti < s->meshes[m].index_count; ti += 3) {
                        const u32 id = (m << 24) | (ti / 3);
                        if (id == hid) continue;
                        TriangleHitData th{FLT_MAX, vec2{}};
                        if (!RayTriangle(ray.origin, ray.direction, s->GetVertex(m, s->GetIndex(m, ti)).pos, s->GetVertex(m, s->GetIndex(m, ti + 1)).pos,
                                         s->GetVertex(m, s->GetIndex(m, ti + 2)).pos, th)) continue;
                        const float dt = std::fabs(th.RayT - hit.t);
                        if (dt <= 1e-4f * hit.t) ++close;
                        if (dt < best) { best = dt; rival = id; rival_t = th.RayT; }
                    }
            }
            memcpy(rec + 0, &hid, 4);
            rec[1] = any ? hit.t : 0.f;
            memcpy(rec + 2, &rival, 4);
            rec[3] = rival_t;
            rec[4] = float(close);
            rec[5] = 0.f;
        }
        if (!any) break;
        vec3 attenuation, emission;
        Ray next;
        Scatter(*s, s->materials[hit.material], ray, hit, attenuation, emission, next, rng);
        ray = next;
        throughput *= attenuation;
        if (s->flags & TRH_FLAG_RUSSIAN_ROULETTE) {
            float p = EPS + fmax_(throughput.x, fmax_(throughput.y, throughput.z));
            if (fastrand(rng) > p) break;
            throughput *= rcp(p);
        }
    }
    return iters;
}

// CpuTrace::OnUpdate (cpu_trace.cpp:33-70) called spp_count times: frame f traces sample
// index spp_begin+f of every pixel and blends it with CPUDetails::UpdateOutput
// (cpu_details.cpp:187-199). Pixel loop is x-major with idx = y*w+x as in the reference.
void oref_render(oref_scene* s, uint32_t spp_begin, uint32_t spp_count, uint64_t seed, int count_traversal) {
    const i32 w = (i32)s->width, h = (i32)s->height;
    for (u32 f = 0; f < spp_count; ++f) {
        const u32 sample = spp_begin + f;
        const float blend_factor = s->frame_counter / (s->frame_counter + 1.f);
        u64 rays = 0, boxes = 0, tris = 0;
#pragma omp parallel for collapse(2) schedule(dynamic, 64) reduction(+ : rays, boxes, tris)
        for (i32 x = 0; x < w; ++x) {
            for (i32 y = 0; y < h; ++y) {
                const i32 idx = y * w + x;
                u32 rng = path_rng_state((u32)idx, sample, seed);
                float u = (x + fastrand(rng)) / float(w);
                float v = (y + fastrand(rng)) / float(h);
                TraversalCounters cnt;
                vec3 c = Trace(*s, GetRayFrom(*s, u, v), rng, rays, count_traversal ? &cnt : nullptr, nullptr);
                boxes += cnt.box;
                tris += cnt.tri;
                s->output[idx] = lerp(c, s->output[idx], blend_factor);
            }
        }
        s->raycount += rays;
        s->box_tests += boxes;
        s->tri_tests += tris;
        ++s->frame_counter;  // UpdateBitmap (cpu_details.cpp:215)
    }
}

// The literal reference behaviour: one RNG state for the whole run, seeded 0xABCDEF once and
// never reseeded (cpu_trace.cpp:42), pixels visited x-major. Single-threaded by construction.
void oref_render_serial_rng(oref_scene* s, uint32_t frames) {
    const i32 w = (i32)s->width, h = (i32)s->height;
    for (u32 f = 0; f < frames; ++f) {
        const float blend_factor = s->frame_counter / (s->frame_counter + 1.f);
        for (i32 x = 0; x < w; ++x) {
            for (i32 y = 0; y < h; ++y) {
                const i32 idx = y * w + x;
                float u = (x + fastrand(s->serial_rng)) / float(w);
                float v = (y + fastrand(s->serial_rng)) / float(h);
                vec3 c = Trace(*s, GetRayFrom(*s, u, v), s->serial_rng, s->raycount, nullptr, nullptr);
                s->output[idx] = lerp(c, s->output[idx], blend_factor);
            }
        }
        ++s->frame_counter;
    }
}

// output[] as width*height*3 floats (row 0 = bottom row, like the reference)
void oref_get_output(const oref_scene* s, float* rgb_out) { memcpy(rgb_out, s->output.data(), s->output.size() * sizeof(vec3)); }

void oref_get_counters(oref_scene* s, uint64_t* rays, uint64_t* frames, uint64_t* box_tests, uint64_t* tri_tests, int reset) {
    if (rays) *rays = s->raycount;
    if (frames) *frames = s->frame_counter;
    if (box_tests) *box_tests = s->box_tests;
    if (tri_tests) *tri_tests = s->tri_tests;
    if (reset) s->raycount = s->box_tests = s->tri_tests = 0;
}

// Known-answer helpers for unit tests of the primitives.
float oref_fastrand(uint32_t* state) { return fastrand(*state); }
uint32_t oref_path_rng_state(uint32_t pixel, uint32_t sample, uint64_t seed) { return path_rng_state(pixel, sample, seed); }
int oref_ray_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3], float* t_inout, float uv_out[2]) {
    TriangleHitData th{*t_inout, vec2{}};
    bool hit = RayTriangle(vec3(o[0], o[1], o[2]), vec3(d[0], d[1], d[2]), vec3(v0[0], v0[1], v0[2]), vec3(v1[0], v1[1], v1[2]), vec3(v2[0], v2[1], v2[2]), th);
    if (hit) { *t_inout = th.RayT; uv_out[0] = th.TriangleUV.x; uv_out[1] = th.TriangleUV.y; }
    return hit ? 1 : 0;
}
int oref_ray_aabb(const float o[3], const float d[3], const float bmin[3], const float bmax[3], float tmax) {
    Ray r(vec3(o[0], o[1], o[2]), vec3(d[0], d[1], d[2]));
    return RayAABB(r.origin, r.inv_direction, vec3(bmin[0], bmin[1], bmin[2]), vec3(bmax[0], bmax[1], bmax[2]), tmax) ? 1 : 0;
}
// CPUDetails::Tonemap + Bitmap::SetPixel (cpu_details.cpp:218-243, bitmap.cpp:58-69) on a linear
// RGB image whose row 0 is the bottom row (the module's output[]): out row 0 is the top row,
// pixels 0x00RRGGBB. mode: 0 none, 1 sRGB, 2 Reinhard, 3 ACES (curves: cuda_details.cpp:36-41;
// srgb = glm convertLinearToSRGB, the in-tree stand-in for cclib's, common.h:152).
void oref_tonemap_bgra8(const float* rgb, uint32_t w, uint32_t h, int mode, float exposure, uint32_t* out) {
    auto srgb = [](float c) {
        c = std::min(std::max(c, 0.f), 1.f);
        return c < 0.0031308f ? c * 12.92f : std::pow(c, 0.41666f) * 1.055f - 0.055f;
    };
    auto curve = [mode](float x) {
        if (mode == 2) return std::min(std::max(x / (1.f + x), 0.f), 1.f);
        if (mode == 3) return std::min(std::max((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.f), 1.f);
        return x;
    };
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const float* p = rgb + ((size_t)y * w + x) * 3;
            uint32_t b[3];
            for (int k = 0; k < 3; ++k) {
                float v = p[k];
                if (mode != 0) v = srgb(curve(v * exposure));   // the no-tonemap build does not apply kExposure (cpu_details.cpp:236-238)
                v = std::min(std::max(255.99f * v, 0.f), 255.f);
                b[k] = (uint32_t)(uint8_t)v;
            }
            out[(size_t)(h - 1 - y) * w + x] = b[2] | (b[1] << 8) | (b[0] << 16);
        }
}

int oref_num_threads(void) {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
