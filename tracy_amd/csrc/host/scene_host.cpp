// scene_host.cpp -- host-side scene loader of the hip module's standalone hosts.
// Implements include/tracy_scene.h. Behavioural restatement of the reference's input
// side (Scene::Init and friends, src/scene.cpp:50-545; Mesh post-processing,
// src/mesh.h:116-195; Camera, src/camera.h:22-55; Texture conversion, src/texture.h:93-108).
// Geometry produced here decides hit ids, so the reference's quirks are kept on purpose
// (per-quad unshared sphere vertices, position-only OBJ vertex dedupe, "last triangle wins"
// flat normals, the += / = tangent accumulation, Transform applied even for identity).
#include "image_decode.h"
#include "tracy_scene.h"
#include "hmath.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>
#include <unordered_map>
#include <algorithm>

using namespace hm;

namespace {

thread_local std::string g_err;
void set_err(const std::string& e) { g_err = e; }

struct Vertex {  // src/vertex.h:36-40
    vec3 pos, normal;
    vec2 uv0;
    vec3 tangent, bitangent;
};
static_assert(sizeof(Vertex) == TRH_VERTEX_FLOATS * sizeof(float), "Vertex must be 14 packed floats");

struct Mesh {
    std::vector<Vertex> v;
    std::vector<uint32_t> idx;
    uint32_t material = 0;
    vec3 bmin{FLT_MAX, FLT_MAX, FLT_MAX}, bmax{-FLT_MAX, -FLT_MAX, -FLT_MAX};

    // src/mesh.h:116-125
    Mesh& Transform(const mat4& m) {
        const mat4 nm = transpose(inverse(m));
        for (auto& vt : v) {
            vec4 p = mul(m, vec4{vt.pos.x, vt.pos.y, vt.pos.z, 1.f});
            vt.pos = {p.x, p.y, p.z};
            vec4 n = mul(nm, vec4{vt.normal.x, vt.normal.y, vt.normal.z, 1.f});
            vt.normal = normalize(vec3{n.x, n.y, n.z});
        }
        return *this;
    }
    // src/mesh.h:127-140 (flat normal written to all three vertices; last triangle wins)
    Mesh& ComputeNormals() {
        for (size_t i = 0; i + 2 < idx.size(); i += 3) {
            Vertex& v1 = v[idx[i + 0]];
            Vertex& v2 = v[idx[i + 1]];
            Vertex& v3 = v[idx[i + 2]];
            vec3 n = normalize(cross(v2.pos - v1.pos, v3.pos - v1.pos));
            v1.normal = v2.normal = v3.normal = n;
        }
        return *this;
    }
    // src/mesh.h:142-152
    Mesh& ComputeBoundingBox() {
        bmin = vec3(FLT_MAX);
        bmax = vec3(-FLT_MAX);
        for (auto& vt : v) {
            bmin = pmin(bmin, vt.pos);
            bmax = pmax(bmax, vt.pos);
        }
        return *this;
    }
    // src/mesh.h:154-195
    Mesh& ComputeTangentsAndBitangents() {
        for (size_t i = 0; i + 2 < idx.size(); i += 3) {
            Vertex& v1 = v[idx[i + 0]];
            Vertex& v2 = v[idx[i + 1]];
            Vertex& v3 = v[idx[i + 2]];
            vec3 dp1 = v2.pos - v1.pos;
            vec3 dp2 = v3.pos - v1.pos;
            vec2 duv1 = v2.uv0 - v1.uv0;
            vec2 duv2 = v3.uv0 - v1.uv0;
            float r = 1.f / (duv1.x * duv2.y - duv1.y * duv2.x);
            vec3 tangent = (dp1 * duv2.y - dp2 * duv1.y) * r;
            vec3 bitangent = (dp2 * duv1.x - dp1 * duv2.x) * r;
            // "v1.tangent += v2.tangent = v3.tangent = tangent" (mesh.h:178-179)
            v3.tangent = tangent;
            v2.tangent = tangent;
            v1.tangent += tangent;
            v3.bitangent = bitangent;
            v2.bitangent = bitangent;
            v1.bitangent += bitangent;
        }
        for (auto& vt : v) {
            vt.tangent = normalize(vt.tangent - vt.normal * dot(vt.normal, vt.tangent));
            float sign = (dot(cross(vt.normal, vt.tangent), vt.bitangent) < 0.0f) ? -1.0f : 1.0f;
            vt.bitangent = sign * cross(vt.normal, vt.tangent);
        }
        return *this;
    }
};

struct TextureSlot {
    std::string path;
    bool srgb = false;
    bool is_hdr = false;
    uint32_t w = 0, h = 0;
    std::vector<float> rgba;
};

}  // namespace

struct tsc_scene {
    std::string name;
    std::string asset_root;
    uint32_t width = 0, height = 0;
    tsc_camera cam{};
    std::vector<Mesh> objects;
    std::vector<trh_material> materials;
    std::vector<TextureSlot> textures;
    // flattened views for tsc_fill_desc
    std::vector<float> flat_vertices;
    std::vector<uint32_t> flat_indices;
    std::vector<trh_mesh> flat_meshes;
    std::vector<trh_texture> flat_textures;
};

namespace {

trh_material make_material(vec3 color = vec3(), float roughness = 1.f, float metalness = 0.f, float ior = 1.f,
                           float emissive = 0.f, float translucency = 0.f) {
    // Material::Material (src/material.h:19-26)
    trh_material m{};
    m.albedo[0] = color.x; m.albedo[1] = color.y; m.albedo[2] = color.z;
    m.roughness = roughness;
    m.metalness = metalness;
    m.ior = ior;
    m.emissive[0] = emissive * color.x; m.emissive[1] = emissive * color.y; m.emissive[2] = emissive * color.z;
    m.translucent = translucency;
    for (auto& t : m.tex) t = TRH_NO_TEXTURE;
    return m;
}

void camera_update(tsc_camera& c) {
    // Camera::UpdateProjection / UpdateView (src/camera.h:37-55)
    mat4 proj = perspective(radians(c.fov_deg), c.ratio, .1f, 10000.f);
    mat4 view = lookAt(vec3{c.eye[0], c.eye[1], c.eye[2]}, vec3{c.center[0], c.center[1], c.center[2]},
                       vec3{c.up[0], c.up[1], c.up[2]});
    mat4 ivp = inverse(mul(proj, view));
    memcpy(c.view, view.c, sizeof(c.view));
    memcpy(c.projection, proj.c, sizeof(c.projection));
    memcpy(c.inv_view_proj, ivp.c, sizeof(c.inv_view_proj));
}

// Scene::AddSphere (src/scene.cpp:50-131)
Mesh& add_sphere(tsc_scene& s, vec3 center, float radius, uint32_t steps = 32) {
    Mesh m;
    auto emit = [&](float theta, float phi, float du, float dv) {
        vec3 pos{std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta)};
        Vertex vt{};
        vt.pos = center + pos * radius;
        vt.normal = pos;
        vt.uv0 = vec2{du, dv};
        m.v.push_back(vt);
    };
    for (uint32_t lon = 0; lon < steps; ++lon) {
        float delta_theta1 = (float)lon / steps;
        float delta_theta2 = (float)(lon + 1) / steps;
        float theta1 = delta_theta1 * PI;
        float theta2 = delta_theta2 * PI;
        for (uint32_t lat = 0; lat < steps; ++lat) {
            float delta_phi1 = (float)lat / steps;
            float delta_phi2 = (float)(lat + 1) / steps;
            float phi1 = delta_phi1 * 2.f * PI;
            float phi2 = delta_phi2 * 2.f * PI;
            emit(theta1, phi1, delta_phi1, delta_theta1);
            emit(theta1, phi2, delta_phi2, delta_theta1);
            emit(theta2, phi2, delta_phi2, delta_theta2);
            emit(theta2, phi1, delta_phi1, delta_theta2);
            uint32_t b = static_cast<uint32_t>(m.v.size()) - 4;
            if (lon == 0) {
                m.idx.insert(m.idx.end(), {b + 0, b + 3, b + 2});
            } else if (lon + 1 == steps) {
                m.idx.insert(m.idx.end(), {b + 2, b + 1, b + 0});
            } else {
                m.idx.insert(m.idx.end(), {b + 0, b + 3, b + 1});
                m.idx.insert(m.idx.end(), {b + 1, b + 3, b + 2});
            }
        }
    }
    for (auto& vt : m.v) vt.normal = normalize(vt.normal);
    s.objects.push_back(std::move(m));
    return s.objects.back().ComputeBoundingBox().ComputeTangentsAndBitangents();
}

// Scene::AddBox (src/scene.cpp:133-214)
Mesh& add_box(tsc_scene& s, vec3 bottom, vec3 top, const mat4& transform) {
    const vec3 corner[8] = {{top.x, top.y, bottom.z},    {top.x, bottom.y, bottom.z},    {top.x, top.y, top.z},
                            {top.x, bottom.y, top.z},    {bottom.x, top.y, bottom.z},    {bottom.x, bottom.y, bottom.z},
                            {bottom.x, top.y, top.z},    {bottom.x, bottom.y, top.z}};
    const vec2 uv[4] = {{0.f, 0.f}, {1.f, 0.f}, {0.f, 1.f}, {1.f, 1.f}};
    const vec3 nrm[6] = {{0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}, {-1.f, 0.f, 0.f}, {0.f, -1.f, 0.f}, {1.f, 0.f, 0.f}, {0.f, 0.f, -1.f}};
    // (corner, normal, uv) of the 24 vertices in emission order
    static const int vtab[24][3] = {
        {4, 0, 0}, {2, 0, 3}, {0, 0, 1},  {2, 1, 1}, {7, 1, 2}, {3, 1, 3},  {6, 2, 1}, {5, 2, 2}, {7, 2, 3},
        {1, 3, 3}, {7, 3, 0}, {5, 3, 2},  {0, 4, 1}, {3, 4, 2}, {1, 4, 3},  {4, 5, 1}, {1, 5, 2}, {5, 5, 3},
        {6, 0, 2}, {6, 1, 0}, {4, 2, 0},  {3, 3, 1}, {2, 4, 0}, {0, 5, 0}};
    static const uint32_t itab[36] = {0, 1, 2,  3, 4, 5,  6, 7, 8,  9, 10, 11,  12, 13, 14,  15, 16, 17,
                                      0, 18, 1,  3, 19, 4,  6, 20, 7,  9, 21, 10,  12, 22, 13,  15, 23, 16};
    Mesh m;
    for (auto& t : vtab) {
        Vertex vt{};
        vt.pos = corner[t[0]];
        vt.normal = nrm[t[1]];
        vt.uv0 = uv[t[2]];
        m.v.push_back(vt);
    }
    m.idx.assign(itab, itab + 36);
    s.objects.push_back(std::move(m));
    return s.objects.back().Transform(transform).ComputeBoundingBox().ComputeTangentsAndBitangents();
}

// Scene::AddTriangle (src/scene.cpp:216-222)
Mesh& add_triangle(tsc_scene& s, vec3 a, vec3 b, vec3 c) {
    Mesh m;
    for (vec3 p : {a, b, c}) {
        Vertex vt{};
        vt.pos = p;
        m.v.push_back(vt);
    }
    m.idx = {0, 1, 2};
    s.objects.push_back(std::move(m));
    return s.objects.back().ComputeBoundingBox().ComputeNormals().ComputeTangentsAndBitangents();
}

// Scene::AddMesh (src/scene.cpp:224-229)
Mesh& add_mesh(tsc_scene& s, Mesh&& mesh, const mat4& transform, bool compute_normals) {
    s.objects.push_back(std::move(mesh));
    Mesh& r = s.objects.back().Transform(transform).ComputeBoundingBox();
    return compute_normals ? r.ComputeNormals().ComputeTangentsAndBitangents() : r.ComputeTangentsAndBitangents();
}

// ---------------------------------------------------------------------------------------
// Wavefront OBJ reader with the behaviour of tinyobj::LoadObj as Scene::Init uses it (src/scene.cpp:471-527;
// ext/tiny_obj_loader.h 2.0.0, default arguments: triangulate = true, real_t = float). What decides hit ids is
// restated, and pinned against the real tiny_obj_loader by tests/golden/ref_obj_*.npz (oracle/ref_dump.cpp):
//   * numbers: tinyobj's own decimal parser (tiny_obj_loader.h:865-1003), not strtod -- the integer digits are
//     accumulated as m = 10 m + d, fraction digits added as d * 10^-k (a table for k < 8, pow beyond), a decimal
//     exponent applied as ldexp(m * 5^e, e), all in double, then one rounding to float;
//   * corners "v", "v/vt", "v//vn", "v/vt/vn" with atoi semantics, index 0 invalid, negative = relative to the
//     counts at that line (:799-820, :1145-1200);
//   * faces: a triangle as it is; a QUAD split along its SHORTER diagonal ([0,1,2][0,2,3] if |v2-v0|^2 < |v3-v1|^2,
//     else [0,1,3][1,2,3], :1457-1565); five and more corners by tinyobj's ear clipping in the plane of the first
//     non-degenerate corner (:1566-1852);
//   * shapes split at 'g' and 'o' lines, shapes without faces dropped (:2813-2888, :3007-3016).
// Deviation, on purpose: a corner whose index lies outside the arrays read so far makes the load FAIL with a message;
// tinyobj lets such triangles through (a warning at most) and Scene::Init then reads attrib.vertices out of bounds.
// ---------------------------------------------------------------------------------------
struct ObjIndex { int v, vt, vn; };
struct ObjShape { std::vector<ObjIndex> idx; };
struct ObjData {
    std::vector<float> v, vn, vt;
    std::vector<ObjShape> shapes;
};

inline bool obj_is_space(char c) { return c == ' ' || c == '\t'; }
inline bool obj_is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool obj_is_newline(char c) { return c == '\r' || c == '\n' || c == '\0'; }

// tryParseDouble (tiny_obj_loader.h:865-1003) on [s, s_end): false leaves *result alone
bool obj_parse_double(const char* s, const char* s_end, double* result) {
    if (s >= s_end) return false;
    double mantissa = 0.0;
    int exponent = 0;
    bool negative = false, exp_negative = false, leading_dot = false;
    const char* c = s;
    int read = 0;
    if (*c == '+' || *c == '-') {
        negative = *c == '-';
        ++c;
        if (c != s_end && *c == '.') leading_dot = true;
    } else if (*c == '.') {
        leading_dot = true;
    } else if (!obj_is_digit(*c)) {
        return false;
    }
    if (!leading_dot) {
        while (c != s_end && obj_is_digit(*c)) {
            mantissa = mantissa * 10 + (*c - '0');
            ++c;
            ++read;
        }
        if (read == 0) return false;
    }
    bool has_exponent = false;
    if (c != s_end) {
        if (*c == '.') {
            static const double kPow10[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
            ++c;
            read = 1;
            while (c != s_end && obj_is_digit(*c)) {
                mantissa += (*c - '0') * (read < 8 ? kPow10[read] : std::pow(10.0, -read));
                ++read;
                ++c;
            }
            has_exponent = c != s_end && (*c == 'e' || *c == 'E');
        } else if (*c == 'e' || *c == 'E') {
            has_exponent = true;
        }
    }
    if (has_exponent) {
        ++c;
        if (c != s_end && (*c == '+' || *c == '-')) {
            exp_negative = *c == '-';
            ++c;
        } else if (!obj_is_digit(*c)) {
            return false;   // an empty exponent is not a number
        }
        read = 0;
        while (c != s_end && obj_is_digit(*c)) {
            if (exponent > 2147483647 / 10) return false;
            exponent = exponent * 10 + (*c - '0');
            ++c;
            ++read;
        }
        if (exp_negative) exponent = -exponent;
        if (read == 0) return false;
    }
    *result = (negative ? -1 : 1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}

// parseReal (tiny_obj_loader.h:1004-1012): a token that is no number is 0
float obj_parse_real(const char*& token) {
    token += strspn(token, " \t");
    const char* end = token + strcspn(token, " \t\r");
    double val = 0.0;
    (void)obj_parse_double(token, end, &val);
    token = end;
    return static_cast<float>(val);
}

// fixIndex (tiny_obj_loader.h:799-820)
bool obj_fix_index(int idx, int n, int* out) {
    if (idx > 0) { *out = idx - 1; return true; }
    if (idx == 0) return false;
    *out = n + idx;
    return true;
}

// parseTriple (tiny_obj_loader.h:1145-1200)
bool parse_obj_index(const char*& p, int nv, int nvt, int nvn, ObjIndex& out) {
    out = {-1, -1, -1};
    if (!obj_fix_index(atoi(p), nv, &out.v)) return false;
    p += strcspn(p, "/ \t\r");
    if (*p != '/') return true;
    ++p;
    if (*p == '/') {   // v//vn
        ++p;
        if (!obj_fix_index(atoi(p), nvn, &out.vn)) return false;
        p += strcspn(p, "/ \t\r");
        return true;
    }
    if (!obj_fix_index(atoi(p), nvt, &out.vt)) return false;   // v/vt or v/vt/vn
    p += strcspn(p, "/ \t\r");
    if (*p != '/') return true;
    ++p;
    if (!obj_fix_index(atoi(p), nvn, &out.vn)) return false;
    p += strcspn(p, "/ \t\r");
    return true;
}

// pnpoly as tinyobj uses it on a triangle (tiny_obj_loader.h:1414-1427)
bool obj_point_in_triangle(const float* vx, const float* vy, float tx, float ty) {
    bool inside = false;
    for (int i = 0, j = 2; i < 3; j = i++)
        if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) inside = !inside;
    return inside;
}

// exportGroupsToShape's triangulation of one face (tiny_obj_loader.h:1446-1855). `v` = the positions read so far.
void obj_triangulate(const std::vector<ObjIndex>& face, const std::vector<float>& v, std::vector<ObjIndex>& out) {
    size_t n = face.size();
    if (n < 3) return;   // "Degenerated face found"
    auto pos = [&](const ObjIndex& ix, int a) { return v[(size_t)ix.v * 3 + a]; };
    if (n == 4) {
        float d02 = 0.f, d13 = 0.f;
        float e02[3], e13[3];
        for (int a = 0; a < 3; ++a) {
            e02[a] = pos(face[2], a) - pos(face[0], a);
            e13[a] = pos(face[3], a) - pos(face[1], a);
        }
        d02 = e02[0] * e02[0] + e02[1] * e02[1] + e02[2] * e02[2];
        d13 = e13[0] * e13[0] + e13[1] * e13[1] + e13[2] * e13[2];
        if (d02 < d13) out.insert(out.end(), {face[0], face[1], face[2], face[0], face[2], face[3]});
        else out.insert(out.end(), {face[0], face[1], face[3], face[1], face[2], face[3]});
        return;
    }
    // the two axes of the projection plane: those of the first corner whose edges are not parallel
    size_t axes[2] = {1, 2};
    for (size_t k = 0; k < n; ++k) {
        const ObjIndex &i0 = face[k % n], &i1 = face[(k + 1) % n], &i2 = face[(k + 2) % n];
        const float e0x = pos(i1, 0) - pos(i0, 0), e0y = pos(i1, 1) - pos(i0, 1), e0z = pos(i1, 2) - pos(i0, 2);
        const float e1x = pos(i2, 0) - pos(i1, 0), e1y = pos(i2, 1) - pos(i1, 1), e1z = pos(i2, 2) - pos(i1, 2);
        const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
        const float eps = std::numeric_limits<float>::epsilon();
        if (cx > eps || cy > eps || cz > eps) {
            if (!(cx > cy && cx > cz)) {
                axes[0] = 0;
                if (cz > cx && cz > cy) axes[1] = 1;
            }
            break;
        }
    }
    std::vector<ObjIndex> rest = face;
    size_t guess = 0, iterations_left = face.size(), previous = rest.size();
    while (rest.size() > 3 && iterations_left > 0) {
        n = rest.size();
        if (guess >= n) guess -= n;
        if (previous != n) { previous = n; iterations_left = n; } else { --iterations_left; }
        ObjIndex ind[3];
        float vx[3], vy[3];
        for (size_t k = 0; k < 3; ++k) {
            ind[k] = rest[(guess + k) % n];
            vx[k] = pos(ind[k], (int)axes[0]);
            vy[k] = pos(ind[k], (int)axes[1]);
        }
        const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
        const float cross = e0x * e1y - e0y * e1x;
        const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
        if (cross * area < 0.f) { ++guess; continue; }   // "an internal angle"
        bool overlap = false;
        for (size_t other = 3; other < n; ++other) {
            const size_t idx = (guess + other) % n;
            if (obj_point_in_triangle(vx, vy, pos(rest[idx], (int)axes[0]), pos(rest[idx], (int)axes[1]))) { overlap = true; break; }
        }
        if (overlap) { ++guess; continue; }
        out.insert(out.end(), {ind[0], ind[1], ind[2]});   // an ear: clip its middle corner
        rest.erase(rest.begin() + (long)((guess + 1) % n));
    }
    if (rest.size() == 3) out.insert(out.end(), {rest[0], rest[1], rest[2]});
}

bool load_obj(const std::string& path, ObjData& out, std::string& err) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) { err = "cannot open " + path; return false; }
    std::string text;
    {
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), fp)) > 0) text.append(buf, got);
    }
    fclose(fp);
    ObjShape cur;
    std::vector<std::vector<ObjIndex>> faces;   // the open face group (triangulated when the shape is flushed)
    auto export_faces = [&]() {
        for (const auto& f : faces) obj_triangulate(f, out.v, cur.idx);
        faces.clear();
    };
    auto flush = [&]() {
        export_faces();
        if (!cur.idx.empty()) out.shapes.push_back(std::move(cur));
        cur = ObjShape{};
    };
    size_t line_no = 0;
    for (size_t at = 0; at < text.size();) {
        size_t eol = text.find('\n', at);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(at, eol - at);
        at = eol + 1;
        ++line_no;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const char* p = line.c_str();
        p += strspn(p, " \t");
        if (*p == '\0' || *p == '#') continue;
        if (p[0] == 'v' && obj_is_space(p[1])) {
            p += 2;
            for (int k = 0; k < 3; ++k) out.v.push_back(obj_parse_real(p));
        } else if (p[0] == 'v' && p[1] == 'n' && obj_is_space(p[2])) {
            p += 3;
            for (int k = 0; k < 3; ++k) out.vn.push_back(obj_parse_real(p));
        } else if (p[0] == 'v' && p[1] == 't' && obj_is_space(p[2])) {
            p += 3;
            for (int k = 0; k < 2; ++k) out.vt.push_back(obj_parse_real(p));
        } else if (p[0] == 'f' && obj_is_space(p[1])) {
            p += 2;
            p += strspn(p, " \t");
            std::vector<ObjIndex> face;
            const int nv = (int)out.v.size() / 3, nvt = (int)out.vt.size() / 2, nvn = (int)out.vn.size() / 3;
            while (!obj_is_newline(*p)) {
                ObjIndex ix;
                if (!parse_obj_index(p, nv, nvt, nvn, ix)) {   // tinyobj: LoadObj fails ("zero value for face index")
                    err = path + ": line " + std::to_string(line_no) + ": not a face corner (index 0 or no number)";
                    return false;
                }
                if (ix.v < 0 || ix.v >= nv || ix.vt < -1 || ix.vt >= nvt || ix.vn < -1 || ix.vn >= nvn) {
                    err = path + ": line " + std::to_string(line_no) + ": face corner index outside the " + std::to_string(nv) + " positions / " +
                          std::to_string(nvt) + " texture coordinates / " + std::to_string(nvn) + " normals read so far";
                    return false;
                }
                face.push_back(ix);
                p += strspn(p, " \t\r");
            }
            faces.push_back(std::move(face));
        } else if (strncmp(p, "usemtl", 6) == 0) {
            export_faces();   // a material change closes the face group but not the shape (tiny_obj_loader.h:2747-2755)
        } else if ((p[0] == 'g' || p[0] == 'o') && obj_is_space(p[1])) {
            flush();
        }
    }
    flush();
    return true;
}

std::string join_path(const std::string& root, const std::string& rel) {
    if (rel.empty() || rel[0] == '/' || root.empty()) return rel;
    return root + "/" + rel;
}

bool ends_with(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    if (s.size() < n) return false;
    std::string tail = s.substr(s.size() - n);
    std::transform(tail.begin(), tail.end(), tail.begin(), ::tolower);
    return tail == suf;
}

}  // namespace

extern "C" {

const char* tsc_last_error(void) { return g_err.c_str(); }

tsc_scene* tsc_load(const char* scn_path, const char* asset_root, int override_w, int override_h) {
    if (!scn_path) { set_err("tsc_load: scn_path is NULL"); return nullptr; }
    const std::string root = asset_root ? asset_root : "";
    FILE* fp = fopen(scn_path, "r");
    if (!fp) { set_err(std::string("tsc_load: cannot open ") + scn_path); return nullptr; }

    auto* s = new tsc_scene();
    s->asset_root = root;
    s->materials.push_back(make_material());  // materials_{1}: default sky (src/scene.h:67)
    // main() defaults (src/win_raytracer.cpp:439-440), overridden by OUT
    uint32_t width = 640, height = 480;
    if (override_w > 0 && override_h > 0) { width = (uint32_t)override_w; height = (uint32_t)override_h; }
    s->width = width;
    s->height = height;

    std::unordered_map<std::string, uint32_t> material_id, texture_id;
    auto add_texture = [&](const char* file, bool srgb) -> uint32_t {
        TextureSlot t;
        t.path = file;
        t.srgb = srgb;
        t.is_hdr = ends_with(t.path, ".hdr");
        s->textures.push_back(std::move(t));
        return (uint32_t)s->textures.size() - 1;
    };

    static const size_t LINE = 512;
    char line[LINE], params[LINE];
    while (fgets(line, LINE, fp)) {
        if (line[0] == '#' || line[0] == '\n') continue;
        char id[3];
        if (sscanf(line, "%c%c%c %[^\n]", &id[0], &id[1], &id[2], params) != 4) continue;
        const std::string tag(id, 3);
        if (tag == "SCN") {
            s->name = params;
        } else if (tag == "OUT") {
            int w, h;
            if (sscanf(params, "%d %d", &w, &h) == 2) {
                if (!(override_w > 0 && override_h > 0)) { width = (uint32_t)w; height = (uint32_t)h; }
                s->width = width;
                s->height = height;
            }
        } else if (tag == "CAM") {
            vec3 eye, center, up;
            float fov;
            float ratio = float(width) / float(std::max(height, 1u));
            if (sscanf(params, "(%f,%f,%f) (%f,%f,%f) (%f,%f,%f) %f", &eye.x, &eye.y, &eye.z, &center.x, &center.y, &center.z,
                       &up.x, &up.y, &up.z, &fov) == 10) {
                tsc_camera& c = s->cam;
                c.eye[0] = eye.x; c.eye[1] = eye.y; c.eye[2] = eye.z;
                c.center[0] = center.x; c.center[1] = center.y; c.center[2] = center.z;
                c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
                c.fov_deg = fov;
                c.ratio = ratio;
                camera_update(c);
            }
        } else if (tag == "MTL") {
            char mat_name[260];
            vec3 albedo;
            float roughness, metalness, ior = 1.f, emissive = 0.f, translucency = 0.f;
            if (sscanf(params, "%259s (%f,%f,%f) %f %f %f %f %f", mat_name, &albedo.x, &albedo.y, &albedo.z, &roughness, &metalness,
                       &ior, &emissive, &translucency) >= 6) {
                material_id[mat_name] = (uint32_t)s->materials.size();
                s->materials.push_back(make_material(albedo, roughness, metalness, ior, emissive, translucency));
            }
        } else if (tag == "TEX") {
            char mat_name[260], tex_type = 0, file_name[260], srgb_flag[8] = {0};
            int num = sscanf(params, "%259s %c %259s %4s", mat_name, &tex_type, file_name, srgb_flag);
            int slot = -1;
            switch (tex_type) {
                case 'B': slot = TRH_TEX_BASECOLOR; break;
                case 'N': slot = TRH_TEX_NORMAL; break;
                case 'R': slot = TRH_TEX_ROUGHNESS; break;
                case 'M': slot = TRH_TEX_METALNESS; break;
                case 'E': slot = TRH_TEX_EMISSIVE; break;
            }
            if (num >= 3) {
                if (texture_id.count(file_name) == 0)
                    texture_id[file_name] = add_texture(file_name, num == 4 && strncmp(srgb_flag, "SRGB", 4) == 0);
                if (slot >= 0) s->materials[material_id[mat_name]].tex[slot] = texture_id[file_name];
            }
        } else if (tag == "SKY") {
            vec3 albedo;
            if (sscanf(params, "(%f,%f,%f)", &albedo.x, &albedo.y, &albedo.z) == 3) {
                s->materials[0] = make_material(albedo, .0f, .0f, .0f, 1.f);
            } else {
                char file_name[260], srgb_flag[8] = {0};
                int num = sscanf(params, "%259s %4s", file_name, srgb_flag);
                if (num >= 1) {
                    if (texture_id.count(file_name) == 0)
                        texture_id[file_name] = add_texture(file_name, num == 2 && strncmp(srgb_flag, "SRGB", 4) == 0);
                    s->materials[0].tex[TRH_TEX_EMISSIVE] = texture_id[file_name];
                }
            }
        } else if (tag == "OBJ") {
            char obj_type, sub[LINE];
            if (sscanf(params, "%c %[^\n]", &obj_type, sub) == 2) {
                char mat_name[260];
                if (obj_type == 'S') {
                    vec3 c;
                    float r;
                    if (sscanf(sub, "(%f,%f,%f) %f %259s", &c.x, &c.y, &c.z, &r, mat_name) == 5)
                        add_sphere(*s, c, r).material = material_id[mat_name];
                } else if (obj_type == 'B') {
                    vec3 mn, mx, rot;
                    if (sscanf(sub, "(%f,%f,%f) (%f,%f,%f) (%f,%f,%f) %259s", &mn.x, &mn.y, &mn.z, &mx.x, &mx.y, &mx.z, &rot.x, &rot.y,
                               &rot.z, mat_name) == 10) {
                        vec3 ctr = (mn + mx) * .5f;
                        mat4 t = mat4::identity();
                        t = translate(t, ctr);
                        t = rotate(t, radians(rot.x), {1, 0, 0});
                        t = rotate(t, radians(rot.y), {0, 1, 0});
                        t = rotate(t, radians(rot.z), {0, 0, 1});
                        t = translate(t, -ctr);
                        add_box(*s, mn, mx, t).material = material_id[mat_name];
                    } else if (sscanf(sub, "(%f,%f,%f) (%f,%f,%f) %259s", &mn.x, &mn.y, &mn.z, &mx.x, &mx.y, &mx.z, mat_name) == 7) {
                        add_box(*s, mn, mx, mat4::identity()).material = material_id[mat_name];
                    }
                } else if (obj_type == 'T') {
                    vec3 a, b, c;
                    if (sscanf(sub, "(%f,%f,%f) (%f,%f,%f) (%f,%f,%f) %259s", &a.x, &a.y, &a.z, &b.x, &b.y, &b.z, &c.x, &c.y, &c.z,
                               mat_name) == 10)
                        add_triangle(*s, a, b, c).material = material_id[mat_name];
                }
            }
        } else if (tag == "TRI") {
            char file_name[260], mat_name[260];
            vec3 tr, rot;
            float scaling = 1.f;
            if (sscanf(params, "%259s %259s (%f, %f, %f) (%f, %f, %f) %f", file_name, mat_name, &tr.x, &tr.y, &tr.z, &rot.x, &rot.y,
                       &rot.z, &scaling) >= 2) {
                ObjData obj;
                std::string obj_err;
                if (!load_obj(join_path(root, file_name), obj, obj_err)) {
                    set_err("tsc_load: cannot read OBJ: " + obj_err);
                    fclose(fp);
                    delete s;
                    return nullptr;
                }
                mat4 t = mat4::identity();
                t = rotate(t, radians(rot.x), {1, 0, 0});
                t = rotate(t, radians(rot.y), {0, 1, 0});
                t = rotate(t, radians(rot.z), {0, 0, 1});
                t = translate(t, tr);
                t = scale(t, vec3(scaling));
                for (const ObjShape& shape : obj.shapes) {
                    Mesh m;
                    bool recompute_normals = true;
                    std::unordered_map<int, int> remap;  // position index only (src/scene.cpp:490-523)
                    for (const ObjIndex& ix : shape.idx) {
                        auto it = remap.find(ix.v);
                        if (it != remap.end()) { m.idx.push_back((uint32_t)it->second); continue; }
                        Vertex vt{};
                        vt.pos = vec3{obj.v[3 * ix.v], obj.v[3 * ix.v + 1], obj.v[3 * ix.v + 2]};
                        if (ix.vn != -1) {
                            recompute_normals = false;
                            vt.normal = vec3{obj.vn[3 * ix.vn], obj.vn[3 * ix.vn + 1], obj.vn[3 * ix.vn + 2]};
                        }
                        if (ix.vt != -1) vt.uv0 = vec2{obj.vt[2 * ix.vt], obj.vt[2 * ix.vt + 1]};
                        m.v.push_back(vt);
                        int last = (int)m.v.size() - 1;
                        m.idx.push_back((uint32_t)last);
                        remap[ix.v] = last;
                    }
                    add_mesh(*s, std::move(m), t, recompute_normals).material = material_id[mat_name];
                }
            }
        }
    }
    fclose(fp);
    if (s->objects.size() > 256) {
        set_err("tsc_load: more than 256 meshes (mesh id is 8 bit in the packed hit id)");
        delete s;
        return nullptr;
    }
    return s;
}

void tsc_free(tsc_scene* s) { delete s; }

const char* tsc_name(const tsc_scene* s) { return s ? s->name.c_str() : ""; }

uint32_t tsc_triangle_count(const tsc_scene* s) {
    uint32_t n = 0;
    if (s) for (auto& m : s->objects) n += (uint32_t)m.idx.size() / 3;
    return n;
}

int tsc_get_camera(const tsc_scene* s, tsc_camera* out) {
    if (!s || !out) { set_err("tsc_get_camera: NULL argument"); return TRH_ERR_INVALID_ARG; }
    *out = s->cam;
    return TRH_OK;
}

int tsc_set_view(tsc_scene* s, const float eye[3], const float center[3], const float up[3]) {
    if (!s || !eye || !center || !up) { set_err("tsc_set_view: NULL argument"); return TRH_ERR_INVALID_ARG; }
    memcpy(s->cam.eye, eye, 12);
    memcpy(s->cam.center, center, 12);
    memcpy(s->cam.up, up, 12);
    camera_update(s->cam);
    return TRH_OK;
}

uint32_t tsc_texture_count(const tsc_scene* s) { return s ? (uint32_t)s->textures.size() : 0; }

int tsc_get_texture_info(const tsc_scene* s, uint32_t i, tsc_texture_info* out) {
    if (!s || !out || i >= s->textures.size()) { set_err("tsc_get_texture_info: bad argument"); return TRH_ERR_INVALID_ARG; }
    const TextureSlot& t = s->textures[i];
    out->path = t.path.c_str();
    out->srgb = t.srgb;
    out->is_hdr = t.is_hdr;
    out->width = t.w;
    out->height = t.h;
    return TRH_OK;
}

// Texture::Texture<T> (src/texture.h:93-108)
int tsc_set_texture_u8(tsc_scene* s, uint32_t i, uint32_t w, uint32_t h, const uint8_t* rgba8) {
    if (!s || i >= s->textures.size() || !rgba8 || !w || !h) { set_err("tsc_set_texture_u8: bad argument"); return TRH_ERR_INVALID_ARG; }
    TextureSlot& t = s->textures[i];
    t.w = w;
    t.h = h;
    t.rgba.resize((size_t)w * h * 4);
    for (size_t k = 0; k < (size_t)w * h * 4; ++k) {
        float c = float(rgba8[k]) / 255.f;
        t.rgba[k] = t.srgb ? srgb_to_linear(c) : c;
    }
    return TRH_OK;
}

int tsc_set_texture_f32(tsc_scene* s, uint32_t i, uint32_t w, uint32_t h, const float* rgba) {
    if (!s || i >= s->textures.size() || !rgba || !w || !h) { set_err("tsc_set_texture_f32: bad argument"); return TRH_ERR_INVALID_ARG; }
    TextureSlot& t = s->textures[i];
    t.w = w;
    t.h = h;
    t.rgba.resize((size_t)w * h * 4);
    for (size_t k = 0; k < (size_t)w * h * 4; ++k) t.rgba[k] = t.srgb ? srgb_to_linear(rgba[k] / 1.f) : rgba[k];
    return TRH_OK;
}

// Stand-in for the missing data/sky.hdr: closed-form 1024x512 equirect probe
// (vertical gradient + ground + one HDR sun lobe); formula recorded in BASELINE.md section 3.
int tsc_set_texture_synthetic_sky(tsc_scene* s, uint32_t i) {
    if (!s || i >= s->textures.size()) { set_err("tsc_set_texture_synthetic_sky: bad argument"); return TRH_ERR_INVALID_ARG; }
    const uint32_t W = 1024, H = 512;
    std::vector<float> px((size_t)W * H * 4);
    const double pi = 3.14159265358979323846;
    const double sun_az = 0.9, sun_el = 0.7;
    const double sx = std::cos(sun_el) * std::cos(sun_az), sy = std::sin(sun_el), sz = std::cos(sun_el) * std::sin(sun_az);
    for (uint32_t j = 0; j < H; ++j) {
        // Texture::GetPixel flips v (src/texture.h:54): row 0 is v = 1 (straight up)
        double v = 1.0 - (j + 0.5) / H;
        double el = (v - 0.5) * pi;
        for (uint32_t k = 0; k < W; ++k) {
            double u = (k + 0.5) / W;
            double az = (u - 0.5) * 2.0 * pi;
            double dx = std::cos(el) * std::cos(az), dy = std::sin(el), dz = std::cos(el) * std::sin(az);
            double r, g, b;
            if (dy >= 0.0) {
                double t = std::sqrt(dy);
                r = 0.90 + (0.25 - 0.90) * t;
                g = 0.90 + (0.45 - 0.90) * t;
                b = 0.95 + (0.90 - 0.95) * t;
            } else {
                double t = std::sqrt(-dy);
                r = 0.90 + (0.30 - 0.90) * t;
                g = 0.90 + (0.28 - 0.90) * t;
                b = 0.95 + (0.25 - 0.95) * t;
            }
            double c = dx * sx + dy * sy + dz * sz;
            if (c > 0.0) {
                double lobe = 40.0 * std::pow(c, 512.0);
                r += lobe;
                g += lobe * 0.95;
                b += lobe * 0.85;
            }
            float* o = &px[((size_t)j * W + k) * 4];
            o[0] = (float)r; o[1] = (float)g; o[2] = (float)b; o[3] = 1.f;
        }
    }
    TextureSlot& t = s->textures[i];
    bool srgb = t.srgb;
    t.srgb = false;  // the probe is defined in linear radiance
    int rc = tsc_set_texture_f32(s, i, W, H, px.data());
    t.srgb = srgb;
    return rc;
}

int tsc_subdivide(tsc_scene* s, uint32_t min_triangles, uint32_t levels) {
    if (!s) { set_err("tsc_subdivide: NULL scene"); return TRH_ERR_INVALID_ARG; }
    for (uint32_t l = 0; l < levels; ++l) {
        for (Mesh& m : s->objects) {
            const size_t ntri = m.idx.size() / 3;
            if (ntri < min_triangles) continue;
            if (ntri * 4 > (1u << 24)) { set_err("tsc_subdivide: more than 2^24 triangles in one mesh (24-bit triangle id)"); return TRH_ERR_LIMIT; }
            std::unordered_map<uint64_t, uint32_t> mid;
            mid.reserve(ntri * 2);
            auto midpoint = [&](uint32_t a, uint32_t b) -> uint32_t {
                const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
                auto it = mid.find(key);
                if (it != mid.end()) return it->second;
                const Vertex& va = m.v[a];
                const Vertex& vb = m.v[b];
                Vertex n;
                n.pos = (va.pos + vb.pos) * .5f;
                n.normal = (va.normal + vb.normal) * .5f;
                n.uv0 = vec2{(va.uv0.x + vb.uv0.x) * .5f, (va.uv0.y + vb.uv0.y) * .5f};
                n.tangent = (va.tangent + vb.tangent) * .5f;
                n.bitangent = (va.bitangent + vb.bitangent) * .5f;
                const uint32_t id = (uint32_t)m.v.size();
                m.v.push_back(n);
                mid.emplace(key, id);
                return id;
            };
            std::vector<uint32_t> out;
            out.reserve(m.idx.size() * 4);
            for (size_t t = 0; t < ntri; ++t) {
                const uint32_t a = m.idx[t * 3], b = m.idx[t * 3 + 1], c = m.idx[t * 3 + 2];
                const uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
                const uint32_t tri[12] = {a, ab, ca, ab, b, bc, ca, bc, c, ab, bc, ca};
                out.insert(out.end(), tri, tri + 12);
            }
            m.idx.swap(out);
        }
    }
    return TRH_OK;
}

int tsc_decode_textures(tsc_scene* s, int synthetic_missing_hdr) {
    if (!s) { set_err("tsc_decode_textures: NULL scene"); return TRH_ERR_INVALID_ARG; }
    for (uint32_t i = 0; i < s->textures.size(); ++i) {
        TextureSlot& t = s->textures[i];
        if (!t.rgba.empty()) continue;
        const std::string full = join_path(s->asset_root, t.path);
        std::vector<uint8_t> bytes;
        if (!imgdec::read_file(full, bytes)) {
            if (t.is_hdr && synthetic_missing_hdr) {
                int rc = tsc_set_texture_synthetic_sky(s, i);
                if (rc) return rc;
                continue;
            }
            set_err("tsc_decode_textures: cannot read '" + full + "'");
            return TRH_ERR_INVALID_ARG;
        }
        std::string err;
        int rc;
        if (t.is_hdr) {
            imgdec::ImageF img;
            if (!imgdec::decode_hdr(bytes.data(), bytes.size(), img, err)) { set_err("tsc_decode_textures: '" + t.path + "': " + err); return TRH_ERR_INVALID_ARG; }
            rc = tsc_set_texture_f32(s, i, img.width, img.height, img.rgba.data());
        } else if (ends_with(t.path, ".jpg") || ends_with(t.path, ".jpeg")) {
            imgdec::Image8 img;
            if (!imgdec::decode_jpeg(bytes.data(), bytes.size(), img, err)) { set_err("tsc_decode_textures: '" + t.path + "': " + err); return TRH_ERR_INVALID_ARG; }
            rc = tsc_set_texture_u8(s, i, img.width, img.height, img.rgba.data());
        } else {
            set_err("tsc_decode_textures: '" + t.path + "': no built-in decoder for this format (JPEG and Radiance HDR only); decode it and call tsc_set_texture_*");
            return TRH_ERR_INVALID_ARG;
        }
        if (rc) return rc;
    }
    return TRH_OK;
}

int tsc_decode_jpeg_file(const char* path, uint32_t* w_out, uint32_t* h_out, uint8_t** rgba8_out) {
    if (!path || !w_out || !h_out || !rgba8_out) { set_err("tsc_decode_jpeg_file: NULL argument"); return TRH_ERR_INVALID_ARG; }
    std::vector<uint8_t> bytes;
    if (!imgdec::read_file(path, bytes)) { set_err(std::string("tsc_decode_jpeg_file: cannot read ") + path); return TRH_ERR_INVALID_ARG; }
    imgdec::Image8 img;
    std::string err;
    if (!imgdec::decode_jpeg(bytes.data(), bytes.size(), img, err)) { set_err("tsc_decode_jpeg_file: " + err); return TRH_ERR_INVALID_ARG; }
    *rgba8_out = (uint8_t*)malloc(img.rgba.size());
    if (!*rgba8_out) { set_err("tsc_decode_jpeg_file: out of memory"); return TRH_ERR_LIMIT; }
    memcpy(*rgba8_out, img.rgba.data(), img.rgba.size());
    *w_out = img.width;
    *h_out = img.height;
    return TRH_OK;
}

int tsc_decode_hdr_file(const char* path, uint32_t* w_out, uint32_t* h_out, float** rgba_out) {
    if (!path || !w_out || !h_out || !rgba_out) { set_err("tsc_decode_hdr_file: NULL argument"); return TRH_ERR_INVALID_ARG; }
    std::vector<uint8_t> bytes;
    if (!imgdec::read_file(path, bytes)) { set_err(std::string("tsc_decode_hdr_file: cannot read ") + path); return TRH_ERR_INVALID_ARG; }
    imgdec::ImageF img;
    std::string err;
    if (!imgdec::decode_hdr(bytes.data(), bytes.size(), img, err)) { set_err("tsc_decode_hdr_file: " + err); return TRH_ERR_INVALID_ARG; }
    *rgba_out = (float*)malloc(img.rgba.size() * sizeof(float));
    if (!*rgba_out) { set_err("tsc_decode_hdr_file: out of memory"); return TRH_ERR_LIMIT; }
    memcpy(*rgba_out, img.rgba.data(), img.rgba.size() * sizeof(float));
    *w_out = img.width;
    *h_out = img.height;
    return TRH_OK;
}

void tsc_free_pixels(void* p) { free(p); }

int tsc_parse_obj_file(const char* path, tsc_obj_counts* counts, float** v_out, float** vt_out, float** vn_out, int32_t** shape_sizes_out, int32_t** corners_out) {
    if (!path || !counts) { set_err("tsc_parse_obj_file: NULL argument"); return TRH_ERR_INVALID_ARG; }
    ObjData obj;
    std::string err;
    if (!load_obj(path, obj, err)) { set_err("tsc_parse_obj_file: " + err); return TRH_ERR_INVALID_ARG; }
    size_t corners = 0;
    for (const ObjShape& sh : obj.shapes) corners += sh.idx.size();
    counts->v_floats = (uint32_t)obj.v.size();
    counts->vt_floats = (uint32_t)obj.vt.size();
    counts->vn_floats = (uint32_t)obj.vn.size();
    counts->shapes = (uint32_t)obj.shapes.size();
    counts->corners = (uint32_t)corners;
    auto dup = [](const void* src, size_t bytes) -> void* {
        void* p = malloc(bytes ? bytes : 1);
        if (p && bytes) memcpy(p, src, bytes);
        return p;
    };
    if (v_out) *v_out = (float*)dup(obj.v.data(), obj.v.size() * 4);
    if (vt_out) *vt_out = (float*)dup(obj.vt.data(), obj.vt.size() * 4);
    if (vn_out) *vn_out = (float*)dup(obj.vn.data(), obj.vn.size() * 4);
    if (shape_sizes_out) {
        std::vector<int32_t> sizes;
        for (const ObjShape& sh : obj.shapes) sizes.push_back((int32_t)sh.idx.size());
        *shape_sizes_out = (int32_t*)dup(sizes.data(), sizes.size() * 4);
    }
    if (corners_out) {
        std::vector<int32_t> flat;
        flat.reserve(corners * 3);
        for (const ObjShape& sh : obj.shapes)
            for (const ObjIndex& ix : sh.idx) { flat.push_back(ix.v); flat.push_back(ix.vt); flat.push_back(ix.vn); }
        *corners_out = (int32_t*)dup(flat.data(), flat.size() * 4);
    }
    return TRH_OK;
}

int tsc_fill_desc(const tsc_scene* cs, trh_scene_desc* out) {
    if (!cs || !out) { set_err("tsc_fill_desc: NULL argument"); return TRH_ERR_INVALID_ARG; }
    tsc_scene* s = const_cast<tsc_scene*>(cs);
    s->flat_vertices.clear();
    s->flat_indices.clear();
    s->flat_meshes.clear();
    s->flat_textures.clear();
    for (const Mesh& m : s->objects) {
        trh_mesh fm{};
        fm.first_vertex = (uint32_t)(s->flat_vertices.size() / TRH_VERTEX_FLOATS);
        fm.vertex_count = (uint32_t)m.v.size();
        fm.first_index = (uint32_t)s->flat_indices.size();
        fm.index_count = (uint32_t)m.idx.size();
        fm.material = m.material;
        fm.aabb_min[0] = m.bmin.x; fm.aabb_min[1] = m.bmin.y; fm.aabb_min[2] = m.bmin.z;
        fm.aabb_max[0] = m.bmax.x; fm.aabb_max[1] = m.bmax.y; fm.aabb_max[2] = m.bmax.z;
        const float* vf = reinterpret_cast<const float*>(m.v.data());
        s->flat_vertices.insert(s->flat_vertices.end(), vf, vf + m.v.size() * TRH_VERTEX_FLOATS);
        s->flat_indices.insert(s->flat_indices.end(), m.idx.begin(), m.idx.end());
        s->flat_meshes.push_back(fm);
    }
    for (size_t i = 0; i < s->textures.size(); ++i) {
        const TextureSlot& t = s->textures[i];
        if (t.rgba.empty()) {
            set_err("tsc_fill_desc: texture '" + t.path + "' has no pixels (decode it and call tsc_set_texture_*)");
            return TRH_ERR_STATE;
        }
        s->flat_textures.push_back(trh_texture{t.w, t.h, t.rgba.data()});
    }
    memset(out, 0, sizeof(*out));
    out->vertices = s->flat_vertices.data();
    out->vertex_count = (uint32_t)(s->flat_vertices.size() / TRH_VERTEX_FLOATS);
    out->indices = s->flat_indices.data();
    out->index_count = (uint32_t)s->flat_indices.size();
    out->meshes = s->flat_meshes.data();
    out->mesh_count = (uint32_t)s->flat_meshes.size();
    out->materials = s->materials.data();
    out->material_count = (uint32_t)s->materials.size();
    out->textures = s->flat_textures.data();
    out->texture_count = (uint32_t)s->flat_textures.size();
    out->width = s->width;
    out->height = s->height;
    out->max_bounces = 5;
    out->flags = TRH_FLAG_DEFAULT;
    return TRH_OK;
}

}  // extern "C"
