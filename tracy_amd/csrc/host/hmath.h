// hmath.h -- small float vector/matrix helpers for the HOST side of the hip module
// (scene loading, camera set-up, BVH build). The reference takes these from the
// un-vendored `cclib` (src/common.h:171-216); the semantics used here are the ones the
// reference's interchangeable glm branch documents (src/common.h:100-166): column-major
// mat4, right-handed lookAt, clip z in [-1,1], post-multiplying translate/rotate/scale.
#pragma once
#include <cmath>
#include <cfloat>
#include <cstdint>

namespace hm {

constexpr float PI = 3.1415926535897932f;
constexpr float EPS = 1.e-8f;

struct vec2 { float x = 0.f, y = 0.f; };
struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    vec3() = default;
    constexpr vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit constexpr vec3(float a) : x(a), y(a), z(a) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct vec4 { float x = 0.f, y = 0.f, z = 0.f, w = 0.f; };

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }

inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
// glm::normalize: v * inversesqrt(dot(v, v))
inline vec3 normalize(vec3 a) { return a * (1.f / std::sqrt(dot(a, a))); }
inline vec3 pmin(vec3 a, vec3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline vec3 pmax(vec3 a, vec3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// column-major 4x4: c[col][row]
struct mat4 {
    float c[4][4];
    static mat4 identity() {
        mat4 m{};
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.c[i][j] = (i == j) ? 1.f : 0.f;
        return m;
    }
};

inline vec4 mul(const mat4& m, vec4 v) {
    // glm operator*(mat4, vec4): (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
    vec4 r;
    float* o = &r.x;
    for (int i = 0; i < 4; ++i) {
        float a0 = m.c[0][i] * v.x + m.c[1][i] * v.y;
        float a1 = m.c[2][i] * v.z + m.c[3][i] * v.w;
        o[i] = a0 + a1;
    }
    return r;
}

inline mat4 mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) {
            r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2] + a.c[3][row] * b.c[col][3];
        }
    return r;
}

inline mat4 transpose(const mat4& m) {
    mat4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.c[i][j] = m.c[j][i];
    return r;
}

// general inverse by cofactors, evaluated in double and rounded once to float
inline mat4 inverse(const mat4& m) {
    double a[16], inv[16];
    for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row) a[col * 4 + row] = m.c[col][row];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    double r = 1.0 / det;
    mat4 o;
    for (int col = 0; col < 4; ++col) for (int row = 0; row < 4; ++row) o.c[col][row] = static_cast<float>(inv[col * 4 + row] * r);
    return o;
}

// glm::translate(m, v): m[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]
inline mat4 translate(const mat4& m, vec3 v) {
    mat4 r = m;
    for (int i = 0; i < 4; ++i) r.c[3][i] = m.c[0][i] * v.x + m.c[1][i] * v.y + m.c[2][i] * v.z + m.c[3][i];
    return r;
}

// glm::scale(m, v)
inline mat4 scale(const mat4& m, vec3 v) {
    mat4 r = m;
    for (int i = 0; i < 4; ++i) { r.c[0][i] = m.c[0][i] * v.x; r.c[1][i] = m.c[1][i] * v.y; r.c[2][i] = m.c[2][i] * v.z; }
    return r;
}

// glm::rotate(m, angle, axis): m * R(angle, normalize(axis))
inline mat4 rotate(const mat4& m, float angle, vec3 v) {
    const float c = std::cos(angle), s = std::sin(angle);
    vec3 axis = normalize(v);
    vec3 temp = axis * (1.f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;          R[0][1] = temp.x * axis.y + s * axis.z; R[0][2] = temp.x * axis.z - s * axis.y;
    R[1][0] = temp.y * axis.x - s * axis.z; R[1][1] = c + temp.y * axis.y;          R[1][2] = temp.y * axis.z + s * axis.x;
    R[2][0] = temp.z * axis.x + s * axis.y; R[2][1] = temp.z * axis.y - s * axis.x; R[2][2] = c + temp.z * axis.z;
    mat4 r;
    for (int i = 0; i < 4; ++i) {
        r.c[0][i] = m.c[0][i] * R[0][0] + m.c[1][i] * R[0][1] + m.c[2][i] * R[0][2];
        r.c[1][i] = m.c[0][i] * R[1][0] + m.c[1][i] * R[1][1] + m.c[2][i] * R[1][2];
        r.c[2][i] = m.c[0][i] * R[2][0] + m.c[1][i] * R[2][1] + m.c[2][i] * R[2][2];
        r.c[3][i] = m.c[3][i];
    }
    return r;
}

// glm::perspective (right-handed, clip z in [-1, 1])
inline mat4 perspective(float fovy, float aspect, float zn, float zf) {
    const float t = std::tan(fovy / 2.f);
    mat4 r{};
    r.c[0][0] = 1.f / (aspect * t);
    r.c[1][1] = 1.f / t;
    r.c[2][2] = -(zf + zn) / (zf - zn);
    r.c[2][3] = -1.f;
    r.c[3][2] = -(2.f * zf * zn) / (zf - zn);
    return r;
}

// glm::lookAt (right-handed)
inline mat4 lookAt(vec3 eye, vec3 center, vec3 up) {
    vec3 f = normalize(center - eye);
    vec3 s = normalize(cross(f, up));
    vec3 u = cross(s, f);
    mat4 r = mat4::identity();
    r.c[0][0] = s.x; r.c[1][0] = s.y; r.c[2][0] = s.z;
    r.c[0][1] = u.x; r.c[1][1] = u.y; r.c[2][1] = u.z;
    r.c[0][2] = -f.x; r.c[1][2] = -f.y; r.c[2][2] = -f.z;
    r.c[3][0] = -dot(s, eye); r.c[3][1] = -dot(u, eye); r.c[3][2] = dot(f, eye);
    return r;
}

// glm::convertSRGBToLinear (gamma 2.4, exact piecewise)
inline float srgb_to_linear(float c) {
    return (c <= 0.04045f) ? c * 0.07739938080495356037151702786378f
                           : std::pow((c + 0.055f) * 0.94786729857819905213270142180095f, 2.4f);
}

}  // namespace hm
