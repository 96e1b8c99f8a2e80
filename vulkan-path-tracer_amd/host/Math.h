// Math.h — the handful of glm calls the reference's host code makes (glm::perspective, lookAt, inverse,
// radians; PathTracer.cpp:173-188,578, FlyCamera.cpp:84-94), restated without glm.  Matrices are column-major
// float[16] exactly as glm stores a mat4, which is also what the C-ABI takes.
#pragma once
#include <cmath>
#include <cstring>

namespace vpthost {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3& operator+=(Vec3& a, Vec3 b) { a = a + b; return a; }
inline Vec3& operator-=(Vec3& a, Vec3 b) { a = a - b; return a; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalize(Vec3 a) { float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline float degrees(float rad) { return rad * 57.295779513082320876798154814105f; }

struct Mat4 {
    float m[16];  // m[col*4 + row]
    Mat4() { std::memset(m, 0, sizeof(m)); m[0] = m[5] = m[10] = m[15] = 1.0f; }
    float& at(int row, int col) { return m[col * 4 + row]; }
    float at(int row, int col) const { return m[col * 4 + row]; }
};

inline Mat4 multiply(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int c = 0; c < 4; c++)
        for (int rw = 0; rw < 4; rw++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (double)a.at(rw, k) * (double)b.at(k, c);
            r.at(rw, c) = (float)s;
        }
    return r;
}

// glm::inverse, evaluated in double (Gauss-Jordan with partial pivoting) and rounded once.
inline Mat4 inverse(const Mat4& a) {
    double w[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { w[r][c] = a.at(r, c); w[r][c + 4] = r == c ? 1.0 : 0.0; }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++) if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (p != c) for (int k = 0; k < 8; k++) std::swap(w[p][k], w[c][k]);
        double d = w[c][c];
        for (int k = 0; k < 8; k++) w[c][k] /= d;
        for (int r = 0; r < 4; r++) if (r != c) { double f = w[r][c]; for (int k = 0; k < 8; k++) w[r][k] -= f * w[c][k]; }
    }
    Mat4 out;
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out.at(r, c) = (float)w[r][c + 4];
    return out;
}

// glm::perspective (right-handed, depth -1..1), fovy in radians.
inline Mat4 perspective(float fovy, float aspect, float zn, float zf) {
    Mat4 r; std::memset(r.m, 0, sizeof(r.m));
    const double f = 1.0 / std::tan((double)fovy / 2.0);
    r.at(0, 0) = (float)(f / aspect);
    r.at(1, 1) = (float)f;
    r.at(2, 2) = (float)(((double)zf + zn) / ((double)zn - zf));
    r.at(2, 3) = (float)(2.0 * zf * zn / ((double)zn - zf));
    r.at(3, 2) = -1.0f;
    return r;
}

// glm::lookAt (right-handed).
inline Mat4 lookAt(Vec3 eye, Vec3 center, Vec3 up) {
    Vec3 f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
    Mat4 r;
    r.at(0, 0) = s.x; r.at(0, 1) = s.y; r.at(0, 2) = s.z; r.at(0, 3) = -dot(s, eye);
    r.at(1, 0) = u.x; r.at(1, 1) = u.y; r.at(1, 2) = u.z; r.at(1, 3) = -dot(u, eye);
    r.at(2, 0) = -f.x; r.at(2, 1) = -f.y; r.at(2, 2) = -f.z; r.at(2, 3) = dot(f, eye);
    return r;
}

}  // namespace vpthost
