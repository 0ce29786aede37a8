// Mat44<T>: 4x4 matrix in the row-vector convention of the reference
// (cudaTracer/Mat.cuh:7-387): points are row vectors, p' = p * M, translation
// lives in row 3, and `A * B` applies A first.  multPoint divides by w,
// multVec ignores both translation and w (Mat.cuh:127-149), so the one-argument
// scale(s) - which only writes 1/s into [3][3] (Mat.cuh:381-386) - scales
// points but not directions.
#ifndef RT_HOST_MAT_H
#define RT_HOST_MAT_H

#include <cmath>
#include <utility>

#include "Vec.h"

namespace rt {

constexpr float PI = 3.1415926536f;  // cudaTracer/globals.h:6
inline float ToRadian(float degree) { return degree * (PI / 180.0f); }

template <typename T>
class Mat44 {
   public:
    T x[4][4];

    constexpr Mat44() : x{} {}
    Mat44(const T r0[4], const T r1[4], const T r2[4], const T r3[4]) {
        const T* rows[4] = {r0, r1, r2, r3};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) x[i][j] = rows[i][j];
    }
    constexpr Mat44(T a, T b, T c, T d, T e, T f, T g, T h, T i, T j, T k, T l, T m, T n, T o, T p)
        : x{{a, b, c, d}, {e, f, g, h}, {i, j, k, l}, {m, n, o, p}} {}

    const T* operator[](int i) const { return x[i]; }
    T* operator[](int i) { return x[i]; }

    static void multiply(const Mat44& a, const Mat44& b, Mat44& c) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                c[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j] + a[i][3] * b[3][j];
    }
    Mat44 operator*(const Mat44& v) const {
        Mat44 tmp;
        multiply(*this, v, tmp);
        return tmp;
    }
    friend bool operator==(const Mat44& a, const Mat44& b) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if (a[i][j] != b[i][j]) return false;
        return true;
    }
    friend bool operator!=(const Mat44& a, const Mat44& b) { return !(a == b); }

    Mat44 transposed() const {
        Mat44 t;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) t[i][j] = x[j][i];
        return t;
    }
    Mat44& transpose() { return *this = transposed(); }

    template <typename S>
    Vec<S, 3> multPoint(const Vec<S, 3>& src) const {
        S r[4];
        for (int j = 0; j < 4; ++j) r[j] = src[0] * x[0][j] + src[1] * x[1][j] + src[2] * x[2][j] + x[3][j];
        return Vec<S, 3>(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
    }
    template <typename S>
    Vec<S, 3> multVec(const Vec<S, 3>& src) const {
        S r[3];
        for (int j = 0; j < 3; ++j) r[j] = src[0] * x[0][j] + src[1] * x[1][j] + src[2] * x[2][j];
        return Vec<S, 3>(r[0], r[1], r[2]);
    }

    // Gauss-Jordan elimination with row pivoting; a singular input yields the
    // identity (Mat.cuh:169-247).
    Mat44 inversed() const {
        Mat44 s = identity();
        Mat44 t(*this);
        auto swapRows = [](Mat44& m, int a, int b) {
            for (int j = 0; j < 4; ++j) std::swap(m[a][j], m[b][j]);
        };
        auto eliminate = [&](int dst, int src, T f) {
            for (int k = 0; k < 4; ++k) {
                t[dst][k] -= f * t[src][k];
                s[dst][k] -= f * s[src][k];
            }
        };
        for (int i = 0; i < 3; ++i) {  // forward elimination
            int pivot = i;
            T pivotsize = t[i][i] < 0 ? -t[i][i] : t[i][i];
            for (int j = i + 1; j < 4; ++j) {
                T tmp = t[j][i] < 0 ? -t[j][i] : t[j][i];
                if (tmp > pivotsize) {
                    pivot = j;
                    pivotsize = tmp;
                }
            }
            if (pivotsize == 0) return identity();
            if (pivot != i) {
                swapRows(t, i, pivot);
                swapRows(s, i, pivot);
            }
            for (int j = i + 1; j < 4; ++j) eliminate(j, i, t[j][i] / t[i][i]);
        }
        for (int i = 3; i >= 0; --i) {  // backward substitution
            T f = t[i][i];
            if (f == 0) return identity();
            for (int j = 0; j < 4; ++j) {
                t[i][j] /= f;
                s[i][j] /= f;
            }
            for (int j = 0; j < i; ++j) eliminate(j, i, t[j][i]);
        }
        return s;
    }
    const Mat44& invert() { return *this = inversed(); }

    static Mat44 identity() { return Mat44(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1); }

    static Mat44 rotation(const Vec3f& axis, const float& angle) {
        const float s = sin(ToRadian(angle)), c = cos(ToRadian(angle));
        const float ux = axis[0], uy = axis[1], uz = axis[2];
        Mat44 m;
        m[0][0] = c + ux * ux * (1.0f - c);
        m[0][1] = ux * uy * (1 - c) - uz * s;
        m[0][2] = ux * uz * (1 - c) + uy * s;
        m[1][0] = uy * ux * (1 - c) + uz * s;
        m[1][1] = c + uy * uy * (1 - c);
        m[1][2] = uy * uz * (1 - c) - ux * s;
        m[2][0] = uz * ux * (1 - c) - uy * s;
        m[2][1] = uz * uy * (1 - c) + ux * s;
        m[2][2] = c + uz * uz * (1 - c);
        m[3][3] = 1;
        return m;
    }
    static Mat44 rotationX(const float& angle) {
        const float s = sin(ToRadian(angle)), c = cos(ToRadian(angle));
        return Mat44(1.0f, 0.0f, 0.0f, 0.0f, 0.0f, c, -s, 0.0f, 0.0f, s, c, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f);
    }
    static Mat44 rotationY(const float& angle) {
        const float s = sin(ToRadian(angle)), c = cos(ToRadian(angle));
        return Mat44(c, 0.0f, s, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, -s, 0.0f, c, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f);
    }
    static Mat44 rotationZ(const float& angle) {
        const float s = sin(ToRadian(angle)), c = cos(ToRadian(angle));
        return Mat44(c, -s, 0.0f, 0.0f, s, c, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f);
    }
    static Mat44 translate(const float tx, const float ty, const float tz) {
        return Mat44(1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, tx, ty, tz, 1.0f);
    }
    static Mat44 scale(const float sx, const float sy, const float sz) {
        return Mat44(sx, 0.0f, 0.0f, 0.0f, 0.0f, sy, 0.0f, 0.0f, 0.0f, 0.0f, sz, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f);
    }
    static Mat44 scale(const float s) {
        return Mat44(1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f / s);
    }
};

typedef Mat44<float> Mat44f;

}  // namespace rt
#endif
