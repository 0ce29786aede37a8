// Vec<T, Size>: the small fixed-size vector the host API is written in.
// Keeps the behaviour of the reference's class (cudaTracer/Vec.cuh:12-263),
// including its quirks, because scene-building code relies on them:
//   * a constructor call with fewer than Size values zero-fills the rest, so
//     Vec3f(x) is (x, 0, 0), not a splat (Vec.cuh:21-22);
//   * scalar operands are always `float`, also for integer vectors, so
//     Vec3i(10, 20, 30) / 3 truncates 3.33 -> 3 (Vec.cuh:54-60);
//   * `r - v` and `r / v` with the scalar on the left compute v - r and v / r
//     (Vec.cuh:177-197);
//   * normalized() returns a zero vector unchanged (Vec.cuh:220-230);
//   * dot() accumulates x, y, z in order starting from 0 (Vec.cuh:232-238).
#ifndef RT_HOST_VEC_H
#define RT_HOST_VEC_H

#include <cassert>
#include <cmath>
#include <type_traits>

namespace rt {

namespace vectorAxes {
enum EUCLIDEAN { X, Y, Z, W };
enum UV { U, V };
enum ST { S, T };
}  // namespace vectorAxes

template <typename T, int Size>
class Vec {
   public:
    T data[Size];
    enum AXIS { X, Y, Z, W };

    constexpr Vec() : data{} {}
    template <typename... Arg, typename = std::enable_if_t<(sizeof...(Arg) >= 1) && (sizeof...(Arg) <= Size) &&
                                                           (std::is_arithmetic_v<Arg> && ...)>>
    constexpr Vec(Arg... args) : data{static_cast<T>(args)...} {}
    explicit Vec(const T* d) {
        for (int i = 0; i < Size; ++i) data[i] = d[i];
    }

    T& operator[](int idx) { return data[idx]; }
    const T& operator[](int idx) const { return data[idx]; }

#define RT_VEC_OPERATOR(OP)                                       \
    Vec& operator OP##=(const float& r) {                         \
        for (int i = 0; i < Size; ++i) data[i] OP## = r;          \
        return *this;                                             \
    }                                                             \
    Vec& operator OP##=(const Vec& o) {                           \
        for (int i = 0; i < Size; ++i) data[i] OP## = o.data[i];  \
        return *this;                                             \
    }                                                             \
    Vec operator OP(const float& r) const { return Vec(*this) OP## = r; } \
    Vec operator OP(const Vec& o) const { return Vec(*this) OP## = o; }   \
    /* scalar on the left: same as scalar on the right, for every operator */ \
    friend Vec operator OP(const float& r, const Vec& v) { return v OP r; }
    RT_VEC_OPERATOR(+)
    RT_VEC_OPERATOR(-)
    RT_VEC_OPERATOR(*)
    RT_VEC_OPERATOR(/)
#undef RT_VEC_OPERATOR

    Vec operator-() const {
        Vec v;
        for (int i = 0; i < Size; ++i) v.data[i] = -data[i];
        return v;
    }
    friend bool operator==(const Vec& a, const Vec& b) {
        for (int i = 0; i < Size; ++i)
            if (a.data[i] != b.data[i]) return false;
        return true;
    }
    friend bool operator!=(const Vec& a, const Vec& b) { return !(a == b); }

    float magnitude() const {
        float mag2 = 0.0f;
        for (int i = 0; i < Size; ++i) mag2 += data[i] * data[i];
        return sqrtf(mag2);
    }
    Vec normalized() const {
        float mag2 = 0.0f;
        for (int i = 0; i < Size; ++i) mag2 += data[i] * data[i];
        if (mag2 > 0) {
            float invMag = 1 / sqrtf(mag2);
            return *this * invMag;
        }
        return *this;
    }
    T dot(const Vec& o) const {
        T sum = 0;
        for (int i = 0; i < Size; ++i) sum += data[i] * o.data[i];
        return sum;
    }
    Vec cross(const Vec& o) const {
        static_assert(Size == 3, "cross product needs three components");
        return Vec(data[1] * o[2] - data[2] * o[1], data[2] * o[0] - data[0] * o[2], data[0] * o[1] - data[1] * o[0]);
    }
    // I - N * 2 * dot(I, N), evaluated as (N * 2) * dot (Vec.cuh:249-251)
    Vec reflect(const Vec& N) const { return *this - N * 2 * dot(N); }
};

typedef Vec<float, 3> Vec3f;
typedef Vec<float, 2> Vec2f;
typedef Vec<int, 3> Vec3i;
typedef Vec<int, 2> Vec2i;

}  // namespace rt
#endif
