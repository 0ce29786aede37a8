// Known-answer tests of the host API classes.  Every case is one TEST_CASE /
// SECTION of the reference's own unit tests, with the same operands and the same
// expected values, applied to this library's Vec / Mat44 / BoundingBox /
// Triangle / BVH::PreNode:
//   cudaTracer/VectorTest.cpp:6-341   cudaTracer/Tests.cpp:45-243
//   cudaTracer/BoundingBoxTest.cpp:6-240   cudaTracer/BvhTest.cpp:7-19
// (Tests.cpp:10-43, the GlobalCudaArray allocation test, is device memory
// plumbing that the C-ABI replaces and is covered by the GPU tests.)
#include <cmath>
#include <cstdio>
#include <limits>
#include <string>

#include "BVH.h"
#include "BoundingBox.h"
#include "Mat.h"
#include "Vec.h"

using namespace rt;
using namespace rt::vectorAxes;

namespace {
struct Runner {
    std::string out;
    int failures = 0;
    const char* current = "";
    bool ok = true;
    void begin(const char* name) {
        end();
        current = name;
        ok = true;
    }
    void end() {
        if (*current) {
            out += std::string(ok ? "PASS " : "FAIL ") + current + "\n";
            if (!ok) failures++;
        }
        current = "";
    }
    void require(bool cond) { ok = ok && cond; }
};
// Catch2's Approx with its default epsilon (100 * float epsilon, relative) and an optional absolute margin
bool approx(double value, double target, double margin = 0.0) {
    const double eps = std::numeric_limits<float>::epsilon() * 100.0;
    return std::fabs(value - target) <= margin || std::fabs(value - target) <= eps * std::fabs(target);
}
}  // namespace

#define REQUIRE(c) r.require(c)

extern "C" int rth_selftest(char* log, int logLen) {
    Runner r;
    {
        Vec3i x(1, 0, 0), y(0, 1, 0), z(0, 0, 1), v(1, 2, 3);
        r.begin("Vectors/ctor + copy ctor (VectorTest.cpp:13-25)");
        { Vec3i a(1, 2, 3); REQUIRE(a[Vec3i::X] == 1 && a[Vec3i::Y] == 2 && a[Vec3i::Z] == 3); Vec3i b(v); REQUIRE(b == a); }
        r.begin("Vectors/operator * with scalar (VectorTest.cpp:27-37)");
        REQUIRE(x * 3 == Vec3i(3, 0, 0)); REQUIRE(v * 3 == Vec3i(3, 6, 9));
        r.begin("Vectors/operator * with vector (VectorTest.cpp:39-49)");
        REQUIRE(x * y == Vec3i(0, 0, 0)); REQUIRE(v * z == Vec3i(0, 0, 3));
        r.begin("Vectors/operator / with scalar (VectorTest.cpp:51-61)");
        REQUIRE(x / 1 == Vec3i(1, 0, 0)); REQUIRE(v * 6 / 3 == Vec3i(2, 4, 6));
        r.begin("Vectors/operator / with vector (VectorTest.cpp:63-78)");
        REQUIRE(x / v == Vec3i(1, 0, 0)); REQUIRE(z / v == Vec3i(0, 0, 0)); REQUIRE(Vec3i(10, 20, 30) / v == Vec3i(10, 10, 10));
        r.begin("Vectors/operator - with scalar (VectorTest.cpp:80-90)");
        REQUIRE(x - 3 == Vec3i(-2, -3, -3)); REQUIRE(v - 3 == Vec3i(-2, -1, 0));
        r.begin("Vectors/operator - with vector (VectorTest.cpp:92-102)");
        REQUIRE(x - y == Vec3i(1, -1, 0)); REQUIRE(v - z == Vec3i(1, 2, 2));
        r.begin("Vectors/operator + with scalar (VectorTest.cpp:104-114)");
        REQUIRE(x + 3 == Vec3i(4, 3, 3)); REQUIRE(v + 3 == Vec3i(4, 5, 6));
        r.begin("Vectors/operator + with vector (VectorTest.cpp:116-126)");
        REQUIRE(x + y == Vec3i(1, 1, 0)); REQUIRE(v + z == Vec3i(1, 2, 4));
        r.begin("Vectors/negation (VectorTest.cpp:128-138)");
        REQUIRE(-x == Vec3i(-1, 0, 0)); REQUIRE(-v == Vec3i(-1, -2, -3));
        r.begin("Vectors/operator += (VectorTest.cpp:140-166)");
        { Vec3i a(x); a += 3; REQUIRE(a == Vec3i(4, 3, 3)); Vec3i b(v); b += 3; REQUIRE(b == Vec3i(4, 5, 6));
          Vec3i c(x); c += y; REQUIRE(c == Vec3i(1, 1, 0)); Vec3i d(v); d += z; REQUIRE(d == Vec3i(1, 2, 4)); }
        r.begin("Vectors/operator -= (VectorTest.cpp:168-194)");
        { Vec3i a(x); a -= 3; REQUIRE(a == Vec3i(-2, -3, -3)); Vec3i b(v); b -= 3; REQUIRE(b == Vec3i(-2, -1, 0));
          Vec3i c(x); c -= y; REQUIRE(c == Vec3i(1, -1, 0)); Vec3i d(v); d -= z; REQUIRE(d == Vec3i(1, 2, 2)); }
        r.begin("Vectors/operator *= (VectorTest.cpp:196-222)");
        { Vec3i a(x); a *= 3; REQUIRE(a == Vec3i(3, 0, 0)); Vec3i b(v); b *= 3; REQUIRE(b == Vec3i(3, 6, 9));
          Vec3i c(x); c *= y; REQUIRE(c == Vec3i(0, 0, 0)); Vec3i d(v); d *= z; REQUIRE(d == Vec3i(0, 0, 3)); }
        r.begin("Vectors/operator /= (VectorTest.cpp:224-260)");
        { Vec3i a(x); a /= 3; REQUIRE(a == Vec3i(0, 0, 0)); Vec3i b(v); b /= 3; REQUIRE(b == Vec3i(0, 0, 1));
          REQUIRE(Vec3i(10, 20, 30) / 3 == Vec3i(3, 6, 10)); REQUIRE(Vec3i(10, 20, 30) / v == Vec3i(10, 10, 10)); }
        r.begin("Vectors/[] write and read (VectorTest.cpp:262-271)");
        { Vec3i w(x); w[Vec3i::X] = 2; REQUIRE(w[Vec3i::X] == 2); REQUIRE(x[Vec3i::X] == 1 && x[Vec3i::Y] == 0 && x[Vec3i::Z] == 0); }
        r.begin("Vectors/vector-vector forms of + - * / (VectorTest.cpp:273-299)");
        REQUIRE(z + v == Vec3i(1, 2, 4)); REQUIRE(z - v == Vec3i(-1, -2, -2)); REQUIRE(z * v == Vec3i(0, 0, 3));
        REQUIRE(Vec3i(10, 4, 90) / v == Vec3i(10, 2, 30));
        r.begin("Vectors/== and != (VectorTest.cpp:301-303)");
        REQUIRE(v == Vec3i(1, 2, 3)); REQUIRE(x != y);
        r.begin("Vectors/magnitude (VectorTest.cpp:305-308)");
        REQUIRE(approx(x.magnitude(), 1)); REQUIRE(approx(v.magnitude(), 3.75, 0.01));
        r.begin("Vectors/normalized incl. integer truncation (VectorTest.cpp:310-317)");
        REQUIRE(x.normalized() == Vec3i(1, 0, 0)); REQUIRE(v.normalized() == Vec3i(0, 0, 0));
        { Vec3f a = Vec3f(1.0f, 2.0f, 3.0f).normalized(); REQUIRE(approx(a[0], 0.267, 0.01) && approx(a[1], 0.535, 0.01) && approx(a[2], 0.802, 0.01)); }
        r.begin("Vectors/dot (VectorTest.cpp:319-324)");
        REQUIRE(x.dot(y) == 0 && x.dot(v) == 1 && y.dot(v) == 2 && v.dot(v) == 14);
        r.begin("Vectors/cross (VectorTest.cpp:326-331)");
        REQUIRE(x.cross(y) == z); REQUIRE(y.cross(x) == -z);
        r.begin("Vectors/(v*2)-1 changes every axis (VectorTest.cpp:333-340)");
        { Vec3f n(0.5f, 1.0f, 0.1f); n = ((n * 2.0f) - 1.0f); REQUIRE(approx(n[0], 0.0, 1e-12) && approx(n[1], 1.0) && approx(n[2], -0.8)); }
        r.begin("Vectors/scalar-on-the-left forms and short constructor (Vec.cuh:21-22,170-197)");
        { Vec3f q = 1.0f - Vec3f(0.25f, 0.5f, 2.0f); REQUIRE(q == Vec3f(-0.75f, -0.5f, 1.0f)); REQUIRE(Vec3f(7.0f) == Vec3f(7.0f, 0.0f, 0.0f));
          REQUIRE(2.0f / Vec3f(4.0f, 2.0f, 1.0f) == Vec3f(2.0f, 1.0f, 0.5f)); REQUIRE(Vec3f().normalized() == Vec3f(0.0f, 0.0f, 0.0f)); }
    }
    {
        Mat44<int> I = Mat44<int>::identity();
        r.begin("Matrix multiplication/I*I=I (Tests.cpp:47-50)");
        REQUIRE(I * I == I);
        r.begin("Matrix multiplication/R*I=I*R=R (Tests.cpp:52-56)");
        { Mat44<int> R = Mat44<int>::rotationZ(90.0f); REQUIRE(I * R == R); REQUIRE(R * I == R); }
        r.begin("Matrix multiplication/M*N != N*M (Tests.cpp:58-64)");
        { Mat44<int> M(1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 1, 2, 3, 4, 5, 6), N(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 1, 2, 3, 4, 5); REQUIRE(N * M != M * N); }
    }
    {
        Vec<int, 3> origo(0, 0, 0), fours(4, 4, 4);
        Mat44<int> translation(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 4, 16, 44, 4);
        Mat44<int> scalingXYZ(1, 0, 0, 0, 0, 2, 0, 0, 0, 0, 4, 0, 0, 0, 0, 1);
        Mat44<int> scalingW(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 4);
        r.begin("Matrix-vector/Points: translation applies, w divides (Tests.cpp:71-88)");
        REQUIRE(translation.multPoint(origo) == Vec3i(1, 4, 11)); REQUIRE(translation.multPoint(fours) == Vec3i(2, 5, 12));
        r.begin("Matrix-vector/Points: XYZ scaling (Tests.cpp:90-107)");
        REQUIRE(scalingXYZ.multPoint(origo) == Vec3i(0, 0, 0)); REQUIRE(scalingXYZ.multPoint(fours) == Vec3i(4, 8, 16));
        r.begin("Matrix-vector/Points: W scaling (Tests.cpp:109-126)");
        REQUIRE(scalingW.multPoint(origo) == Vec3i(0, 0, 0)); REQUIRE(scalingW.multPoint(fours) == Vec3i(1, 1, 1));
        r.begin("Matrix-vector/Vectors: no translation (Tests.cpp:132-149)");
        REQUIRE(translation.multVec(origo) == Vec3i(0, 0, 0)); REQUIRE(translation.multVec(fours) == Vec3i(4, 4, 4));
        r.begin("Matrix-vector/Vectors: XYZ scaling (Tests.cpp:151-168)");
        REQUIRE(scalingXYZ.multVec(origo) == Vec3i(0, 0, 0)); REQUIRE(scalingXYZ.multVec(fours) == Vec3i(4, 8, 16));
        r.begin("Matrix-vector/Vectors: W ignored (Tests.cpp:170-187)");
        REQUIRE(scalingW.multVec(origo) == Vec3i(0, 0, 0)); REQUIRE(scalingW.multVec(fours) == Vec3i(4, 4, 4));
    }
    {
        const Vec3f right(1.0f, 0.0f, 0.0f), up(0.0f, 1.0f, 0.0f), forward(0.0f, 0.0f, 1.0f);
        auto near3 = [](const Vec3f& a, const Vec3f& b) { return approx(a[0], b[0], 0.1) && approx(a[1], b[1], 0.1) && approx(a[2], b[2], 0.1); };
        r.begin("Rotation matrix/X (Tests.cpp:199-212)");
        { Vec3f a = Mat44f::rotation(right, 90.0f).multVec(up), b = Mat44f::rotationX(90.0f).multVec(up); REQUIRE(near3(a, Vec3f(0.0f, 0.0f, -1.0f)) && near3(a, b)); }
        r.begin("Rotation matrix/Y (Tests.cpp:214-227)");
        { Vec3f a = Mat44f::rotation(up, 90.0f).multVec(right), b = Mat44f::rotationY(90.0f).multVec(right); REQUIRE(near3(a, Vec3f(0.0f, 0.0f, 1.0f)) && near3(a, b)); }
        r.begin("Rotation matrix/Z (Tests.cpp:229-242)");
        { Vec3f a = Mat44f::rotation(forward, 90.0f).multVec(right), b = Mat44f::rotationZ(90.0f).multVec(right); REQUIRE(near3(a, Vec3f(0.0f, -1.0f, 0.0f)) && near3(a, b)); }
    }
    {
        r.begin("Triangle/ctors (BoundingBoxTest.cpp:6-21)");
        { REQUIRE(Triangle()[0] == -1); Triangle t(1, 2, 3); REQUIRE(t[0] == 1 && t[1] == 2 && t[2] == 3);
          int values[5] = {0, 1, 2, 3, 4}; Triangle u(1, values); REQUIRE(u[0] == 1 && u[1] == 2 && u[2] == 3); }
        r.begin("BoundingBox/from vectors (BoundingBoxTest.cpp:24-34)");
        { BoundingBox bb(Vec3f(0.0f, 1.0f, 2.0f), Vec3f(3.0f, 4.0f, 5.0f)); REQUIRE(bb.bbmin == Vec3f(0.0f, 1.0f, 2.0f) && bb.bbmax == Vec3f(3.0f, 4.0f, 5.0f)); }
        r.begin("BoundingBox/from triangle + centroid (BoundingBoxTest.cpp:36-52)");
        { std::vector<Vertex> vs = {Vertex(-2.0f, -1.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.1f, 0.2f), Vertex(-1.0f, 0.0f, 1.0f, 0.0f, 1.0f, 0.0f, 0.3f, 0.4f),
                                    Vertex(1.0f, 2.0f, 3.0f, 1.0f, 0.0f, 1.0f, 0.5f, 0.6f)};
          Triangle tri = {0, 1, 2}; BoundingBox bb(tri, vs);
          REQUIRE(bb.bbmin == Vec3f(-2.0f, -1.0f, 0.0f) && bb.bbmax == Vec3f(1.0f, 2.0f, 3.0f)); REQUIRE(bb.centroid() == Vec3f(-0.5f, 0.5f, 1.5f)); }
        r.begin("BoundingBox/centroid (BoundingBoxTest.cpp:54-63)");
        REQUIRE(BoundingBox(Vec3f(0.0f, 1.0f, 2.0f), Vec3f(1.0f, 4.0f, 5.0f)).centroid() == Vec3f(0.5f, 2.5f, 3.5f));
        r.begin("BoundingBox/diagonal (BoundingBoxTest.cpp:65-95)");
        REQUIRE(BoundingBox(Vec3f(1.0f, 0.0f, 0.0f), Vec3f(5.0f, 2.0f, 1.0f)).diagonal() == Vec3f(4.0f, 2.0f, 1.0f));
        REQUIRE(BoundingBox(Vec3f(0.0f, 1.0f, 2.0f), Vec3f(1.0f, 4.0f, 5.0f)).diagonal() == Vec3f(1.0f, 3.0f, 3.0f));
        REQUIRE(BoundingBox(Vec3f(1.0f, 0.0f, 0.0f), Vec3f(2.0f, 1.0f, 1.0f)).diagonal() == Vec3f(1.0f, 1.0f, 1.0f));
        r.begin("BoundingBox/maxExtent ties go to Z (BoundingBoxTest.cpp:97-113)");
        REQUIRE(BoundingBox(Vec3f(1.0f, 0.0f, 0.0f), Vec3f(5.0f, 2.0f, 1.0f)).maxExtent() == 0);
        REQUIRE(BoundingBox(Vec3f(0.0f, 1.0f, 2.0f), Vec3f(1.0f, 4.0f, 5.0f)).maxExtent() == 2);
        REQUIRE(BoundingBox(Vec3f(1.0f, 0.0f, 0.0f), Vec3f(2.0f, 1.0f, 1.0f)).maxExtent() == 2);
        r.begin("BoundingBox/union chain from the empty box (BoundingBoxTest.cpp:115-149)");
        { BoundingBox bb1; Vec3f a(1.0f, 1.0f, 1.0f), b(2.0f, 2.0f, 2.0f), c(3.0f, 3.0f, 3.0f), d(4.0f, 4.0f, 4.0f);
          BoundingBox bb5 = BoundingBox::unionn(bb1, BoundingBox(a, b)); REQUIRE(bb5.bbmin == a && bb5.bbmax == b);
          BoundingBox bb6 = BoundingBox::unionn(bb5, BoundingBox(b, c)); REQUIRE(bb6.bbmin == a && bb6.bbmax == c);
          BoundingBox bb7 = BoundingBox::unionn(bb6, BoundingBox(c, d)); REQUIRE(bb7.bbmin == a && bb7.bbmax == d); }
        r.begin("BoundingBox/union does not swap inverted extents (BoundingBoxTest.cpp:151-166)");
        { Vec3f a(0.0f, 1.0f, 2.0f), b(-1.0f, 4.0f, 5.0f); BoundingBox u = BoundingBox::unionn(BoundingBox(a, b), BoundingBox(a + 2, b + 2));
          REQUIRE(!approx(u.bbmin[X], -1.0) && approx(u.bbmin[X], -0.0, 1e-12)); REQUIRE(approx(u.bbmin[Y], 1.0) && approx(u.bbmin[Z], 2.0));
          REQUIRE(u.bbmax == Vec3f(1.0f, 6.0f, 7.0f)); }
        r.begin("BoundingBox/union with a point (BoundingBoxTest.cpp:168-180)");
        { Vec3f a(0.0f, 1.0f, 2.0f), b(1.0f, 4.0f, 5.0f); BoundingBox u = BoundingBox::unionn(BoundingBox(a, b), b * 3);
          REQUIRE(u.bbmin == a && u.bbmax == Vec3f(3.0f, 12.0f, 15.0f)); }
        for (int sign = 1; sign >= -1; sign -= 2) {
            r.begin(sign > 0 ? "BoundingBox/fast intersect, ray +Z (BoundingBoxTest.cpp:183-210)" : "BoundingBox/fast intersect, ray -Z (BoundingBoxTest.cpp:212-239)");
            Ray ray(Vec3f(0.0f, 0.0f, 0.0f), Vec3f(0.0f, 0.0f, (float)sign));
            Vec3f invDir(1.0f / ray.dir[X], 1.0f / ray.dir[Y], 1.0f / ray.dir[Z]);
            int dirIsNeg[3] = {invDir[X] < 0, invDir[Y] < 0, invDir[Z] < 0};
            REQUIRE(BoundingBox(Vec3f(-1.0f, -1.0f, -1.0f), Vec3f(1.0f, 1.0f, 1.0f)).intersect(ray, invDir, dirIsNeg));
            if (sign > 0) {
                REQUIRE(BoundingBox(Vec3f(-1.0f, -1.0f, 1.0f), Vec3f(1.0f, 1.0f, 3.0f)).intersect(ray, invDir, dirIsNeg));
                REQUIRE(!BoundingBox(Vec3f(0.5f, 0.5f, 1.0f), Vec3f(1.0f, 1.0f, 3.0f)).intersect(ray, invDir, dirIsNeg));
            } else {
                REQUIRE(BoundingBox(Vec3f(-1.0f, -1.0f, -1.0f), Vec3f(1.0f, 1.0f, -3.0f)).intersect(ray, invDir, dirIsNeg));
                REQUIRE(!BoundingBox(Vec3f(0.5f, 0.5f, -1.0f), Vec3f(1.0f, 1.0f, -3.0f)).intersect(ray, invDir, dirIsNeg));
            }
        }
        r.begin("PreNode (BvhTest.cpp:7-19)");
        { BoundingBox bb(Vec3f(-3.0f, -2.0f, -1.0f), Vec3f(0.0f, 1.0f, 2.0f)); BVH::PreNode n(0, bb);
          REQUIRE(n.bb.bbmin == bb.bbmin && n.bb.bbmax == bb.bbmax); REQUIRE(n.centroid == Vec3f(-1.5f, -0.5f, 0.5f)); }
    }
    r.end();
    if (log && logLen > 0) std::snprintf(log, logLen, "%s", r.out.c_str());
    return r.failures;
}
