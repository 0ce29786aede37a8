// oracle.cpp - CPU restatement of the reference renderer's trace/shade path.
//
// TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP kernels: it
// may be called from tests/, from __graft_entry__.smoke() and from bench.py's
// cpu_baseline leg, and from nowhere else.  Nothing in the shipped library
// links, loads or includes it.
//
// What it follows: the reference's device code, function by function, with the
// same operation order in fp32 (build with -ffp-contract=off, no fast-math):
//   cudaTracer/Main.cu:43-98        per-pixel ray generation, gamma, store
//   cudaTracer/tracer.cu:15-215     castRay, fresnel, refract, shading, bias
//   cudaTracer/Scene.cuh:21-87      scene-BVH walk / linear scan
//   cudaTracer/Shape.cuh:42-87      sphere / mesh dispatch
//   cudaTracer/Mesh.cuh:65-212      object-space transform, mesh-BVH walk,
//                                   Moller-Trumbore, attribute interpolation
//   cudaTracer/Sphere.cuh:9-57      quadratic + sphere test
//   cudaTracer/BoundingBox.cuh:17-139 box type + slab test
//   cudaTracer/Texture.cuh:23-69    bilinear RGBA8 sampling
//   cudaTracer/Vec.cuh, Mat.cuh     vector / matrix arithmetic
//   cudaTracer/BVH.h:63-181         median-split BVH build + flatten
//
// Pinning status: the reference cannot be compiled in this environment without
// writing stand-ins for <cuda_runtime.h> / thrust, which is not done.  The
// restatement is pinned by the known-answer assertions of the reference's own
// unit tests (VectorTest.cpp, Tests.cpp, BoundingBoxTest.cpp, BvhTest.cpp),
// restated in oracle_selftest() below.  Those tests cover Vec, Mat44,
// BoundingBox (incl. the fast slab test), Triangle and PreNode; they do NOT
// cover traversal, Moller-Trumbore, the sphere test, texture filtering or
// shading, for which the reference holds no vectors: that part is PARITY
// UNPINNED (line-by-line restatement only).
//
// Deliberate deviations from the reference (all fence undefined behaviour):
//   * WRAP addressing with a negative texel index reads out of bounds in the
//     reference (cudaTracer/Texture.cuh:59-61, C '%' keeps the sign); here the
//     index wraps to [0, size).
//   * `sqrt` in solveQuadratic (cudaTracer/Sphere.cuh:18) is the float
//     overload under nvcc; sqrtf is used here.
//   * samples > 1 uses a jitter RNG that is racy and unseeded per pixel in the
//     reference (cudaTracer/Main.cu:60-61,108-110); rt_sample_jitter
//     (include/rt_capi.h) stands in for it, so frames with samples > 1 have no
//     counterpart in the reference at all (PARITY UNPINNED by construction).
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../include/rt_capi.h"

namespace ora {

// ---------------------------------------------------------------- Vec (Vec.cuh)
struct V3 {
    float d[3];
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
};
struct V2 {
    float d[2];
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
};
static inline V3 mk(float x, float y, float z) { return V3{{x, y, z}}; }
static inline V3 mk(const float* p) { return V3{{p[0], p[1], p[2]}}; }
// Vec.cuh:38-100: element-wise loops
static inline V3 muls(const V3& a, float r) { return mk(a[0] * r, a[1] * r, a[2] * r); }
static inline V3 divs(const V3& a, float r) { return mk(a[0] / r, a[1] / r, a[2] / r); }
static inline V3 subs(const V3& a, float r) { return mk(a[0] - r, a[1] - r, a[2] - r); }
static inline V3 adds(const V3& a, float r) { return mk(a[0] + r, a[1] + r, a[2] + r); }
static inline V3 mul(const V3& a, const V3& b) { return mk(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
static inline V3 sub(const V3& a, const V3& b) { return mk(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
static inline V3 add(const V3& a, const V3& b) { return mk(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
static inline V3 neg(const V3& a) { return mk(-a[0], -a[1], -a[2]); }
// Vec.cuh:232-238: sum starts at 0 and adds x, y, z products in order
static inline float dot(const V3& a, const V3& b) {
    float sum = 0;
    for (int i = 0; i < 3; ++i) sum += a[i] * b[i];
    return sum;
}
// Vec.cuh:240-247
static inline V3 cross(const V3& a, const V3& b) {
    return mk(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
// Vec.cuh:212-218
static inline float magnitude(const V3& a) {
    float mag2 = 0.0f;
    for (int i = 0; i < 3; ++i) mag2 += a[i] * a[i];
    return sqrtf(mag2);
}
// Vec.cuh:220-230: zero (and NaN) vectors are returned unchanged
static inline V3 normalized(const V3& a) {
    float mag2 = 0.0f;
    for (int i = 0; i < 3; ++i) mag2 += a[i] * a[i];
    if (mag2 > 0) {
        float invMag = 1 / sqrtf(mag2);
        return muls(a, invMag);
    }
    return a;
}
// Vec.cuh:249-251: *this - N * 2 * dot(N)  == I - ((N*2)*dot)
static inline V3 reflect(const V3& I, const V3& N) { return sub(I, muls(muls(N, 2), dot(I, N))); }

// ---------------------------------------------------------------- Mat44 (Mat.cuh)
struct M44 {
    float x[4][4];
};
static inline M44 fromRowMajor(const float* p) {
    M44 m;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) m.x[i][j] = p[i * 4 + j];
    return m;
}
static M44 identity() {
    M44 m;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) m.x[i][j] = (i == j) ? 1.0f : 0.0f;
    return m;
}
// Mat.cuh:81-89
static M44 multiply(const M44& a, const M44& b) {
    M44 c;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            c.x[i][j] = a.x[i][0] * b.x[0][j] + a.x[i][1] * b.x[1][j] + a.x[i][2] * b.x[2][j] +
                        a.x[i][3] * b.x[3][j];
    return c;
}
// Mat.cuh:127-134: row vector times matrix, translation row added, divide by w
static inline V3 multPoint(const M44& m, const V3& s) {
    float a = s[0] * m.x[0][0] + s[1] * m.x[1][0] + s[2] * m.x[2][0] + m.x[3][0];
    float b = s[0] * m.x[0][1] + s[1] * m.x[1][1] + s[2] * m.x[2][1] + m.x[3][1];
    float c = s[0] * m.x[0][2] + s[1] * m.x[1][2] + s[2] * m.x[2][2] + m.x[3][2];
    float w = s[0] * m.x[0][3] + s[1] * m.x[1][3] + s[2] * m.x[2][3] + m.x[3][3];
    return mk(a / w, b / w, c / w);
}
// Mat.cuh:143-149: no translation, no w
static inline V3 multVec(const M44& m, const V3& s) {
    float a = s[0] * m.x[0][0] + s[1] * m.x[1][0] + s[2] * m.x[2][0];
    float b = s[0] * m.x[0][1] + s[1] * m.x[1][1] + s[2] * m.x[2][1];
    float c = s[0] * m.x[0][2] + s[1] * m.x[1][2] + s[2] * m.x[2][2];
    return mk(a, b, c);
}
// Mat.cuh:169-247: Gauss-Jordan with partial pivoting; singular -> identity
static M44 inversed(const M44& in) {
    int i, j, k;
    M44 s = identity();
    M44 t = in;
    for (i = 0; i < 3; i++) {
        int pivot = i;
        float pivotsize = t.x[i][i];
        if (pivotsize < 0) pivotsize = -pivotsize;
        for (j = i + 1; j < 4; j++) {
            float tmp = t.x[j][i];
            if (tmp < 0) tmp = -tmp;
            if (tmp > pivotsize) {
                pivot = j;
                pivotsize = tmp;
            }
        }
        if (pivotsize == 0) return identity();
        if (pivot != i) {
            for (j = 0; j < 4; j++) {
                std::swap(t.x[i][j], t.x[pivot][j]);
                std::swap(s.x[i][j], s.x[pivot][j]);
            }
        }
        for (j = i + 1; j < 4; j++) {
            float f = t.x[j][i] / t.x[i][i];
            for (k = 0; k < 4; k++) {
                t.x[j][k] -= f * t.x[i][k];
                s.x[j][k] -= f * s.x[i][k];
            }
        }
    }
    for (i = 3; i >= 0; --i) {
        float f;
        if ((f = t.x[i][i]) == 0) return identity();
        for (j = 0; j < 4; j++) {
            t.x[i][j] /= f;
            s.x[i][j] /= f;
        }
        for (j = 0; j < i; j++) {
            f = t.x[j][i];
            for (k = 0; k < 4; k++) {
                t.x[j][k] -= f * t.x[i][k];
                s.x[j][k] -= f * s.x[i][k];
            }
        }
    }
    return s;
}
// globals.h:6-7
static const float kPI = 3.1415926536f;
static inline float toRadian(float deg) { return deg * (kPI / 180.0f); }
// Mat.cuh:306-336
static M44 rotation(const V3& axis, float angle) {
    M44 m;
    float sinA = sin(toRadian(angle));
    float cosA = cos(toRadian(angle));
    float ux = axis[0], uy = axis[1], uz = axis[2];
    m.x[0][0] = cosA + ux * ux * (1.0f - cosA);
    m.x[0][1] = ux * uy * (1 - cosA) - uz * sinA;
    m.x[0][2] = ux * uz * (1 - cosA) + uy * sinA;
    m.x[0][3] = 0;
    m.x[1][0] = uy * ux * (1 - cosA) + uz * sinA;
    m.x[1][1] = cosA + uy * uy * (1 - cosA);
    m.x[1][2] = uy * uz * (1 - cosA) - ux * sinA;
    m.x[1][3] = 0;
    m.x[2][0] = uz * ux * (1 - cosA) - uy * sinA;
    m.x[2][1] = uz * uy * (1 - cosA) + ux * sinA;
    m.x[2][2] = cosA + uz * uz * (1 - cosA);
    m.x[2][3] = 0;
    m.x[3][0] = m.x[3][1] = m.x[3][2] = 0;
    m.x[3][3] = 1;
    return m;
}
static M44 fromValues(float a, float b, float c, float d, float e, float f, float g, float h, float i,
                      float j, float k, float l, float mm, float n, float o, float p) {
    M44 m = {{{a, b, c, d}, {e, f, g, h}, {i, j, k, l}, {mm, n, o, p}}};
    return m;
}
// Mat.cuh:338-363
static M44 rotationX(float angle) {
    float s = sin(toRadian(angle)), c = cos(toRadian(angle));
    return fromValues(1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1);
}
static M44 rotationY(float angle) {
    float s = sin(toRadian(angle)), c = cos(toRadian(angle));
    return fromValues(c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1);
}
static M44 rotationZ(float angle) {
    float s = sin(toRadian(angle)), c = cos(toRadian(angle));
    return fromValues(c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1);
}

// ---------------------------------------------------------------- Ray (Ray.cuh)
struct Ray {
    V3 origin, dir;
    float tMax;
};
static inline Ray mkRay(const V3& o, const V3& d, float tMax = FLT_MAX) { return Ray{o, d, tMax}; }
static inline V3 rayAt(const Ray& r) { return add(r.origin, muls(r.dir, r.tMax)); }  // Ray.cuh:16

// ---------------------------------------------------------------- BoundingBox
// BoundingBox.cuh:14-19
static const float kMachineEpsilon = std::numeric_limits<float>::epsilon() * 0.5;
static inline float gammaN(int n) { return (n * kMachineEpsilon) / (1 - n * kMachineEpsilon); }

struct BB {
    V3 bbmin, bbmax;
};
// BoundingBox.cuh:25-27: max starts at FLT_MIN (smallest positive normal), not -FLT_MAX
static inline BB bbDefault() { return BB{mk(FLT_MAX, FLT_MAX, FLT_MAX), mk(FLT_MIN, FLT_MIN, FLT_MIN)}; }
// BoundingBox.cuh:29-43
static BB bbOfTriangle(const RtTriangle& t, const RtVertex* v) {
    BB b = bbDefault();
    for (int vi = 0; vi < 3; ++vi) {
        const float* pos = v[t.i[vi]].position;
        for (int a = 0; a < 3; ++a) {
            if (pos[a] > b.bbmax[a]) b.bbmax[a] = pos[a];
            if (pos[a] < b.bbmin[a]) b.bbmin[a] = pos[a];
        }
    }
    return b;
}
static inline V3 bbCentroid(const BB& b) { return divs(add(b.bbmin, b.bbmax), 2.0f); }  // :55
static inline V3 bbDiagonal(const BB& b) { return sub(b.bbmax, b.bbmin); }              // :57
// BoundingBox.cuh:59-68: ties go to Z, then Y
static inline int bbMaxExtent(const BB& b) {
    V3 d = bbDiagonal(b);
    if (d[0] > d[1] && d[0] > d[2]) return 0;
    if (d[1] > d[2]) return 1;
    return 2;
}
// BoundingBox.cuh:70-86
static inline BB bbUnion(const BB& a, const BB& b) {
    return BB{mk((std::min)(a.bbmin[0], b.bbmin[0]), (std::min)(a.bbmin[1], b.bbmin[1]),
                 (std::min)(a.bbmin[2], b.bbmin[2])),
              mk((std::max)(a.bbmax[0], b.bbmax[0]), (std::max)(a.bbmax[1], b.bbmax[1]),
                 (std::max)(a.bbmax[2], b.bbmax[2]))};
}
static inline BB bbUnion(const BB& b, const V3& p) {
    return BB{mk((std::min)(b.bbmin[0], p[0]), (std::min)(b.bbmin[1], p[1]), (std::min)(b.bbmin[2], p[2])),
              mk((std::max)(b.bbmax[0], p[0]), (std::max)(b.bbmax[1], p[1]), (std::max)(b.bbmax[2], p[2]))};
}
// BoundingBox.cuh:115-139: pbrt slab test; x and y far planes are padded, z is not
static inline bool bbIntersect(const float* bbmin, const float* bbmax, const Ray& ray, const V3& invDir,
                               const int dirIsNeg[3]) {
    const float* bounds[2] = {bbmin, bbmax};
    float tMin = (bounds[dirIsNeg[0]][0] - ray.origin[0]) * invDir[0];
    float tMax = (bounds[1 - dirIsNeg[0]][0] - ray.origin[0]) * invDir[0];
    float tyMin = (bounds[dirIsNeg[1]][1] - ray.origin[1]) * invDir[1];
    float tyMax = (bounds[1 - dirIsNeg[1]][1] - ray.origin[1]) * invDir[1];
    tMax *= 1 + 2 * gammaN(3);
    tyMax *= 1 + 2 * gammaN(3);
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = (bounds[dirIsNeg[2]][2] - ray.origin[2]) * invDir[2];
    float tzMax = (bounds[1 - dirIsNeg[2]][2] - ray.origin[2]) * invDir[2];
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    return (tMin < ray.tMax) && (tMax > 0);
}

// ---------------------------------------------------------------- BVH build (BVH.h)
struct PreNode {  // BVH.h:16-26
    int primIdx;
    BB bb;
    V3 centroid;
};
struct InterNode {  // BVH.h:28-61
    BB bb;
    InterNode* children[2] = {nullptr, nullptr};
    int splitAxis = -1, firstPrimOffset = -1, primitiveCnt = -1;
    bool isLeaf = false;
};
struct Builder {
    std::vector<PreNode> info;
    std::vector<int> ordered;  // original primitive index per leaf-order slot
    std::vector<RtLinearNode> nodes;
    std::vector<InterNode*> pool;
    int threshold = 1;
    bool tight = false;  // not in the reference: unions start from a true empty box
    BB empty() const { return tight ? BB{mk(FLT_MAX, FLT_MAX, FLT_MAX), mk(-FLT_MAX, -FLT_MAX, -FLT_MAX)} : bbDefault(); }

    // BVH.h:104-134
    InterNode* recursive(int start, int end, int& totalNodes) {
        InterNode* node = new InterNode();
        pool.push_back(node);
        totalNodes++;
        BB bb = empty();
        for (int i = start; i < end; ++i) bb = bbUnion(bb, info[i].bb);
        int primCnt = end - start;
        if (primCnt <= threshold) {
            leaf(bb, primCnt, start, end, node);
            return node;
        }
        BB centroidBounds = empty();
        for (int i = start; i < end; ++i) centroidBounds = bbUnion(centroidBounds, bbCentroid(info[i].bb));
        int dim = bbMaxExtent(centroidBounds);
        if (centroidBounds.bbmax[dim] == centroidBounds.bbmin[dim]) {
            leaf(bb, primCnt, start, end, node);
            return node;
        }
        // BVH.h:137-150: equal-count split by nth_element on the box centroid.
        int mid = (start + end) / 2;
        std::nth_element(&info[start], &info[mid], &info[end - 1] + 1, [dim](const PreNode& a, const PreNode& b) {
            return bbCentroid(a.bb)[dim] < bbCentroid(b.bb)[dim];
        });
        // The two recursive() calls are function arguments in the reference
        // (BVH.h:146-149), so their order is the compiler's choice; both MSVC
        // (the reference's toolchain) and g++ evaluate right to left, which
        // only decides the numbering of leaf slots, never the tree.
        InterNode* c1 = recursive(mid, end, totalNodes);
        InterNode* c0 = recursive(start, mid, totalNodes);
        node->children[0] = c0;
        node->children[1] = c1;
        node->bb = bbUnion(c0->bb, c1->bb);
        node->splitAxis = dim;
        node->primitiveCnt = primCnt;
        return node;
    }
    // BVH.h:152-160
    void leaf(const BB& bb, int primCnt, int start, int end, InterNode* node) {
        int first = (int)ordered.size();
        for (int i = start; i < end; ++i) ordered.push_back(info[i].primIdx);
        node->firstPrimOffset = first;
        node->primitiveCnt = primCnt;
        node->bb = bb;
        node->isLeaf = true;
    }
    // BVH.h:162-180: depth-first pre-order, first child at self + 1
    int flatten(InterNode* node, int* offset) {
        RtLinearNode* ln = &nodes[*offset];
        for (int a = 0; a < 3; ++a) {
            ln->bbmin[a] = node->bb.bbmin[a];
            ln->bbmax[a] = node->bb.bbmax[a];
        }
        int my = (*offset)++;
        if (node->isLeaf) {
            ln->offset = node->firstPrimOffset;
            ln->primCnt = (uint16_t)node->primitiveCnt;
        } else {
            ln->axis = (uint8_t)node->splitAxis;
            ln->primCnt = 0;
            flatten(node->children[0], offset);
            ln->offset = flatten(node->children[1], offset);
        }
        return my;
    }
    void run() {
        int total = 0;
        InterNode* root = recursive(0, (int)info.size(), total);
        nodes.assign(total, RtLinearNode{});
        int off = 0;
        flatten(root, &off);
        for (auto* p : pool) delete p;
        pool.clear();
    }
};

// ---------------------------------------------------------------- scene view
struct Counters {
    uint64_t c[RT_COUNTER_CNT] = {0};
};
struct SurfaceData {  // SurfaceData.cuh:10-19
    const RtShape* hit = nullptr;
    V3 hitPoint;
    RtTriangle triangle = {{-1, -1, -1}};
    V2 uv = {{0, 0}}, st = {{0, 0}};
    V3 n;
};
struct Ctx {
    const RtSceneDesc* scene;
    RtDeviceOpts opts;
    V3 orig;
    M44 camera;
    bool shadowAnyHit;
    uint32_t sampleSeed;
    Counters* cnt;
};

// ---------------------------------------------------------------- Sphere.cuh
static inline bool solveQuadratic(float a, float b, float c, float& x0, float& x1) {  // :9-28
    float discr = b * b - 4 * a * c;
    if (discr < 0) {
        return false;
    } else if (discr == 0) {
        x0 = x1 = -0.5f * b / a;
    } else {
        float q = (b > 0) ? -0.5f * (b + sqrtf(discr)) : -0.5f * (b - sqrtf(discr));
        x0 = q / a;
        x1 = c / q;
    }
    if (x0 > x1) std::swap(x0, x1);
    return true;
}
static inline bool sphereIntersect(const RtShape& s, Ray& ray) {  // :41-54
    V3 L = sub(ray.origin, mk(s.center));
    float a = dot(ray.dir, ray.dir);
    float b = 2 * dot(ray.dir, L);
    float c = dot(L, L) - s.radius2;
    float t0, t1;
    if (!solveQuadratic(a, b, c, t0, t1)) return false;
    if (t0 < 0) t0 = t1;
    if (t0 < 0) return false;
    if (t0 > ray.tMax) return false;
    ray.tMax = t0;
    return true;
}

// ---------------------------------------------------------------- Texture.cuh
static inline V3 texGetCol(const RtTexture& t, int x, int y) {  // :50-69
    if (t.mode == RT_TEX_CLAMP) {
        x = x < 0 ? 0 : x;
        x = x > t.width - 1 ? t.width - 1 : x;
        y = y < 0 ? 0 : y;
        y = y > t.height - 1 ? t.height - 1 : y;
    } else {
        x = x % t.width;
        y = y % t.height;
        if (x < 0) x += t.width;  // deviation: the reference reads out of bounds here
        if (y < 0) y += t.height;
    }
    float r = t.data[y * t.width * 4 + x * 4] / 255.0f;
    float g = t.data[y * t.width * 4 + x * 4 + 1] / 255.0f;
    float b = t.data[y * t.width * 4 + x * 4 + 2] / 255.0f;
    return mk(r, g, b);
}
static inline V3 texSample(const RtTexture& t, float u, float v) {  // :33-48
    u = u * t.width - 0.5;  // float product, double subtraction, rounded back to float
    v = v * t.height - 0.5;
    int x = floor(u);
    int y = floor(v);
    double u_ratio = u - x;  // float subtraction widened to double
    double v_ratio = v - y;
    double u_opposite = 1 - u_ratio;
    double v_opposite = 1 - v_ratio;
    // Vec::operator*(const float&): every double weight is rounded to float first
    V3 top = add(muls(texGetCol(t, x, y), (float)u_opposite), muls(texGetCol(t, x + 1, y), (float)u_ratio));
    V3 bot = add(muls(texGetCol(t, x, y + 1), (float)u_opposite), muls(texGetCol(t, x + 1, y + 1), (float)u_ratio));
    return add(muls(top, (float)v_opposite), muls(bot, (float)v_ratio));
}

// ---------------------------------------------------------------- Mesh.cuh
static inline bool rayTriangleIntersect(const RtMesh& m, const RtTriangle& tri, Ray& ray, SurfaceData& data) {  // :182-212
    V3 v0 = mk(m.vertices[tri.i[0]].position);
    V3 v1 = mk(m.vertices[tri.i[1]].position);
    V3 v2 = mk(m.vertices[tri.i[2]].position);
    V3 edge1 = sub(v1, v0);
    V3 edge2 = sub(v2, v0);
    V3 pvec = cross(ray.dir, edge2);
    float det = dot(edge1, pvec);
    if (det == 0 || det < 0) return false;
    V3 tvec = sub(ray.origin, v0);
    float u = dot(tvec, pvec);
    if (u < 0 || u > det) return false;
    V3 qvec = cross(tvec, edge1);
    float v = dot(ray.dir, qvec);
    if (v < 0 || u + v > det) return false;
    float invDet = 1 / det;
    float t = dot(edge2, qvec) * invDet;
    if (t < 0.0f || ray.tMax < t) return false;
    ray.tMax = t;
    data.uv[0] = u * invDet;
    data.uv[1] = v * invDet;
    data.triangle = tri;
    return true;
}

// Shared by Mesh::intersectWithBvh (Mesh.cuh:127-171) and
// Scene::intersectWithBvh (Scene.cuh:29-76): same walk, different leaf action.
template <class Leaf>
static inline bool walkBvh(const RtLinearNode* nodes, Ray& ray, Counters* cnt, bool anyHit, Leaf&& leafFn) {
    bool hit = false;
    V3 invDir = mk(1.0f / ray.dir[0], 1.0f / ray.dir[1], 1.0f / ray.dir[2]);
    int dirIsNeg[3] = {invDir[0] < 0, invDir[1] < 0, invDir[2] < 0};
    int toVisitOffset = 0;
    int currentNodeIdx = 0;
    int nodesToVisit[64];
    while (true) {
        const RtLinearNode* node = &nodes[currentNodeIdx];
        if (cnt) cnt->c[RT_CNT_NODE]++;
        if (bbIntersect(node->bbmin, node->bbmax, ray, invDir, dirIsNeg)) {
            if (node->primCnt > 0) {
                for (int i = 0; i < node->primCnt; ++i) {
                    if (leafFn(node->offset + i)) {
                        hit = true;
                        if (anyHit) return true;
                    }
                }
                if (toVisitOffset == 0) break;
                currentNodeIdx = nodesToVisit[--toVisitOffset];
            } else {
                if (dirIsNeg[node->axis]) {
                    nodesToVisit[toVisitOffset++] = currentNodeIdx + 1;
                    currentNodeIdx = node->offset;
                } else {
                    nodesToVisit[toVisitOffset++] = node->offset;
                    currentNodeIdx = currentNodeIdx + 1;
                }
            }
        } else {
            if (toVisitOffset == 0) break;
            currentNodeIdx = nodesToVisit[--toVisitOffset];
        }
    }
    return hit;
}

static inline bool meshIntersect(const Ctx& cx, const RtMesh& m, Ray& ray, SurfaceData& surface, bool anyHit) {  // :115-126
    M44 inv = fromRowMajor(m.transformInverse);
    Ray tr = mkRay(multPoint(inv, ray.origin), multVec(inv, ray.dir), ray.tMax);
    if (cx.cnt) cx.cnt->c[RT_CNT_MESH_ENTRY]++;
    bool hit = false;
    if (cx.opts.useMeshBvhs) {
        hit = walkBvh(m.nodes, tr, cx.cnt, anyHit, [&](int idx) {
            if (cx.cnt) cx.cnt->c[RT_CNT_TRI]++;
            return rayTriangleIntersect(m, m.triangles[idx], tr, surface);
        });
    } else {  // :173-179
        for (int i = 0; i < m.triangleCnt; ++i) {
            if (cx.cnt) cx.cnt->c[RT_CNT_TRI]++;
            hit |= rayTriangleIntersect(m, m.triangles[i], tr, surface);
            if (hit && anyHit) break;
        }
    }
    ray.tMax = tr.tMax;
    return hit;
}
template <int N, class T>
static inline T interpolate(const V2& uv, const T& v0, const T& v1, const T& v2) {  // :101-109
    T r;
    float w = 1 - uv[0] - uv[1];
    for (int i = 0; i < N; ++i) r.d[i] = v0.d[i] * w + v1.d[i] * uv[0] + v2.d[i] * uv[1];
    return r;
}
static inline V2 meshStCoords(const RtMesh& m, const RtTriangle& t, const V2& uv) {  // :65-71
    const RtVertex& a = m.vertices[t.i[0]];
    const RtVertex& b = m.vertices[t.i[1]];
    const RtVertex& c = m.vertices[t.i[2]];
    return interpolate<2>(uv, V2{{a.texCoord[0], a.texCoord[1]}}, V2{{b.texCoord[0], b.texCoord[1]}},
                          V2{{c.texCoord[0], c.texCoord[1]}});
}
static inline V3 meshNormal(const RtMesh& m, const RtTriangle& t, const V2& uv, const V2& st) {  // :73-95
    const RtVertex& a = m.vertices[t.i[0]];
    const RtVertex& b = m.vertices[t.i[1]];
    const RtVertex& c = m.vertices[t.i[2]];
    V3 tg = interpolate<3>(uv, mk(a.tangent), mk(b.tangent), mk(c.tangent));
    V3 bt = interpolate<3>(uv, mk(a.bitangent), mk(b.bitangent), mk(c.bitangent));
    V3 n = interpolate<3>(uv, mk(a.normal), mk(b.normal), mk(c.normal));
    V3 ns = texSample(m.normals, st[0], st[1]);
    ns = subs(muls(ns, 2.0f), 1.0f);
    // object-space normal; the reference never brings it to world space
    return normalized(add(add(n, muls(tg, ns[0])), muls(bt, ns[1])));
}

// ---------------------------------------------------------------- Shape.cuh
static inline bool shapeIntersect(const Ctx& cx, const RtShape& s, Ray& ray, SurfaceData& data, bool anyHit) {  // :81-87
    if (s.kind == RT_SHAPE_SPHERE) {
        if (cx.cnt) cx.cnt->c[RT_CNT_SPHERE]++;
        return sphereIntersect(s, ray);
    }
    return meshIntersect(cx, cx.scene->meshes[s.meshIdx], ray, data, anyHit);
}

// ---------------------------------------------------------------- Scene.cuh
static bool sceneIntersect(const Ctx& cx, Ray& ray, SurfaceData& surface, bool anyHit) {  // :21-87
    const RtSceneDesc& sc = *cx.scene;
    if (cx.opts.useSceneBvh) {
        return walkBvh(sc.nodes, ray, cx.cnt, anyHit, [&](int idx) {
            const RtShape* prim = &sc.shapes[idx];
            if (shapeIntersect(cx, *prim, ray, surface, anyHit)) {
                surface.hit = prim;
                return true;
            }
            return false;
        });
    }
    bool hit = false;
    for (int i = 0; i < sc.shapeCnt; ++i) {
        if (shapeIntersect(cx, sc.shapes[i], ray, surface, anyHit)) {
            hit = true;
            surface.hit = &sc.shapes[i];
            if (anyHit) break;
        }
    }
    return hit;
}

// ---------------------------------------------------------------- tracer.cu
static inline float clampf(float lo, float hi, float v) { return fmaxf(lo, fminf(hi, v)); }  // :54-56
static float fresnel(const V3& i, const V3& n, float ior) {                                 // :67-89
    float cosi = clampf(-1, 1, dot(i, n));
    float etai = 1, etat = ior;
    if (cosi > 0.0f) std::swap(etai, etat);
    float sint = etai / etat * sqrtf(fmaxf(0.f, 1 - cosi * cosi));
    if (sint >= 1) return 1;
    float cost = sqrtf(fmaxf(0.f, 1.0f - sint * sint));
    cosi = fabsf(cosi);
    float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return (Rs * Rs + Rp * Rp) * 0.5f;
}
static V3 refract(const V3& I, const V3& N, float ior) {  // :102-115
    float cosi = clampf(-1, 1, dot(I, N));
    float etai = 1, etat = ior;
    V3 n = N;
    if (cosi < 0) {
        cosi = -cosi;
    } else {
        std::swap(etai, etat);
        n = neg(N);
    }
    float eta = etai / etat;
    float k = 1 - eta * eta * (1 - cosi * cosi);
    if (k < 0) return mk(0, 0, 0);  // total internal reflection: Vec3f(0)
    return add(muls(I, eta), muls(n, eta * cosi - sqrtf(k)));
}
static inline V3 biasedOrigin(const Ctx& cx, const V3& dir, const SurfaceData& d) {  // :202-208
    if (dot(dir, d.n) < 0.0f) return add(d.hitPoint, muls(d.n, cx.opts.shadowBias));
    return sub(d.hitPoint, muls(d.n, cx.opts.shadowBias));
}
static inline V3 invBiasedOrigin(const Ctx& cx, const V3& dir, const SurfaceData& d) {  // :209-215
    if (dot(dir, d.n) < 0.0f) return sub(d.hitPoint, muls(d.n, cx.opts.shadowBias));
    return add(d.hitPoint, muls(d.n, cx.opts.shadowBias));
}

static V3 castRay(const Ctx& cx, Ray& ray, int depth, int32_t* primaryKey);

static V3 diffuseAndGlossy(const Ctx& cx, const V3& dir, SurfaceData& data) {  // :169-200
    const RtShape& shape = *data.hit;
    const RtSceneDesc& sc = *cx.scene;
    const bool isMesh = shape.kind == RT_SHAPE_MESH;
    V3 diffuseColor = isMesh ? texSample(sc.meshes[shape.meshIdx].diffuse, data.st[0], data.st[1])
                             : mk(shape.material.diffuseColor);
    V3 diffuseLight = mk(0, 0, 0), specularLight = mk(0, 0, 0);
    V3 shadowPointOrig = biasedOrigin(cx, dir, data);
    for (int i = 0; i < sc.lightCnt; ++i) {
        const RtLight& light = sc.lights[i];
        V3 lightDir = sub(mk(light.position), data.hitPoint);
        float distance = magnitude(lightDir);
        lightDir = normalized(lightDir);
        float lambert = fmaxf(0.f, dot(lightDir, data.n));
        Ray shadowRay = mkRay(shadowPointOrig, lightDir, distance);
        SurfaceData scratch;
        if (cx.cnt) cx.cnt->c[RT_CNT_SHADOW]++;
        bool visible = !sceneIntersect(cx, shadowRay, scratch, cx.shadowAnyHit);
        if (visible) {
            float specularFac = 0.0f;
            if (lambert > 0.0f) {
                // blinnPhong, :144-149
                V3 halfDir = normalized(sub(lightDir, dir));
                float specAngle = fmaxf(dot(halfDir, data.n), 0.0f);
                specularFac = powf(specAngle, shape.material.specularExponent);
            }
            float specMask = isMesh ? texSample(sc.meshes[shape.meshIdx].specular, data.st[0], data.st[1])[0]
                                    : shape.material.Ks;
            V3 lightComponent = divs(muls(mk(light.intensity), light.power), distance);
            diffuseLight = add(diffuseLight, muls(lightComponent, lambert));
            specularLight = add(specularLight, muls(muls(lightComponent, specMask), specularFac));
        }
    }
    return add(mul(diffuseLight, diffuseColor), specularLight);
}

static V3 castRay(const Ctx& cx, Ray& ray, int depth, int32_t* primaryKey) {  // :15-48
    V3 bg = mk(cx.opts.backgroundColor);
    if (depth > cx.opts.maxDepth) return bg;
    if (depth > 0 && cx.cnt) cx.cnt->c[RT_CNT_SECONDARY]++;
    SurfaceData sd;
    if (!sceneIntersect(cx, ray, sd, false)) return bg;
    if (primaryKey) {
        primaryKey[0] = sd.hit->tag;
        const bool isMesh = sd.hit->kind == RT_SHAPE_MESH;
        for (int k = 0; k < 3; ++k) primaryKey[1 + k] = isMesh ? sd.triangle.i[k] : -1;
    }
    const RtShape& shape = *sd.hit;
    sd.hitPoint = rayAt(ray);
    if (shape.kind == RT_SHAPE_SPHERE) {
        sd.st = V2{{1.0f, 1.0f}};                                   // Shape.cuh:44-46
        sd.n = normalized(sub(sd.hitPoint, mk(shape.center)));    // Sphere.cuh:55-57
        if (cx.cnt) cx.cnt->c[RT_CNT_SHADE_SPHERE]++;
    } else {
        const RtMesh& m = cx.scene->meshes[shape.meshIdx];
        sd.st = meshStCoords(m, sd.triangle, sd.uv);
        sd.n = meshNormal(m, sd.triangle, sd.uv, sd.st);
        if (cx.cnt) cx.cnt->c[RT_CNT_SHADE_MESH]++;
    }
    const V3 dir = ray.dir;
    switch (shape.material.materialType) {
        case RT_MAT_REFLECTION_AND_REFRACTION: {  // :117-132
            // note the swapped operands: the normal is reflected about the view direction
            V3 reflDir = normalized(reflect(sd.n, dir));
            Ray reflRay = mkRay(invBiasedOrigin(cx, reflDir, sd), reflDir);
            V3 reflColor = castRay(cx, reflRay, depth + 1, nullptr);
            const float ior = shape.material.ior;
            V3 refrDir = normalized(refract(dir, sd.n, ior));
            Ray refrRay = mkRay(invBiasedOrigin(cx, refrDir, sd), refrDir);
            V3 refrColor = castRay(cx, refrRay, depth + 1, nullptr);
            float kr = fresnel(dir, sd.n, ior);
            return add(muls(reflColor, kr), muls(refrColor, 1 - kr));
        }
        case RT_MAT_REFLECTION: {  // :134-142
            V3 reflDir = normalized(reflect(dir, sd.n));
            Ray reflRay = mkRay(invBiasedOrigin(cx, reflDir, sd), reflDir);
            V3 reflColor = castRay(cx, reflRay, depth + 1, nullptr);
            float kr = fresnel(dir, sd.n, shape.material.ior);
            return muls(reflColor, kr);
        }
        case RT_MAT_DIFFUSE_AND_GLOSSY:
            return diffuseAndGlossy(cx, dir, sd);
        default:
            return mk(1.0f, 0.5f, 0.25f);
    }
}

// Main.cu:43-98 for one pixel.  With samples == 1 both jitters are 0.0f; with more, the jitter is
// rt_sample_jitter (include/rt_capi.h), which replaces the reference's irreproducible curand draw.
static void renderPixel(const Ctx& cx, int x, int y, float* rgba, int32_t* key) {
    const RtDeviceOpts& o = cx.opts;
    float widthScale = 1 / (float)o.width;
    float heightScale = 1 / (float)o.height;
    float ndcX = (2.0f * (x + 0.5f) * widthScale - 1.0f) * o.imageAspectRatio * o.scale;
    float ndcY = (1.0f - 2.0f * (y + 0.5f) * heightScale) * o.scale;
    V3 pixCol = mk(0, 0, 0);
    if (key) key[0] = key[1] = key[2] = key[3] = -1;
    for (int i = 0; i < o.samples; ++i) {
        const uint32_t pixel = (uint32_t)y * (uint32_t)o.width + (uint32_t)x;
        float xJitter = o.samples > 1 ? rt_sample_jitter(cx.sampleSeed, pixel, (uint32_t)i, 0u) : 0.0f;
        float yJitter = o.samples > 1 ? rt_sample_jitter(cx.sampleSeed, pixel, (uint32_t)i, 1u) : 0.0f;
        V3 dir = normalized(mk(ndcX + xJitter * widthScale, ndcY + yJitter * heightScale, 1));
        dir = multVec(cx.camera, dir);
        dir = normalized(dir);
        Ray ray = mkRay(cx.orig, dir);
        if (cx.cnt) cx.cnt->c[RT_CNT_PRIMARY]++;
        pixCol = add(pixCol, castRay(cx, ray, 0, i == 0 ? key : nullptr));  // the hit key is sample 0's
    }
    for (int i = 0; i < 3; ++i) pixCol[i] /= o.samples;
    if (o.gammaCorrection) {
        const float screenGamma = 2.2;
        const float sgInv = 1.0f / screenGamma;
        rgba[0] = powf(pixCol[0], sgInv);
        rgba[1] = powf(pixCol[1], sgInv);
        rgba[2] = powf(pixCol[2], sgInv);
    } else {
        rgba[0] = pixCol[0];
        rgba[1] = pixCol[1];
        rgba[2] = pixCol[2];
    }
    rgba[3] = 1.0f;
}

}  // namespace ora

// =========================================================================
// C entry points (loaded with ctypes by the tests and the bench baseline)
// =========================================================================
static uint32_t g_sampleSeed = 1234;  // the product's default (rt_set_sample_seed)

extern "C" {

void oracle_set_sample_seed(uint32_t seed) { g_sampleSeed = seed; }

// Renders rows y = y0, y0+yStep, ... < y1 of the frame into tight
// width*height*4 float / int32 planes (other rows untouched).  Rows are dealt
// round-robin to nThreads host threads.  counters (RT_COUNTER_CNT uint64) are
// added to.  shadowAnyHit != 0 ends a shadow traversal at the first accepted
// primitive (same boolean as the reference's full traversal, fewer tests).
int oracle_render(const RtSceneDesc* scene, const RtDeviceOpts* opts, const RtCamera* cam, float* rgba,
                  int32_t* hitKeys, uint64_t* counters, int y0, int y1, int yStep, int nThreads,
                  int shadowAnyHit) {
    if (!scene || !opts || !cam || !rgba || opts->samples < 1 || yStep < 1) return -1;
    if (nThreads < 1) nThreads = 1;
    std::vector<int> rows;
    for (int y = y0; y < y1 && y < opts->height; y += yStep) rows.push_back(y);
    std::vector<ora::Counters> cnts(nThreads);
    auto work = [&](int tid) {
        ora::Ctx cx;
        cx.scene = scene;
        cx.opts = *opts;
        cx.orig = ora::mk(cam->orig);
        cx.camera = ora::fromRowMajor(cam->camera);
        cx.shadowAnyHit = shadowAnyHit != 0;
        cx.sampleSeed = g_sampleSeed;
        cx.cnt = counters ? &cnts[tid] : nullptr;
        for (size_t r = tid; r < rows.size(); r += nThreads) {
            int y = rows[r];
            for (int x = 0; x < opts->width; ++x) {
                size_t p = (size_t)y * opts->width + x;
                ora::renderPixel(cx, x, y, rgba + 4 * p, hitKeys ? hitKeys + 4 * p : nullptr);
            }
        }
    };
    if (nThreads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nThreads; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (counters)
        for (auto& c : cnts)
            for (int i = 0; i < RT_COUNTER_CNT; ++i) counters[i] += c.c[i];
    return 0;
}

// BVH over triangles (BVH.h:72-87): nodesOut receives at most 2*nt-1 nodes,
// orderOut[slot] = index into `tris` of the triangle stored at that leaf slot.
int oracle_bvh_triangles(const RtVertex* verts, const RtTriangle* tris, int nt, int threshold, int tight,
                         RtLinearNode* nodesOut, int* nodeCntOut, int* orderOut) {
    ora::Builder b;
    b.threshold = threshold;
    b.tight = tight != 0;
    b.info.resize(nt);
    for (int i = 0; i < nt; ++i) {
        ora::BB bb = ora::bbOfTriangle(tris[i], verts);
        if (tight) {
            bb = b.empty();
            for (int k = 0; k < 3; ++k) bb = ora::bbUnion(bb, ora::mk(verts[tris[i].i[k]].position));
        }
        b.info[i] = ora::PreNode{i, bb, ora::bbCentroid(bb)};
    }
    b.run();
    std::memcpy(nodesOut, b.nodes.data(), b.nodes.size() * sizeof(RtLinearNode));
    *nodeCntOut = (int)b.nodes.size();
    std::memcpy(orderOut, b.ordered.data(), b.ordered.size() * sizeof(int));
    return 0;
}
// BVH over boxes (BVH.h:89-102, threshold 1): boxes = n * {min[3], max[3]}.
int oracle_bvh_boxes(const float* boxes, int n, int tight, RtLinearNode* nodesOut, int* nodeCntOut, int* orderOut) {
    ora::Builder b;
    b.threshold = 1;
    b.tight = tight != 0;
    b.info.resize(n);
    for (int i = 0; i < n; ++i) {
        ora::BB bb{ora::mk(boxes + 6 * i), ora::mk(boxes + 6 * i + 3)};
        b.info[i] = ora::PreNode{i, bb, ora::bbCentroid(bb)};
    }
    b.run();
    std::memcpy(nodesOut, b.nodes.data(), b.nodes.size() * sizeof(RtLinearNode));
    *nodeCntOut = (int)b.nodes.size();
    std::memcpy(orderOut, b.ordered.data(), b.ordered.size() * sizeof(int));
    return 0;
}

// ---- single-function probes (golden-vector generation and unit parity) ----
int oracle_box_intersect(const float bbmin[3], const float bbmax[3], const float o[3], const float d[3],
                         float tMax) {
    ora::Ray r = ora::mkRay(ora::mk(o), ora::mk(d), tMax);
    ora::V3 inv = ora::mk(1.0f / d[0], 1.0f / d[1], 1.0f / d[2]);
    int neg[3] = {inv[0] < 0, inv[1] < 0, inv[2] < 0};
    return ora::bbIntersect(bbmin, bbmax, r, inv, neg) ? 1 : 0;
}
// out = {t, u, v}; returns 1 on hit
int oracle_triangle_intersect(const float v0[3], const float v1[3], const float v2[3], const float o[3],
                              const float d[3], float tMax, float out[3]) {
    RtVertex vs[3];
    std::memset(vs, 0, sizeof vs);
    std::memcpy(vs[0].position, v0, 12);
    std::memcpy(vs[1].position, v1, 12);
    std::memcpy(vs[2].position, v2, 12);
    RtMesh m;
    std::memset(&m, 0, sizeof m);
    m.vertices = vs;
    RtTriangle t = {{0, 1, 2}};
    ora::Ray r = ora::mkRay(ora::mk(o), ora::mk(d), tMax);
    ora::SurfaceData sd;
    bool hit = ora::rayTriangleIntersect(m, t, r, sd);
    out[0] = r.tMax;
    out[1] = sd.uv[0];
    out[2] = sd.uv[1];
    return hit ? 1 : 0;
}
int oracle_sphere_intersect(const float c[3], float radius, const float o[3], const float d[3], float tMax,
                            float* tOut) {
    RtShape s;
    std::memset(&s, 0, sizeof s);
    std::memcpy(s.center, c, 12);
    s.radius = radius;
    s.radius2 = radius * radius;
    ora::Ray r = ora::mkRay(ora::mk(o), ora::mk(d), tMax);
    bool hit = ora::sphereIntersect(s, r);
    *tOut = r.tMax;
    return hit ? 1 : 0;
}
float oracle_fresnel(const float i[3], const float n[3], float ior) { return ora::fresnel(ora::mk(i), ora::mk(n), ior); }
void oracle_refract(const float i[3], const float n[3], float ior, float out[3]) {
    ora::V3 r = ora::refract(ora::mk(i), ora::mk(n), ior);
    std::memcpy(out, r.d, 12);
}
void oracle_tex_sample(const RtTexture* t, float u, float v, float out[3]) {
    ora::V3 r = ora::texSample(*t, u, v);
    std::memcpy(out, r.d, 12);
}
void oracle_mat_inverse(const float in[16], float out[16]) {
    ora::M44 m = ora::inversed(ora::fromRowMajor(in));
    std::memcpy(out, m.x, 64);
}
void oracle_mat_mult_point(const float m[16], const float p[3], float out[3]) {
    ora::V3 r = ora::multPoint(ora::fromRowMajor(m), ora::mk(p));
    std::memcpy(out, r.d, 12);
}
void oracle_mat_mult_vec(const float m[16], const float p[3], float out[3]) {
    ora::V3 r = ora::multVec(ora::fromRowMajor(m), ora::mk(p));
    std::memcpy(out, r.d, 12);
}
float oracle_box_pad_factor(void) { return 1 + 2 * ora::gammaN(3); }

// ---- the reference's own unit-test assertions, restated ------------------
// Returns the number of failed checks; `log` receives one line per section.
int oracle_selftest(char* log, int logLen) {
    using namespace ora;
    std::string out;
    int failures = 0;
    auto approx = [](double a, double b, double margin = 0.0) {
        // Catch2 Approx: |a-b| <= margin or <= eps*(scale+|b|), eps = 100*FLT_EPSILON, scale = 0
        double eps = std::numeric_limits<float>::epsilon() * 100.0;
        return std::fabs(a - b) <= margin || std::fabs(a - b) <= eps * std::fabs(b);
    };
    auto section = [&](const char* name, bool ok) {
        out += std::string(ok ? "PASS " : "FAIL ") + name + "\n";
        if (!ok) failures++;
    };
    auto eq3 = [&](const V3& a, double x, double y, double z, double m = 0.0) {
        return approx(a[0], x, m) && approx(a[1], y, m) && approx(a[2], z, m);
    };
    {  // VectorTest.cpp:27-37,51-61,80-138 float forms of the operator table
        V3 x = mk(1, 0, 0), y = mk(0, 1, 0), z = mk(0, 0, 1), v = mk(1, 2, 3);
        section("Vec * scalar (VectorTest.cpp:27-37)", eq3(muls(x, 3), 3, 0, 0) && eq3(muls(v, 3), 3, 6, 9));
        section("Vec * Vec (VectorTest.cpp:39-49)", eq3(mul(x, y), 0, 0, 0) && eq3(mul(v, z), 0, 0, 3));
        section("Vec / scalar (VectorTest.cpp:51-61)", eq3(divs(x, 1), 1, 0, 0) && eq3(divs(muls(v, 6), 3), 2, 4, 6));
        section("Vec - scalar touches all axes (VectorTest.cpp:80-90)", eq3(subs(x, 3), -2, -3, -3) && eq3(subs(v, 3), -2, -1, 0));
        section("Vec - Vec (VectorTest.cpp:92-102)", eq3(sub(x, y), 1, -1, 0) && eq3(sub(v, z), 1, 2, 2));
        section("Vec + scalar (VectorTest.cpp:104-114)", eq3(adds(x, 3), 4, 3, 3) && eq3(adds(v, 3), 4, 5, 6));
        section("Vec + Vec (VectorTest.cpp:116-126)", eq3(add(x, y), 1, 1, 0) && eq3(add(v, z), 1, 2, 4));
        section("negation (VectorTest.cpp:128-138)", eq3(neg(x), -1, 0, 0) && eq3(neg(v), -1, -2, -3));
        section("magnitude (VectorTest.cpp:305-308)", approx(magnitude(x), 1) && approx(magnitude(v), 3.75, 0.01));
        V3 a = normalized(mk(1.0f, 2.0f, 3.0f));
        section("normalized (VectorTest.cpp:310-317)", eq3(normalized(x), 1, 0, 0) && approx(a[0], 0.267, 0.01) &&
                                                           approx(a[1], 0.535, 0.01) && approx(a[2], 0.802, 0.01));
        section("dot (VectorTest.cpp:319-324)", dot(x, y) == 0 && dot(x, v) == 1 && dot(y, v) == 2 && dot(v, v) == 14);
        section("cross (VectorTest.cpp:326-331)", eq3(cross(x, y), 0, 0, 1) && eq3(cross(y, x), 0, 0, -1));
        V3 ns = subs(muls(mk(0.5f, 1.0f, 0.1f), 2.0f), 1.0f);
        section("(v*2)-1 on every axis (VectorTest.cpp:333-340)",
                approx(ns[0], 0.0, 1e-12) && approx(ns[1], 1.0) && approx(ns[2], -0.8));
        section("normalized(0) stays 0 (Vec.cuh:225-229)", eq3(normalized(mk(0, 0, 0)), 0, 0, 0));
    }
    {  // Tests.cpp:45-65
        M44 I = identity();
        M44 R = rotationZ(90.0f);
        auto same = [](const M44& a, const M44& b) { return std::memcmp(&a, &b, sizeof a) == 0; };
        section("I*I == I (Tests.cpp:47-50)", same(multiply(I, I), I));
        // with float entries I*R reproduces R exactly (1*x + 0*y...)
        bool ir = true;
        M44 a = multiply(I, R), b = multiply(R, I);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) ir = ir && a.x[i][j] == R.x[i][j] && b.x[i][j] == R.x[i][j];
        section("I*R == R*I == R (Tests.cpp:52-56)", ir);
        M44 M = fromValues(1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 1, 2, 3, 4, 5, 6);
        M44 N = fromValues(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 1, 2, 3, 4, 5);
        section("M*N != N*M (Tests.cpp:58-64)", !same(multiply(N, M), multiply(M, N)));
    }
    {  // Tests.cpp:67-189
        V3 origo = mk(0, 0, 0), fours = mk(4, 4, 4);
        M44 tr = fromValues(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 4, 16, 44, 4);
        M44 sc = fromValues(1, 0, 0, 0, 0, 2, 0, 0, 0, 0, 4, 0, 0, 0, 0, 1);
        M44 sw = fromValues(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 4);
        section("multPoint translates and divides by w (Tests.cpp:71-88)",
                eq3(multPoint(tr, origo), 1, 4, 11) && eq3(multPoint(tr, fours), 2, 5, 12));
        section("multPoint xyz scale (Tests.cpp:90-107)", eq3(multPoint(sc, origo), 0, 0, 0) && eq3(multPoint(sc, fours), 4, 8, 16));
        section("multPoint w scale (Tests.cpp:109-126)", eq3(multPoint(sw, origo), 0, 0, 0) && eq3(multPoint(sw, fours), 1, 1, 1));
        section("multVec ignores translation (Tests.cpp:132-149)", eq3(multVec(tr, origo), 0, 0, 0) && eq3(multVec(tr, fours), 4, 4, 4));
        section("multVec xyz scale (Tests.cpp:151-168)", eq3(multVec(sc, origo), 0, 0, 0) && eq3(multVec(sc, fours), 4, 8, 16));
        section("multVec ignores w (Tests.cpp:170-187)", eq3(multVec(sw, origo), 0, 0, 0) && eq3(multVec(sw, fours), 4, 4, 4));
    }
    {  // Tests.cpp:191-243
        V3 right = mk(1, 0, 0), up = mk(0, 1, 0), forward = mk(0, 0, 1);
        V3 r = multVec(rotation(right, 90.0f), up), rx = multVec(rotationX(90.0f), up);
        section("rotation about X (Tests.cpp:199-212)", eq3(r, 0, 0, -1, 0.1) && eq3(r, rx[0], rx[1], rx[2], 0.1));
        r = multVec(rotation(up, 90.0f), right);
        V3 ry = multVec(rotationY(90.0f), right);
        section("rotation about Y (Tests.cpp:214-227)", eq3(r, 0, 0, 1, 0.1) && eq3(r, ry[0], ry[1], ry[2], 0.1));
        r = multVec(rotation(forward, 90.0f), right);
        V3 rz = multVec(rotationZ(90.0f), right);
        section("rotation about Z (Tests.cpp:229-242)", eq3(r, 0, -1, 0, 0.1) && eq3(r, rz[0], rz[1], rz[2], 0.1));
    }
    {  // BoundingBoxTest.cpp:23-180
        RtVertex vs[3];
        std::memset(vs, 0, sizeof vs);
        float p[3][3] = {{-2, -1, 0}, {-1, 0, 1}, {1, 2, 3}};
        for (int i = 0; i < 3; ++i) std::memcpy(vs[i].position, p[i], 12);
        RtTriangle t = {{0, 1, 2}};
        BB bb = bbOfTriangle(t, vs);
        section("box of triangle + centroid (BoundingBoxTest.cpp:36-52)",
                eq3(bb.bbmin, -2, -1, 0) && eq3(bb.bbmax, 1, 2, 3) && eq3(bbCentroid(bb), -0.5, 0.5, 1.5));
        section("centroid (BoundingBoxTest.cpp:54-63)", eq3(bbCentroid(BB{mk(0, 1, 2), mk(1, 4, 5)}), 0.5, 2.5, 3.5));
        section("diagonal (BoundingBoxTest.cpp:65-95)", eq3(bbDiagonal(BB{mk(1, 0, 0), mk(5, 2, 1)}), 4, 2, 1) &&
                                                            eq3(bbDiagonal(BB{mk(0, 1, 2), mk(1, 4, 5)}), 1, 3, 3) &&
                                                            eq3(bbDiagonal(BB{mk(1, 0, 0), mk(2, 1, 1)}), 1, 1, 1));
        section("maxExtent tie rules (BoundingBoxTest.cpp:97-113)", bbMaxExtent(BB{mk(1, 0, 0), mk(5, 2, 1)}) == 0 &&
                                                                        bbMaxExtent(BB{mk(0, 1, 2), mk(1, 4, 5)}) == 2 &&
                                                                        bbMaxExtent(BB{mk(1, 0, 0), mk(2, 1, 1)}) == 2);
        BB bb1 = bbDefault();
        BB bb5 = bbUnion(bb1, BB{mk(1, 1, 1), mk(2, 2, 2)});
        BB bb6 = bbUnion(bb5, BB{mk(2, 2, 2), mk(3, 3, 3)});
        BB bb7 = bbUnion(bb6, BB{mk(3, 3, 3), mk(4, 4, 4)});
        section("union chain from the default box (BoundingBoxTest.cpp:115-149)",
                eq3(bb5.bbmin, 1, 1, 1) && eq3(bb5.bbmax, 2, 2, 2) && eq3(bb6.bbmax, 3, 3, 3) && eq3(bb7.bbmin, 1, 1, 1) &&
                    eq3(bb7.bbmax, 4, 4, 4));
        V3 a = mk(0, 1, 2), b = mk(-1, 4, 5);
        BB u = bbUnion(BB{a, b}, BB{adds(a, 2), adds(b, 2)});
        section("union keeps inverted extents (BoundingBoxTest.cpp:151-166)",
                !approx(u.bbmin[0], -1.0) && approx(u.bbmin[0], -0.0, 1e-12) && approx(u.bbmin[1], 1) && approx(u.bbmin[2], 2) &&
                    eq3(u.bbmax, 1, 6, 7));
        BB up = bbUnion(BB{mk(0, 1, 2), mk(1, 4, 5)}, muls(mk(1, 4, 5), 3));
        section("union with a point (BoundingBoxTest.cpp:168-180)", eq3(up.bbmin, 0, 1, 2) && eq3(up.bbmax, 3, 12, 15));
        section("default box max is FLT_MIN (BoundingBox.cuh:25-27)", bb1.bbmax[0] == FLT_MIN && bb1.bbmin[0] == FLT_MAX);
    }
    {  // BoundingBoxTest.cpp:182-240
        auto hit = [](V3 mn, V3 mx, V3 d) {
            float o[3] = {0, 0, 0};
            return oracle_box_intersect(mn.d, mx.d, o, d.d, FLT_MAX) != 0;
        };
        V3 f = mk(0, 0, 1), bk = mk(0, 0, -1);
        section("slab test +Z (BoundingBoxTest.cpp:183-210)", hit(mk(-1, -1, -1), mk(1, 1, 1), f) && hit(mk(-1, -1, 1), mk(1, 1, 3), f) &&
                                                                  !hit(mk(0.5f, 0.5f, 1), mk(1, 1, 3), f));
        section("slab test -Z (BoundingBoxTest.cpp:212-239)", hit(mk(-1, -1, -1), mk(1, 1, 1), bk) && hit(mk(-1, -1, -1), mk(1, 1, -3), bk) &&
                                                                  !hit(mk(0.5f, 0.5f, -1), mk(1, 1, -3), bk));
        section("far-plane pad factor is 1+3*2^-23", oracle_box_pad_factor() == 1.00000035762786865234375f);
    }
    {  // BvhTest.cpp:7-19
        BB bb{mk(-3, -2, -1), mk(0, 1, 2)};
        PreNode n{0, bb, bbCentroid(bb)};
        section("PreNode copies the box and takes its centroid (BvhTest.cpp:7-19)",
                eq3(n.bb.bbmin, -3, -2, -1) && eq3(n.bb.bbmax, 0, 1, 2) && eq3(n.centroid, -1.5, -0.5, 0.5));
    }
    {  // BoundingBoxTest.cpp:6-21: Triangle constructors (plain-data form)
        RtTriangle t = {{1, 2, 3}};
        int values[5] = {0, 1, 2, 3, 4};
        RtTriangle t2 = {{values[1], values[2], values[3]}};
        section("Triangle index triple (BoundingBoxTest.cpp:6-21)", t.i[0] == 1 && t.i[2] == 3 && t2.i[0] == 1 && t2.i[1] == 2 && t2.i[2] == 3);
    }
    if (log && logLen > 0) {
        std::snprintf(log, logLen, "%s", out.c_str());
    }
    return failures;
}

}  // extern "C"
