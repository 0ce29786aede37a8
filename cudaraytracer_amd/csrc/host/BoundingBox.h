// Axis-aligned box + the flattened 32-byte BVH node of the host API.
// Behaviour follows cudaTracer/BoundingBox.cuh:21-139 and
// cudaTracer/LinearNode.cuh:4-13.  Two properties matter to BVH topology and
// are kept on purpose:
//   * the empty box starts its max corner at FLT_MIN (the smallest positive
//     normal float), not at -FLT_MAX (BoundingBox.cuh:25-27), so boxes and
//     centroid bounds of all-negative geometry are clamped to ~0 on top;
//   * maxExtent() breaks ties towards Z, then Y (BoundingBox.cuh:59-68).
#ifndef RT_HOST_BOUNDINGBOX_H
#define RT_HOST_BOUNDINGBOX_H

#include <algorithm>
#include <limits>
#include <vector>

#include "Types.h"

namespace rt {

static constexpr float MachineEpsilon = std::numeric_limits<float>::epsilon() * 0.5;
inline constexpr float gamma(int n) { return (n * MachineEpsilon) / (1 - n * MachineEpsilon); }

struct BoundingBox {
    Vec3f bbmin;
    Vec3f bbmax;

    BoundingBox() : bbmin(FLT_MAX, FLT_MAX, FLT_MAX), bbmax(FLT_MIN, FLT_MIN, FLT_MIN) {}
    BoundingBox(const Vec3f& bbmin, const Vec3f& bbmax) : bbmin(bbmin), bbmax(bbmax) {}
    BoundingBox(const Triangle& t, const std::vector<Vertex>& v) : BoundingBox() {
        for (int corner = 0; corner < 3; ++corner) {
            const Vec3f& pos = v[t[corner]].position;
            for (int axis = 0; axis < 3; ++axis) {
                if (pos[axis] > bbmax[axis]) bbmax[axis] = pos[axis];
                if (pos[axis] < bbmin[axis]) bbmin[axis] = pos[axis];
            }
        }
    }

    const Vec3f& operator[](int i) const { return i == 0 ? bbmin : bbmax; }
    Vec3f& operator[](int i) { return i == 0 ? bbmin : bbmax; }

    Vec3f centroid() const { return (bbmin + bbmax) / 2.0f; }
    Vec3f diagonal() const { return bbmax - bbmin; }
    int maxExtent() const {
        const Vec3f d = diagonal();
        if (d[0] > d[1] && d[0] > d[2]) return 0;
        return d[1] > d[2] ? 1 : 2;
    }

    static BoundingBox unionn(BoundingBox a, BoundingBox b) {
        BoundingBox r(a);
        for (int k = 0; k < 3; ++k) {
            r.bbmin[k] = (std::min)(a.bbmin[k], b.bbmin[k]);
            r.bbmax[k] = (std::max)(a.bbmax[k], b.bbmax[k]);
        }
        return r;
    }
    static BoundingBox unionn(BoundingBox b, Vec3f p) {
        BoundingBox r(b);
        for (int k = 0; k < 3; ++k) {
            r.bbmin[k] = (std::min)(b.bbmin[k], p[k]);
            r.bbmax[k] = (std::max)(b.bbmax[k], p[k]);
        }
        return r;
    }

    // pbrt-style slab test with precomputed reciprocal direction
    // (BoundingBox.cuh:115-139).  Host-side twin of the kernel's node test, kept
    // for API parity and for host-side culling; x/y far planes are widened by
    // 1 + 2*gamma(3), the z far plane is not.
    bool intersect(const Ray& ray, const Vec3f& invDir, const int dirIsNeg[3]) const {
        const BoundingBox& bounds = *this;
        float tMin = (bounds[dirIsNeg[0]][0] - ray.origin[0]) * invDir[0];
        float tMax = (bounds[1 - dirIsNeg[0]][0] - ray.origin[0]) * invDir[0];
        float tyMin = (bounds[dirIsNeg[1]][1] - ray.origin[1]) * invDir[1];
        float tyMax = (bounds[1 - dirIsNeg[1]][1] - ray.origin[1]) * invDir[1];
        tMax *= 1 + 2 * gamma(3);
        tyMax *= 1 + 2 * gamma(3);
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
    // Slab test that reports the parametric interval (BoundingBox.cuh:88-113).
    bool intersect(const Ray& ray, float* hitt0, float* hitt1) const {
        float t0 = 0, t1 = ray.tMax;
        for (int i = 0; i < 3; ++i) {
            float invRayDir = 1 / ray.dir[i];
            float tNear = (bbmin[i] - ray.origin[i]) * invRayDir;
            float tFar = (bbmax[i] - ray.origin[i]) * invRayDir;
            if (tNear > tFar) std::swap(tNear, tFar);
            t0 = tNear > t0 ? tNear : t0;
            t1 = tFar < t1 ? tFar : t1;
            if (t0 > t1) return false;
        }
        if (hitt0) *hitt0 = t0;
        if (hitt1) *hitt1 = t1;
        return true;
    }
};

struct LinearNode {
    BoundingBox bb;
    union {
        int primitivesOffset;   // leaf: first primitive slot
        int secondChildOffset;  // interior: index of the second child (the first is self + 1)
    };
    uint16_t primtiveCnt = 0;  // 0 marks an interior node
    uint8_t axis = 0;          // interior: split axis
    uint8_t pad[1] = {0};
    LinearNode() : primitivesOffset(0) {}
};

static_assert(sizeof(BoundingBox) == 24, "BoundingBox layout");
static_assert(sizeof(LinearNode) == sizeof(RtLinearNode) && sizeof(LinearNode) == 32, "LinearNode layout");

}  // namespace rt
#endif
