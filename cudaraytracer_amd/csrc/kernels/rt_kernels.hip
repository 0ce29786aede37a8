// rt_kernels.hip - gfx950 render kernels.
//
// Structure: a frame is a short sequence of launches over "path nodes" (one
// node = one invocation of the reference's recursive castRay):
//
//   traceKernel<closest>  persistent 64-lane wavefronts; every LANE owns one ray
//                         and gets the next one from an atomic work queue after
//                         its ray ends (wave-level chunks of the queue keep
//                         neighbouring pixels in one wave; the queue is eight
//                         counters, so that a launch does not queue up at one
//                         address); two-level BVH walk, box tests per lane, leaf
//                         triangles tested by all 64 lanes together, per-lane
//                         stack in an LDS ring.  Large frames: two launches,
//                         the box test in min / max form for rays with a
//                         finite reciprocal direction, literally for the rest
//   shadeKernel           hit -> surface (st, normal, textures) -> either a
//                         shading context + shadow work, or child rays
//                         (ballot-compacted into the next level's queue)
//   traceKernel<any hit>  shadow rays: (diffuse hit, light) pairs, ray rebuilt
//                         from the shading context; once the queue is dry, idle
//                         lanes take over subtrees of the rays still walked
//   accumKernel           Blinn-Phong sum over the lights in light order
//   resolveKernel         mirror / glass nodes combined bottom-up
//   finalizeKernel        sample average, gamma, float4 store
//
// Arithmetic contract: every fp32 operation below is written in the operation
// order of the reference's device code (cited per function) and this file is
// compiled with -ffp-contract=off, IEEE division and square root, denormals on,
// so hit decisions are bit-identical to a host evaluation of the same formulas.
// Only powf (gamma, Blinn-Phong) may differ in the last ulp.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>

#include "../rt_device_format.h"
#include "rt_trace_probes.h"

// No expression in this file may be contracted to a fused multiply-add, whatever the command line says.
#pragma clang fp contract(off)

namespace {

constexpr int kWave = 64;
#ifndef RT_CHUNK
#define RT_CHUNK 64
#endif
constexpr unsigned kChunk = RT_CHUNK;  // rays a wavefront takes from the global queue per atomic
#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 32  // a wavefront refills once this many lanes have finished their rays
#endif
#ifndef RT_TRI_LANES
#define RT_TRI_LANES 12  // a triangle phase starts once this many lanes wait in a leaf
#endif
#ifndef RT_MIN_WAVES_ANY
#define RT_MIN_WAVES_ANY 1  // the same for the shadow-ray variants
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 1  // __launch_bounds__ waves per SIMD for the trace kernel
#endif

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};

__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 mk(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 mk(float4 v) { return V3{v.x, v.y, v.z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 muls(V3 a, float r) { return mk(a.x * r, a.y * r, a.z * r); }
__device__ __forceinline__ V3 divs(V3 a, float r) { return mk(a.x / r, a.y / r, a.z / r); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// cudaTracer/Vec.cuh:232-238: the sum starts at 0 (keeps the sign-of-zero behaviour)
__device__ __forceinline__ float dot(V3 a, V3 b) { return ((0.0f + a.x * b.x) + a.y * b.y) + a.z * b.z; }
// cudaTracer/Vec.cuh:240-247
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// squares are never -0, so 0 + x*x == x*x and the leading zero can be dropped here
__device__ __forceinline__ float mag2(V3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
// cudaTracer/Vec.cuh:212-230
__device__ __forceinline__ float magnitude(V3 a) { return sqrtf(mag2(a)); }
__device__ __forceinline__ V3 normalized(V3 a) {
    const float m = mag2(a);
    if (m > 0) {
        const float invMag = 1.0f / sqrtf(m);
        return muls(a, invMag);
    }
    return a;
}
// cudaTracer/Vec.cuh:249-251: I - (N*2)*dot(I,N)
__device__ __forceinline__ V3 reflect(V3 I, V3 N) { return sub(I, muls(muls(N, 2.0f), dot(I, N))); }

__device__ __forceinline__ unsigned laneRank(unsigned long long mask) {
    // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// ---------------------------------------------------------------------------
// cudaTracer/BoundingBox.cuh:115-139, literally, split in two: everything that does not depend on the ray's current
// reach (slab overlap and "far plane in front of the origin") comes back as `ok`, the entry distance as tMin; a box
// is visited iff  ok && tMin < ray.tMax  - the reference's final expression.  A box is {min.x, min.y, min.z, max.x},
// {max.y, max.z}; every value is formed by the reference's sequence of roundings.  The far planes of x and y are
// padded by 1 + 2*gamma(3) == 1 + 3*2^-23 in fp32, z is not.
__device__ __forceinline__ void boxLiteral(const float4 lo, const float hiY, const float hiZ, const V3 o, const V3 inv, const bool nx, const bool ny,
                                           const bool nz, bool& okOut, float& tMinOut) {
    constexpr float kPad = 0x1.000006p+0f;
    float tMin = ((nx ? lo.w : lo.x) - o.x) * inv.x;
    float tMax = (((nx ? lo.x : lo.w) - o.x) * inv.x) * kPad;
    const float tyMin = ((ny ? hiY : lo.y) - o.y) * inv.y;
    const float tyMax = (((ny ? lo.y : hiY) - o.y) * inv.y) * kPad;
    bool ok = !(tMin > tyMax || tyMin > tMax);
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    const float tzMin = ((nz ? hiZ : lo.z) - o.z) * inv.z;
    const float tzMax = ((nz ? lo.z : hiZ) - o.z) * inv.z;
    ok = ok && !(tMin > tzMax || tzMin > tMax);
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    tMinOut = tMin;
    okOut = ok && (tMax > 0);
}
__device__ __forceinline__ void boxPairSlabs(const float4 n0, const float4 n1, const float4 n2, const float4 n3, const V3 o, const V3 inv,
                                             const bool nx, const bool ny, const bool nz, bool& ok0, bool& ok1, float& tMin0, float& tMin1) {
    boxLiteral(n0, n1.x, n1.y, o, inv, nx, ny, nz, ok0, tMin0);
    boxLiteral(n2, n3.x, n3.y, o, inv, nx, ny, nz, ok1, tMin1);
}

// The same test in min / max form: near / far plane distance = the smaller / larger of the two plane distances of an
// axis, and "largest entry <= smallest exit" for the reference's chain of comparisons.  It agrees with the literal
// form, bit for bit, whenever no plane distance is a NaN (finite ray origin, finite reciprocal direction) and the
// boxes have min <= max: then (min - o) <= (max - o), a product with inv keeps or reverses that order, and the
// smaller product is the reference's near-plane distance, formed by the same operations; the reference's comparisons
// say "every entry <= every exit of ANOTHER axis", and entry <= exit of the SAME axis can only fail when a padded exit
// distance is negative, where the box fails "exit > 0" in both forms.  A ray lying exactly in a box face (0 * inf) can
// come out differently, which is why rays with an infinite reciprocal direction are walked by a launch of
// their own with the literal form (see traceKernel).  Both forms inside ONE kernel cost more registers than the
// shorter form saves instructions (measured: 1 615 against 1 640 Mrays/s).
__device__ __forceinline__ void boxPairMinMax(const float4 n0, const float4 n1, const float4 n2, const float4 n3, const V3 o, const V3 inv,
                                              float& tMin0, float& tMax0, float& tMin1, float& tMax1) {
    constexpr float kPad = 0x1.000006p+0f;
    {
        const float ax = (n0.x - o.x) * inv.x, bx = (n0.w - o.x) * inv.x;
        const float ay = (n0.y - o.y) * inv.y, by = (n1.x - o.y) * inv.y;
        const float az = (n0.z - o.z) * inv.z, bz = (n1.y - o.z) * inv.z;
        tMin0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        tMax0 = fminf(fminf(fmaxf(ax, bx) * kPad, fmaxf(ay, by) * kPad), fmaxf(az, bz));
    }
    {
        const float ax = (n2.x - o.x) * inv.x, bx = (n2.w - o.x) * inv.x;
        const float ay = (n2.y - o.y) * inv.y, by = (n3.x - o.y) * inv.y;
        const float az = (n2.z - o.z) * inv.z, bz = (n3.y - o.z) * inv.z;
        tMin1 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        tMax1 = fminf(fminf(fmaxf(ax, bx) * kPad, fmaxf(ay, by) * kPad), fmaxf(az, bz));
    }
}
// cudaTracer/Sphere.cuh:9-54
__device__ __forceinline__ bool sphereHit(const DevShape& s, const V3 o, const V3 d, float& tMax) {
    const V3 L = sub(o, mk(s.center));
    const float a = dot(d, d);
    const float b = 2.0f * dot(d, L);
    const float c = dot(L, L) - s.radius2;
    float t0, t1;
    const float discr = b * b - 4.0f * a * c;
    if (discr < 0.0f) {
        return false;
    } else if (discr == 0.0f) {
        t0 = t1 = -0.5f * b / a;
    } else {
        const float q = (b > 0.0f) ? -0.5f * (b + sqrtf(discr)) : -0.5f * (b - sqrtf(discr));
        t0 = q / a;
        t1 = c / q;
    }
    if (t0 > t1) {
        const float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    if (t0 < 0.0f) t0 = t1;
    if (t0 < 0.0f) return false;
    if (t0 > tMax) return false;
    tMax = t0;
    return true;
}

// cudaTracer/tracer.cu:202-215
__device__ __forceinline__ V3 biasedOrigin(V3 dir, V3 P, V3 n, float bias) {
    return dot(dir, n) < 0.0f ? add(P, muls(n, bias)) : sub(P, muls(n, bias));
}
__device__ __forceinline__ V3 invBiasedOrigin(V3 dir, V3 P, V3 n, float bias) {
    return dot(dir, n) < 0.0f ? sub(P, muls(n, bias)) : add(P, muls(n, bias));
}

// ---------------------------------------------------------------------------
// Level-0 node -> pixel.  With S = opts.samples, node s of this call is lane (s & 63) of sample (s >> 6) % S
// of local tile (s >> 6) / S; local tile k is frame tile tileFirst + k * tileStride (row-major), or
// tileOrder[k] when the host supplies an order (the same set of tiles, expensive image regions first).
// `single`: true in code compiled for frames of one sample per pixel (no division by the sample count).
// The tile's row and column come from a multiplication by P.tilesXMagic = floor((2^32 - 1) / tilesX): the product's high
// word is the quotient or one less (the estimate is short by less than tile / 2^32 < 1), which one comparison settles.
// (A division by a kernel argument makes the compiler keep a reciprocal in a vector register for the whole kernel.)
__device__ __forceinline__ bool slotPixel(const FrameParams& P, int slot, int& x, int& y, const bool single = false) {
    const int group = slot >> 6;
    const int local = (single || P.opts.samples == 1) ? group : group / P.opts.samples;
    const unsigned tile = (unsigned)(P.tileOrder ? P.tileOrder[local] : P.tileFirst + local * P.tileStride);
    const int l = slot & 63;
    unsigned row = __umulhi(tile, P.tilesXMagic);
    unsigned col = tile - row * (unsigned)P.tilesX;
    if (col >= (unsigned)P.tilesX) {
        col -= (unsigned)P.tilesX;
        row += 1u;
    }
    x = (int)col * RT_TILE + (l & (RT_TILE - 1));
    y = (int)row * RT_TILE + (l >> 3);
    return x < P.opts.width && y < P.opts.height;
}
__device__ __forceinline__ int slotSample(const FrameParams& P, int slot) {
    return P.opts.samples == 1 ? 0 : (slot >> 6) % P.opts.samples;
}

// Samples a pixel holds from earlier calls when accumulating over frames: its own count in the running-sum surface, so
// that a frame rendered shard by shard (or tile list by tile list) advances pixel by pixel.  0 without accumulation.
__device__ __forceinline__ uint32_t pixelSampleBase(const FrameParams& P, int x, int y) {
    return P.accum ? (uint32_t) reinterpret_cast<const float*>(P.accum + (size_t)y * P.accumPitch + (size_t)x * 16)[3] : 0u;
}

// Primary ray direction, cudaTracer/Main.cu:57-75.  With one sample both jitters are exactly 0.0f.  `jitter` is
// true in the code paths compiled for frames that have a jitter (several samples, or accumulation over frames); the trace
// kernels of all other frames are compiled without that branch and the registers it holds on to (about 2 % of the
// rays per second with frames in flight).
// P.invWidth / P.invHeight are the reference's 1.0f / width and 1.0f / height (Main.cu:57-58), divided on the host: the same
// correctly rounded quotient, without two vector registers that hold it for the whole kernel.
__device__ __forceinline__ V3 primaryDir(const FrameParams& P, int x, int y, int sample, const bool jitter = true) {
    const float widthScale = P.invWidth;
    const float heightScale = P.invHeight;
    const float ndcX = (2.0f * ((float)x + 0.5f) * widthScale - 1.0f) * P.opts.imageAspectRatio * P.opts.scale;
    const float ndcY = (1.0f - 2.0f * ((float)y + 0.5f) * heightScale) * P.opts.scale;
    float xJitter = 0.0f, yJitter = 0.0f;
    if (jitter && (P.opts.samples > 1 || P.accum)) {
        // accumulating over frames, this call's samples continue the numbering of the pixel's earlier ones
        const uint32_t pixel = (uint32_t)y * (uint32_t)P.opts.width + (uint32_t)x;
        const uint32_t k = pixelSampleBase(P, x, y) + (uint32_t)sample;
        xJitter = rt_sample_jitter(P.sampleSeed, pixel, k, 0u);
        yJitter = rt_sample_jitter(P.sampleSeed, pixel, k, 1u);
    }
    // (without a jitter the two sums would add +0, which changes no value: ndcX and ndcY are never -0)
    V3 dir = jitter ? normalized(mk(ndcX + xJitter * widthScale, ndcY + yJitter * heightScale, 1.0f)) : normalized(mk(ndcX, ndcY, 1.0f));
    const float* C = P.camera;
    dir = mk((dir.x * C[0] + dir.y * C[4]) + dir.z * C[8], (dir.x * C[1] + dir.y * C[5]) + dir.z * C[9],
             (dir.x * C[2] + dir.y * C[6]) + dir.z * C[10]);
    return normalized(dir);
}

// Ray of a path node: level 0 is rebuilt from the pixel, deeper nodes are stored.
__device__ __forceinline__ void nodeRay(const FrameParams& P, int node, V3& o, V3& d, const bool jitter = true) {
    if (node < P.slots) {
        int x, y;
        slotPixel(P, node, x, y, !jitter);
        o = mk(P.orig);
        d = primaryDir(P, x, y, jitter ? slotSample(P, node) : 0, jitter);
    } else {
        const float4* r = P.secRay + 2 * (size_t)(node - P.slots);
        o = mk(r[0]);
        d = mk(r[1]);
    }
}

// Shadow ray of (diffuse hit k, light i), cudaTracer/tracer.cu:175-183: origin
// biased along the normal by the sign of view . n, direction and reach taken
// from the UNBIASED hit point.
// `unlit`: the light cannot contribute whatever the ray finds - lambert (tracer.cu:181, formed as accumKernel forms it)
// is 0, and lightComponent (tracer.cu:194) is certainly finite, so that its products with lambert and with the specular
// factor, which is 0 with lambert, are zeros.
__device__ __forceinline__ void shadowRay(const FrameParams& P, unsigned k, int light, V3& o, V3& d, float& tMax, bool& unlit) {
    const float4* c = P.ctx + 4 * (size_t)k;
    const float4 c0 = c[0], c1 = c[1], c2 = c[2];
    const V3 hitPoint = mk(c0.x, c0.y, c0.z);
    const V3 n = mk(c0.w, c1.x, c1.y);
    const V3 view = mk(c1.z, c1.w, c2.x);
    const RtLight* L = reinterpret_cast<const RtLight*>(P.blob + P.offLights) + light;
    V3 lightDir = sub(mk(L->position), hitPoint);
    tMax = magnitude(lightDir);
    d = normalized(lightDir);
    o = biasedOrigin(view, hitPoint, n, P.opts.shadowBias);
    unlit = false;
    if (P.skipUnlit != 0 && !(dot(d, n) > 0.0f)) {
        // lightComponent = intensity * power / distance below 1e38 in size, and its product with the specular mask too
        const V3 scaled = muls(mk(L->intensity), L->power);
        const float big = 1e18f, specMask = c[3].x;
        unlit = tMax > 1e-20f && __builtin_fabsf(scaled.x) < big && __builtin_fabsf(scaled.y) < big && __builtin_fabsf(scaled.z) < big &&
                __builtin_fabsf(specMask) <= 1.0f;
    }
}

// What a lane does next is held in ONE word, `cur`:
//   bit 31 clear          an interior node reference: expand it (fetch its child pair)
//   10 | next (30 bits)   inside a mesh-BVH leaf: triangle records next .. end-1 are to be tested
//   11 | next (30 bits)   inside a scene-BVH leaf: shapes next .. end-1 are to be tested
//   CUR_POP               take the next entry from the traversal stack
//   CUR_DONE              the ray is finished (also the value of a lane without a ray)
constexpr unsigned CUR_POP = 0xFFFFFFFFu, CUR_DONE = 0xFFFFFFFEu;
// Population count of a lane mask from its two halves: a 64-bit count would be compared as a 64-bit number, for which
// the scalar unit has no ordering compare, and the comparison of two wave-uniform numbers would go to the vector unit.
__device__ __forceinline__ int pop64(unsigned long long m) { return __builtin_popcount((unsigned)m) + __builtin_popcount((unsigned)(m >> 32)); }

constexpr unsigned TAG_TRI = 0x80000000u, TAG_SHAPE = 0xC0000000u, IDX_MASK = 0x3FFFFFFFu;
constexpr unsigned STACK_MESH_EXIT = 0xFFFFFFFDu;  // stack marker: the mesh walk is over, restore the world-space ray
constexpr int HIT_PENDING = -3;                    // hShape while the hit mesh is still being walked
constexpr int HIT_DEFERRED = -4;                   // P.hit[].w of a ray left to the launch with the literal box test
constexpr unsigned char VIS_DEFERRED = 2;          // P.vis[] of such a shadow ray

#ifndef RT_POP_WHILE
#define RT_POP_WHILE if  // one pop attempt per iteration of the main loop (`while` = pop until something is entered)
#endif
#ifndef RT_TRI_PART
#define RT_TRI_PART 16  // rays whose leaves are tested in one cooperative triangle phase (8 lanes each, 8 rays per round)
#endif
constexpr int kTriPart = RT_TRI_PART;
#ifndef RT_TRI_GROUP
#define RT_TRI_GROUP 8  // lanes that test one ray's leaf side by side (one triangle each)
#endif
constexpr int kGrp = RT_TRI_GROUP, kPerRound = 64 / kGrp, kRounds = kTriPart / kPerRound;
static_assert((kGrp == 8 || kGrp == 4) && kTriPart % kPerRound == 0 && kRounds >= 1, "triangle phase geometry");
static_assert(RT_MESH_ENTRY_SLOTS >= 6, "the mesh-entry save area holds six entries");
constexpr size_t kTriLdsBytes = (size_t)kTriPart * (2 * sizeof(float4) + 8 * sizeof(float) + 8 * sizeof(float2));

__host__ __device__ constexpr size_t traceLdsBytes(int stackLds) { return (size_t)(stackLds + 1) * kWave * sizeof(uint2) + kTriLdsBytes; }

// cudaTracer/Mesh.cuh:182-212 on a precomputed {v0, e1, e2} record, WITHOUT the final comparison
// against the ray's reach: returns false if one of the determinant / barycentric tests rejects the
// triangle, else t, u, v exactly as the reference forms them.
__device__ __forceinline__ bool triangleSolve(const float4 r0, const float4 r1, const float e2z, const V3 o, const V3 d,
                                              float& tOut, float& uOut, float& vOut) {
    const V3 v0 = mk(r0.x, r0.y, r0.z);
    const V3 e1 = mk(r0.w, r1.x, r1.y);
    const V3 e2 = mk(r1.z, r1.w, e2z);
    const V3 pvec = cross(d, e2);
    const float det = dot(e1, pvec);
    const V3 tvec = sub(o, v0);
    const float u = dot(tvec, pvec);
    const V3 qvec = cross(tvec, e1);
    const float v = dot(d, qvec);
    const bool reject = (det == 0.0f || det < 0.0f) || (u < 0.0f || u > det) || (v < 0.0f || u + v > det);
    if (reject) return false;
    const float invDet = 1.0f / det;
    tOut = dot(e2, qvec) * invDet;
    uOut = u * invDet;
    vOut = v * invDet;
    return true;
}

// ---------------------------------------------------------------------------
// Persistent traversal kernel.  ANY = false: closest hit for the path nodes of
// `level` (cudaTracer/Scene.cuh:21-87 + cudaTracer/Mesh.cuh:115-179).  ANY =
// true: shadow rays of the level's diffuse hits; the walk ends at the first
// accepted primitive, which yields the same boolean as the reference's full
// walk because acceptance of that primitive does not depend on later tests.
//
// Every LANE owns one ray.  Per iteration the wavefront votes for ONE kind of
// work, so that each (long) block of code runs with as many lanes as possible:
//
//   node step       every lane that sits on an interior node expands it: ONE
//                   64-byte fetch brings both child boxes; the near child (by the
//                   sign of the ray along the split axis) is tested against the
//                   ray's current reach and entered, the far child is pushed
//                   together with its entry distance
//   triangle phase  the lanes that sit in a mesh leaf (up to RT_TRI_PART of
//                   them) publish their rays in LDS and ALL 64 lanes test the
//                   leaves' triangles, 8 lanes per ray, one triangle per lane
//                   (coalesced 48-byte record reads); the results go back through
//                   LDS and each owner applies them IN LEAF ORDER with the
//                   reference's rule (a hit at t <= reach replaces the current
//                   one, so of equal distances the later triangle wins)
//
// followed by the rare per-lane work: scene-leaf shapes (sphere test, or the
// change to a mesh's object space) and stack pops.  A popped entry is entered
// iff its stored entry distance is still below the ray's reach - the same
// decision the reference makes when it tests that box at pop time, since the
// reach is the only input of the box test that changes along the walk.
//
// The stack (8 bytes per entry: node reference + entry distance) lives in LDS,
// one column per lane, with a global-memory continuation for deep entries; it
// is shared by the scene BVH and the mesh BVH; a marker entry separates the two.
// Visiting order, near/far choice and the <=-tie rule are the reference's, so
// results do not depend on which lane or wavefront a ray lands in.
//
// Counting (COUNT): a box test is tallied when the reference would perform it -
// the near child at once, a far child when popped - so the totals equal the
// reference walk's even when a shadow ray stops early.  For that, this variant
// also pushes far children that are already known to fail (marked by an
// infinite entry distance).
//
// Box test (LIT).  The reference's slab test picks the near and the far plane of every axis by the sign of the ray's
// direction and chains its comparisons so that a NaN (0 * inf: a ray that lies exactly in a box face) falls a
// certain way.  Without a NaN the smaller / larger of the two plane distances IS the near / far distance, formed by
// the same operations (boxPairMinMax), which saves a quarter of a node step's vector instructions.  A NaN needs an
// infinite reciprocal direction, so the LIT = false launch walks every ray whose reciprocal
// direction is finite - in world space and in every mesh's object space it enters - and marks the others
// (HIT_DEFERRED / VIS_DEFERRED, counted in CTL_DEFER_*); a second launch with the literal test (LIT = true, `rescue`)
// walks the marked rays from the start and ends at once when there is none.  Counter launches, small frames and
// reflection levels use the literal test for every ray, in one launch (launchTrace).
// JIT: the frame's primary rays are jittered (closest-hit launches of frames with several samples or accumulation).
template <bool ANY, bool COUNT, bool LIT, bool JIT>
__global__ void __launch_bounds__(kWave, ANY ? RT_MIN_WAVES_ANY : RT_MIN_WAVES) traceKernel(const FrameParams P, const int level, const int rescue) {
    extern __shared__ uint2 s_dyn[];
    const int lane = threadIdx.x;
    // The top of each lane's traversal stack is a ring of ringN (a power of two) entries in LDS: entry e lives in
    // stk[(e & ringMask) * 64].  Entries [lo, sp) are in the ring, older ones [0, lo) in this wavefront's slice of a
    // global array, so global memory is touched only when a walk gets more than ringN entries ahead of itself.
    // Slot ringN of the LDS column is a write-only dump slot.
    const int ringN = P.stackLds, ringMask = ringN - 1;
    uint2* const stk = s_dyn + lane;
    float4* const rayBuf = reinterpret_cast<float4*>(s_dyn + (size_t)(ringN + 1) * kWave);  // [kTriPart][2]
    float* const tBuf = reinterpret_cast<float*>(rayBuf + 2 * kTriPart);                    // [kTriPart][8]
    float2* const uvBuf = reinterpret_cast<float2*>(tBuf + 8 * kTriPart);                   // [kTriPart][8]
    uint2* const gstk = P.stackSpill + ((size_t)blockIdx.x * (size_t)(P.stackEntries + RT_MESH_ENTRY_SLOTS)) * kWave + lane;
    // the world-space ray and the rest of the scene leaf wait here while a mesh is walked
    uint2* const gsave = gstk + (size_t)P.stackEntries * kWave;

    constexpr bool kDefer = !LIT;  // rays with an infinite reciprocal direction are left to the rescue launch
    // The launches with the min / max box test walk quads (four grandchildren per fetch, rt_device_format.h); counter
    // launches and the launches with the literal box test walk the sibling pairs, test for test as the reference does.
    constexpr bool kQuad = !LIT && !COUNT;
    // the rays left to the rescue launch: listed behind the control words while the list lasts, marked in any case
    const unsigned deferred = rescue ? P.ctl[(ANY ? CTL_DEFER_SHADOW : CTL_DEFER_CLOSEST)] : 0u;
    if (rescue && deferred == 0u) return;
    uint32_t* const deferList = P.ctl + RT_DEFER_WORD(ANY);
    const bool fromList = rescue && deferred <= (unsigned)RT_DEFER_LIST;
    const unsigned total = fromList ? deferred : (ANY ? P.ctl[CTL_DIFFUSE_CNT] * (unsigned)P.lightCnt : P.ctl[CTL_LEVEL_COUNT + level]);
    const unsigned first = (ANY || fromList) ? 0u : P.ctl[CTL_LEVEL_START + level];
    unsigned* const queues = P.ctl + RT_QUEUE_WORD((rescue ? 2 : 0) + (ANY ? 1 : 0), 0);
    const unsigned chunks = (total + kChunk - 1) / kChunk;
    // True if the min / max box test may differ from the literal one for a ray with this reciprocal direction and
    // origin: an infinite component of the former (0 * inf = NaN for a ray in a box face), or a NaN in either (min / max
    // drop a NaN operand where the reference's chain of comparisons keeps it and misses).
    auto special = [](const V3 r, const V3 from) {
        const float big = __builtin_huge_valf();
        return !(__builtin_fabsf(r.x) < big) || !(__builtin_fabsf(r.y) < big) || !(__builtin_fabsf(r.z) < big) || from.x != from.x ||
               from.y != from.y || from.z != from.z;
    };
    // With little work, only as many wavefronts take part as get P.waveChunks chunks each.  One chunk each spreads
    // the rays over the most wavefronts (shortest launch when nothing else runs); several make a wavefront refill
    // its lanes instead of running a single chunk dry, and leave slots free for the kernels of other frames in flight.
    const unsigned waveChunks = (unsigned)(total >= 6000u * kChunk ? P.waveChunksLarge : P.waveChunks);
    if ((unsigned long long)blockIdx.x * (kChunk * waveChunks) >= total && blockIdx.x != 0) return;

    const float4* const blob4 = reinterpret_cast<const float4*>(P.blob);
    const float4* const records = reinterpret_cast<const float4*>(P.blob + P.offRecords);
    const DevShape* const shapes = reinterpret_cast<const DevShape*>(P.blob + P.offShapes);
    const DevMesh* const meshes = reinterpret_cast<const DevMesh*>(P.blob + P.offMeshes);
    const bool useSceneBvh = P.opts.useSceneBvh != 0;
    const bool useMeshBvhs = P.opts.useMeshBvhs != 0;
    const float kInf = __builtin_huge_valf();

    unsigned cNode = 0, cTri = 0, cSphere = 0, cMesh = 0, cRays = 0, cPrimary = 0;
    unsigned rayTests = 0;  // cNode + cTri when the lane's current ray started (COUNT: per-tile cost)
    // wave-level statistics (meaningful in lane 0): iterations, node steps and their lanes, triangle phases and their
    // rays, lanes holding a ray
    unsigned long long wSteps = 0, wNodeLanes = 0, wTriLanes = 0, wBusyLanes = 0, wNodeSteps = 0, wTriSteps = 0;
    unsigned long long wTailSteps = 0, wTailLanes = 0;  // the same after the work queue ran dry

    // wave-uniform queue state
    unsigned chunkNext = 0, chunkEnd = 0;
    unsigned dryQueues = 0;  // counters this wavefront has found dry, starting with its own (blockIdx % RT_QUEUES)
    bool exhausted = false;
    unsigned drainSteps = 0;

    // per-lane ray state
    int item = -1;  // work item (node index or shadow index); -1 = no ray
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1), inv = mk(0, 0, 0);
    float tMax = 0.0f;
    unsigned cur = CUR_DONE, end = 0;
    int sp = 0, lo = 0;  // stack entries [lo, sp) are in the LDS ring, [0, lo) in global memory
    int hShape = -1, hTri = -1;
    float hu = 0.0f, hv = 0.0f;
    unsigned sgn = 0;  // bit a set = the ray runs towards smaller coordinates along axis a, in the space it is walked in
    auto signsOf = [](const V3 r) { return (r.x < 0.0f ? 1u : 0u) | (r.y < 0.0f ? 2u : 0u) | (r.z < 0.0f ? 4u : 0u); };
    // Splitting shadow rays.  A launch ends with its longest rays: one ray in a thousand makes ten times the average
    // number of tests, and when such a ray starts late, the whole machine waits for one lane.  Whether a shadow ray is
    // blocked does not depend on the order of its tests, so once the queue is dry, lanes without a ray take over the
    // oldest stack entry (the largest untouched subtree) of lanes that still walk: same ray, same space, a walk of
    // its own that ends with that subtree.  The ray's visibility byte is set to "visible" when the ray starts and
    // cleared by whichever part finds an occluder; a hit ends the other parts of the ray in the same wavefront.
    // Counter launches walk every ray in one piece, in the reference's order.
    constexpr bool kSplit = ANY && !COUNT;
    int floorAt = 0;  // first stack entry of the space the ray is walked in now (behind the marker while inside a mesh)

    RT_PROBE_STATE();  // diagnostic builds only (rt_trace_probes.h)

    auto push = [&](unsigned x, unsigned y) {
        if (sp - lo >= ringN) {
            gstk[(size_t)lo * kWave] = stk[(lo & ringMask) * kWave];
            ++lo;
        }
        stk[(sp & ringMask) * kWave] = make_uint2(x, y);
        ++sp;
    };
    auto pop = [&]() -> uint2 {
        // the ring slot is read in any case (an LDS read the common path needs anyway); entries older than the
        // ring come back from global memory
        --sp;
        uint2 e = stk[(sp & ringMask) * kWave];
        asm volatile("" : "+v"(e.x), "+v"(e.y));  // keeps this an LDS read of its own instead of one generic-address load
        if (sp < lo) {
            lo = sp;
            e = gstk[(size_t)sp * kWave];
        }
        return e;
    };
    // enter the node described by reference r (its box test has passed)
    auto enter = [&](unsigned r) {
        const unsigned f = RT_REF_LEAF_FIRST(r);
        const bool leaf = RT_REF_IS_LEAF(r);
        cur = leaf ? ((r & RT_REF_TAG_MASK) | f) : r;
        end = f + RT_REF_LEAF_COUNT(r);
    };

    // A finished part of a shadow ray (kSplit): an occluder clears the ray's visibility byte, a miss leaves it alone.
    auto publishShadow = [&]() {
        if (kDefer && hShape == HIT_DEFERRED) {
            const unsigned at = atomicAdd(P.ctl + CTL_DEFER_SHADOW, 1u);
            if (at < (unsigned)RT_DEFER_LIST) deferList[at] = (uint32_t)item;
            P.vis[item] = VIS_DEFERRED;
        } else if (hShape != -1) {
            P.vis[item] = 0;
        }
        item = -1;
    };

    while (true) {
        // ---- hand new rays to lanes without one ----------------------------
        const unsigned long long idle = __ballot(item < 0);
        if (idle != 0ull && !exhausted) {
            const unsigned want = (unsigned)pop64(idle);
            const unsigned rank = laneRank(idle);
            unsigned given = 0;
            while (given < want) {
                if (chunkNext == chunkEnd) {
                    unsigned c = chunks;
                    while (dryQueues < (unsigned)RT_QUEUES) {
                        const unsigned q = (blockIdx.x + dryQueues) % (unsigned)RT_QUEUES;
                        const unsigned held = (chunks + RT_QUEUES - 1 - q) / RT_QUEUES;  // chunks q, q + RT_QUEUES, ...
                        unsigned i = 0;
                        if (lane == 0) i = atomicAdd(queues + q * RT_QUEUE_STRIDE, 1u);
                        i = __builtin_amdgcn_readfirstlane(i);
                        // (i cannot wrap: a counter is asked at most once per chunk it holds plus once per wavefront)
                        if (i < held) {
                            c = q + i * RT_QUEUES;
                            break;
                        }
                        ++dryQueues;
                    }
                    if (c >= chunks) {
                        exhausted = true;
                        break;
                    }
                    chunkNext = c * kChunk;
                    chunkEnd = (total - chunkNext < kChunk) ? total : chunkNext + kChunk;
                }
                const unsigned take = (want - given < chunkEnd - chunkNext) ? want - given : chunkEnd - chunkNext;
                if (item < 0 && rank >= given && rank < given + take) item = (int)(first + chunkNext + (rank - given));
                chunkNext += take;
                given += take;
            }
            if (item >= 0 && cur == CUR_DONE) {
                // ---- start this lane's new ray
                bool valid = true;
                if (fromList) item = (int)deferList[item];
                if (rescue && (ANY ? P.vis[item] != VIS_DEFERRED : __float_as_int(P.hit[item].w) != HIT_DEFERRED)) {
                    item = -1;  // walked by the first launch
                } else if (ANY) {
                    const unsigned k = (unsigned)item / (unsigned)P.lightCnt;
                    const int li = (int)((unsigned)item - k * (unsigned)P.lightCnt);
                    bool unlit;
                    shadowRay(P, k, li, o, d, tMax, unlit);
                    if (!COUNT && !rescue && unlit) {
                        P.vis[item] = 0;  // "hidden": accumKernel passes over the light, which is what adding its zeros comes to
                        item = -1;
                    }
                } else {
                    if (item < P.slots) {
                        int x, y;
                        valid = slotPixel(P, item, x, y, !JIT);
                        if (valid && COUNT) cPrimary++;
                    }
                    if (valid) {
                        nodeRay(P, item, o, d, JIT);
                        tMax = 3.402823466e+38f;
                    }
                }
                if (item >= 0 && valid) {
                    inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    sgn = signsOf(inv);
                }
                if (item < 0) {
                    // nothing to walk
                } else if (kDefer && valid && special(inv, o)) {
                    hShape = HIT_DEFERRED;  // published as such below; cur stays CUR_DONE
                } else if (valid) {
                    if (COUNT) {
                        cRays++;
                        rayTests = cNode + cTri;
                    }
                    hShape = -1;
                    hTri = -1;
                    hu = hv = 0.0f;
                    floorAt = 0;
                    if (kSplit) P.vis[item] = 1;
                    sp = 0;
                    lo = 0;
                    // cudaTracer/Scene.cuh:38-39: the walk starts with the root's box test
                    cur = useSceneBvh ? (kQuad ? P.sceneRootQuad : P.sceneRootRef) : TAG_SHAPE;
                    end = (unsigned)P.shapeCnt;
                } else {
                    // tile pixel outside the frame: no ray, marked so that later stages skip it
                    P.hit[item] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-2));
                    item = -1;
                }
            }
        }
        if (__ballot(item >= 0) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // rays that were not started (lights that cannot contribute) leave their lanes free: fill up again before walking
        if (ANY && !COUNT && P.skipUnlit != 0 && !exhausted && pop64(__ballot(item < 0)) >= RT_REFILL_LANES / 2) continue;

        STAMP(0);  // refill
        // ---- advance the rays until enough lanes have finished --------------
        while (true) {
            const unsigned long long nodeMask = __ballot((int)cur >= 0);
            const unsigned long long triMask = __ballot((cur >> 30) == 2u);
            const int nTri = pop64(triMask);
            const bool triPhase = nTri != 0 && (nTri >= RT_TRI_LANES || nodeMask == 0ull);
            if (COUNT) {
                const unsigned nb = (unsigned)pop64(__ballot(cur != CUR_DONE));
                wSteps++;
                wBusyLanes += nb;
                if (exhausted) {
                    wTailSteps++;
                    wTailLanes += nb;
                }
            }
            STAMP(1);  // vote
            if (triPhase) {
                // ================= cooperative triangle phase =================
                const bool isTri = (cur >> 30) == 2u;
                const unsigned rank = laneRank(triMask);
                const bool part = isTri && rank < (unsigned)kTriPart;
                const unsigned next = cur & IDX_MASK;
                const int nPart = nTri < kTriPart ? nTri : kTriPart;
                if (COUNT) {
                    wTriSteps++;
                    wTriLanes += nPart;
                }
                if (part) {
                    rayBuf[2 * rank] = make_float4(o.x, o.y, o.z, __uint_as_float(next));
                    rayBuf[2 * rank + 1] = make_float4(d.x, d.y, d.z, __uint_as_float(end));
                    if (ANY) tBuf[rank] = tMax;  // a shadow ray's reach never shrinks: the testing lanes can apply it themselves
                }
                __syncthreads();
                STAMP(4);  // triangle phase: rays published
                const int rounds = (nPart + kPerRound - 1) / kPerRound;
                // ANY: lanes whose triangle the ray accepts, per round (bit = lane: 8 bits per participating ray)
                unsigned long long accepted[kRounds] = {};
#pragma unroll
                for (int r = 0; r < kRounds; ++r) {
                    if (r >= rounds) break;
                    const int rr = r * kPerRound + lane / kGrp;
                    const unsigned k = (unsigned)lane % (unsigned)kGrp;
                    const float4 A = rayBuf[2 * rr], B = rayBuf[2 * rr + 1];
                    const unsigned idx = __float_as_uint(A.w) + k;
                    float t = -1.0f, u = 0.0f, v = 0.0f;
                    bool solved = false;
                    if (rr < nPart && idx < __float_as_uint(B.w)) {
                        const float4* rec = records + 3 * (size_t)idx;
                        const float4 r0 = rec[0], r1 = rec[1];
                        const float e2z = reinterpret_cast<const float*>(rec + 2)[0];
                        STAMP(5);  // triangle phase: records arrived
                        float tt, uu, vv;
                        if (triangleSolve(r0, r1, e2z, mk(A.x, A.y, A.z), mk(B.x, B.y, B.z), tt, uu, vv)) {
                            solved = true;
                            t = tt;
                            u = uu;
                            v = vv;
                        }
                    }
                    if (ANY) {
                        // cudaTracer/Mesh.cuh:205-208 against the reach the ray came with
                        const float reach = tBuf[rr < nPart ? rr : 0];
                        accepted[r] = __ballot(solved && !(t < 0.0f) && !(reach < t));
                    } else {
                        // a rejected or absent triangle is stored as t = -1, which the owner's "t < 0" test discards
                        tBuf[rr * 8 + k] = t;
                        uvBuf[rr * 8 + k] = make_float2(u, v);
                    }
                }
                if (!ANY) __syncthreads();
                STAMP(6);  // triangle phase: tests done, results in LDS
                if (part) {
                    int win = -1;
                    if (ANY) {
                        // the first accepted triangle of the leaf ends the walk (later ones cannot change the answer)
                        unsigned long long m = accepted[0];
#pragma unroll
                        for (int r = 1; r < kRounds; ++r) m = (rank / (unsigned)kPerRound) == (unsigned)r ? accepted[r] : m;
                        const unsigned bits = (unsigned)(m >> ((rank % (unsigned)kPerRound) * (unsigned)kGrp)) & ((1u << kGrp) - 1u);
                        win = bits ? __ffs((int)bits) - 1 : -1;
                    } else {
                        // cudaTracer/Mesh.cuh:141-146 + 205-208: leaf triangles in order; a hit needs t >= 0 and t <= reach,
                        // so a later hit at equal t replaces an earlier one
                        const float4 ta = reinterpret_cast<const float4*>(tBuf)[2 * rank], tb = reinterpret_cast<const float4*>(tBuf)[2 * rank + 1];
                        const float ts[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
#pragma unroll
                        for (int k = 0; k < kGrp; ++k) {
                            const bool acc = !(ts[k] < 0.0f) && !(tMax < ts[k]);
                            tMax = acc ? ts[k] : tMax;
                            win = acc ? k : win;
                        }
                    }
                    const unsigned left = end - next;
                    if (COUNT) cTri += (ANY && win >= 0) ? (unsigned)win + 1u : (left < (unsigned)kGrp ? left : (unsigned)kGrp);
                    if (win >= 0) {
                        if (!ANY) {
                            const float2 uv = uvBuf[rank * 8 + win];
                            hu = uv.x;
                            hv = uv.y;
                            hTri = (int)(next + (unsigned)win);
                        }
                        hShape = HIT_PENDING;
                    }
                    const unsigned nn = next + (unsigned)kGrp;
                    cur = (ANY && win >= 0) ? CUR_DONE : (nn >= end ? CUR_POP : (TAG_TRI | nn));
                }
                STAMP(7);  // triangle phase: results applied
            } else if (nodeMask != 0ull) {
                // ========================= node step ==========================
                if (COUNT) {
                    wNodeSteps++;
                    wNodeLanes += pop64(nodeMask);
                }
                RT_PROBE_QUAD_VOTE();
                if (kQuad && (int)cur >= 0) {
                    // One 128-byte fetch: the child pairs of both children of the node (halves A and B).  The children's
                    // own boxes are not tested: a grandchild's box lies inside its parent's, so a ray that passes the
                    // grandchild's test would have passed the parent's, with a smaller entry distance.
                    const float4* p = P.quads + ((size_t)RT_PROBE_QUAD_UNITS(cur) << 3);
                    const float4 A0 = p[0], A1 = p[1], A2 = p[2], A3 = p[3], B0 = p[4], B1 = p[5], B2 = p[6], B3 = p[7];
                    const auto B = [](bool c) { return __builtin_amdgcn_ballot_w64(c); };
                    float ta0, fa0, ta1, fa1, tb0, fb0, tb1, fb1;
                    boxPairMinMax(A0, A1, A2, A3, o, inv, ta0, fa0, ta1, fa1);
                    boxPairMinMax(B0, B1, B2, B3, o, inv, tb0, fb0, tb1, fb1);
                    // Split axes as one-hot words: of the node (orders the halves) and of its two children (order inside a
                    // half; 0 where a half holds fewer than two records).  Records that are not there have NaN boxes, which
                    // pass no test.  The root's pseudo reference has axis 3: its bit meets none of the ray's three.
                    const unsigned ohP = 1u << RT_REF_AXIS(cur), ohA = __float_as_uint(A1.w), ohB = __float_as_uint(B1.w);
                    const unsigned long long mNP = B((sgn & ohP) != 0u), mNA = B((sgn & ohA) != 0u), mNB = B((sgn & ohB) != 0u);
                    const unsigned long long hA0 = B(ta0 <= fa0) & B(fa0 > 0.0f) & B(ta0 < tMax);
                    const unsigned long long hA1 = B(ta1 <= fa1) & B(fa1 > 0.0f) & B(ta1 < tMax);
                    const unsigned long long hB0 = B(tb0 <= fb0) & B(fb0 > 0.0f) & B(tb0 < tMax);
                    const unsigned long long hB1 = B(tb1 <= fb1) & B(fb1 > 0.0f) & B(tb1 < tMax);
                    // cudaTracer/Mesh.cuh:155-161 twice over: inside a half the record on the ray's side of the child's
                    // split comes first, and the half on the ray's side of the node's split comes first
                    const bool negA = __builtin_amdgcn_inverse_ballot_w64(mNA), negB = __builtin_amdgcn_inverse_ballot_w64(mNB);
                    const bool negP = __builtin_amdgcn_inverse_ballot_w64(mNP);
                    const unsigned rAf = __float_as_uint(negA ? A3.z : A1.z), rAs = __float_as_uint(negA ? A1.z : A3.z);
                    const unsigned rBf = __float_as_uint(negB ? B3.z : B1.z), rBs = __float_as_uint(negB ? B1.z : B3.z);
                    const float tAf = negA ? ta1 : ta0, tAs = negA ? ta0 : ta1, tBf = negB ? tb1 : tb0, tBs = negB ? tb0 : tb1;
                    const unsigned long long hAf = (mNA & hA1) | (~mNA & hA0), hAs = (mNA & hA0) | (~mNA & hA1);
                    const unsigned long long hBf = (mNB & hB1) | (~mNB & hB0), hBs = (mNB & hB0) | (~mNB & hB1);
                    const unsigned long long h0 = (mNP & hBf) | (~mNP & hAf), h1 = (mNP & hBs) | (~mNP & hAs);
                    const unsigned long long h2 = (mNP & hAf) | (~mNP & hBf), h3 = (mNP & hAs) | (~mNP & hBs);
                    unsigned r0 = negP ? rBf : rAf, r1 = negP ? rBs : rAs, r2 = negP ? rAf : rBf, r3 = negP ? rAs : rBs;
                    RT_PROBE_QUAD_STEP();
                    const float t1 = negP ? tBs : tAs, t2 = negP ? tAf : tBf, t3 = negP ? tAs : tBs;
                    // the first of the four that is hit is entered, the later ones wait on the stack, the next one on top
                    if (__builtin_amdgcn_inverse_ballot_w64(h3 & (h0 | h1 | h2))) push(r3, __float_as_uint(t3));
                    if (__builtin_amdgcn_inverse_ballot_w64(h2 & (h0 | h1))) push(r2, __float_as_uint(t2));
                    if (__builtin_amdgcn_inverse_ballot_w64(h1 & h0)) push(r1, __float_as_uint(t1));
                    const bool e0 = __builtin_amdgcn_inverse_ballot_w64(h0), e1 = __builtin_amdgcn_inverse_ballot_w64(h1);
                    const bool e2 = __builtin_amdgcn_inverse_ballot_w64(h2), e3 = __builtin_amdgcn_inverse_ballot_w64(h3);
                    enter(e0 ? r0 : (e1 ? r1 : (e2 ? r2 : r3)));
                    if (!(e0 || e1 || e2 || e3)) cur = CUR_POP;
                } else if ((int)cur >= 0) {
                    const float4* p = blob4 + ((size_t)RT_REF_PAIR_UNITS(cur) << 2);
                    const float4 L0 = p[0], L1 = p[1], L2 = p[2], L3 = p[3];
                    STAMP(2);  // node step: child pair arrived
                    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
                    RT_PROBE_NODE_STEP();
                    // Lane predicates live in scalar mask registers: every mask below is the ballot of ONE comparison (a
                    // vector compare that writes a scalar register pair) and all the logic on them is scalar arithmetic.
                    const auto B = [](bool c) { return __builtin_amdgcn_ballot_w64(c); };
                    float t0, t1;
                    unsigned long long mH0, mH1;  // child 0 / 1 is hit within the ray's reach
                    if (LIT) {
                        bool g0, g1;
                        boxPairSlabs(L0, L1, L2, L3, o, inv, nx, ny, nz, g0, g1, t0, t1);
                        mH0 = __ballot(g0 && t0 < tMax);
                        mH1 = __ballot(g1 && t1 < tMax);
                    } else {
                        float f0, f1;
                        boxPairMinMax(L0, L1, L2, L3, o, inv, t0, f0, t1, f1);
                        mH0 = B(t0 <= f0) & B(f0 > 0.0f) & B(t0 < tMax);
                        mH1 = B(t1 <= f1) & B(f1 > 0.0f) & B(t1 < tMax);
                    }
                    const unsigned axis = RT_REF_AXIS(cur);
                    // cudaTracer/Mesh.cuh:155-161: with a negative direction along the split axis the
                    // second child is nearer; the other one waits on the stack.  A root record has no sibling.
                    const unsigned long long mNeg = B(((sgn >> axis) & 1u) != 0u);  // (the root's axis 3 meets no sign bit)
                    const unsigned long long mSingle = B(axis == (unsigned)RT_REF_ROOT_AXIS);
                    const bool single = __builtin_amdgcn_inverse_ballot_w64(mSingle);
                    const bool negAxis = __builtin_amdgcn_inverse_ballot_w64(mNeg);
                    const unsigned nearRef = __float_as_uint(negAxis ? L3.z : L1.z);
                    const unsigned farRef = __float_as_uint(negAxis ? L1.z : L3.z);
                    // near / far outcome = the two boxes' outcomes swapped where negAxis holds
                    const bool nearHit = __builtin_amdgcn_inverse_ballot_w64((mNeg & mH1) | (~mNeg & mH0));
                    const bool farHit = __builtin_amdgcn_inverse_ballot_w64(((mNeg & mH0) | (~mNeg & mH1)) & ~mSingle);
                    const float farT = negAxis ? t0 : t1;
                    const bool doPush = COUNT ? !single : farHit;
                    // the entry is written in any case (above the top of the stack it is harmless; with a full
                    // ring and nothing to push it goes to the dump slot), which keeps the common path free of branches
                    const uint2 e = make_uint2(farRef, __float_as_uint(farHit ? farT : kInf));
                    const bool full = sp - lo >= ringN;
                    if (doPush && full) {
                        gstk[(size_t)lo * kWave] = stk[(lo & ringMask) * kWave];
                        ++lo;
                    }
                    stk[((!doPush && full) ? ringN : (sp & ringMask)) * kWave] = e;
                    sp += doPush ? 1 : 0;
                    if (COUNT) cNode++;
                    enter(nearRef);
                    if (!nearHit) cur = CUR_POP;
                }
                STAMP(3);  // node step: boxes tested, far child pushed
            }
            // ================= scene-leaf shapes (per lane) ==================
            if (cur >= TAG_SHAPE && cur < STACK_MESH_EXIT) {
                const unsigned idx = cur & IDX_MASK;
                const unsigned nextShape = idx + 1u;
                const DevShape& s = shapes[idx];
                if (s.kind == RT_SHAPE_SPHERE) {
                    if (COUNT) cSphere++;
                    const bool hit = sphereHit(s, o, d, tMax);
                    if (hit) hShape = (int)idx;
                    cur = (ANY && hit) ? CUR_DONE : (nextShape < end ? (TAG_SHAPE | nextShape) : CUR_POP);
                } else {
                    // cudaTracer/Mesh.cuh:115-117 + cudaTracer/Mat.cuh:127-149: the origin is
                    // divided by w, the direction is neither translated nor renormalised
                    const DevMesh* m = meshes + s.meshIdx;
                    const float4* M = reinterpret_cast<const float4*>(m->inv);
                    const float4 m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3];
                    if (COUNT) cMesh++;
                    gsave[0 * kWave] = make_uint2(__float_as_uint(o.x), __float_as_uint(o.y));
                    gsave[1 * kWave] = make_uint2(__float_as_uint(o.z), __float_as_uint(d.x));
                    gsave[2 * kWave] = make_uint2(__float_as_uint(d.y), __float_as_uint(d.z));
                    gsave[3 * kWave] = make_uint2(__float_as_uint(inv.x), __float_as_uint(inv.y));
                    gsave[4 * kWave] = make_uint2(__float_as_uint(inv.z), (unsigned)floorAt);
                    gsave[5 * kWave] = make_uint2(TAG_SHAPE | nextShape, end);
                    push(STACK_MESH_EXIT, idx);
                    floorAt = sp;
                    const V3 wo = o, wd = d;
                    const float pa = ((wo.x * m0.x + wo.y * m1.x) + wo.z * m2.x) + m3.x;
                    const float pb = ((wo.x * m0.y + wo.y * m1.y) + wo.z * m2.y) + m3.y;
                    const float pc = ((wo.x * m0.z + wo.y * m1.z) + wo.z * m2.z) + m3.z;
                    const float pw = ((wo.x * m0.w + wo.y * m1.w) + wo.z * m2.w) + m3.w;
                    o = mk(pa / pw, pb / pw, pc / pw);
                    d = mk((wd.x * m0.x + wd.y * m1.x) + wd.z * m2.x, (wd.x * m0.y + wd.y * m1.y) + wd.z * m2.y,
                           (wd.x * m0.z + wd.y * m1.z) + wd.z * m2.z);
                    inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    sgn = signsOf(inv);
                    // cudaTracer/Mesh.cuh:137-138: root box test, or all triangles in order without a BVH
                    cur = useMeshBvhs ? (kQuad ? P.meshRootQuad[s.meshIdx] : m->rootRef) : (TAG_TRI | m->recordBase);
                    end = m->recordBase + (unsigned)m->triangleCnt;
                    if (kDefer && special(inv, o)) {
                        // in this mesh's space the ray runs along an axis plane: the whole ray goes to the rescue launch
                        hShape = HIT_DEFERRED;
                        cur = CUR_DONE;
                    }
                }
            }
            STAMP(8);  // scene-leaf shapes
            // ========================= stack pops ============================
            RT_POP_WHILE (cur == CUR_POP) {
                if (sp == 0) {
                    cur = CUR_DONE;
                } else {
                    const uint2 e = pop();
                    if (e.x == STACK_MESH_EXIT) {
                        // mesh walk finished: back to world space (t is shared between the spaces)
                        if (hShape == HIT_PENDING) hShape = (int)e.y;
                        const uint2 e0 = gsave[0 * kWave], e1 = gsave[1 * kWave], e2 = gsave[2 * kWave], e3 = gsave[3 * kWave],
                                    e4 = gsave[4 * kWave], c = gsave[5 * kWave];
                        o = mk(__uint_as_float(e0.x), __uint_as_float(e0.y), __uint_as_float(e1.x));
                        d = mk(__uint_as_float(e1.y), __uint_as_float(e2.x), __uint_as_float(e2.y));
                        inv = mk(__uint_as_float(e3.x), __uint_as_float(e3.y), __uint_as_float(e4.x));
                        sgn = signsOf(inv);
                        floorAt = (int)e4.y;
                            end = c.y;
                        cur = (c.x & IDX_MASK) < c.y ? c.x : CUR_POP;
                    } else {
                        if (COUNT) cNode++;
                        if (kSplit && sp < floorAt) floorAt = sp;  // popped past entries given away earlier
                        if (__uint_as_float(e.y) < tMax) enter(e.x);
                    }
                }
            }
            STAMP(9);  // pops
            // (every fourth iteration: entries are handed on a little later, and the iterations in between stay short)
            if (kSplit && exhausted && (++drainSteps & 3u) == 0u) {
                // ---- a part that has found an occluder ends the other parts of its ray; finished parts free their lanes
                const bool fin = item >= 0 && cur == CUR_DONE;
                unsigned long long hitM = __ballot(fin && hShape != -1 && hShape != HIT_DEFERRED);
                while (hitM != 0ull) {
                    const int from = __ffsll((long long)hitM) - 1;
                    hitM &= hitM - 1ull;
                    const int blocked = __builtin_amdgcn_readlane(item, from);
                    if (item == blocked && cur != CUR_DONE) {
                        cur = CUR_DONE;
                        item = -1;
                    }
                }
                if (fin) publishShadow();
                // ---- lanes without a ray take over the oldest entry of lanes that walk
                const bool taker = item < 0;
                const bool giver = item >= 0 && cur != CUR_DONE && sp > floorAt;
                const unsigned long long takeM = __ballot(taker), giveM = __ballot(giver);
                if (takeM != 0ull && giveM != 0ull) {
                    unsigned n = (unsigned)pop64(takeM) < (unsigned)pop64(giveM) ? (unsigned)pop64(takeM) : (unsigned)pop64(giveM);
                    n = n < (unsigned)kTriPart ? n : (unsigned)kTriPart;
                    const unsigned gr = laneRank(giveM), tr = laneRank(takeM);
                    if (giver && gr < n) {
                        const int at = floorAt;
                        const bool inRing = at >= lo;
                        const uint2 e = inRing ? stk[(at & ringMask) * kWave] : gstk[(size_t)at * kWave];
                        // the entry stays where it is, as one that is never entered
                        const uint2 dead = make_uint2(e.x, __float_as_uint(kInf));
                        if (inRing)
                            stk[(at & ringMask) * kWave] = dead;
                        else
                            gstk[(size_t)at * kWave] = dead;
                        floorAt = at + 1;
                        rayBuf[2 * gr] = make_float4(o.x, o.y, o.z, tMax);
                        rayBuf[2 * gr + 1] = make_float4(d.x, d.y, d.z, __uint_as_float(e.x));
                        tBuf[2 * gr] = __int_as_float(item);
                        tBuf[2 * gr + 1] = __uint_as_float(e.y);
                    }
                    __syncthreads();
                    if (taker && tr < n) {
                        const float4 A = rayBuf[2 * tr], B = rayBuf[2 * tr + 1];
                        const float entry = tBuf[2 * tr + 1];
                        o = mk(A.x, A.y, A.z);
                        tMax = A.w;
                        d = mk(B.x, B.y, B.z);
                        inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        sgn = signsOf(inv);
                        hShape = -1;
                        sp = 0;
                        lo = 0;
                        floorAt = 0;
                        if (entry < tMax) {  // the box test the entry stands for, against the ray's reach
                            item = __float_as_int(tBuf[2 * tr]);
                            enter(__float_as_uint(B.w));
                        }
                    }
                    __syncthreads();
                }
            }
            // keep going while most lanes still work; once the queue is dry, until all are done
            const unsigned long long busy = __ballot(cur != CUR_DONE);
            if (busy == 0ull) break;
            if (!exhausted && pop64(busy) <= kWave - RT_REFILL_LANES && __ballot(item >= 0) != busy) break;
        }

        // ---- finished lanes publish their result and give up their ray -----
        if (kSplit) {
            if (item >= 0 && cur == CUR_DONE) publishShadow();
        } else if (item >= 0 && cur == CUR_DONE) {
            if (COUNT && P.tileCost) {
                // box + triangle tests of this ray, booked on the frame tile of its pixel (primary rays and the shadow
                // rays of primary hits; deeper rays are not attributed)
                const int node = ANY ? P.ctxNode[(unsigned)item / (unsigned)P.lightCnt] : item;
                if (node < P.slots) {
                    const int local = P.opts.samples == 1 ? node >> 6 : (node >> 6) / P.opts.samples;
                    const int tile = P.tileOrder ? P.tileOrder[local] : P.tileFirst + local * P.tileStride;
                    atomicAdd(P.tileCost + tile, cNode + cTri - rayTests);
                }
            }
            const bool leftOver = kDefer && hShape == HIT_DEFERRED;
            if (leftOver) {
                const unsigned at = atomicAdd(P.ctl + (ANY ? CTL_DEFER_SHADOW : CTL_DEFER_CLOSEST), 1u);
                if (at < (unsigned)RT_DEFER_LIST) deferList[at] = (uint32_t)item;
            }
            if (ANY) {
                P.vis[item] = leftOver ? VIS_DEFERRED : (hShape != -1 ? 0 : 1);
            } else {
                int tri = hTri;
                if (hShape >= 0 && shapes[hShape].kind == RT_SHAPE_MESH) tri -= (int)meshes[shapes[hShape].meshIdx].recordBase;
                P.hit[item] = make_float4(tMax, hu, hv, __int_as_float(hShape >= 0 ? hShape : (leftOver ? HIT_DEFERRED : -1)));
                P.hitTri[item] = tri;
            }
            item = -1;
        }
        STAMP(10);  // results published
    }

    RT_PROBE_QUAD_FLUSH();
    if (COUNT && P.counters) {
        RT_PROBE_FLUSH();
        unsigned vals[6] = {cNode, cTri, cSphere, cMesh, cRays, cPrimary};
        const int where[6] = {RT_CNT_NODE, RT_CNT_TRI, RT_CNT_SPHERE, RT_CNT_MESH_ENTRY,
                              ANY ? RT_CNT_SHADOW : (level > 0 ? RT_CNT_SECONDARY : -1), ANY ? -1 : RT_CNT_PRIMARY};
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            unsigned v = vals[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
            if (lane == 0 && v && where[i] >= 0) atomicAdd(P.counters + where[i], (unsigned long long)v);
            if (ANY && i < 4 && lane == 0 && v) atomicAdd(P.counters + RT_CNT_SHADOW_NODE + i, (unsigned long long)v);
        }
        if (lane == 0) {
            const int b = ANY ? RT_CNT_STAT_SHADOW : RT_CNT_STAT_CLOSEST;
            atomicAdd(P.counters + b + 0, wSteps);
            atomicAdd(P.counters + b + 1, wNodeSteps);
            atomicAdd(P.counters + b + 2, wNodeLanes);
            atomicAdd(P.counters + b + 3, wTriSteps);
            atomicAdd(P.counters + b + 4, wTriLanes);
            atomicAdd(P.counters + b + 5, wBusyLanes);
            atomicAdd(P.counters + (ANY ? RT_CNT_STAT_TAIL_SHADOW : RT_CNT_STAT_TAIL_CLOSEST) + 0, wTailSteps);
            atomicAdd(P.counters + (ANY ? RT_CNT_STAT_TAIL_SHADOW : RT_CNT_STAT_TAIL_CLOSEST) + 1, wTailLanes);
        }
    }
}

// ---------------------------------------------------------------------------
// Shading.

// cudaTracer/Texture.cuh:50-69.  WRAP wraps negative indices into range (the
// reference's C '%' leaves them negative and reads out of bounds).
__device__ __forceinline__ V3 texel(const unsigned char* __restrict__ data, int w, int h, int mode, int x, int y) {
    if (mode == RT_TEX_CLAMP) {
        x = x < 0 ? 0 : x;
        x = x > w - 1 ? w - 1 : x;
        y = y < 0 ? 0 : y;
        y = y > h - 1 ? h - 1 : y;
    } else {
        x = x % w;
        y = y % h;
        if (x < 0) x += w;
        if (y < 0) y += h;
    }
    const uchar4 px = *reinterpret_cast<const uchar4*>(data + ((size_t)y * w + x) * 4);
    return mk((float)px.x / 255.0f, (float)px.y / 255.0f, (float)px.z / 255.0f);
}
// cudaTracer/Texture.cuh:33-48: texel-centre offset and the two "1 - ratio"
// weights go through double; every weight is rounded to float before it scales a colour.
__device__ __forceinline__ V3 texSample(const unsigned char* __restrict__ data, int w, int h, int mode, float u, float v) {
    u = (float)((double)(u * (float)w) - 0.5);
    v = (float)((double)(v * (float)h) - 0.5);
    const int x = (int)floorf(u);
    const int y = (int)floorf(v);
    const float ur = u - (float)x;
    const float vr = v - (float)y;
    const float uo = (float)(1.0 - (double)ur);
    const float vo = (float)(1.0 - (double)vr);
    const V3 top = add(muls(texel(data, w, h, mode, x, y), uo), muls(texel(data, w, h, mode, x + 1, y), ur));
    const V3 bot = add(muls(texel(data, w, h, mode, x, y + 1), uo), muls(texel(data, w, h, mode, x + 1, y + 1), ur));
    return add(muls(top, vo), muls(bot, vr));
}
__device__ __forceinline__ V3 meshTex(const FrameParams& P, const DevMesh* m, int which, V2 st) {
    return texSample(P.blob + m->offTex[which], m->texW[which], m->texH[which], m->texMode[which], st.x, st.y);
}

// cudaTracer/tracer.cu:54-115
__device__ __forceinline__ float clampf(float lo, float hi, float v) { return fmaxf(lo, fminf(hi, v)); }
__device__ __forceinline__ float fresnel(V3 i, V3 n, float ior) {
    float cosi = clampf(-1.0f, 1.0f, dot(i, n));
    float etai = 1.0f, etat = ior;
    if (cosi > 0.0f) {
        const float tmp = etat;
        etat = etai;
        etai = tmp;
    }
    const float sint = etai / etat * sqrtf(fmaxf(0.f, 1.0f - cosi * cosi));
    if (sint >= 1.0f) return 1.0f;
    const float cost = sqrtf(fmaxf(0.f, 1.0f - sint * sint));
    cosi = fabsf(cosi);
    const float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    const float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return (Rs * Rs + Rp * Rp) * 0.5f;
}
__device__ __forceinline__ V3 refract(V3 I, V3 N, float ior) {
    float cosi = clampf(-1.0f, 1.0f, dot(I, N));
    float etai = 1.0f, etat = ior;
    V3 n = N;
    if (cosi < 0.0f) {
        cosi = -cosi;
    } else {
        const float tmp = etai;
        etai = etat;
        etat = tmp;
        n = neg(N);
    }
    const float eta = etai / etat;
    const float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
    if (k < 0.0f) return mk(0.0f, 0.0f, 0.0f);
    return add(muls(I, eta), muls(n, eta * cosi - sqrtf(k)));
}

// cudaTracer/Mesh.cuh:101-109
__device__ __forceinline__ float lerp3(float a, float b, float c, float w, float u, float v) { return (a * w + b * u) + c * v; }

// st + normal of a mesh hit (cudaTracer/Mesh.cuh:65-95); the normal stays in object space.
__device__ __forceinline__ void meshSurface(const FrameParams& P, const DevMesh* m, const TriRecord* rec, float hu, float hv,
                                            V2& st, V3& nOut) {
    const RtVertex* verts = reinterpret_cast<const RtVertex*>(P.blob + m->offVertices);
    const RtVertex* a = verts + rec->idx[0];
    const RtVertex* b = verts + rec->idx[1];
    const RtVertex* c = verts + rec->idx[2];
    const float w = (1.0f - hu) - hv;
    st.x = lerp3(a->texCoord[0], b->texCoord[0], c->texCoord[0], w, hu, hv);
    st.y = lerp3(a->texCoord[1], b->texCoord[1], c->texCoord[1], w, hu, hv);
    V3 t, bt, n;
    t.x = lerp3(a->tangent[0], b->tangent[0], c->tangent[0], w, hu, hv);
    t.y = lerp3(a->tangent[1], b->tangent[1], c->tangent[1], w, hu, hv);
    t.z = lerp3(a->tangent[2], b->tangent[2], c->tangent[2], w, hu, hv);
    bt.x = lerp3(a->bitangent[0], b->bitangent[0], c->bitangent[0], w, hu, hv);
    bt.y = lerp3(a->bitangent[1], b->bitangent[1], c->bitangent[1], w, hu, hv);
    bt.z = lerp3(a->bitangent[2], b->bitangent[2], c->bitangent[2], w, hu, hv);
    n.x = lerp3(a->normal[0], b->normal[0], c->normal[0], w, hu, hv);
    n.y = lerp3(a->normal[1], b->normal[1], c->normal[1], w, hu, hv);
    n.z = lerp3(a->normal[2], b->normal[2], c->normal[2], w, hu, hv);
    V3 ns = meshTex(P, m, 1, st);
    ns = mk(ns.x * 2.0f - 1.0f, ns.y * 2.0f - 1.0f, ns.z * 2.0f - 1.0f);
    nOut = normalized(add(add(n, muls(t, ns.x)), muls(bt, ns.y)));
}

// ---------------------------------------------------------------------------
// Shade the nodes of `level`: the body of castRay after trace()
// (cudaTracer/tracer.cu:22-47).  Diffuse hits leave a shading context and
// become shadow work; mirror / glass hits spawn their child rays into the next
// level (slots taken with one atomic per wavefront via ballot compaction), or,
// at the last level, take the background as the child result exactly as the
// reference's depth test does (cudaTracer/tracer.cu:16-18).
template <bool COUNT>
__global__ void __launch_bounds__(256) shadeKernel(const FrameParams P, const int level) {
    const unsigned start = P.ctl[CTL_LEVEL_START + level];
    const unsigned count = P.ctl[CTL_LEVEL_COUNT + level];
    const unsigned nextStart = start + count;
    const V3 bg = mk(P.opts.backgroundColor);
    const DevShape* shapes = reinterpret_cast<const DevShape*>(P.blob + P.offShapes);
    unsigned cMesh = 0, cSphere = 0;
    // per round: diffuse hits and child rays of the block's four wavefronts, and the first slot of each kind
    __shared__ unsigned s_count[4], s_want[4], s_base, s_childBase;
    const unsigned stride = gridDim.x * blockDim.x;
    // every thread of a block runs the same number of iterations (the ballots and barriers below need that)
    const unsigned rounds = (count + stride - 1) / stride;
    for (unsigned r = 0; r < rounds; ++r) {
        const unsigned idx = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = idx < count;
        const int node = (int)(start + (live ? idx : 0));
        int want = 0;          // child rays this lane needs
        bool diffuse = false;  // becomes a diffuse hit
        V3 hitPoint = mk(0, 0, 0), n = mk(0, 0, 0), rd = mk(0, 0, 0), diffuseColor = mk(0, 0, 0);
        float specMask = 0.0f, specExp = 0.0f, ior = 1.0f;
        int materialType = -1;
        if (live) {
            const float4 h = P.hit[node];
            const int shapeSlot = __float_as_int(h.w);
            if (shapeSlot == -2) {
                // pixel of a partial tile outside the frame
            } else if (shapeSlot < 0) {
                P.color[node] = make_float4(bg.x, bg.y, bg.z, 0.0f);
                P.kind[node] = NODE_DONE;
                if (level == 0 && P.hitKeys && slotSample(P, node) == 0) {
                    int x, y;
                    slotPixel(P, node, x, y);
                    if (pixelSampleBase(P, x, y) == 0u) *reinterpret_cast<int4*>(P.hitKeys + ((size_t)y * P.opts.width + x) * 4) = make_int4(-1, -1, -1, -1);
                }
            } else {
                V3 ro;
                nodeRay(P, node, ro, rd);
                const DevShape& shape = shapes[shapeSlot];
                hitPoint = add(ro, muls(rd, h.x));
                V2 st = V2{1.0f, 1.0f};
                int4 key = make_int4(shape.tag, -1, -1, -1);
                const bool isMesh = shape.kind == RT_SHAPE_MESH;
                const DevMesh* m = reinterpret_cast<const DevMesh*>(P.blob + P.offMeshes) + (isMesh ? shape.meshIdx : 0);
                if (!isMesh) {
                    n = normalized(sub(hitPoint, mk(shape.center)));
                    if (COUNT) cSphere++;
                } else {
                    const TriRecord* rec = reinterpret_cast<const TriRecord*>(P.blob + m->offTriRecords) + P.hitTri[node];
                    meshSurface(P, m, rec, h.y, h.z, st, n);
                    key.y = rec->idx[0];
                    key.z = rec->idx[1];
                    key.w = rec->idx[2];
                    if (COUNT) cMesh++;
                }
                if (level == 0 && P.hitKeys && slotSample(P, node) == 0) {
                    int x, y;
                    slotPixel(P, node, x, y);
                    if (pixelSampleBase(P, x, y) == 0u) *reinterpret_cast<int4*>(P.hitKeys + ((size_t)y * P.opts.width + x) * 4) = key;
                }
                materialType = shape.materialType;
                ior = shape.ior;
                if (materialType == RT_MAT_DIFFUSE_AND_GLOSSY) {
                    diffuse = true;
                    diffuseColor = isMesh ? meshTex(P, m, 0, st) : mk(shape.diffuseColor);
                    specMask = isMesh ? meshTex(P, m, 2, st).x : shape.Ks;
                    specExp = shape.specularExponent;
                } else if (materialType == RT_MAT_REFLECTION) {
                    want = 1;
                } else if (materialType == RT_MAT_REFLECTION_AND_REFRACTION) {
                    want = 2;
                } else {
                    P.color[node] = make_float4(1.0f, 0.5f, 0.25f, 0.0f);
                    P.kind[node] = NODE_DONE;
                }
            }
        }
        // ---- mirror / glass hits: directions and Fresnel term; at the last level the children are the background
        V3 reflDir = mk(0, 0, 0);
        float kr = 0.0f;
        if (want > 0) {
            // cudaTracer/tracer.cu:117-142.  Glass reflects the NORMAL about the view direction
            // (operands swapped with respect to the mirror case), as the reference does.
            reflDir = normalized(materialType == RT_MAT_REFLECTION ? reflect(rd, n) : reflect(n, rd));
            kr = fresnel(rd, n, ior);
            if (level >= P.opts.maxDepth) {
                // children would be deeper than maxDepth: each returns the background
                const V3 c = (want == 1) ? muls(bg, kr) : add(muls(bg, kr), muls(bg, 1.0f - kr));
                P.color[node] = make_float4(c.x, c.y, c.z, 0.0f);
                P.kind[node] = NODE_DONE;
                want = 0;
            } else {
                P.kr[node] = kr;
                P.kind[node] = want == 1 ? NODE_MIRROR : NODE_GLASS;
            }
        }
        // ---- slots in the level's shading-context list (diffuse hits) and in the next level (child rays): ballot
        // compaction per wavefront, then ONE atomic of each kind per block and round - tens of thousands of atomics
        // on one address take longer than everything else this kernel does
        const unsigned long long dmask = __ballot(diffuse);
        const unsigned long long m1 = __ballot(want >= 1), m2 = __ballot(want == 2);  // want is 0, 1 or 2
        const unsigned wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63u) == 0u) {
            s_count[wave] = (unsigned)pop64(dmask);
            s_want[wave] = (unsigned)pop64(m1) + (unsigned)pop64(m2);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned hits = s_count[0] + s_count[1] + s_count[2] + s_count[3];
            const unsigned children = s_want[0] + s_want[1] + s_want[2] + s_want[3];
            s_base = hits ? atomicAdd(P.ctl + CTL_DIFFUSE_CNT, hits) : 0u;
            s_childBase = children ? atomicAdd(P.ctl + CTL_LEVEL_COUNT + level + 1, children) : 0u;
        }
        __syncthreads();
        unsigned base = s_base, childBase = s_childBase;
        for (unsigned w = 0; w < wave; ++w) {
            base += s_count[w];
            childBase += s_want[w];
        }
        __syncthreads();  // the next round overwrites the counts
        if (diffuse) {
            const unsigned k = base + laneRank(dmask);
            float4* c = P.ctx + 4 * (size_t)k;
            c[0] = make_float4(hitPoint.x, hitPoint.y, hitPoint.z, n.x);
            c[1] = make_float4(n.y, n.z, rd.x, rd.y);
            c[2] = make_float4(rd.z, diffuseColor.x, diffuseColor.y, diffuseColor.z);
            c[3] = make_float4(specMask, specExp, 0.0f, 0.0f);
            P.ctxNode[k] = node;
            P.kind[node] = NODE_DONE;
        }
        if (want > 0) {
            // exclusive prefix of `want` over the wave from the two ballots
            const unsigned mine = nextStart + childBase + laneRank(m1) + laneRank(m2);
            if (mine + (unsigned)want > (unsigned)P.nodeCap) {
                // no room: report, and fall back to the background so that nothing is read out of bounds.
                // Requests are handed out in increasing order, so every later one fails too and at most one
                // request straddles the arrays' end: the slots below it are the level's last nodes.
                atomicOr(P.ctl + CTL_OVERFLOW, 1u);
                if (mine < (unsigned)P.nodeCap) P.ctl[CTL_LEVEL_FIT + level + 1] = (mine - nextStart) + 1u;
                const V3 c = (want == 1) ? muls(bg, kr) : add(muls(bg, kr), muls(bg, 1.0f - kr));
                P.color[node] = make_float4(c.x, c.y, c.z, 0.0f);
                P.kind[node] = NODE_DONE;
            } else {
                P.child[node] = (int)mine;
                float4* r = P.secRay + 2 * (size_t)(mine - (unsigned)P.slots);
                const V3 reflOrig = invBiasedOrigin(reflDir, hitPoint, n, P.opts.shadowBias);
                r[0] = make_float4(reflOrig.x, reflOrig.y, reflOrig.z, 0.0f);
                r[1] = make_float4(reflDir.x, reflDir.y, reflDir.z, 0.0f);
                if (want == 2) {
                    const V3 refrDir = normalized(refract(rd, n, ior));
                    const V3 refrOrig = invBiasedOrigin(refrDir, hitPoint, n, P.opts.shadowBias);
                    r[2] = make_float4(refrOrig.x, refrOrig.y, refrOrig.z, 0.0f);
                    r[3] = make_float4(refrDir.x, refrDir.y, refrDir.z, 0.0f);
                }
            }
        }
    }
    if (COUNT && P.counters) {
        for (int off = 32; off > 0; off >>= 1) {
            cMesh += __shfl_down(cMesh, off, kWave);
            cSphere += __shfl_down(cSphere, off, kWave);
        }
        if ((threadIdx.x & 63) == 0) {
            if (cMesh) atomicAdd(P.counters + RT_CNT_SHADE_MESH, (unsigned long long)cMesh);
            if (cSphere) atomicAdd(P.counters + RT_CNT_SHADE_SPHERE, (unsigned long long)cSphere);
        }
    }
}

// diffuseAndGlossy (cudaTracer/tracer.cu:169-200, blinnPhong :144-149) for every
// diffuse hit of the level, lights summed in table order.
__global__ void __launch_bounds__(256) accumKernel(const FrameParams P) {
    const unsigned count = P.ctl[CTL_DIFFUSE_CNT];
    const RtLight* lights = reinterpret_cast<const RtLight*>(P.blob + P.offLights);
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const float4* c = P.ctx + 4 * (size_t)k;
        const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
        const V3 hitPoint = mk(c0.x, c0.y, c0.z);
        const V3 n = mk(c0.w, c1.x, c1.y);
        const V3 dir = mk(c1.z, c1.w, c2.x);
        const V3 diffuseColor = mk(c2.y, c2.z, c2.w);
        const float specMask = c3.x, specExp = c3.y;
        V3 diffuseLight = mk(0, 0, 0), specularLight = mk(0, 0, 0);
        for (int i = 0; i < P.lightCnt; ++i) {
            if (!P.vis[(size_t)k * P.lightCnt + i]) continue;
            const RtLight& light = lights[i];
            V3 lightDir = sub(mk(light.position), hitPoint);
            const float distance = magnitude(lightDir);
            lightDir = normalized(lightDir);
            const float lambert = fmaxf(0.f, dot(lightDir, n));
            float specularFac = 0.0f;
            if (lambert > 0.0f) {
                const V3 halfDir = normalized(sub(lightDir, dir));
                const float specAngle = fmaxf(dot(halfDir, n), 0.0f);
                specularFac = powf(specAngle, specExp);
            }
            const V3 lightComponent = divs(muls(mk(light.intensity), light.power), distance);
            diffuseLight = add(diffuseLight, muls(lightComponent, lambert));
            specularLight = add(specularLight, muls(muls(lightComponent, specMask), specularFac));
        }
        const V3 out = add(mul(diffuseLight, diffuseColor), specularLight);
        P.color[P.ctxNode[k]] = make_float4(out.x, out.y, out.z, 0.0f);
    }
}

// Combines mirror / glass nodes of `level` with their finished children
// (cudaTracer/tracer.cu:131,141).
__global__ void __launch_bounds__(256) resolveKernel(const FrameParams P, const int level) {
    const unsigned start = P.ctl[CTL_LEVEL_START + level];
    const unsigned count = P.ctl[CTL_LEVEL_COUNT + level];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const unsigned node = start + i;
        if (node < (unsigned)P.slots && __float_as_int(P.hit[node].w) == -2) continue;
        const int kind = P.kind[node];
        if (kind == NODE_DONE) continue;
        const float kr = P.kr[node];
        const int ch = P.child[node];
        const V3 refl = mk(P.color[ch]);
        V3 out;
        if (kind == NODE_MIRROR) {
            out = muls(refl, kr);
        } else {
            const V3 refr = mk(P.color[ch + 1]);
            out = add(muls(refl, kr), muls(refr, 1.0f - kr));
        }
        P.color[node] = make_float4(out.x, out.y, out.z, 0.0f);
    }
}

// cudaTracer/Main.cu:66-98: sample sum (from 0), average, gamma, store.
__global__ void __launch_bounds__(256) finalizeKernel(const FrameParams P) {
    RT_PROBE_QUAD_REPORT();
    const int S = P.opts.samples;
    const unsigned pixels = (unsigned)(P.slots / S);  // 64 per tile
    for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += gridDim.x * blockDim.x) {
        const unsigned first = ((p >> 6) * (unsigned)S << 6) | (p & 63u);  // the pixel's node of sample 0
        int x, y;
        if (!slotPixel(P, (int)first, x, y)) continue;
        // cudaTracer/Main.cu:66-82: the samples are added up in order, starting from zero, then divided
        // Accumulating over frames (rt_set_accumulation), the sum continues where the pixel's previous call left it, sample
        // by sample in the same order, so that N calls of one sample add up exactly like one call of N samples.  (A new
        // accumulation starts from a cleared surface: sum 0, count 0.)
        V3 col = mk(0.0f, 0.0f, 0.0f);
        float4* const sum = P.accum ? reinterpret_cast<float4*>(P.accum + (size_t)y * P.accumPitch + (size_t)x * 16) : nullptr;
        uint32_t base = 0u;
        if (sum) {
            const float4 sofar = *sum;
            col = mk(sofar);
            base = (uint32_t)sofar.w;
        }
        for (int i = 0; i < S; ++i) col = add(col, mk(P.color[first + ((unsigned)i << 6)]));
        const float samples = (float)(base + (uint32_t)S);
        if (sum) *sum = make_float4(col.x, col.y, col.z, samples);
        col = mk(col.x / samples, col.y / samples, col.z / samples);
        float4 out;
        if (P.opts.gammaCorrection) {
            const float sgInv = 1.0f / 2.2f;
            out = make_float4(powf(col.x, sgInv), powf(col.y, sgInv), powf(col.z, sgInv), 1.0f);
        } else {
            out = make_float4(col.x, col.y, col.z, 1.0f);
        }
        *reinterpret_cast<float4*>(P.surface + (size_t)y * P.pitch + (size_t)x * 16) = out;
    }
}

// Opens `level`: fixes its first node and its node count, and clears the per-level queues.  Level 0 also clears the
// rest of the control block (all but the sticky overflow word).  A level never holds more nodes than have a slot
// AND a ray: requests that passed the arrays' end are dropped here (their parents took the background instead).
__global__ void advanceKernel(const FrameParams P, const int level) {
    if (blockIdx.x != 0) return;
    if (level == 0) {
        for (int w = threadIdx.x; w < CTL_WORDS; w += blockDim.x)
            if (w != CTL_OVERFLOW) P.ctl[w] = 0u;
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    if (level == 0) {
        P.ctl[CTL_LEVEL_COUNT] = (unsigned)P.slots;
    } else {
        const unsigned start = P.ctl[CTL_LEVEL_START + level - 1] + P.ctl[CTL_LEVEL_COUNT + level - 1];
        const unsigned room = start < (unsigned)P.nodeCap ? (unsigned)P.nodeCap - start : 0u;
        unsigned count = P.ctl[CTL_LEVEL_COUNT + level];
        const unsigned fit = P.ctl[CTL_LEVEL_FIT + level];
        if (count > room) count = room;
        if (fit != 0u && count > fit - 1u) count = fit - 1u;
        P.ctl[CTL_LEVEL_START + level] = start;
        P.ctl[CTL_LEVEL_COUNT + level] = count;
    }
    for (int kind = 0; kind < 4; ++kind)
        for (int q = 0; q < RT_QUEUES; ++q) P.ctl[RT_QUEUE_WORD(kind, q)] = 0;
    P.ctl[CTL_DIFFUSE_CNT] = 0;
    P.ctl[CTL_DEFER_CLOSEST] = 0;
    P.ctl[CTL_DEFER_SHADOW] = 0;
}

// Counter launches walk every ray with the literal box test.
constexpr bool kCountLit = true;

// Resident wavefronts per CU of one trace-kernel variant (register and LDS limits).  Asked once per LDS size and kept in a
// small table; contexts of several host threads (one per GPU) share it under a lock.  The figure is a property of the
// code object and the LDS size, the same on every gfx950 device.
template <bool ANY, bool COUNT, bool LIT, bool JIT>
int wavesPerCu(size_t lds) {
    static std::mutex lock;
    static std::map<size_t, int> known;
    std::lock_guard<std::mutex> guard(lock);
    const auto it = known.find(lds);
    if (it != known.end()) return it->second;
    int perCu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, traceKernel<ANY, COUNT, LIT, JIT>, kWave, lds) != hipSuccess || perCu < 1) perCu = 8;
    perCu = perCu > 32 ? 32 : perCu;
    known[lds] = perCu;
    if (getenv("RT_TRACE_DEBUG"))
        fprintf(stderr, "traceKernel<%d,%d,%d,%d>: %d resident wavefronts per CU with %zu B LDS\n", (int)ANY, (int)COUNT, (int)LIT, (int)JIT, perCu, lds);
    return perCu;
}

// The grid is persistent: exactly as many wavefronts as stay resident for this variant, capped by
// p.traceWaves (which also sizes the stack spill buffer).
template <bool ANY, bool COUNT, bool LIT>
void launchOne(const FrameParams& p, int level, int cuCount, hipStream_t s, int rescue) {
    const size_t lds = traceLdsBytes(p.stackLds);
    // shadow rays are rebuilt from their shading contexts: only closest-hit launches have a variant with jitter
    const bool jit = !ANY && (p.opts.samples > 1 || p.accum != nullptr);
    int waves = (jit ? wavesPerCu<ANY, COUNT, LIT, !ANY>(lds) : wavesPerCu<ANY, COUNT, LIT, false>(lds)) * cuCount;
    if (waves > p.traceWaves) waves = p.traceWaves;
    // A rescue launch nearly always finds nothing to do: two wavefronts per CU keep it cheap to start and still walk a
    // frame's worth of marked rays, if it ever comes to that, at a twelfth of the full rate.
    if (rescue && waves > 2 * cuCount) waves = 2 * cuCount;
    if (jit)
        hipLaunchKernelGGL((traceKernel<ANY, COUNT, LIT, !ANY>), dim3(waves), dim3(kWave), lds, s, p, level, rescue);
    else
        hipLaunchKernelGGL((traceKernel<ANY, COUNT, LIT, false>), dim3(waves), dim3(kWave), lds, s, p, level, rescue);
}
template <bool ANY, bool COUNT>
void launchTrace(const FrameParams& p, int level, int cuCount, hipStream_t s) {
    if (COUNT) {
        launchOne<ANY, true, kCountLit>(p, level, cuCount, s, 0);
        return;
    }
    // Two launches pay where the rays are: the primary rays of a large frame and their shadow rays.  A kernel
    // boundary costs 15-25 us of an in-order frame, more than the shorter box test saves on a small frame or on the
    // few rays of a reflection level: those are walked by one launch with the literal test.
    if (level == 0 && p.slots >= p.splitSlots) {
        launchOne<ANY, false, false>(p, level, cuCount, s, 0);
        launchOne<ANY, false, true>(p, level, cuCount, s, 1);  // the rays the first launch left to the literal box test
    } else {
        launchOne<ANY, false, true>(p, level, cuCount, s, 0);
    }
}

}  // namespace

int rtk_trace_waves_per_cu(int stackLds) {
    const size_t lds = traceLdsBytes(stackLds);
    // the largest residency of the variants: each launch takes its own, this one sizes the spill buffer
    int w = wavesPerCu<false, false, false, false>(lds);
    w = std::max(w, wavesPerCu<true, false, false, false>(lds));
    w = std::max(w, wavesPerCu<false, true, kCountLit, false>(lds));
    w = std::max(w, wavesPerCu<true, true, kCountLit, false>(lds));
    w = std::max(w, wavesPerCu<false, false, true, false>(lds));
    w = std::max(w, wavesPerCu<true, false, true, false>(lds));
    w = std::max(w, wavesPerCu<false, false, false, true>(lds));
    w = std::max(w, wavesPerCu<false, true, kCountLit, true>(lds));
    w = std::max(w, wavesPerCu<false, false, true, true>(lds));
    return w;
}

// Workspace of rt_render_noalloc: one array per GPU the library runs on, held until the process ends.  In host
// code the array's name only stands for the symbol; its address on the current device comes from
// hipGetSymbolAddress and is what kernels and memory calls are given.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

int rtk_scratch(void** devPtr, size_t* bytes) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    *devPtr = p;
    *bytes = sizeof(scratch);
    return 0;
}

// Enqueues the kernel sequence of one frame on `stream`.
int rtk_render_frame(const FrameParams* frame, int hasSecondary, int count, int cuCount, void* stream, void** stageEvents) {
    const FrameParams& p = *frame;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t* ev = reinterpret_cast<hipEvent_t*>(stageEvents);
    const int lastLevel = hasSecondary ? p.opts.maxDepth : 0;
    const dim3 wide(cuCount * 8), block(256);
    if (ev) (void)hipEventRecord(ev[0], s);
    for (int level = 0; level <= lastLevel; ++level) {
        hipLaunchKernelGGL(advanceKernel, dim3(1), dim3(64), 0, s, p, level);
        if (count)
            launchTrace<false, true>(p, level, cuCount, s);
        else
            launchTrace<false, false>(p, level, cuCount, s);
        if (ev && level == 0) (void)hipEventRecord(ev[1], s);
        if (count)
            hipLaunchKernelGGL((shadeKernel<true>), wide, block, 0, s, p, level);
        else
            hipLaunchKernelGGL((shadeKernel<false>), wide, block, 0, s, p, level);
        if (ev && level == 0) (void)hipEventRecord(ev[2], s);
        if (p.lightCnt > 0) {
            if (count)
                launchTrace<true, true>(p, level, cuCount, s);
            else
                launchTrace<true, false>(p, level, cuCount, s);
        }
        if (ev && level == 0) (void)hipEventRecord(ev[3], s);
        hipLaunchKernelGGL(accumKernel, wide, block, 0, s, p);
    }
    for (int level = lastLevel - 1; level >= 0; --level) hipLaunchKernelGGL(resolveKernel, wide, block, 0, s, p, level);
    hipLaunchKernelGGL(finalizeKernel, wide, block, 0, s, p);
    if (ev) (void)hipEventRecord(ev[4], s);
    return (int)hipGetLastError();
}
