// rt_kernels.hip - gfx950 render kernel: one 64-lane wavefront per 8x8-pixel
// tile, persistent wavefronts pulling tiles from an atomic queue, two-level BVH
// traversal with a per-lane stack in LDS, shading and shadow / bounce rays in
// the same launch.
//
// Arithmetic contract: every fp32 operation below is written in the operation
// order of the reference's device code (cited per function) and this file is
// compiled with -ffp-contract=off, IEEE division and square root, denormals on,
// so hit decisions are bit-identical to a host evaluation of the same formulas.
// Only powf (gamma, Blinn-Phong) may differ in the last ulp.
//
// Data layout (rt_device_format.h): 32-byte nodes fetched as two 16-byte loads,
// 48-byte triangle records {v0, e1, e2, indices} fetched as three, 64-byte
// shapes, all inside one blob addressed by offsets.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../rt_device_format.h"

namespace {

constexpr int kWave = 64;
constexpr int kMaxFrames = RT_MAX_DEPTH + 1;

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};

__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
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

struct Counters {
    unsigned int c[RT_COUNTER_CNT];
};

struct Hit {
    int shape;  // slot in the shape table
    int tri;    // triangle record of the last accepted triangle
    float u, v;
};

// ---------------------------------------------------------------------------
// cudaTracer/BoundingBox.cuh:115-139.  n0 = {min.x, min.y, min.z, max.x},
// n1 = {max.y, max.z, offset, meta}.  1 + 2*gamma(3) == 1 + 3*2^-23 in fp32.
__device__ __forceinline__ bool boxHit(const float4 n0, const float4 n1, const V3 o, const V3 inv, const bool nx,
                                       const bool ny, const bool nz, const float rayTMax) {
    constexpr float kPad = 0x1.000006p+0f;
    float tMin = ((nx ? n0.w : n0.x) - o.x) * inv.x;
    float tMax = ((nx ? n0.x : n0.w) - o.x) * inv.x;
    const float tyMin = ((ny ? n1.x : n0.y) - o.y) * inv.y;
    float tyMax = ((ny ? n0.y : n1.x) - o.y) * inv.y;
    tMax *= kPad;
    tyMax *= kPad;
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    const float tzMin = ((nz ? n1.y : n0.z) - o.z) * inv.z;
    const float tzMax = ((nz ? n0.z : n1.y) - o.z) * inv.z;
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    return (tMin < rayTMax) && (tMax > 0);
}

// cudaTracer/Mesh.cuh:182-212 on a precomputed {v0, e1, e2} record.
__device__ __forceinline__ bool triangleHit(const float4 r0, const float4 r1, const float e2z, const V3 o, const V3 d,
                                            float& tMax, float& uOut, float& vOut) {
    const V3 v0 = mk(r0.x, r0.y, r0.z);
    const V3 e1 = mk(r0.w, r1.x, r1.y);
    const V3 e2 = mk(r1.z, r1.w, e2z);
    const V3 pvec = cross(d, e2);
    const float det = dot(e1, pvec);
    if (det == 0.0f || det < 0.0f) return false;
    const V3 tvec = sub(o, v0);
    const float u = dot(tvec, pvec);
    if (u < 0.0f || u > det) return false;
    const V3 qvec = cross(tvec, e1);
    const float v = dot(d, qvec);
    if (v < 0.0f || u + v > det) return false;
    const float invDet = 1.0f / det;
    const float t = dot(e2, qvec) * invDet;
    if (t < 0.0f || tMax < t) return false;
    tMax = t;
    uOut = u * invDet;
    vOut = v * invDet;
    return true;
}

// cudaTracer/Sphere.cuh:9-54
__device__ __forceinline__ bool sphereHit(const DevShape& s, const V3 o, const V3 d, float& tMax) {
    const V3 L = sub(o, mk(s.center[0], s.center[1], s.center[2]));
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

// ---------------------------------------------------------------------------
// Mesh BVH walk (cudaTracer/Mesh.cuh:115-179).  `stk` is this lane's column of
// the LDS stack, already advanced past the scene-level entries.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool meshHit(const RenderParams& P, const DevMesh* __restrict__ m, const V3 wo, const V3 wd,
                                        float& tMax, Hit& h, int* stk, Counters& cnt) {
    const float4* M = reinterpret_cast<const float4*>(m->inv);
    const float4 m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3];
    // cudaTracer/Mat.cuh:127-149: row vector * matrix; the point is divided by w,
    // the direction is neither translated nor renormalised, so t is shared.
    const float pa = ((wo.x * m0.x + wo.y * m1.x) + wo.z * m2.x) + m3.x;
    const float pb = ((wo.x * m0.y + wo.y * m1.y) + wo.z * m2.y) + m3.y;
    const float pc = ((wo.x * m0.z + wo.y * m1.z) + wo.z * m2.z) + m3.z;
    const float pw = ((wo.x * m0.w + wo.y * m1.w) + wo.z * m2.w) + m3.w;
    const V3 o = mk(pa / pw, pb / pw, pc / pw);
    const V3 d = mk((wd.x * m0.x + wd.y * m1.x) + wd.z * m2.x, (wd.x * m0.y + wd.y * m1.y) + wd.z * m2.y,
                    (wd.x * m0.z + wd.y * m1.z) + wd.z * m2.z);
    if (COUNT) cnt.c[RT_CNT_MESH_ENTRY]++;
    const float4* tris = reinterpret_cast<const float4*>(P.blob + m->offTriRecords);
    bool hit = false;
    float t = tMax;
    if (P.opts.useMeshBvhs) {
        const float4* nodes = reinterpret_cast<const float4*>(P.blob + m->offNodes);
        const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
        int sp = 0;
        int cur = 0;
        while (true) {
            const float4 n0 = nodes[2 * cur];
            const float4 n1 = nodes[2 * cur + 1];
            if (COUNT) cnt.c[RT_CNT_NODE]++;
            if (boxHit(n0, n1, o, inv, nx, ny, nz, t)) {
                const unsigned meta = __float_as_uint(n1.w);
                const int off = __float_as_int(n1.z);
                const int primCnt = (int)(meta & 0xffffu);
                if (primCnt > 0) {
                    for (int i = 0; i < primCnt; ++i) {
                        const float4* r = tris + 3 * (size_t)(off + i);
                        const float4 r0 = r[0];
                        const float4 r1 = r[1];
                        const float e2z = reinterpret_cast<const float*>(r + 2)[0];
                        if (COUNT) cnt.c[RT_CNT_TRI]++;
                        if (triangleHit(r0, r1, e2z, o, d, t, h.u, h.v)) {
                            h.tri = off + i;
                            hit = true;
                            if (ANY) {
                                tMax = t;
                                return true;
                            }
                        }
                    }
                    if (sp == 0) break;
                    cur = stk[(--sp) * kWave];
                } else {
                    const unsigned axis = (meta >> 16) & 0xffu;
                    const bool negAxis = axis == 0 ? nx : (axis == 1 ? ny : nz);
                    if (negAxis) {
                        stk[sp * kWave] = cur + 1;
                        cur = off;
                    } else {
                        stk[sp * kWave] = off;
                        cur = cur + 1;
                    }
                    ++sp;
                }
            } else {
                if (sp == 0) break;
                cur = stk[(--sp) * kWave];
            }
        }
    } else {
        for (int i = 0; i < m->triangleCnt; ++i) {
            const float4* r = tris + 3 * (size_t)i;
            if (COUNT) cnt.c[RT_CNT_TRI]++;
            if (triangleHit(r[0], r[1], reinterpret_cast<const float*>(r + 2)[0], o, d, t, h.u, h.v)) {
                h.tri = i;
                hit = true;
                if (ANY) break;
            }
        }
    }
    tMax = t;
    return hit;
}

// cudaTracer/Shape.cuh:81-87
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool shapeHit(const RenderParams& P, const DevShape* __restrict__ shapes, const int slot,
                                         const V3 o, const V3 d, float& tMax, Hit& h, int* stk, Counters& cnt) {
    const DevShape& s = shapes[slot];
    if (s.kind == RT_SHAPE_SPHERE) {
        if (COUNT) cnt.c[RT_CNT_SPHERE]++;
        return sphereHit(s, o, d, tMax);
    }
    const DevMesh* m = reinterpret_cast<const DevMesh*>(P.blob + P.offMeshes) + s.meshIdx;
    return meshHit<ANY, COUNT>(P, m, o, d, tMax, h, stk, cnt);
}

// Scene-level walk (cudaTracer/Scene.cuh:21-87).  With ANY the walk ends at the
// first accepted primitive: the boolean is the same as the full walk's, because
// acceptance of that primitive does not depend on what is tested after it.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool sceneHit(const RenderParams& P, const V3 o, const V3 d, float& tMax, Hit& h, int* stk,
                                         Counters& cnt) {
    const DevShape* shapes = reinterpret_cast<const DevShape*>(P.blob + P.offShapes);
    bool hit = false;
    if (P.opts.useSceneBvh) {
        const float4* nodes = reinterpret_cast<const float4*>(P.blob + P.offSceneNodes);
        const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        const bool nx = inv.x < 0, ny = inv.y < 0, nz = inv.z < 0;
        int sp = 0;
        int cur = 0;
        while (true) {
            const float4 n0 = nodes[2 * cur];
            const float4 n1 = nodes[2 * cur + 1];
            if (COUNT) cnt.c[RT_CNT_NODE]++;
            if (boxHit(n0, n1, o, inv, nx, ny, nz, tMax)) {
                const unsigned meta = __float_as_uint(n1.w);
                const int off = __float_as_int(n1.z);
                const int primCnt = (int)(meta & 0xffffu);
                if (primCnt > 0) {
                    for (int i = 0; i < primCnt; ++i) {
                        if (shapeHit<ANY, COUNT>(P, shapes, off + i, o, d, tMax, h, stk + sp * kWave, cnt)) {
                            h.shape = off + i;
                            hit = true;
                            if (ANY) return true;
                        }
                    }
                    if (sp == 0) break;
                    cur = stk[(--sp) * kWave];
                } else {
                    const unsigned axis = (meta >> 16) & 0xffu;
                    const bool negAxis = axis == 0 ? nx : (axis == 1 ? ny : nz);
                    if (negAxis) {
                        stk[sp * kWave] = cur + 1;
                        cur = off;
                    } else {
                        stk[sp * kWave] = off;
                        cur = cur + 1;
                    }
                    ++sp;
                }
            } else {
                if (sp == 0) break;
                cur = stk[(--sp) * kWave];
            }
        }
    } else {
        for (int i = 0; i < P.shapeCnt; ++i) {
            if (shapeHit<ANY, COUNT>(P, shapes, i, o, d, tMax, h, stk, cnt)) {
                h.shape = i;
                hit = true;
                if (ANY) break;
            }
        }
    }
    return hit;
}

// ---------------------------------------------------------------------------
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
__device__ __forceinline__ V3 meshTex(const RenderParams& P, const DevMesh* m, int which, V2 st) {
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
// cudaTracer/tracer.cu:202-215
__device__ __forceinline__ V3 biasedOrigin(V3 dir, V3 P, V3 n, float bias) {
    return dot(dir, n) < 0.0f ? add(P, muls(n, bias)) : sub(P, muls(n, bias));
}
__device__ __forceinline__ V3 invBiasedOrigin(V3 dir, V3 P, V3 n, float bias) {
    return dot(dir, n) < 0.0f ? sub(P, muls(n, bias)) : add(P, muls(n, bias));
}

struct Surface {
    V3 hitPoint;
    V3 n;
    V2 st;
};

// cudaTracer/Mesh.cuh:101-109
__device__ __forceinline__ float lerp3(float a, float b, float c, float w, float u, float v) { return (a * w + b * u) + c * v; }

// st + normal of a mesh hit (cudaTracer/Mesh.cuh:65-95); the normal stays in object space.
__device__ __forceinline__ void meshSurface(const RenderParams& P, const DevMesh* m, const Hit& h, Surface& s) {
    const TriRecord* rec = reinterpret_cast<const TriRecord*>(P.blob + m->offTriRecords) + h.tri;
    const RtVertex* verts = reinterpret_cast<const RtVertex*>(P.blob + m->offVertices);
    const RtVertex* a = verts + rec->idx[0];
    const RtVertex* b = verts + rec->idx[1];
    const RtVertex* c = verts + rec->idx[2];
    const float w = (1.0f - h.u) - h.v;
    s.st.x = lerp3(a->texCoord[0], b->texCoord[0], c->texCoord[0], w, h.u, h.v);
    s.st.y = lerp3(a->texCoord[1], b->texCoord[1], c->texCoord[1], w, h.u, h.v);
    V3 t, bt, n;
    t.x = lerp3(a->tangent[0], b->tangent[0], c->tangent[0], w, h.u, h.v);
    t.y = lerp3(a->tangent[1], b->tangent[1], c->tangent[1], w, h.u, h.v);
    t.z = lerp3(a->tangent[2], b->tangent[2], c->tangent[2], w, h.u, h.v);
    bt.x = lerp3(a->bitangent[0], b->bitangent[0], c->bitangent[0], w, h.u, h.v);
    bt.y = lerp3(a->bitangent[1], b->bitangent[1], c->bitangent[1], w, h.u, h.v);
    bt.z = lerp3(a->bitangent[2], b->bitangent[2], c->bitangent[2], w, h.u, h.v);
    n.x = lerp3(a->normal[0], b->normal[0], c->normal[0], w, h.u, h.v);
    n.y = lerp3(a->normal[1], b->normal[1], c->normal[1], w, h.u, h.v);
    n.z = lerp3(a->normal[2], b->normal[2], c->normal[2], w, h.u, h.v);
    V3 ns = meshTex(P, m, 1, s.st);
    ns = mk(ns.x * 2.0f - 1.0f, ns.y * 2.0f - 1.0f, ns.z * 2.0f - 1.0f);
    s.n = normalized(add(add(n, muls(t, ns.x)), muls(bt, ns.y)));
}

// cudaTracer/tracer.cu:169-200 (+ blinnPhong :144-149)
template <bool COUNT>
__device__ __forceinline__ V3 diffuseAndGlossy(const RenderParams& P, const DevShape& shape, const DevMesh* m, const V3 dir,
                                               const Surface& s, int* stk, Counters& cnt) {
    const bool isMesh = shape.kind == RT_SHAPE_MESH;
    const V3 diffuseColor =
        isMesh ? meshTex(P, m, 0, s.st) : mk(shape.diffuseColor[0], shape.diffuseColor[1], shape.diffuseColor[2]);
    V3 diffuseLight = mk(0, 0, 0), specularLight = mk(0, 0, 0);
    const V3 shadowOrig = biasedOrigin(dir, s.hitPoint, s.n, P.opts.shadowBias);
    const RtLight* lights = reinterpret_cast<const RtLight*>(P.blob + P.offLights);
    float specMask = 0.0f;
    bool haveMask = false;
    for (int i = 0; i < P.lightCnt; ++i) {
        const RtLight& light = lights[i];
        V3 lightDir = sub(mk(light.position[0], light.position[1], light.position[2]), s.hitPoint);
        const float distance = magnitude(lightDir);
        lightDir = normalized(lightDir);
        const float lambert = fmaxf(0.f, dot(lightDir, s.n));
        float tMax = distance;
        Hit scratch;
        if (COUNT) cnt.c[RT_CNT_SHADOW]++;
        const bool visible = !sceneHit<true, COUNT>(P, shadowOrig, lightDir, tMax, scratch, stk, cnt);
        if (visible) {
            float specularFac = 0.0f;
            if (lambert > 0.0f) {
                const V3 halfDir = normalized(sub(lightDir, dir));
                const float specAngle = fmaxf(dot(halfDir, s.n), 0.0f);
                specularFac = powf(specAngle, shape.specularExponent);
            }
            if (!haveMask) {  // pure function of st: sampled once, on first use
                specMask = isMesh ? meshTex(P, m, 2, s.st).x : shape.Ks;
                haveMask = true;
            }
            const V3 lightComponent =
                divs(muls(mk(light.intensity[0], light.intensity[1], light.intensity[2]), light.power), distance);
            diffuseLight = add(diffuseLight, muls(lightComponent, lambert));
            specularLight = add(specularLight, muls(muls(lightComponent, specMask), specularFac));
        }
    }
    return add(mul(diffuseLight, diffuseColor), specularLight);
}

// One pending level of the reference's recursion (cudaTracer/tracer.cu:117-142).
struct Frame {
    V3 ro, rd;  // refraction ray still to trace (glass)
    V3 refl;    // colour returned by the reflection child (glass)
    float kr;
    int state;
};
enum { FRAME_MIRROR = 0, FRAME_GLASS_REFLECTING = 1, FRAME_GLASS_REFRACTING = 2 };

// castRay (cudaTracer/tracer.cu:15-48) with the recursion unrolled into an
// explicit post-order walk: children are evaluated in the reference's order
// (reflection first) and combined with the same expressions.
template <bool SECONDARY, bool COUNT>
__device__ __forceinline__ V3 castRay(const RenderParams& P, V3 ro, V3 rd, int* stk, Counters& cnt, int* key) {
    const V3 bg = mk(P.opts.backgroundColor[0], P.opts.backgroundColor[1], P.opts.backgroundColor[2]);
    const DevShape* shapes = reinterpret_cast<const DevShape*>(P.blob + P.offShapes);
    Frame frames[SECONDARY ? kMaxFrames : 1];
    int depth = 0;
    V3 ret = bg;
    while (true) {
        // ---- "call": trace (ro, rd) at `depth`
        bool spawned = false;
        if (depth > P.opts.maxDepth) {
            ret = bg;
        } else {
            if (COUNT && depth > 0) cnt.c[RT_CNT_SECONDARY]++;
            float tMax = 3.402823466e+38f;
            Hit h;
            h.shape = -1;
            h.tri = -1;
            h.u = h.v = 0.0f;
            if (!sceneHit<false, COUNT>(P, ro, rd, tMax, h, stk, cnt)) {
                ret = bg;
            } else {
                const DevShape& shape = shapes[h.shape];
                const DevMesh* m = reinterpret_cast<const DevMesh*>(P.blob + P.offMeshes) + (shape.kind == RT_SHAPE_MESH ? shape.meshIdx : 0);
                Surface s;
                s.hitPoint = add(ro, muls(rd, tMax));
                if (shape.kind == RT_SHAPE_SPHERE) {
                    s.st = V2{1.0f, 1.0f};
                    s.n = normalized(sub(s.hitPoint, mk(shape.center[0], shape.center[1], shape.center[2])));
                    if (COUNT) cnt.c[RT_CNT_SHADE_SPHERE]++;
                } else {
                    meshSurface(P, m, h, s);
                    if (COUNT) cnt.c[RT_CNT_SHADE_MESH]++;
                }
                if (key && depth == 0) {
                    key[0] = shape.tag;
                    if (shape.kind == RT_SHAPE_MESH) {
                        const TriRecord* rec = reinterpret_cast<const TriRecord*>(P.blob + m->offTriRecords) + h.tri;
                        key[1] = rec->idx[0];
                        key[2] = rec->idx[1];
                        key[3] = rec->idx[2];
                    }
                }
                const int mt = shape.materialType;
                if (mt == RT_MAT_DIFFUSE_AND_GLOSSY) {
                    ret = diffuseAndGlossy<COUNT>(P, shape, m, rd, s, stk, cnt);
                } else if (SECONDARY && mt == RT_MAT_REFLECTION) {
                    const V3 reflDir = normalized(reflect(rd, s.n));
                    Frame& f = frames[depth];
                    f.state = FRAME_MIRROR;
                    f.kr = fresnel(rd, s.n, shape.ior);
                    ro = invBiasedOrigin(reflDir, s.hitPoint, s.n, P.opts.shadowBias);
                    rd = reflDir;
                    spawned = true;
                } else if (SECONDARY && mt == RT_MAT_REFLECTION_AND_REFRACTION) {
                    // the reference reflects the NORMAL about the view direction here
                    const V3 reflDir = normalized(reflect(s.n, rd));
                    const V3 refrDir = normalized(refract(rd, s.n, shape.ior));
                    Frame& f = frames[depth];
                    f.state = FRAME_GLASS_REFLECTING;
                    f.kr = fresnel(rd, s.n, shape.ior);
                    f.ro = invBiasedOrigin(refrDir, s.hitPoint, s.n, P.opts.shadowBias);
                    f.rd = refrDir;
                    ro = invBiasedOrigin(reflDir, s.hitPoint, s.n, P.opts.shadowBias);
                    rd = reflDir;
                    spawned = true;
                } else {
                    ret = mk(1.0f, 0.5f, 0.25f);
                }
            }
        }
        if (SECONDARY && spawned) {
            ++depth;
            continue;
        }
        // ---- "return": hand `ret` to the pending parents
        if (!SECONDARY) break;
        bool resume = false;
        while (depth > 0) {
            Frame& f = frames[depth - 1];
            if (f.state == FRAME_MIRROR) {
                ret = muls(ret, f.kr);
                --depth;
            } else if (f.state == FRAME_GLASS_REFLECTING) {
                f.refl = ret;
                f.state = FRAME_GLASS_REFRACTING;
                ro = f.ro;
                rd = f.rd;
                resume = true;  // trace the refraction child at the same depth
                break;
            } else {
                ret = add(muls(f.refl, f.kr), muls(ret, 1.0f - f.kr));
                --depth;
            }
        }
        if (!resume) break;
    }
    return ret;
}

// cudaTracer/Main.cu:43-98 for samples == 1 (both jitters are exactly 0.0f)
template <bool SECONDARY, bool COUNT>
__device__ __forceinline__ void renderPixel(const RenderParams& P, int x, int y, int* stk, Counters& cnt) {
    const float widthScale = 1.0f / (float)P.opts.width;
    const float heightScale = 1.0f / (float)P.opts.height;
    const float ndcX = (2.0f * ((float)x + 0.5f) * widthScale - 1.0f) * P.opts.imageAspectRatio * P.opts.scale;
    const float ndcY = (1.0f - 2.0f * ((float)y + 0.5f) * heightScale) * P.opts.scale;
    V3 dir = normalized(mk(ndcX + 0.0f * widthScale, ndcY + 0.0f * heightScale, 1.0f));
    const float* C = P.camera;
    dir = mk((dir.x * C[0] + dir.y * C[4]) + dir.z * C[8], (dir.x * C[1] + dir.y * C[5]) + dir.z * C[9],
             (dir.x * C[2] + dir.y * C[6]) + dir.z * C[10]);
    dir = normalized(dir);
    int key[4] = {-1, -1, -1, -1};
    if (COUNT) cnt.c[RT_CNT_PRIMARY]++;
    V3 col = castRay<SECONDARY, COUNT>(P, mk(P.orig[0], P.orig[1], P.orig[2]), dir, stk, cnt, P.hitKeys ? key : nullptr);
    col = add(mk(0.0f, 0.0f, 0.0f), col);
    const float samples = (float)P.opts.samples;
    col = mk(col.x / samples, col.y / samples, col.z / samples);
    float4 out;
    if (P.opts.gammaCorrection) {
        const float sgInv = 1.0f / 2.2f;
        out = make_float4(powf(col.x, sgInv), powf(col.y, sgInv), powf(col.z, sgInv), 1.0f);
    } else {
        out = make_float4(col.x, col.y, col.z, 1.0f);
    }
    *reinterpret_cast<float4*>(P.surface + (size_t)y * P.pitch + (size_t)x * 16) = out;
    if (P.hitKeys) {
        *reinterpret_cast<int4*>(P.hitKeys + ((size_t)y * P.opts.width + x) * 4) = make_int4(key[0], key[1], key[2], key[3]);
    }
}

template <bool SECONDARY, bool COUNT>
__global__ void __launch_bounds__(kWave) renderKernel(const RenderParams P) {
    __shared__ int s_stack[RT_STACK_ENTRIES * kWave];
    const int lane = threadIdx.x;
    int* stk = s_stack + lane;
    Counters cnt;
    if (COUNT) {
#pragma unroll
        for (int i = 0; i < RT_COUNTER_CNT; ++i) cnt.c[i] = 0;
    }
    while (true) {
        // one atomic per wavefront and tile; every wave ends when the queue is drained
        unsigned int k = 0;
        if (lane == 0) k = atomicAdd(P.tileQueue, 1u);
        k = __builtin_amdgcn_readfirstlane(k);
        if (k >= (unsigned)P.tileCnt) break;
        const int tile = P.tileFirst + (int)k * P.tileStride;
        const int tx = tile % P.tilesX;
        const int ty = tile / P.tilesX;
        const int x = tx * RT_TILE + (lane & (RT_TILE - 1));
        const int y = ty * RT_TILE + (lane >> 3);
        if (x < P.opts.width && y < P.opts.height) renderPixel<SECONDARY, COUNT>(P, x, y, stk, cnt);
    }
    if (COUNT && P.counters) {
#pragma unroll
        for (int i = 0; i < RT_COUNTER_CNT; ++i) {
            unsigned int v = cnt.c[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
            if (lane == 0 && v) atomicAdd(P.counters + i, (unsigned long long)v);
        }
    }
}

template <bool S, bool C>
int launch(const RenderParams* p, int gridBlocks, hipStream_t stream) {
    hipLaunchKernelGGL((renderKernel<S, C>), dim3(gridBlocks), dim3(kWave), 0, stream, *p);
    return (int)hipGetLastError();
}

}  // namespace

int rtk_launch_render(const RenderParams* p, int secondary, int count, int gridBlocks, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (secondary) return count ? launch<true, true>(p, gridBlocks, s) : launch<true, false>(p, gridBlocks, s);
    return count ? launch<false, true>(p, gridBlocks, s) : launch<false, false>(p, gridBlocks, s);
}

int rtk_persistent_blocks(int secondary, int count, int cuCount) {
    int perCu = 0;
    hipError_t e;
    if (secondary)
        e = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, renderKernel<true, true>, kWave, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, renderKernel<true, false>, kWave, 0);
    else
        e = count ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, renderKernel<false, true>, kWave, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, renderKernel<false, false>, kWave, 0);
    if (e != hipSuccess || perCu < 1) perCu = 8;
    if (perCu > 32) perCu = 32;
    return perCu * cuCount;
}
