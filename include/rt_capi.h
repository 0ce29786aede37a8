/*
 * rt_capi.h - C-ABI of the gfx950 ray-trace library (librt_gfx950.so).
 *
 * This is the drop-in boundary for the per-frame render call of the reference
 * renderer.  The reference exposes one free function per frame plus two RNG
 * helpers (cudaTracer/main.cuh:7-12):
 *
 *     void cudamain(const DeviceOpts&, const Scene&, const void* surface,
 *                   size_t pitch, int blockDim, unsigned char* rngStates);
 *     unsigned char* cu_initCurand(int w, int h);  void cu_cleanCurand(...);
 *
 * where `Scene` carries raw device pointers that the reference's
 * GlobalCudaVector members own (cudaTracer/Scene.cuh:12-19,
 * cudaTracer/GlobalCudaVector.h:42-59).  Here the library owns all device
 * memory behind opaque handles; the caller hands over host arrays laid out
 * exactly like the reference's device structs (same field order and sizes),
 * and no torch / C++ types cross the boundary.
 *
 * All entry points return RT_OK (0) or a negative error code and never call
 * exit(); the message of the last failure on a context is available from
 * rt_last_error().  (The reference prints and exits: cudaTracer/cudaUtils.h:13-21.)
 */
#ifndef RT_CAPI_H
#define RT_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)  /* bad argument / malformed scene            */
#define RT_ERR_DEVICE (-2)   /* a HIP call failed                         */
#define RT_ERR_LIMIT (-3)    /* scene exceeds a compiled-in kernel limit  */

/* Deepest recursion level the kernel's bounce stack holds (levels 0..maxDepth
 * are traced, cudaTracer/tracer.cu:16). */
#define RT_MAX_DEPTH 15
/* Largest per-ray traversal stack (scene BVH + mesh BVH share one stack; the
 * launch sizes it to scene depth + deepest mesh depth + 1).  The reference
 * reserves int[64] per level (cudaTracer/Scene.cuh:36, cudaTracer/Mesh.cuh:135);
 * a median-split tree over 10 M triangles is 21 levels deep.                 */
#define RT_STACK_ENTRIES 64

/* ---- plain-data mirrors of the reference's device structs --------------- */

/* cudaTracer/Vertex.cuh:6-13 (56 bytes) */
typedef struct RtVertex {
    float normal[3];
    float tangent[3];
    float bitangent[3];
    float texCoord[2];
    float position[3];
} RtVertex;

/* cudaTracer/Triangle.cuh:6-7 (12 bytes) */
typedef struct RtTriangle {
    int32_t i[3];
} RtTriangle;

/* cudaTracer/LinearNode.cuh:4-13 (32 bytes).  `offset` is primitivesOffset
 * for a leaf and secondChildOffset for an interior node; primCnt == 0 marks an
 * interior node. */
typedef struct RtLinearNode {
    float bbmin[3];
    float bbmax[3];
    int32_t offset;
    uint16_t primCnt;
    uint8_t axis;
    uint8_t pad;
} RtLinearNode;

/* cudaTracer/Texture.cuh:6-11.  `data` is RGBA8, row-major, width*height*4
 * bytes of host memory. */
enum { RT_TEX_WRAP = 0, RT_TEX_CLAMP = 1 };
typedef struct RtTexture {
    int32_t width;
    int32_t height;
    const unsigned char* data;
    int32_t mode;
} RtTexture;

/* cudaTracer/Material.cuh:6-17 (32 bytes) */
enum { RT_MAT_DIFFUSE_AND_GLOSSY = 0, RT_MAT_REFLECTION_AND_REFRACTION = 1, RT_MAT_REFLECTION = 2 };
typedef struct RtMaterial {
    int32_t materialType;
    float ior;
    float Kd, Ks;
    float diffuseColor[3];
    float specularExponent;
} RtMaterial;

/* cudaTracer/Light.cuh:6-17 (28 bytes) */
typedef struct RtLight {
    float position[3];
    float intensity[3];
    float power;
} RtLight;

/* cudaTracer/Mesh.cuh:17-26.  `triangles` are in mesh-BVH leaf order
 * (cudaTracer/CudaMesh.h:60-63); `transformInverse` is row-major, row-vector
 * convention (cudaTracer/Mat.cuh:127-149).  `nodes` is the tree as
 * cudaTracer/BVH.h:152-180 flattens it: `nodeCnt` entries, all of them reached
 * from entry 0 (an array with stray entries is rejected), and an interior
 * entry's box holds the boxes of its two children, as the union that builder
 * forms does (the trace kernels test grandchildren without their parents). */
typedef struct RtMesh {
    int32_t triangleCnt;
    const RtTriangle* triangles;
    int32_t vertexCnt;
    const RtVertex* vertices;
    RtTexture diffuse;
    RtTexture normals;
    RtTexture specular;
    int32_t nodeCnt;
    const RtLinearNode* nodes;
    float transformInverse[16];
} RtMesh;

/* cudaTracer/Shape.cuh:8-16.  The reference embeds the Mesh by value in a
 * union; here a mesh shape names its mesh by index into RtSceneDesc.meshes.
 * `tag` is the shape's index in creation order (before the scene-BVH reorder,
 * cudaTracer/RayTracer.cpp:72-75); it is what hit keys report. */
enum { RT_SHAPE_SPHERE = 0, RT_SHAPE_MESH = 1 };
typedef struct RtShape {
    int32_t kind;
    float center[3]; /* sphere (cudaTracer/Sphere.cuh:30-32) */
    float radius, radius2;
    int32_t meshIdx; /* mesh */
    RtMaterial material;
    float bbmin[3], bbmax[3];
    int32_t tag;
} RtShape;

/* cudaTracer/Scene.cuh:12-19 without the camera (passed per frame).  `shapes`
 * are in scene-BVH leaf order; `nodes` is the scene BVH. */
typedef struct RtSceneDesc {
    const RtLight* lights;
    int32_t lightCnt;
    const RtShape* shapes;
    int32_t shapeCnt;
    const RtMesh* meshes;
    int32_t meshCnt;
    const RtLinearNode* nodes;
    int32_t nodeCnt;
} RtSceneDesc;

/* cudaTracer/Options.cuh:23-36, field for field (48 bytes). */
typedef struct RtDeviceOpts {
    float imageAspectRatio;
    float scale;
    int32_t width;
    int32_t height;
    float backgroundColor[3];
    float shadowBias;
    int32_t blockSize;
    int32_t maxDepth;
    int32_t samples;
    uint8_t gammaCorrection;
    uint8_t useSceneBvh;
    uint8_t useMeshBvhs;
    uint8_t pad_;
} RtDeviceOpts;

/* Sub-pixel jitter of sample `sample` (0-based) of pixel `pixel` (= y * width + x) along `axis` (0 = x, 1 = y),
 * in [-0.5, 0.5); used when RtDeviceOpts.samples > 1.  The reference draws its jitter from curand states that are
 * shared between thread blocks and divides in integers (cudaTracer/Main.cu:60-61,68-69,108-110), so its values are
 * neither reproducible nor in [-0.5, 0.5); this counter-based hash replaces it.  `seed` is rt_set_sample_seed's. */
#if defined(__HIPCC__)
#define RT_HOST_DEVICE __host__ __device__
#else
#define RT_HOST_DEVICE
#endif
static inline RT_HOST_DEVICE uint32_t rt_hash32(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
static inline RT_HOST_DEVICE float rt_sample_jitter(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t axis) {
    const uint32_t h = rt_hash32(rt_hash32(pixel ^ seed) + (sample * 2u + axis) * 0x9E3779B9u);
    return (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

/* Per-frame camera: Scene::orig and Scene::camera (cudaTracer/Scene.cuh:17-18),
 * filled by the caller as in cudaTracer/RayTracer.cpp:201-202. */
typedef struct RtCamera {
    float orig[3];
    float camera[16]; /* inverse view matrix, row-major */
} RtCamera;

/* Optional per-frame outputs and the screen-tile shard of this call.
 * All pointers are DEVICE pointers (or NULL).
 *   hitKeys  : width*height * 4 int32 {shape tag, i0, i1, i2} of the primary
 *              hit (sphere: i = -1; miss: all -1).
 *   counters : RT_COUNTER_CNT uint64 totals, ADDED to by the call.
 *   tileFirst/tileStride : the call renders 8x8-pixel tiles t = tileFirst,
 *              tileFirst+tileStride, ... in row-major tile order; 0/1 = whole
 *              frame.  Pixels of other tiles are left untouched.            */
enum {
    RT_CNT_PRIMARY = 0,   /* primary rays                                    */
    RT_CNT_SHADOW = 1,    /* shadow rays (diffuse hits x lights, any depth)  */
    RT_CNT_SECONDARY = 2, /* reflection + refraction rays                    */
    RT_CNT_NODE = 3,      /* BVH node box tests (both levels)                */
    RT_CNT_TRI = 4,       /* ray/triangle tests                              */
    RT_CNT_SPHERE = 5,    /* ray/sphere tests                                */
    RT_CNT_MESH_ENTRY = 6,/* ray -> object-space transforms                  */
    RT_CNT_SHADE_MESH = 7,/* shaded mesh hits                                */
    RT_CNT_SHADE_SPHERE = 8,
    /* the share of NODE / TRI / SPHERE / MESH_ENTRY spent on shadow rays (four words, same order) */
    RT_CNT_SHADOW_NODE = 9,
    RT_CNT_SHADOW_TRI = 10,
    RT_CNT_SHADOW_SPHERE = 11,
    RT_CNT_SHADOW_MESH_ENTRY = 12,
    /* wave-step statistics of the trace kernels (diagnostics; six words each):
     * iterations, node steps, lanes in node steps, triangle phases, rays in triangle phases, lanes holding a ray */
    RT_CNT_STAT_CLOSEST = 16,
    RT_CNT_STAT_SHADOW = 22,
    /* iterations after the work queue ran dry, lanes holding a ray in them (two words each) */
    RT_CNT_STAT_TAIL_CLOSEST = 28,
    RT_CNT_STAT_TAIL_SHADOW = 30,
    RT_COUNTER_CNT = 32
};
typedef struct RtFrameAux {
    int32_t* hitKeys;
    uint64_t* counters;
    int32_t tileFirst;
    int32_t tileStride;
} RtFrameAux;

typedef struct rt_ctx rt_ctx;
typedef struct rt_scene rt_scene;

/* ---- context -------------------------------------------------------------
 * One context per GPU and per host thread (rt_multi, at the end of this file, drives several from one thread).  Replaces the implicit CUDA
 * context + `__device__` globals of cudaTracer/Main.cu:11-13.  `device` is the
 * HIP device ordinal. */
rt_ctx* rt_create(int device);
void rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx); /* ctx == NULL: error of the last failed rt_create */
/* Launch on a caller-owned hipStream_t instead of the context's own stream
 * (NULL restores the own stream). */
int rt_set_stream(rt_ctx* ctx, void* hipStream);
/* Blocks until everything the context enqueued has finished
 * (cudaDeviceSynchronize at cudaTracer/Main.cu:40). */
int rt_sync(rt_ctx* ctx);

/* ---- scene upload --------------------------------------------------------
 * rt_scene_upload replaces the GlobalCudaVector::copyToDevice calls that fill
 * Scene's device pointers (cudaTracer/RayTracer.cpp:72-77,196-203,
 * cudaTracer/CudaMesh.h:49-83).  The scene is flattened into one relocatable
 * blob (index-based tables, 32-byte nodes, 48-byte triangle records); the blob
 * can also be produced on one GPU, copied device-to-device (e.g. an RCCL
 * broadcast) and adopted on another with rt_scene_adopt_blob.  Beside the
 * blob a scene owns a second node array that rt_scene_upload and
 * rt_scene_adopt_blob build on the device from the blob's nodes (128 bytes per
 * interior node: the boxes of a node's four grandchildren in one record, which
 * the trace kernels fetch at once); the blob itself is only read. */
rt_scene* rt_scene_upload(rt_ctx* ctx, const RtSceneDesc* desc);
/* Host-side flattening only: *blob is malloc'ed (free with rt_blob_free). */
int rt_scene_pack(rt_ctx* ctx, const RtSceneDesc* desc, void** blob, size_t* bytes);
void rt_blob_free(void* blob);
/* Adopt a blob that already sits in device memory of ctx's GPU.  The scene
 * does not take ownership of `devBlob`, which must outlive it. */
rt_scene* rt_scene_adopt_blob(rt_ctx* ctx, const void* devBlob, size_t bytes);
/* Device address / size of an uploaded scene's blob (source of a broadcast). */
int rt_scene_blob(rt_scene* scene, const void** devBlob, size_t* bytes);
/* Re-upload the light table (same count): RayTracer::updateLights,
 * cudaTracer/RayTracer.cpp:259-267. */
int rt_scene_set_lights(rt_scene* scene, const RtLight* lights, int lightCnt);
void rt_scene_free(rt_scene* scene);

/* ---- surface -------------------------------------------------------------
 * Pitched float-RGBA frame buffer: cudaMallocPitch in
 * cudaTracer/TextureRenderer.cpp:222-227 and the pitched->tight read-back of
 * cudaTracer/TextureRenderer.cpp:86-95. */
int rt_surface_create(rt_ctx* ctx, int width, int height, void** devSurface, size_t* pitch);
int rt_surface_download(rt_ctx* ctx, const void* devSurface, size_t pitch, int width, int height,
                        float* hostRgba);
void rt_surface_free(rt_ctx* ctx, void* devSurface);
/* Plain device scratch for RtFrameAux buffers (zero-filled). */
int rt_device_alloc(rt_ctx* ctx, size_t bytes, void** devPtr);
int rt_device_download(rt_ctx* ctx, const void* devPtr, size_t bytes, void* host);
int rt_device_zero(rt_ctx* ctx, void* devPtr, size_t bytes);
void rt_device_free(rt_ctx* ctx, void* devPtr);

/* ---- the per-frame call --------------------------------------------------
 * cudamain (cudaTracer/Main.cu:21-41): renders opts->width x opts->height
 * pixels into `devSurface` (row stride `pitch` bytes, 4 floats per pixel,
 * alpha = 1).  `blockDim` is accepted for signature parity and ignored: the
 * kernel runs persistent 64-lane wavefronts over 8x8-pixel tiles.  The call
 * enqueues on the context's stream and returns; rt_sync waits.  `aux` may be
 * NULL.  Parity with the reference is defined for samples == 1 (its jitter
 * RNG, cudaTracer/Main.cu:108-110, is not reproducible); samples 2..RT_MAX_SAMPLES
 * jitter every sample by rt_sample_jitter and average them as
 * cudaTracer/Main.cu:66-82 does. */
int rt_render(rt_ctx* ctx, rt_scene* scene, const RtDeviceOpts* opts, const RtCamera* cam,
              void* devSurface, size_t pitch, int blockDim, const RtFrameAux* aux);

/* rt_render for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments, same pixels; the
 * differences:
 *   - the frame's workspace (path-node arrays, shading contexts, visibility bytes, traversal-stack spill area,
 *     control block) is carved from a static device array of RT_NOALLOC_SCRATCH_BYTES that every GPU the library
 *     runs on holds until the process ends; a call that needs more fails with RT_ERR_LIMIT and says how much
 *     (about 120 bytes + one per light per path node: a 640x480 frame with 4 lights fits; render larger frames in
 *     several calls with RtFrameAux.tileStride);
 *   - it runs on the context's stream, in order, whatever rt_set_frames_in_flight says, takes the tiles in row-major
 *     order, records no timing events, and calls no hipMalloc / hipFree or any other allocating routine;
 *   - every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two contexts, two streams) race;
 *     calls on one stream are ordered and safe.  Nothing in the array needs to survive between calls. */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
int rt_render_noalloc(rt_ctx* ctx, rt_scene* scene, const RtDeviceOpts* opts, const RtCamera* cam,
                      void* devSurface, size_t pitch, int blockDim, const RtFrameAux* aux);

/* HIP-event timing of the rt_render calls since the last reset, measured on the
 * stream they ran on.  Call after rt_sync.  totalMs sums the whole per-frame
 * kernel sequence; stageMs[4] (may be NULL) splits it into: closest-hit trace of
 * the primary rays, shading, shadow-ray trace, everything after.  Without a
 * reset the figures cover the latest 4096 calls; rendering itself never
 * depends on a reset. */
int rt_timing_reset(rt_ctx* ctx);
int rt_timing_read(rt_ctx* ctx, double* totalMs, int* launches);
int rt_timing_read_stages(rt_ctx* ctx, double* totalMs, double* stageMs, int* launches);

/* Mirror / glass materials spawn child rays (cudaTracer/tracer.cu:117-142); the
 * per-frame node arrays hold `nodesPerPixel` x pixels path nodes (default 3).
 * A frame that needs more drops the child rays that do not fit (their parents take the background colour for
 * them) and the next rt_sync returns RT_ERR_LIMIT, however many frames were rendered in between (and whether or not
 * a larger frame made the library regrow its workspace meanwhile).  rt_render_noalloc keeps the flag in the GPU's one
 * scratch array, which all contexts on that GPU share: there the report goes to whichever of them calls rt_sync
 * first after the frame. */
int rt_set_path_budget(rt_ctx* ctx, int nodesPerPixel);

/* Diagnostic: every workspace array allocated from now on gets `bytes` (a multiple of 256, 0 = none) of guard
 * band behind it, filled with a pattern; rt_check_workspace_guards waits for the frames in flight and returns
 * RT_ERR_DEVICE naming the array if a kernel wrote past an array's end.  The reference has no counterpart. */
int rt_set_workspace_guards(rt_ctx* ctx, size_t bytes);
int rt_check_workspace_guards(rt_ctx* ctx);

/* Mesh BVHs built on the device.  A mesh handed to rt_scene_upload with nodes == NULL and nodeCnt == 0 gets its
 * BVH built on the GPU with the algorithm of cudaTracer/BVH.h:104-181 (median cut of the box centroids along the
 * widest axis of the centroid bounds, leaves of at most `threshold` triangles; the reference uses 10,
 * cudaTracer/CudaMesh.h:58): same tree shape, boxes and split axes as the host builder whenever the median
 * centroids are distinct; the order of triangles inside a leaf may differ.  Where centroid coordinates tie at a cut
 * (grids, cubes), the tied triangles go left and right in the order the caller gave them - a deterministic choice
 * among those std::nth_element may make - and a range whose box centroids ALL coincide, which the reference turns
 * into one leaf of any size (cudaTracer/BVH.h:126-128), is cut at its median position like any other range: same
 * hits, except equal-distance ties among such triangles.  `tightBounds` = 0 starts boxes from the reference's
 * (FLT_MAX, FLT_MIN) empty box (cudaTracer/BoundingBox.cuh:25-27), 1 from a true empty box.  The triangles of such a
 * mesh are given in any order; rt_scene_pack (host only) rejects it.  Defaults: threshold 10, tight bounds. */
int rt_set_device_bvh(rt_ctx* ctx, int threshold, int tightBounds);

/* Bounds refit at upload.  The reference's builder starts every box union from an "empty" box whose max is FLT_MIN
 * (cudaTracer/BoundingBox.cuh:25-27 through cudaTracer/BVH.h:72-96), so each of its boxes over negative coordinates
 * reaches up to 0: node arrays taken from the reference as they are (cudaTracer/CudaMesh.h:58-63) cost a ray 20-80 x
 * the box and triangle tests of the same tree with true bounds.  With `on` != 0, rt_scene_upload recomputes the boxes
 * of every tree a scene arrives with, on the device, before anything walks them: a mesh leaf gets the exact bounds of
 * its triangles' vertex positions, a scene-BVH leaf the exact union of its shapes' own boxes (RtShape.bbmin / bbmax,
 * as the caller computed them), an interior node the exact union of its children's boxes.  Topology, split axes,
 * leaf contents and therefore visiting order and tie rule stay the caller's; hits are the same except for rays that
 * lie exactly in a box face (a 0 x inf = NaN plane distance, whose outcome depends on which planes a box has).
 * rt_scene_pack does the same on the host when given a context with the switch on, with the same bytes as result.
 * Default: off (the reference's boxes are walked as they come).
 * rt_bvh_refit / rt_scene_bvh_refit are the same computation on a caller's own node arrays, in place, host code only:
 * for a mesh BVH (`triangles` in leaf order, as in RtMesh) and for the scene BVH (`shapes` in leaf order, as in
 * RtSceneDesc).  Their errors: rt_last_error(NULL). */
int rt_set_upload_refit(rt_ctx* ctx, int on);
/* Rebuild at upload: with `on` != 0, rt_scene_upload builds the BVH of EVERY mesh on the device, as rt_set_device_bvh
 * describes, whether or not the mesh arrives with nodes (the node array is not looked at; the triangles may be in any
 * order), and gives the scene BVH true boxes as rt_set_upload_refit does.  For a host that hands over the reference's
 * own arrays (cudaTracer/CudaMesh.h:58-63): its trees were cut on FLT_MIN-clamped centroid bounds
 * (cudaTracer/BVH.h:122-128 over cudaTracer/BoundingBox.cuh:25-27), which picks the wrong split axis wherever
 * coordinates are negative, so even with refitted boxes a ray makes several times the tests it makes in a tree cut on
 * true bounds.  Same hits as with the caller's trees, except equal-distance ties (the order of the tests is the new
 * tree's).  1 M triangles: 15 ms.  rt_scene_pack (host only) ignores the switch.  Default: off. */
int rt_set_upload_rebuild(rt_ctx* ctx, int on);
/* How the device cuts the meshes it builds (those of rt_set_device_bvh and rt_set_upload_rebuild): RT_BVH_MEDIAN_CUT
 * (default; the reference's rule, cudaTracer/BVH.h:104-150) or RT_BVH_BINNED_SAH - the tree rt_mesh_bvh_build makes
 * with that rule and tight bounds, node for node and leaf slot for leaf slot, with the cuts decided on the device: the
 * host keeps the tree (its shape depends on the data) and hands the device one level of ranges at a time; bins and
 * centroid bounds are gathered with atomics, one thread per range walks the host builder's candidates in its order
 * with its arithmetic, and a stable radix sort moves every range's left side to its front.  Needs tight bounds
 * (rt_set_device_bvh(ctx, threshold, 1)).  10 M triangles: 0.45 s against 2.5 s for the host builder on 16 threads;
 * a 4K frame of that height field takes 3.5 ms on this tree and 4.6 ms on the median cut's.  No counterpart in the
 * reference. */
int rt_set_device_bvh_rule(rt_ctx* ctx, int splitRule);
int rt_bvh_refit(RtLinearNode* nodes, int nodeCnt, const RtVertex* vertices, int vertexCnt, const RtTriangle* triangles, int triangleCnt);
int rt_scene_bvh_refit(RtLinearNode* nodes, int nodeCnt, const RtShape* shapes, int shapeCnt);

/* Mesh BVH built on the host, for callers that hand their meshes over WITH nodes (RtMesh.nodes).  Host code only: no
 * context, no GPU.  RT_BVH_MEDIAN_CUT is the reference's builder (cudaTracer/BVH.h:104-181, called with threshold 10
 * by cudaTracer/CudaMesh.h:58): same nodes, same leaf order.  RT_BVH_BINNED_SAH has no counterpart in the reference:
 * every range is cut where the surface-area heuristic over 32 centroid bins per axis is lowest (median cut when no bin
 * boundary separates the range), so the tree is not balanced and may be deeper than ceil(log2(n / threshold)); hits
 * are the same as with the reference's tree except for equal-distance ties, which depend on visiting order.
 * `leafThreshold` (1..63) is the most triangles a leaf gets unless their centroids coincide.  `order[k]` receives the
 * index (into `triangles`) of the triangle that belongs in leaf slot k: the caller hands RtMesh.triangles over in that
 * order.  `nodes` must hold `nodeCapacity` entries; 2 * triangleCnt - 1 always suffice.
 * Large meshes are built on several threads (as many as the machine reports, at most 16; the environment variable
 * RT_BVH_THREADS overrides the number, 1 = the calling thread alone); the result does not depend on the number.
 * Returns RT_ERR_INVALID / RT_ERR_LIMIT with the message in rt_last_error(NULL). */
#define RT_BVH_MEDIAN_CUT 0
#define RT_BVH_BINNED_SAH 1
int rt_mesh_bvh_build(const RtVertex* vertices, int vertexCnt, const RtTriangle* triangles, int triangleCnt, int splitRule,
                      int leafThreshold, int tightBounds, RtLinearNode* nodes, int nodeCapacity, int* nodeCnt, int32_t* order);

/* Seed of the sub-pixel jitter used with samples > 1 (default 1234, the seed the reference gives curand,
 * cudaTracer/Main.cu:127).  Change it per frame to accumulate independent samples over frames. */
int rt_set_sample_seed(rt_ctx* ctx, uint32_t seed);
#define RT_MAX_SAMPLES 64

/* Progressive accumulation over frames, on the device: the successor of the reference's per-launch sample loop
 * (cudaTracer/Main.cu:66-98).  `devSum` is a float4 surface of the frame's size in device memory (row stride
 * `sumPitch` bytes) that holds, per pixel, {sum of r, g, b over all samples so far, number of samples}.  While it is
 * set, every rt_render / rt_render_noalloc call
 *   - numbers its opts->samples samples behind the `samplesSoFar` of the earlier calls (sample k of a pixel always
 *     gets the jitter rt_sample_jitter(seed, pixel, k, axis), also when a call renders a single sample),
 *   - adds them to the sums one after the other, in order, and
 *   - writes to the frame surface the average over ALL samples so far, gamma-corrected as Main.cu:80-98 does.
 * N calls of one sample therefore give, bit for bit, the frame of one call with N samples (and any other split of
 * N).  Every PIXEL counts its own samples (the fourth float of its sum), so a frame may be rendered in several calls
 * - tile shards (RtFrameAux.tileStride) or tile lists - in any order: each call continues the pixels it renders.
 * `samplesSoFar` = 0 starts a new accumulation: the next render call clears the surface (opts->width x opts->height
 * pixels) before it renders.  Any other value continues the surface as it is - the counts in it are the truth - and is
 * only the caller's statement of the largest count in it, for the limit below.  devSum = NULL ends it.  Frames must
 * run in order (rt_set_frames_in_flight(1)).  At most 2^24 samples per pixel (the count is kept in fp32); the library
 * checks that against samplesSoFar + the samples of all calls since, which over-counts when a frame takes several
 * calls. */
int rt_set_accumulation(rt_ctx* ctx, void* devSum, size_t sumPitch, uint32_t samplesSoFar);

/* Order in which a call's 8x8 tiles enter the trace kernels' work queue.  Which tiles
 * are rendered and every pixel value are the same either way.  CENTER_OUT (default)
 * starts with the tiles nearest the image centre, where rays usually take longest,
 * so that the last rays in flight are short ones; the reference has no counterpart
 * (cudaTracer/Main.cu:46-90 launches one thread per pixel in grid order). */
#define RT_TILE_ORDER_ROW_MAJOR 0
#define RT_TILE_ORDER_CENTER_OUT 1
int rt_set_tile_order(rt_ctx* ctx, int mode);

/* The tiles of the following rt_render calls as an explicit list: `count` indices of 8x8-pixel frame tiles
 * (row-major over the tile grid, each at most once) in host memory; the list is copied.  The calls render exactly
 * these tiles, and hand them to the trace kernels in this order (put the expensive ones first).  It replaces
 * RtFrameAux.tileFirst / tileStride, which must then be 0 / 1; tiles = NULL, count = 0 returns to them.  This is how
 * a multi-GPU driver gives every GPU a share of equal COST instead of an equal number of tiles
 * (cudaraytracer_amd/multi.py: balanced_assignment).  The reference has no counterpart. */
int rt_set_tile_list(rt_ctx* ctx, const int32_t* tiles, int count);

/* Per-tile cost of a frame: `devCost` points to one uint32 per frame tile in device memory (NULL = off).  Calls that
 * also ask for counters (RtFrameAux.counters: the counting build of the kernels) ADD to entry t the box and triangle
 * tests of tile t's primary rays and of the shadow rays of their hits.  Clear it with rt_device_zero. */
int rt_set_tile_cost_buffer(rt_ctx* ctx, uint32_t* devCost);

/* Tuning knob: how many traversal-stack entries per ray are kept in LDS (8 bytes
 * each, 64 rays per wavefront), rounded down to a power of two (at least 2).
 * They form a ring holding the top of the stack; older entries move to a global
 * array.  Fewer entries = more resident wavefronts per CU.  Results do not
 * depend on it. */
int rt_set_lds_stack(rt_ctx* ctx, int entries);

/* Box test.  The reference's slab test (cudaTracer/BoundingBox.cuh:115-139) equals its min / max
 * form - fewer instructions - bit for bit unless a plane distance is a NaN, which takes an infinite reciprocal
 * direction.  The trace of a frame's primary rays, and of their shadow rays, therefore runs as two launches when the
 * call renders at least `minRays` primary rays (default 100000; INT_MAX = never): the first walks the rays with a
 * finite reciprocal direction with the min / max form, four grandchildren per node step, the second the others with
 * the literal form (it ends at once when there are none).  Smaller frames and reflection levels use the literal form
 * in one launch: with many such frames in flight the extra launches cost more than the shorter walk saves.  Results do not depend on it, bit for bit.  No counterpart in the reference. */
int rt_set_box_test_split(rt_ctx* ctx, int minRays);

/* Shadow rays that cannot matter.  The reference walks one shadow ray per diffuse hit and light
 * (cudaTracer/tracer.cu:176-184) and then adds lightComponent * lambert and lightComponent * specMask * specularFac,
 * where specularFac is 0 unless lambert > 0 (tracer.cu:186-196): for a light behind the surface (lambert == 0) both
 * terms are zeros whatever the ray finds, and adding them changes nothing.  With `on` != 0 such a ray is not walked
 * (its light counts as hidden); it is walked as always when the zero is not certain - a light within 1e-20 of the
 * surface point, an intensity * power beyond 1e18 or a specular mask (a sphere's Ks) beyond 1, where the reference's
 * term could be a NaN or an infinity.  Pixels,
 * hit keys and counters are the same bit for bit (counter frames walk every ray, as the reference does); what changes is the
 * time of the shadow launch and the number of rays actually walked, which is below primary + shadow of the counters.
 * Off by default, so that rays per second means rays walked.  No counterpart in the reference. */
int rt_set_unlit_shadow_skip(rt_ctx* ctx, int on);

/* Frames in flight (1..16, default 1).  With 1, rt_render enqueues the frame on the
 * context's stream, in order, as the reference's cudamain does on the default
 * stream (cudaTracer/Main.cu:92-119).  With more, consecutive rt_render calls run
 * on internal streams and overlap: each frame starts behind everything enqueued
 * on the context's stream at the time of its call, but the context's stream does
 * NOT wait for it.  Frames that overlap must be given different surfaces and aux
 * buffers.  rt_sync waits for all of them; rt_stream_wait_frame orders a stream
 * behind one of them.  Overlap fills the wave slots that the persistent trace
 * kernels leave empty while their last rays finish.  Each frame in flight runs on
 * its own HIP stream; HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
 * (default 4, shared with the caller's streams), so more than two frames only
 * overlap when the process sets that environment variable higher (at least the
 * number of frames in flight plus the caller's own streams, e.g. 32) before
 * its first HIP call.  The reference has no counterpart (it synchronises the device
 * after every frame). */
int rt_set_frames_in_flight(rt_ctx* ctx, int frames);

/* Makes `hipStream` (NULL = the context's stream) wait, on the device, for the
 * frame issued `framesAgo` rt_render calls ago (0 = the latest). */
int rt_stream_wait_frame(rt_ctx* ctx, void* hipStream, int framesAgo);

/* Tuning knob: upper bound on persistent trace wavefronts per CU (0 = as many as
 * the hardware keeps resident).  Results do not depend on it. */
int rt_set_trace_waves(rt_ctx* ctx, int wavesPerCu);

/* ---- tile gather ------------------------------------------------------------
 * The two ends of a multi-GPU frame's gather, as device kernels on the context's stream.  rt_tiles_pack copies the
 * pixels of `tileCnt` 8x8 tiles (`devTiles`: frame tile indices, row-major over the tile grid, in device memory) out
 * of a float4 surface into `devOut`, 64 pixels per tile, tile after tile in list order (pixels of an edge tile that
 * lie outside the frame are zero); rt_tiles_scatter writes such a buffer to the tiles' places in a surface.  With
 * `quantize` the packed pixels are 8-bit RGBA by the rule of the reference's image dump, clamp((int)(v * 256), 0,
 * 255) on all four channels (cudaTracer/PicDumper.h:43-55) - a quarter of the bytes to move - and the scatter's
 * target is an 8-bit RGBA surface (4 bytes per pixel).  Order them behind a frame that runs on an internal stream
 * (rt_set_frames_in_flight > 1) with rt_stream_wait_frame.  The reference has one GPU and no counterpart. */
int rt_tiles_pack(rt_ctx* ctx, const void* devSurface, size_t pitch, int width, int height, const int32_t* devTiles, int tileCnt,
                  void* devOut, int quantize);
int rt_tiles_scatter(rt_ctx* ctx, const void* devIn, const int32_t* devTiles, int tileCnt, void* devSurface, size_t pitch, int width,
                     int height, int quantized);

/* ---- several GPUs from one host process -------------------------------------
 * The reference's caller is one process that calls cudamain once per frame (cudaTracer/RayTracer.cpp:188-208,
 * cudaTracer/main.cuh:7-12).  rt_multi keeps that shape for a node with several GPUs: one context per device, the
 * scene flattened and uploaded once to devices[0] and copied from there to every peer (device-to-device, each peer
 * over its own link), every device rendering its own list of 8x8 tiles of the SAME frame, and the tiles gathered
 * into the caller's surface on devices[0] (packed on the peer, one peer copy, scattered on devices[0]).  No second
 * process and no collective.  A device may be named more than once (several contexts on one GPU): the frame is the
 * same, which is how a one-GPU machine exercises the path.
 *   rt_multi_context(m, i)     the i-th device's context, for the per-context knobs (rt_set_upload_refit,
 *                              rt_set_upload_rebuild, rt_set_device_bvh and rt_set_device_bvh_rule on context 0 apply to
 *                              rt_multi_scene_upload; render-time switches such as rt_set_unlit_shadow_skip are each
 *                              context's own; do not change a context's stream, tile list or frames in flight)
 *   rt_multi_scene_upload      replaces the current scene
 *   rt_multi_balance           renders a counter frame and deals the tiles out so that every device gets the same
 *                              number of tiles and (nearly) the same number of box + triangle tests; until it is
 *                              called, and again after the frame size changes, tile t belongs to device t % ndev
 *   rt_multi_set_quantize      on: `devSurface` of rt_multi_render is an 8-bit RGBA surface (4 bytes per pixel, dump
 *                              rule as in rt_tiles_pack) and the tiles travel as bytes
 *   rt_multi_render            cudamain for the node: enqueues the frame on all devices and the gather behind it;
 *                              `devSurface` is memory of devices[0]
 *   rt_multi_sync              waits for everything enqueued; reports what rt_sync reports
 * Errors: rt_multi_last_error(m) (m == NULL: of the last failed rt_multi_create). */
typedef struct rt_multi rt_multi;
rt_multi* rt_multi_create(int ndev, const int* devices);
void rt_multi_destroy(rt_multi* m);
const char* rt_multi_last_error(const rt_multi* m);
int rt_multi_device_count(const rt_multi* m);
rt_ctx* rt_multi_context(rt_multi* m, int i);
int rt_multi_scene_upload(rt_multi* m, const RtSceneDesc* desc);
int rt_multi_scene_set_lights(rt_multi* m, const RtLight* lights, int lightCnt);
int rt_multi_balance(rt_multi* m, const RtDeviceOpts* opts, const RtCamera* cam);
int rt_multi_tile_counts(const rt_multi* m, int* counts); /* tiles per device of the current assignment */
int rt_multi_set_quantize(rt_multi* m, int on);
int rt_multi_render(rt_multi* m, const RtDeviceOpts* opts, const RtCamera* cam, void* devSurface, size_t pitch, int blockDim);
int rt_multi_sync(rt_multi* m);

/* cu_initCurand / cu_cleanCurand (cudaTracer/Main.cu:113-150) have no
 * counterpart: the jitter is a hash of (seed, pixel, sample, axis) and keeps
 * no state. */

#ifdef __cplusplus
}
#endif
#endif /* RT_CAPI_H */
