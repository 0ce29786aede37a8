// Device-side formats shared by the packer / frame driver (rt_capi.cpp) and
// the kernels (kernels/rt_kernels.hip).
//
// SCENE BLOB.  A scene lives in ONE contiguous, relocatable blob: every
// reference inside it is a byte offset from the blob start, so the blob can be
// copied between GPUs (RCCL broadcast) and used as is.
//
//   [BlobHeader 256 B][lights][DevShape x shapeCnt][scene nodes 32 B each]
//   [DevMesh x meshCnt][triangle records 48 B, all meshes back to back]
//   then per mesh: [mesh nodes 32 B][vertices 56 B][diffuse RGBA8][normals RGBA8][specular RGBA8]
//
// Every section starts on a 256-byte boundary.
//
// NODES.  The host hands over the reference's depth-first LinearNode arrays
// (first child at self + 1).  The packer re-lays each tree out as SIBLING PAIRS:
// record 0 is the root (record 1 is padding), and the two children of every
// interior node sit next to each other at an even index, so that one 64-byte
// fetch brings both child boxes.  A record is still 32 bytes:
//   {bbmin[3], bbmax[3], ref, 0}
// with `ref` describing the node itself (see the RT_REF_* macros): a leaf names
// its primitive range, an interior node names its child pair and split axis.
// References are ABSOLUTE within the blob (child pairs in 64-byte units from the
// blob start, triangles as indices into the one record array, shapes as indices
// into the shape table), so a traversal needs no per-tree base pointers.
// Box values, split axes and leaf contents are untouched, so every box test
// and the near/far visiting order are those of the original tree.
//
// QUADS.  Beside the blob, a scene holds one 128-byte QUAD record per sibling pair (built on the device when the scene
// is uploaded or adopted; rtk_build_quads): the pair's two records each replaced by the pair of ITS children, so that
// one fetch brings the boxes of a node's four grandchildren.  Half h (64 bytes, laid out like a pair) is the child
// pair of the pair's record h with 1 << (that record's split axis) in the spare word of its first record; if record h
// is a leaf, the half is the record itself plus an empty one and the word is 0; for the padding beside a root record
// the half is empty.  An empty record has a box of NaNs, which passes no box test.  Interior references inside quads
// name quads (128-byte units from the start of the quad array).  The launches with the min / max box test walk quads: a box that a ray
// hits lies inside its parent's box, so the parent's own test cannot change which leaves are reached, and the four
// grandchildren are visited in the order the two levels of the binary walk would visit them.
//
// FRAME WORKSPACE.  A frame is rendered as a sequence of ray "waves" over path
// nodes.  A path node is one invocation of the reference's recursive castRay
// (cudaTracer/tracer.cu:15-48): level-0 nodes are the pixels of the tiles this
// call renders (64 per 8x8 tile), deeper nodes are reflection / refraction
// children, appended per level.  With S samples per pixel, level-0 node s is lane (s & 63) of
// sample (s >> 6) % S of local tile (s >> 6) / S.  Per level the driver runs
//   trace (closest hit) -> shade -> trace (any hit, shadow rays) -> accumulate
// and finally resolves mirror / glass nodes bottom-up and writes the pixels.
#ifndef RT_DEVICE_FORMAT_H
#define RT_DEVICE_FORMAT_H

#include <stddef.h>
#include <stdint.h>

#include "../../include/rt_capi.h"

#define RT_BLOB_MAGIC 0x52544246u /* "RTBF" */
#define RT_BLOB_VERSION 4u
#define RT_TILE 8 /* 8x8 pixels = 64 level-0 path nodes */
#define RT_MESH_ENTRY_SLOTS 6 /* 8-byte slots per lane that hold the world-space ray while a mesh is walked */

/* 32-bit node reference.
 *   leaf:     bit 31 set | bit 30 set for a scene-BVH leaf (shapes), clear for a mesh-BVH leaf (triangles)
 *             | primitive count (6 bits) << 24 | first primitive (24 bits)
 *   interior: split axis (2 bits) << 29 | blob offset of the child pair / 64 (29 bits)
 *             axis value 3 marks the pseudo reference to a tree's root record: only the first
 *             record of the pair is a node, and it is the "near child". */
#define RT_REF_LEAF 0x80000000u
#define RT_REF_SHAPES 0x40000000u
#define RT_REF_TAG_MASK 0xC0000000u
#define RT_REF_MAX_LEAF_PRIMS 63
#define RT_REF_MAX_PRIMS ((1 << 24) - 16)
#define RT_REF_IS_LEAF(r) (((r)&RT_REF_LEAF) != 0u)
#define RT_REF_LEAF_COUNT(r) (((r) >> 24) & 0x3fu)
#define RT_REF_LEAF_FIRST(r) ((r)&0xffffffu)
#define RT_REF_AXIS(r) (((r) >> 29) & 3u)
#define RT_REF_ROOT_AXIS 3u
#define RT_REF_PAIR_UNITS(r) ((r)&0x1fffffffu) /* 64-byte units from the blob start */

struct BlobHeader {
    uint32_t magic;
    uint32_t version;
    uint64_t totalBytes;
    int32_t lightCnt;
    int32_t shapeCnt;
    int32_t meshCnt;
    int32_t sceneNodeCnt;
    int32_t sceneDepth;    // levels below the scene-BVH root
    int32_t maxMeshDepth;  // deepest mesh BVH
    int32_t hasSecondary;  // any REFLECTION / REFLECTION_AND_REFRACTION material
    int32_t maxLeafShapes;
    uint64_t offLights;
    uint64_t offShapes;
    uint64_t offSceneNodes;
    uint64_t offMeshes;
    uint64_t offRecords;    // triangle records of all meshes
    uint32_t sceneRootRef;  // pseudo reference to the scene BVH's root record
    uint32_t recordCnt;
    uint8_t reserved[256 - 96];
};

// One shape, 64 bytes (the reference's Shape is 232 B with the Mesh embedded).
struct DevShape {
    int32_t kind;
    float center[3];
    float radius2;
    int32_t meshIdx;
    int32_t tag;
    int32_t materialType;
    float ior;
    float Ks;
    float specularExponent;
    float diffuseColor[3];
    int32_t pad[2];
};

// One mesh, 192 bytes.
struct DevMesh {
    float inv[16];  // transformInverse, row-major
    uint64_t offNodes;
    uint64_t offTriRecords;
    uint64_t offVertices;
    uint64_t offTex[3];  // diffuse, normals, specular
    int32_t texW[3];
    int32_t texH[3];
    int32_t texMode[3];
    int32_t triangleCnt;
    int32_t nodeCnt;
    int32_t vertexCnt;
    uint32_t rootRef;     // pseudo reference to the mesh BVH's root record
    uint32_t recordBase;  // index of the mesh's first triangle record in the blob's record array
    int32_t depth;        // levels below the mesh BVH's root
    int32_t pad[5];
};

// Triangle record, 48 bytes: v0 and the two edges exactly as the reference's
// Moller-Trumbore forms them in fp32 (e1 = v1 - v0, e2 = v2 - v0,
// cudaTracer/Mesh.cuh:184-188), followed by the three vertex indices.
struct TriRecord {
    float v0[3];
    float e1[3];
    float e2[3];
    int32_t idx[3];
};

#ifdef __cplusplus
static_assert(sizeof(BlobHeader) == 256, "BlobHeader size");
static_assert(sizeof(DevShape) == 64, "DevShape size");
static_assert(sizeof(DevMesh) == 192, "DevMesh size");
static_assert(sizeof(TriRecord) == 48, "TriRecord size");
#endif

// ---- frame workspace ------------------------------------------------------
// Control block (uint32 words in device memory).  Every word except CTL_OVERFLOW is cleared when a frame opens
// its first level; CTL_OVERFLOW stays set until rt_sync has reported it.
enum {
    CTL_DIFFUSE_CNT = 2,   // diffuse hits of the current level
    CTL_DEFER_CLOSEST = 5,  // rays of the current level left to the launches with the literal box test: closest hit,
    CTL_DEFER_SHADOW = 6,   // shadow rays
    CTL_LEVEL_START = 8,   // [RT_MAX_DEPTH + 2] first node of each level
    CTL_LEVEL_COUNT = 40,  // [RT_MAX_DEPTH + 2] nodes of each level: while the level above is shaded, the slots its
                           // child rays ASKED for (may pass the arrays' end); once the level is opened, the nodes it holds
    CTL_LEVEL_FIT = 72,    // [RT_MAX_DEPTH + 2] 1 + the level's nodes below a request that straddled the arrays' end (0 = none did)
    CTL_OVERFLOW = 127,    // sticky: a child ray did not fit the node arrays
    CTL_WORDS = 128
};
// Behind the control words, in the same array:
//  - the work queues.  Each of the four queue kinds (closest hit, shadow rays, and the same two for the launches with
//    the literal box test) is RT_QUEUES counters, 256 bytes apart: counter q hands out the 64-ray chunks q, q + RT_QUEUES,
//    q + 2 RT_QUEUES, ...; a wavefront starts with counter blockIdx % RT_QUEUES and moves on to the next one when its
//    own is dry, so that the atomics of a launch are spread over RT_QUEUES addresses instead of queuing up at one;
//  - the work items left to the launches with the literal box test, closest hit first, then shadow rays (RT_DEFER_LIST
//    entries each; if a launch leaves more, the second launch looks through all of the level's items for the marks).
#ifndef RT_QUEUES
#define RT_QUEUES 8
#endif
#define RT_QUEUE_STRIDE 64 /* words between two counters */
#define RT_QUEUE_WORDS (4 * RT_QUEUES * RT_QUEUE_STRIDE)
#define RT_QUEUE_WORD(kind, q) (CTL_WORDS + ((kind) * RT_QUEUES + (q)) * RT_QUEUE_STRIDE)
#define RT_DEFER_LIST 16384
#define RT_DEFER_WORD(shadow) (CTL_WORDS + RT_QUEUE_WORDS + ((shadow) ? RT_DEFER_LIST : 0))
#define RT_CTL_BYTES ((CTL_WORDS + RT_QUEUE_WORDS + 2 * RT_DEFER_LIST) * sizeof(uint32_t))

// Node kinds written by the shade kernel.
enum { NODE_DONE = 0, NODE_MIRROR = 1, NODE_GLASS = 2 };

struct FrameParams {
    // scene
    const unsigned char* blob;
    uint64_t offLights, offShapes, offSceneNodes, offMeshes;
    int32_t lightCnt, shapeCnt;
    // per-frame inputs
    RtDeviceOpts opts;
    float orig[3];
    float camera[16];
    // outputs
    unsigned char* surface;
    uint64_t pitch;
    int32_t* hitKeys;
    unsigned long long* counters;
    uint32_t* tileCost;                      // optional [tilesX * tilesY]: counter launches add each tile's box + triangle tests
    // tile shard
    int32_t tilesX, tilesY;
    uint32_t tilesXMagic;                    // floor((2^32 - 1) / tilesX): tile -> (row, column) without a division
    float invWidth, invHeight;               // 1.0f / width, 1.0f / height as the reference forms them (cudaTracer/Main.cu:57-58)
    int32_t tileFirst, tileStride, tileCnt;  // this call renders tiles tileFirst + k*tileStride, k < tileCnt
    const int32_t* tileOrder;                // optional [tileCnt]: frame tile of local tile k (overrides the arithmetic rule)
    int32_t slots;                           // tileCnt * samples * 64 level-0 nodes
    uint32_t sampleSeed;
    unsigned char* accum;                    // optional running sum surface: float4 {sum r, g, b, samples so far} per pixel
    uint64_t accumPitch;
    int32_t nodeCap;                         // capacity of the per-node arrays
    int32_t stackEntries;                    // per-lane traversal stack entries needed (8 bytes each)
    int32_t stackLds;                        // size of the LDS ring holding the top of the stack (a power of two)
    uint32_t sceneRootRef;
    uint32_t sceneRootQuad;                  // the same in the quad array
    const float4* quads;                     // quad records of all BVHs (scene first, then the meshes)
    const uint32_t* meshRootQuad;            // [meshCnt] pseudo reference to each mesh BVH's root quad
    uint64_t offRecords;
    int32_t traceWaves;                      // wavefronts of the persistent trace kernels
    int32_t waveChunks;                      // 64-ray chunks a wavefront should get at least before another one joins in
    int32_t waveChunksLarge;                 // the same for launches of 6000 chunks and more
    int32_t splitSlots;                      // level-0 traces of at least this many rays run as two launches (min / max box test + literal)
    int32_t skipUnlit;                       // shadow rays towards lights that cannot contribute are not walked (rt_set_unlit_shadow_skip)
    // workspace (device pointers)
    uint32_t* ctl;          // CTL_WORDS
    float4* hit;            // [nodeCap] {t, u, v, shape slot as int bits (-1 miss, -2 outside the frame)}
    int32_t* hitTri;        // [nodeCap] triangle of the hit (index within its mesh)
    float4* color;          // [nodeCap] radiance returned by the node
    float4* secRay;         // [2 * (nodeCap - slots)] origin, direction of nodes beyond level 0
    int32_t* child;         // [nodeCap] first child node (mirror: 1 child, glass: 2 consecutive)
    float* kr;              // [nodeCap] Fresnel term of mirror / glass nodes
    int32_t* kind;          // [nodeCap] NODE_*
    float4* ctx;            // [4 * nodeCap] shading context of each diffuse hit of the current level
    int32_t* ctxNode;       // [nodeCap] node of each diffuse hit
    unsigned char* vis;     // [nodeCap * max(lightCnt,1)] 1 = light visible
    uint2* stackSpill;      // [traceWaves * (stackEntries + RT_MESH_ENTRY_SLOTS) * 64] older stack entries + parked world-space rays
};

#define RT_MAX_FRAMES_IN_FLIGHT 16

// Enqueues all kernels of one frame on `stream` (no host synchronisation).
// `stageEvents`: optional hipEvent_t[5] recorded at frame start, after the level-0 closest-hit
// trace, after shading, after the shadow trace, at frame end.  Returns a hipError_t value.
int rtk_render_frame(const FrameParams* frame, int hasSecondary, int count, int cuCount, void* stream, void** stageEvents);
// ---- mesh BVH construction on the device (kernels/rt_bvh_build.hip) -------------
// The tree's shape follows from the primitive count alone (median cuts), so the host writes the
// node records' references and this work table; the device orders the primitives and fills in
// boxes, split axes and triangle records.
struct RtBuildNode {
    int32_t start, end;   // primitive positions of the node's range
    int32_t rec;          // index of the node's 32-byte record in the mesh's node array
    int32_t left, right;  // records of the children (-1 for a leaf)
    uint32_t pathId;      // heap index of the range (root 1, children 2i and 2i + 1)
};
struct RtBvhBuildJob {
    int32_t triangleCnt, threshold, tightBounds;
    int32_t depth;     // levels of cuts = depth of the deepest leaf
    int32_t nodeCnt;   // entries of `nodes`: leafCnt leaves first, then interior nodes by level, deepest level first
    int32_t leafCnt;
    int32_t interiorPerLevel[RT_STACK_ENTRIES];
    const RtBuildNode* nodes;           // host
    const int32_t* devTriangles;        // [3 * triangleCnt] vertex indices, caller's order
    const unsigned char* devVertices;   // RtVertex array
    unsigned char* devNodeRecords;      // the mesh's node records (references already in place)
    unsigned char* devTriRecords;       // the mesh's triangle records
};
// Runs the build on `stream` and waits for it.  Returns a hipError_t value.
int rtk_build_mesh_bvh(const RtBvhBuildJob* job, void* stream);

// Surface-area cuts on the device, driven level by level from the host (kernels/rt_bvh_build.hip, "Surface-area cuts"):
// `hostBmin` / `hostBmax` are the primitives' boxes as float4 arrays.  All return a hipError_t value and wait for `stream`.
struct RtSahBuild;
int rtk_sah_begin(int n, int threshold, const float* hostBmin, const float* hostBmax, RtSahBuild** out, void* stream);
int rtk_sah_level(RtSahBuild* b, int ranges, int threshold, const int32_t* hostStart, const int32_t* hostEnd, int32_t* hostResult, void* stream);
int rtk_sah_order(RtSahBuild* b, uint32_t* hostOrder, void* stream);
void rtk_sah_end(RtSahBuild* b);

// Bounds refit of one BVH inside the blob (rt_set_upload_refit): `pairs` = device address of the tree's pair records,
// `pairBase` = their blob offset / 64, `depth` = levels below the root.  A mesh BVH passes `records` (the blob's
// triangle-record array) and `vertices` (the mesh's vertex array): its leaves get the exact bounds of their triangles.
// The scene BVH passes `shapeBoxes` (device array, six floats {min, max} per shape in leaf order): its leaves get the
// exact union of their shapes' boxes.  Interior boxes become the exact union of their children's.  Enqueued on
// `stream`.  Returns a hipError_t value.
int rtk_refit_tree(void* pairs, uint32_t pairBase, uint32_t pairCnt, int depth, const void* records, const void* vertices, const float* shapeBoxes,
                   void* stream);

// Quad records (see the head of this file) of one BVH: `pairs` = device address of the tree's pair records, `pairBase`
// = their blob offset / 64 (what the references inside them count from), `pairCnt` pairs; quad i of the tree is written
// to quads[quadBase + i] and the interior references inside become quadBase-relative.  Enqueued on `stream`.
int rtk_build_quads(const void* pairs, uint32_t pairBase, uint32_t pairCnt, void* quads, uint32_t quadBase, void* stream);

// Tile gather (kernels/rt_tiles.hip): the pixels of `tileCnt` 8x8 tiles (frame tile indices in device memory) out of a
// float4 surface into a packed buffer (64 pixels per tile; float4, or 8-bit RGBA by the dump rule with `quantize`),
// and from such a buffer into a surface (float4, or 8-bit RGBA with `quantized`).  Enqueued on `stream`; hipError_t.
int rtk_pack_tiles(const void* surface, size_t pitch, int width, int height, const int32_t* tiles, int tileCnt, void* out, int quantize, void* stream);
int rtk_scatter_tiles(const void* in, const int32_t* tiles, int tileCnt, void* surface, size_t pitch, int width, int height, int quantized, void* stream);

// Device address and size of the CURRENT device's copy of the library's static scratch array (the workspace of
// rt_render_noalloc).  Returns a hipError_t value.
int rtk_scratch(void** devPtr, size_t* bytes);

// Resident wavefronts per CU of the trace kernels for a given LDS stack size.
int rtk_trace_waves_per_cu(int stackLds);

#endif
