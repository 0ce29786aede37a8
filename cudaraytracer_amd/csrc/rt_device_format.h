// Device-side scene format shared by the packer (rt_capi.cpp) and the kernels
// (kernels/rt_kernels.hip).  A scene lives in ONE contiguous, relocatable blob:
// every reference inside it is a byte offset from the blob start, so the blob can
// be copied between GPUs (RCCL broadcast) and used as is.
//
//   [BlobHeader 256 B][lights][DevShape x shapeCnt][scene nodes 32 B each]
//   [DevMesh x meshCnt] then per mesh: [mesh nodes 32 B][triangle records 48 B]
//   [vertices 56 B][diffuse RGBA8][normals RGBA8][specular RGBA8]
//
// Every section starts on a 256-byte boundary.
#ifndef RT_DEVICE_FORMAT_H
#define RT_DEVICE_FORMAT_H

#include <stdint.h>

#include "../../include/rt_capi.h"

#define RT_BLOB_MAGIC 0x52544246u /* "RTBF" */
#define RT_BLOB_VERSION 1u
#define RT_TILE 8 /* 8x8 pixels = one 64-lane wavefront */

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
    uint8_t reserved[256 - 80];
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
    int32_t pad[8];
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

// Arguments of one render launch (passed by value as the kernel argument).
struct RenderParams {
    const unsigned char* blob;
    uint64_t offLights, offShapes, offSceneNodes, offMeshes;
    int32_t lightCnt, shapeCnt;
    RtDeviceOpts opts;
    float orig[3];
    float camera[16];
    unsigned char* surface;
    uint64_t pitch;
    int32_t* hitKeys;
    unsigned long long* counters;
    unsigned int* tileQueue;  // zeroed before the launch
    int32_t tilesX, tilesY;
    int32_t tileFirst, tileStride, tileCnt;  // this launch renders tiles tileFirst + k*tileStride, k < tileCnt
};

// Launcher implemented in kernels/rt_kernels.hip.  `secondary` selects the
// variant with the bounce stack, `count` the one that tallies test counters.
// Returns a hipError_t value.
int rtk_launch_render(const RenderParams* p, int secondary, int count, int gridBlocks, void* stream);
// Waves that fit the device for the chosen variant (persistent grid size).
int rtk_persistent_blocks(int secondary, int count, int cuCount);

#endif
