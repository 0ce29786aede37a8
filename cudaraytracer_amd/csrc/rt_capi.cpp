// rt_capi.cpp - implementation of the C-ABI declared in include/rt_capi.h:
// context / stream management, flattening of an RtSceneDesc into the device
// blob (rt_device_format.h), uploads, and the per-frame launch of the render
// kernel.  Host code only; the kernels live in kernels/rt_kernels.hip.
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/rt_capi.h"
#include "host/BVH.h"
#include "rt_device_format.h"

#ifndef RT_LDS_STACK_DEFAULT
#define RT_LDS_STACK_DEFAULT 8
#endif

namespace {

thread_local std::string g_createError;

// Events of one rt_render call: frame start, after the level-0 closest-hit trace,
// after shading, after the shadow trace, frame end.
struct TimedLaunch {
    hipEvent_t ev[5];
};

// Per-context frame workspace (see FrameParams); grown on demand, never shrunk.
struct Workspace {
    int nodeCap = 0;
    int lightCap = 0;
    int secCap = 0;
    void* ctl = nullptr;
    void* hit = nullptr;
    void* hitTri = nullptr;
    void* color = nullptr;
    void* secRay = nullptr;
    void* child = nullptr;
    void* kr = nullptr;
    void* kind = nullptr;
    void* ctx = nullptr;
    void* ctxNode = nullptr;
    void* vis = nullptr;
    void* stackSpill = nullptr;
    size_t spillBytes = 0;
    size_t guardBytes = 0;  // guard band behind every array (rt_set_workspace_guards), filled with kGuardByte
};
constexpr int kGuardByte = 0xA5;

// How meshes without BVH nodes get theirs on the device (rt_set_device_bvh).
struct BuildParams {
    int threshold = 10;
    int tightBounds = 1;
    bool rebuildAll = false;  // rt_set_upload_rebuild: also meshes that arrive WITH nodes get their BVH on the device
    int splitRule = RT_BVH_MEDIAN_CUT;  // rt_set_device_bvh_rule
};
// true if this mesh's BVH is built on the device at upload (its node array, if it has one, is not looked at)
inline bool builtOnDevice(const RtMesh& m, const BuildParams* build) { return (m.nodeCnt == 0 && !m.nodes) || (build && build->rebuildAll); }

}  // namespace

// One frame in flight: its own stream and workspace, so that consecutive frames can overlap.
struct FrameSlot {
    Workspace ws;
    hipStream_t stream = nullptr;  // created on first use; slot 0 of a strictly ordered context uses the context's stream
    hipEvent_t ready = nullptr;    // recorded on the context's stream when the frame is issued
    hipEvent_t done = nullptr;     // recorded behind the frame's last kernel
    hipStream_t lastStream = nullptr;  // stream the slot's latest frame ran on
    bool issued = false;
    bool mayOverflow = false;  // a frame with reflection / refraction rays ran on this slot since the last rt_sync
};

struct rt_ctx {
    int device = 0;
    int cuCount = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    FrameSlot slots[RT_MAX_FRAMES_IN_FLIGHT];
    int framesInFlight = 1;
    unsigned long long frameCounter = 0;  // frames issued so far
    int pathBudget = 3;  // node arrays hold pathBudget x pixels when the scene has mirror / glass materials
    int splitSlots = 100000;              // rt_set_box_test_split
    bool skipUnlit = false;               // rt_set_unlit_shadow_skip
    int ldsStack = RT_LDS_STACK_DEFAULT;  // traversal stack entries kept in LDS (older ones move to global memory)
    int maxWavesPerCu = 0;                // 0 = as many as are resident
    uint32_t sampleSeed = 1234;
    // rt_set_tile_list: the tiles of the following calls, on the device, in the caller's order
    int32_t* tileList = nullptr;
    int tileListCnt = 0, tileListMax = -1;
    uint32_t* tileCost = nullptr;  // rt_set_tile_cost_buffer
    // rt_set_accumulation: running sum surface and the samples it holds
    unsigned char* accum = nullptr;
    size_t accumPitch = 0;
    uint32_t accumSamples = 0;  // upper bound of any pixel's count (every call adds its samples), for the 2^24 limit
    bool accumFresh = false;    // a new accumulation: the next call clears the surface before it renders
    uint32_t overflowSeen = 0;  // overflow word of a workspace that was regrown before rt_sync could read it
    BuildParams bvhBuild;  // for meshes that arrive without BVH nodes
    bool uploadRefit = false;  // rt_set_upload_refit: boxes of the trees a scene arrives with are recomputed from its geometry
    int tileOrderMode = RT_TILE_ORDER_CENTER_OUT;
    // tile orders already on the device, by (tile grid, shard, frame size); a table is never changed once made
    struct TileOrder {
        long long key[6];
        void* dev;
    };
    std::vector<TileOrder> tileOrders;
    int wavesPerCu = 0, wavesForLds = -1;
    // pinned staging for rt_surface_download: two chunks, so that the copy from the device overlaps the copy to the caller
    void* staging[2] = {nullptr, nullptr};
    hipEvent_t stagingDone[2] = {nullptr, nullptr};
    hipStream_t copyStream = nullptr;
    std::string error;
    size_t guardBytes = 0;  // rt_set_workspace_guards
    // rt_render_noalloc: this device's copy of the library's static scratch array, and whether a frame that can
    // overflow its path budget ran in it since the last rt_sync
    unsigned char* scratch = nullptr;
    size_t scratchBytes = 0;
    bool scratchMayOverflow = false;
    // Events of the launches since the last rt_timing_reset.  At most kMaxTimedLaunches sets exist; a caller that
    // never resets keeps rendering, and the timing then covers the latest kMaxTimedLaunches launches.
    static constexpr size_t kMaxTimedLaunches = 4096;
    std::vector<TimedLaunch> launches;
    size_t launchesUsed = 0;
};

struct rt_scene {
    rt_ctx* ctx = nullptr;
    unsigned char* blob = nullptr;
    bool ownsBlob = false;
    size_t bytes = 0;
    BlobHeader header;
    // quad records of all BVHs (rt_device_format.h), followed by one root reference per mesh; owned by the scene
    unsigned char* quads = nullptr;
    size_t quadCnt = 0;
};

namespace {

int fail(rt_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->error = buf;
    else
        g_createError = buf;
    return code;
}

#define RT_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

int bind(rt_ctx* ctx) {
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return RT_OK;
}

size_t alignUp(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Fills fresh device memory and returns only when the fill has finished.  (A plain hipMemset runs on the NULL stream,
// which is not ordered against the non-blocking streams the frames run on and need not have finished when it
// returns: a late fill would wipe what a frame has written meanwhile.)
// Host-to-device copy that has reached the device when it returns (a blocking hipMemcpy from pageable memory is
// allowed to return once the data is staged; the frames' non-blocking streams would not wait for the rest).
hipError_t uploadNow(void* dev, const void* host, size_t bytes, hipStream_t stream) {
    hipError_t e = hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream);
    return e != hipSuccess ? e : hipStreamSynchronize(stream);
}

hipError_t fillNow(void* dev, int value, size_t bytes, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(dev, value, bytes, stream);
    return e != hipSuccess ? e : hipStreamSynchronize(stream);
}

// Levels below the root of a flattened tree; also validates child / primitive ranges.
int treeDepth(const RtLinearNode* nodes, int nodeCnt, int primCnt, int* maxLeafPrims, std::string* err, bool checkBoxes = true) {
    struct Item {
        int idx, level;
    };
    std::vector<Item> todo;
    todo.push_back({0, 0});
    int depth = 0;
    size_t visited = 0;
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        if (it.idx < 0 || it.idx >= nodeCnt || ++visited > (size_t)nodeCnt) {
            *err = "BVH node index out of range or cyclic";
            return -1;
        }
        depth = std::max(depth, it.level);
        const RtLinearNode& n = nodes[it.idx];
        if (n.primCnt > 0) {
            if (n.offset < 0 || (long long)n.offset + n.primCnt > primCnt) {
                *err = "BVH leaf references primitives out of range";
                return -1;
            }
            if (maxLeafPrims) *maxLeafPrims = std::max(*maxLeafPrims, (int)n.primCnt);
        } else {
            if (n.axis > 2) {
                *err = "BVH interior node with bad axis";
                return -1;
            }
            // The walk over quads (rt_device_format.h) tests grandchildren without their parents: a node's box must hold
            // its children's boxes, as the exact union the reference's builder makes it (cudaTracer/BVH.h:90-96) does.
            if (checkBoxes && n.offset > it.idx && n.offset < nodeCnt)
                for (int child : {it.idx + 1, (int)n.offset})
                    for (int k = 0; k < 3; ++k)
                        if (!(n.bbmin[k] <= nodes[child].bbmin[k] && nodes[child].bbmax[k] <= n.bbmax[k])) {
                            *err = "BVH node whose box does not contain its child's box";
                            return -1;
                        }
            todo.push_back({it.idx + 1, it.level + 1});
            todo.push_back({n.offset, it.level + 1});
        }
    }
    if (visited != (size_t)nodeCnt) {
        // the sizes of the device-side node arrays follow from the count
        *err = "BVH node array holds nodes the root does not reach";
        return -1;
    }
    return depth;
}

// Re-lays a depth-first LinearNode tree out as sibling pairs (see rt_device_format.h).
// `out` receives 32-byte records {bbmin, bbmax, ref, 0}; its size is 2 * (interior nodes + 1).
// References are written relative to this tree (pair index / 2, primitive index within the
// tree's primitive list) and made absolute by relocateRefs once the blob offsets are known.
int pairLayout(const RtLinearNode* nodes, std::vector<RtLinearNode>& out, uint32_t leafTag, std::string* err,
               std::vector<int32_t>* recordOfNode = nullptr) {
    struct Item {
        int src, dst;
    };
    out.assign(2, RtLinearNode{});
    std::vector<Item> todo;
    todo.push_back({0, 0});
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        const RtLinearNode& n = nodes[it.src];
        RtLinearNode rec = n;
        uint32_t ref;
        if (n.primCnt > 0) {
            if (n.primCnt > RT_REF_MAX_LEAF_PRIMS) {
                *err = "BVH leaf too large for the packed node reference (more than 63 primitives in a leaf)";
                return RT_ERR_LIMIT;
            }
            ref = RT_REF_LEAF | leafTag | ((uint32_t)n.primCnt << 24) | (uint32_t)n.offset;
        } else {
            const size_t pair = out.size();
            if (pair / 2 >= (1u << 28)) {
                *err = "BVH too large for the packed node reference";
                return RT_ERR_LIMIT;
            }
            out.resize(pair + 2);
            ref = ((uint32_t)n.axis << 29) | (uint32_t)(pair >> 1);
            // second child first on the work stack: the first child's subtree is laid out right behind the pair
            todo.push_back({n.offset, (int)pair + 1});
            todo.push_back({it.src + 1, (int)pair});
        }
        std::memcpy(&rec.offset, &ref, 4);
        rec.primCnt = 0;
        rec.axis = 0;
        rec.pad = 0;
        out[it.dst] = rec;
        if (recordOfNode) (*recordOfNode)[it.src] = it.dst;
    }
    return RT_OK;
}

// Makes the references of a laid-out tree absolute: child pairs in 64-byte units from the blob
// start (`unitBase` = blob offset of the tree / 64), leaf primitives shifted by `primBase`.
int relocateRefs(std::vector<RtLinearNode>& recs, uint64_t unitBase, uint32_t primBase, std::string* err) {
    for (size_t i = 0; i < recs.size(); ++i) {
        if (i == 1) continue;  // padding next to the root
        uint32_t ref;
        std::memcpy(&ref, &recs[i].offset, 4);
        if (RT_REF_IS_LEAF(ref)) {
            const uint64_t first = (uint64_t)RT_REF_LEAF_FIRST(ref) + primBase;
            if (first + RT_REF_LEAF_COUNT(ref) > (uint64_t)RT_REF_MAX_PRIMS) {
                *err = "scene too large for the packed node reference (16M primitives)";
                return RT_ERR_LIMIT;
            }
            ref = (ref & 0xff000000u) | (uint32_t)first;
        } else {
            const uint64_t unit = (uint64_t)RT_REF_PAIR_UNITS(ref) + unitBase;
            if (unit >= (1u << 29)) {
                *err = "scene blob too large for the packed node reference (32 GB)";
                return RT_ERR_LIMIT;
            }
            ref = (ref & 0x60000000u) | (uint32_t)unit;
        }
        std::memcpy(&recs[i].offset, &ref, 4);
    }
    return RT_OK;
}

// part(k, parts) for k = 0 .. parts - 1, on up to 16 threads when there is enough to do; parts that get no thread (a
// process limit) run on the calling thread.
void onThreads(size_t items, const std::function<void(int, int)>& part) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int parts = items < ((size_t)1 << 16) ? 1 : (int)std::min(hw ? hw : 1u, 16u);
    std::vector<std::thread> pool;
    std::vector<int> inlineParts;
    for (int k = 1; k < parts; ++k) {
        try {
            pool.emplace_back([&part, k, parts] { part(k, parts); });
        } catch (const std::system_error&) {
            inlineParts.push_back(k);
        }
    }
    part(0, parts);
    for (int k : inlineParts) part(k, parts);
    for (std::thread& t : pool) t.join();
}

// A mesh that arrives without BVH nodes gets its BVH on the device.  The tree's shape (median cuts down to
// `threshold` primitives, csrc/host/BVH.h) follows from the count alone: this writes the depth-first node array
// with references and leaf ranges but without boxes and axes, and the per-node facts the device build needs.
struct StaticNodeInfo {
    int32_t start, end, level;
    uint32_t pathId;
};
struct DeviceBuild {
    int mesh = 0;
    RtBvhBuildJob job;
    std::vector<RtBuildNode> table;
    uint64_t offNodes = 0, offRecords = 0, offVertices = 0;
};
int staticTree(int n, int threshold, std::vector<RtLinearNode>& dfs, std::vector<StaticNodeInfo>& info, int s, int e, int level, uint32_t pathId) {
    const int idx = (int)dfs.size();
    dfs.push_back(RtLinearNode{});
    info.push_back(StaticNodeInfo{s, e, level, pathId});
    if (e - s <= threshold) {
        // the host builder descends into the second half first, so ordered slots count from the far end
        dfs[idx].offset = n - e;
        dfs[idx].primCnt = (uint16_t)(e - s);
    } else {
        const int mid = (s + e) / 2;
        staticTree(n, threshold, dfs, info, s, mid, level + 1, 2u * pathId);
        const int second = staticTree(n, threshold, dfs, info, mid, e, level + 1, 2u * pathId + 1u);
        dfs[idx].offset = second;
        dfs[idx].primCnt = 0;
    }
    return idx;
}

// Bounds refit on the host (rt_bvh_refit, and rt_scene_pack under rt_set_upload_refit): the same boxes, bit for bit, as
// the device kernels of rt_bvh_build.hip produce.  `nodes` is a validated depth-first array, in which every child
// sits behind its parent: one sweep from the back is bottom-up.
// (minimum / maximum as comparisons: of +0 and -0 the one seen first stays, on the host as on the device)
inline float refitMin(float acc, float x) { return x < acc ? x : acc; }
inline float refitMax(float acc, float x) { return x > acc ? x : acc; }
void refitNodes(RtLinearNode* nodes, int nodeCnt, const RtVertex* vertices, const RtTriangle* triangles, const RtShape* shapes) {
    for (int i = nodeCnt - 1; i >= 0; --i) {
        RtLinearNode& n = nodes[i];
        if (n.primCnt > 0) {
            float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (int t = n.offset; t < n.offset + (int)n.primCnt; ++t) {
                if (triangles) {
                    for (int c = 0; c < 3; ++c) {
                        const float* pos = vertices[triangles[t].i[c]].position;
                        for (int a = 0; a < 3; ++a) {
                            lo[a] = refitMin(lo[a], pos[a]);
                            hi[a] = refitMax(hi[a], pos[a]);
                        }
                    }
                } else {
                    for (int a = 0; a < 3; ++a) {
                        lo[a] = refitMin(lo[a], shapes[t].bbmin[a]);
                        hi[a] = refitMax(hi[a], shapes[t].bbmax[a]);
                    }
                }
            }
            std::memcpy(n.bbmin, lo, sizeof lo);
            std::memcpy(n.bbmax, hi, sizeof hi);
        } else {
            const RtLinearNode &a = nodes[i + 1], &b = nodes[n.offset];
            for (int k = 0; k < 3; ++k) {
                n.bbmin[k] = refitMin(a.bbmin[k], b.bbmin[k]);
                n.bbmax[k] = refitMax(a.bbmax[k], b.bbmax[k]);
            }
        }
    }
}

// Flattens `desc` into a host blob.  Returns RT_OK or an error (message in *err).  `refit`: the boxes of all trees
// are recomputed from the geometry first (refitNodes); the caller's arrays are not touched.
int packScene(const RtSceneDesc* d, std::vector<unsigned char>& out, std::string* err, const BuildParams* build = nullptr,
              std::vector<DeviceBuild>* builds = nullptr, bool refit = false, bool boxesWillBeRefit = false) {
    const bool checkBoxes = !refit && !boxesWillBeRefit;  // a refit makes the boxes nest whatever they were
    auto bad = [&](const char* m) {
        *err = m;
        return RT_ERR_INVALID;
    };
    if (!d) return bad("null scene description");
    if (d->shapeCnt < 1 || !d->shapes) return bad("scene without shapes");
    if (d->lightCnt < 0 || (d->lightCnt > 0 && !d->lights)) return bad("bad light table");
    if (d->nodeCnt < 1 || !d->nodes) return bad("scene without BVH nodes");
    if (d->meshCnt < 0 || (d->meshCnt > 0 && !d->meshes)) return bad("bad mesh table");

    BlobHeader hd;
    std::memset(&hd, 0, sizeof hd);
    hd.magic = RT_BLOB_MAGIC;
    hd.version = RT_BLOB_VERSION;
    hd.lightCnt = d->lightCnt;
    hd.shapeCnt = d->shapeCnt;
    hd.meshCnt = d->meshCnt;
    hd.sceneNodeCnt = d->nodeCnt;

    int maxLeafShapes = 0;
    hd.sceneDepth = treeDepth(d->nodes, d->nodeCnt, d->shapeCnt, &maxLeafShapes, err, checkBoxes);
    if (hd.sceneDepth < 0) return RT_ERR_INVALID;
    hd.maxLeafShapes = maxLeafShapes;

    size_t cursor = sizeof(BlobHeader);
    auto reserve = [&](size_t bytes) {
        size_t at = alignUp(cursor);
        cursor = at + bytes;
        return at;
    };
    hd.offLights = reserve((size_t)std::max(d->lightCnt, 1) * sizeof(RtLight));
    hd.offShapes = reserve((size_t)d->shapeCnt * sizeof(DevShape));
    std::vector<RtLinearNode> scenePairs;
    std::vector<RtLinearNode> refitted;  // a tree's nodes with recomputed boxes, while it is laid out
    const RtLinearNode* sceneNodes = d->nodes;
    if (refit) {
        refitted.assign(d->nodes, d->nodes + d->nodeCnt);
        refitNodes(refitted.data(), d->nodeCnt, nullptr, nullptr, d->shapes);
        sceneNodes = refitted.data();
    }
    if (int r = pairLayout(sceneNodes, scenePairs, RT_REF_SHAPES, err)) return r;
    hd.offSceneNodes = reserve(scenePairs.size() * sizeof(RtLinearNode));
    hd.offMeshes = reserve((size_t)std::max(d->meshCnt, 1) * sizeof(DevMesh));
    // one record array for all meshes (+8 spare records), so that a leaf can name its triangles absolutely
    size_t recordCnt = 0;
    for (int mi = 0; mi < d->meshCnt; ++mi) recordCnt += (size_t)std::max(d->meshes[mi].triangleCnt, 0);
    if (recordCnt > (size_t)RT_REF_MAX_PRIMS) {
        *err = "scene too large for the packed node reference (16M triangles)";
        return RT_ERR_LIMIT;
    }
    hd.recordCnt = (uint32_t)recordCnt;
    hd.offRecords = reserve((recordCnt + 8) * sizeof(TriRecord));
    if (int r = relocateRefs(scenePairs, hd.offSceneNodes / 64, 0, err)) return r;
    hd.sceneRootRef = (RT_REF_ROOT_AXIS << 29) | (uint32_t)(hd.offSceneNodes / 64);
    uint32_t recordBase = 0;

    std::vector<DevMesh> meshes(std::max(d->meshCnt, 0));
    std::vector<std::vector<RtLinearNode>> meshPairs(std::max(d->meshCnt, 0));
    for (int mi = 0; mi < d->meshCnt; ++mi) {
        const RtMesh& m = d->meshes[mi];
        DevMesh& dm = meshes[mi];
        std::memset(&dm, 0, sizeof dm);
        if (m.triangleCnt < 1 || !m.triangles || m.vertexCnt < 1 || !m.vertices) return bad("mesh with missing triangles or vertices");
        const bool onDevice = builtOnDevice(m, builds ? build : nullptr);
        if (onDevice && !(build && builds)) return bad("mesh without BVH nodes: only rt_scene_upload can build them (on the device)");
        if (!onDevice && (m.nodeCnt < 1 || !m.nodes)) return bad("mesh with an inconsistent node array");
        for (int t = 0; t < m.triangleCnt; ++t)
            for (int k = 0; k < 3; ++k)
                if (m.triangles[t].i[k] < 0 || m.triangles[t].i[k] >= m.vertexCnt) return bad("triangle vertex index out of range");
        std::vector<RtLinearNode> staticNodes;
        std::vector<StaticNodeInfo> staticInfo;
        std::vector<int32_t> recordOfNode;
        int depth;
        if (onDevice) {
            staticTree(m.triangleCnt, build->threshold, staticNodes, staticInfo, 0, m.triangleCnt, 0, 1u);
            depth = 0;
            for (const StaticNodeInfo& si : staticInfo) depth = std::max(depth, (int)si.level);
            recordOfNode.assign(staticNodes.size(), -1);
        } else {
            depth = treeDepth(m.nodes, m.nodeCnt, m.triangleCnt, nullptr, err, checkBoxes);
            if (depth < 0) return RT_ERR_INVALID;
        }
        hd.maxMeshDepth = std::max(hd.maxMeshDepth, depth);
        std::memcpy(dm.inv, m.transformInverse, sizeof dm.inv);
        dm.triangleCnt = m.triangleCnt;
        dm.nodeCnt = onDevice ? (int)staticNodes.size() : m.nodeCnt;
        dm.vertexCnt = m.vertexCnt;
        dm.depth = depth;
        const RtLinearNode* meshNodes = onDevice ? staticNodes.data() : m.nodes;
        if (refit && !onDevice) {
            refitted.assign(m.nodes, m.nodes + m.nodeCnt);
            refitNodes(refitted.data(), m.nodeCnt, m.vertices, m.triangles, nullptr);
            meshNodes = refitted.data();
        }
        if (int r = pairLayout(meshNodes, meshPairs[mi], 0u, err, onDevice ? &recordOfNode : nullptr)) return r;
        dm.offNodes = reserve(meshPairs[mi].size() * sizeof(RtLinearNode));
        dm.recordBase = recordBase;
        dm.offTriRecords = hd.offRecords + (uint64_t)recordBase * sizeof(TriRecord);
        if (int r = relocateRefs(meshPairs[mi], dm.offNodes / 64, recordBase, err)) return r;
        dm.rootRef = (RT_REF_ROOT_AXIS << 29) | (uint32_t)(dm.offNodes / 64);
        recordBase += (uint32_t)m.triangleCnt;
        if (onDevice) {
            builds->emplace_back();
            DeviceBuild& b = builds->back();
            b.mesh = mi;
            std::memset(&b.job, 0, sizeof b.job);
            b.job.triangleCnt = m.triangleCnt;
            b.job.threshold = build->threshold;
            b.job.tightBounds = build->tightBounds;
            b.job.depth = depth;
            // leaves first, then interior nodes by level, deepest level first
            for (size_t k = 0; k < staticNodes.size(); ++k)
                if (staticNodes[k].primCnt > 0)
                    b.table.push_back(RtBuildNode{staticInfo[k].start, staticInfo[k].end, recordOfNode[k], -1, -1, staticInfo[k].pathId});
            b.job.leafCnt = (int32_t)b.table.size();
            for (int level = depth - 1; level >= 0; --level)
                for (size_t k = 0; k < staticNodes.size(); ++k)
                    if (staticNodes[k].primCnt == 0 && staticInfo[k].level == level) {
                        b.table.push_back(RtBuildNode{staticInfo[k].start, staticInfo[k].end, recordOfNode[k], recordOfNode[k + 1],
                                                      recordOfNode[staticNodes[k].offset], staticInfo[k].pathId});
                        b.job.interiorPerLevel[level]++;
                    }
            b.job.nodeCnt = (int32_t)b.table.size();
        }
        dm.offVertices = reserve((size_t)m.vertexCnt * sizeof(RtVertex));
        const RtTexture* tex[3] = {&m.diffuse, &m.normals, &m.specular};
        for (int k = 0; k < 3; ++k) {
            if (tex[k]->width < 1 || tex[k]->height < 1 || !tex[k]->data) return bad("mesh texture missing");
            if (tex[k]->mode != RT_TEX_WRAP && tex[k]->mode != RT_TEX_CLAMP) return bad("bad texture addressing mode");
            dm.texW[k] = tex[k]->width;
            dm.texH[k] = tex[k]->height;
            dm.texMode[k] = tex[k]->mode;
            dm.offTex[k] = reserve((size_t)tex[k]->width * tex[k]->height * 4);
        }
    }
    if (hd.sceneDepth + hd.maxMeshDepth + 2 > RT_STACK_ENTRIES) {
        *err = "BVH too deep for the kernel's traversal stack (scene depth + mesh depth + 2 > RT_STACK_ENTRIES)";
        return RT_ERR_LIMIT;
    }
    hd.totalBytes = alignUp(cursor);
    out.assign(hd.totalBytes, 0);
    unsigned char* base = out.data();

    if (d->lightCnt) std::memcpy(base + hd.offLights, d->lights, (size_t)d->lightCnt * sizeof(RtLight));
    std::memcpy(base + hd.offSceneNodes, scenePairs.data(), scenePairs.size() * sizeof(RtLinearNode));
    DevShape* shapes = reinterpret_cast<DevShape*>(base + hd.offShapes);
    for (int si = 0; si < d->shapeCnt; ++si) {
        const RtShape& s = d->shapes[si];
        DevShape& o = shapes[si];
        if (s.kind != RT_SHAPE_SPHERE && s.kind != RT_SHAPE_MESH) return bad("shape kind not set");
        if (s.kind == RT_SHAPE_MESH && (s.meshIdx < 0 || s.meshIdx >= d->meshCnt)) return bad("shape mesh index out of range");
        o.kind = s.kind;
        std::memcpy(o.center, s.center, sizeof o.center);
        o.radius2 = s.radius2;
        o.meshIdx = s.kind == RT_SHAPE_MESH ? s.meshIdx : -1;
        o.tag = s.tag;
        o.materialType = s.material.materialType;
        o.ior = s.material.ior;
        o.Ks = s.material.Ks;
        o.specularExponent = s.material.specularExponent;
        std::memcpy(o.diffuseColor, s.material.diffuseColor, sizeof o.diffuseColor);
        if (o.materialType != RT_MAT_DIFFUSE_AND_GLOSSY) hd.hasSecondary = 1;
    }
    for (int mi = 0; mi < d->meshCnt; ++mi) {
        const RtMesh& m = d->meshes[mi];
        const DevMesh& dm = meshes[mi];
        std::memcpy(base + dm.offNodes, meshPairs[mi].data(), meshPairs[mi].size() * sizeof(RtLinearNode));
        // (large meshes: vertices and records on several threads - two gigabytes for ten million de-indexed triangles)
        const size_t vertexBytes = (size_t)m.vertexCnt * sizeof(RtVertex);
        onThreads((size_t)m.vertexCnt, [&](int k, int parts) {
            const size_t from = vertexBytes * k / parts, to = vertexBytes * (k + 1) / parts;
            std::memcpy(base + dm.offVertices + from, reinterpret_cast<const unsigned char*>(m.vertices) + from, to - from);
        });
        TriRecord* rec = reinterpret_cast<TriRecord*>(base + dm.offTriRecords);
        const bool onDevice = builtOnDevice(m, builds ? build : nullptr);  // the device build writes this mesh's records
        if (!onDevice)
            onThreads((size_t)m.triangleCnt, [&](int part, int parts) {
                const size_t from = (size_t)m.triangleCnt * part / parts, to = (size_t)m.triangleCnt * (part + 1) / parts;
                for (size_t t = from; t < to; ++t) {
                    const float* p0 = m.vertices[m.triangles[t].i[0]].position;
                    const float* p1 = m.vertices[m.triangles[t].i[1]].position;
                    const float* p2 = m.vertices[m.triangles[t].i[2]].position;
                    for (int k = 0; k < 3; ++k) {
                        rec[t].v0[k] = p0[k];
                        rec[t].e1[k] = p1[k] - p0[k];  // same fp32 subtraction the reference does per test
                        rec[t].e2[k] = p2[k] - p0[k];
                        rec[t].idx[k] = m.triangles[t].i[k];
                    }
                }
            });
        const RtTexture* tex[3] = {&m.diffuse, &m.normals, &m.specular};
        for (int k = 0; k < 3; ++k)
            std::memcpy(base + dm.offTex[k], tex[k]->data, (size_t)tex[k]->width * tex[k]->height * 4);
    }
    if (d->meshCnt) std::memcpy(base + hd.offMeshes, meshes.data(), meshes.size() * sizeof(DevMesh));
    if (builds)
        for (DeviceBuild& b : *builds) {
            b.offNodes = meshes[b.mesh].offNodes;
            b.offRecords = meshes[b.mesh].offTriRecords;
            b.offVertices = meshes[b.mesh].offVertices;
        }
    std::memcpy(base, &hd, sizeof hd);
    return RT_OK;
}

void freeWorkspace(Workspace& w) {
    void* all[] = {w.ctl, w.hit, w.hitTri, w.color, w.secRay, w.child, w.kr, w.kind, w.ctx, w.ctxNode, w.vis, w.stackSpill};
    for (void* p : all)
        if (p) hipFree(p);
    w = Workspace();
}

// The workspace's arrays and their sizes for `n` path nodes, `sc` of them beyond level 0, `lc` lights.
struct WsArray {
    void* Workspace::*member;
    size_t bytes;
    const char* name;
};
std::vector<WsArray> workspaceArrays(size_t n, size_t sc, size_t lc) {
    return {{&Workspace::ctl, RT_CTL_BYTES, "ctl"}, {&Workspace::hit, n * 16, "hit"},
            {&Workspace::hitTri, n * 4, "hitTri"},                  {&Workspace::color, n * 16, "color"},
            {&Workspace::secRay, std::max<size_t>(sc * 32, 256), "secRay"}, {&Workspace::child, n * 4, "child"},
            {&Workspace::kr, n * 4, "kr"},                          {&Workspace::kind, n * 4, "kind"},
            {&Workspace::ctx, n * 64, "ctx"},                       {&Workspace::ctxNode, n * 4, "ctxNode"},
            {&Workspace::vis, n * lc, "vis"}};
}

// Makes the workspace hold `nodeCap` path nodes (`slots` of them level 0) and `lights` lights.
int ensureWorkspace(rt_ctx* ctx, Workspace& w, hipStream_t stream, int nodeCap, int slots, int lights, size_t spillBytes) {
    const int secCap = nodeCap - slots;
    const size_t guard = ctx->guardBytes;
    if (spillBytes > w.spillBytes || (guard != w.guardBytes && w.stackSpill)) {
        RT_HIP(ctx, hipStreamSynchronize(stream));
        if (w.stackSpill) hipFree(w.stackSpill);
        w.stackSpill = nullptr;
        spillBytes = std::max(spillBytes, w.spillBytes);
        w.spillBytes = 0;
        RT_HIP(ctx, hipMalloc(&w.stackSpill, spillBytes + guard));
        if (guard) RT_HIP(ctx, fillNow(static_cast<unsigned char*>(w.stackSpill) + spillBytes, kGuardByte, guard, stream));
        w.spillBytes = spillBytes;
    }
    if (w.ctl && nodeCap <= w.nodeCap && lights <= w.lightCap && secCap <= w.secCap && guard == w.guardBytes) return RT_OK;
    RT_HIP(ctx, hipStreamSynchronize(stream));  // the slot's previous frame may still use the old arrays
    if (w.ctl) {
        // the sticky overflow word leaves with the old control block: keep it for rt_sync
        uint32_t word = 0;
        RT_HIP(ctx, hipMemcpy(&word, static_cast<uint32_t*>(w.ctl) + CTL_OVERFLOW, sizeof word, hipMemcpyDeviceToHost));
        ctx->overflowSeen |= word;
    }
    const int nc = std::max(nodeCap, w.nodeCap), lc = std::max(std::max(lights, 1), w.lightCap), sc = std::max(secCap, w.secCap);
    {
        void* keepSpill = w.stackSpill;
        const size_t keepBytes = w.spillBytes;
        w.stackSpill = nullptr;
        freeWorkspace(w);
        w.stackSpill = keepSpill;
        w.spillBytes = keepBytes;
    }
    for (const WsArray& a : workspaceArrays((size_t)nc, (size_t)sc, (size_t)lc)) {
        RT_HIP(ctx, hipMalloc(&(w.*a.member), a.bytes + guard));
        if (guard) RT_HIP(ctx, fillNow(static_cast<unsigned char*>(w.*a.member) + a.bytes, kGuardByte, guard, stream));
    }
    RT_HIP(ctx, fillNow(w.ctl, 0, CTL_WORDS * sizeof(uint32_t), stream));
    w.nodeCap = nc;
    w.lightCap = lc;
    w.secCap = sc;
    w.guardBytes = guard;
    return RT_OK;
}

// Waits (on the host) for every frame issued so far.
int syncFrames(rt_ctx* ctx) {
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (FrameSlot& sl : ctx->slots)
        if (sl.issued && sl.lastStream) RT_HIP(ctx, hipStreamSynchronize(sl.lastStream));
    return RT_OK;
}

int checkHeader(rt_ctx* ctx, const BlobHeader& hd, size_t bytes) {
    if (hd.magic != RT_BLOB_MAGIC || hd.version != RT_BLOB_VERSION) return fail(ctx, RT_ERR_INVALID, "not a scene blob");
    if (hd.totalBytes != bytes) return fail(ctx, RT_ERR_INVALID, "scene blob size mismatch");
    if (hd.sceneDepth + hd.maxMeshDepth + 2 > RT_STACK_ENTRIES) return fail(ctx, RT_ERR_LIMIT, "scene blob exceeds the traversal stack");
    return RT_OK;
}

}  // namespace

extern "C" {

rt_ctx* rt_create(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1) {
        fail(nullptr, RT_ERR_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= count) {
        fail(nullptr, RT_ERR_INVALID, "device %d out of range (0..%d)", device, count - 1);
        return nullptr;
    }
    rt_ctx* ctx = new rt_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking) != hipSuccess) {
        fail(nullptr, RT_ERR_DEVICE, "cannot initialise device %d", device);
        delete ctx;
        return nullptr;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, RT_ERR_DEVICE, "device %d is %s; this library only carries gfx950 code", device, prop.gcnArchName);
        hipStreamDestroy(ctx->ownStream);
        delete ctx;
        return nullptr;
    }
    ctx->cuCount = prop.multiProcessorCount;
    ctx->stream = ctx->ownStream;
    // this device's copy of the static scratch array (rt_render_noalloc); looked up once, with the device current
    void* scratch = nullptr;
    size_t scratchBytes = 0;
    const int se = rtk_scratch(&scratch, &scratchBytes);
    if (se == 0) {
        ctx->scratch = static_cast<unsigned char*>(scratch);
        ctx->scratchBytes = scratchBytes;
    } else {
        // only rt_render_noalloc needs it, and says so when called
        fail(ctx, RT_ERR_DEVICE, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString((hipError_t)se));
    }
    return ctx;
}

void rt_destroy(rt_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& l : ctx->launches)
        for (hipEvent_t e : l.ev) hipEventDestroy(e);
    for (FrameSlot& sl : ctx->slots) {
        if (sl.lastStream) hipStreamSynchronize(sl.lastStream);
        freeWorkspace(sl.ws);
        if (sl.stream) hipStreamDestroy(sl.stream);
        if (sl.ready) hipEventDestroy(sl.ready);
        if (sl.done) hipEventDestroy(sl.done);
    }
    for (auto& t : ctx->tileOrders)
        if (t.dev) hipFree(t.dev);
    if (ctx->tileList) hipFree(ctx->tileList);
    for (int k = 0; k < 2; ++k) {
        if (ctx->staging[k]) hipHostFree(ctx->staging[k]);
        if (ctx->stagingDone[k]) hipEventDestroy(ctx->stagingDone[k]);
    }
    if (ctx->copyStream) hipStreamDestroy(ctx->copyStream);
    hipStreamDestroy(ctx->ownStream);
    delete ctx;
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_createError.c_str(); }

int rt_set_stream(rt_ctx* ctx, void* hipStream) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->stream = hipStream ? reinterpret_cast<hipStream_t>(hipStream) : ctx->ownStream;
    return RT_OK;
}

int rt_sync(rt_ctx* ctx) {
    if (!ctx) return RT_ERR_INVALID;
    if (int r = bind(ctx)) return r;
    if (int r = syncFrames(ctx)) return r;
    uint32_t overflow = ctx->overflowSeen;
    ctx->overflowSeen = 0;
    for (FrameSlot& sl : ctx->slots) {
        // only frames that spawn child rays can run out of path nodes
        if (!sl.ws.ctl || !sl.mayOverflow) continue;
        sl.mayOverflow = false;
        // the word is sticky on the device: it covers every frame the slot ran since the last rt_sync, and is cleared here
        uint32_t word = 0;
        uint32_t* dev = static_cast<uint32_t*>(sl.ws.ctl) + CTL_OVERFLOW;
        RT_HIP(ctx, hipMemcpy(&word, dev, sizeof word, hipMemcpyDeviceToHost));
        if (word) RT_HIP(ctx, fillNow(dev, 0, sizeof word, sl.lastStream ? sl.lastStream : ctx->stream));
        overflow |= word;
    }
    if (ctx->scratchMayOverflow && ctx->scratch) {
        ctx->scratchMayOverflow = false;
        // rt_render_noalloc puts the control block first in the scratch array
        uint32_t word = 0;
        uint32_t* dev = reinterpret_cast<uint32_t*>(ctx->scratch) + CTL_OVERFLOW;
        RT_HIP(ctx, hipMemcpy(&word, dev, sizeof word, hipMemcpyDeviceToHost));
        if (word) RT_HIP(ctx, fillNow(dev, 0, sizeof word, ctx->stream));
        overflow |= word;
    }
    if (overflow)
        return fail(ctx, RT_ERR_LIMIT, "reflection / refraction rays exceeded the path budget of %d x pixels; raise it with rt_set_path_budget",
                    ctx->pathBudget);
    return RT_OK;
}

int rt_set_frames_in_flight(rt_ctx* ctx, int frames) {
    if (!ctx) return RT_ERR_INVALID;
    if (frames < 1 || frames > RT_MAX_FRAMES_IN_FLIGHT) return fail(ctx, RT_ERR_INVALID, "frames in flight must be in 1..%d", RT_MAX_FRAMES_IN_FLIGHT);
    if (int r = bind(ctx)) return r;
    if (int r = syncFrames(ctx)) return r;
    ctx->framesInFlight = frames;
    return RT_OK;
}

int rt_stream_wait_frame(rt_ctx* ctx, void* hipStream, int framesAgo) {
    if (!ctx) return RT_ERR_INVALID;
    if (framesAgo < 0 || framesAgo >= RT_MAX_FRAMES_IN_FLIGHT) return fail(ctx, RT_ERR_INVALID, "framesAgo must be in 0..%d", RT_MAX_FRAMES_IN_FLIGHT - 1);
    if ((unsigned long long)framesAgo >= ctx->frameCounter) return RT_OK;  // nothing issued that far back
    if (framesAgo >= ctx->framesInFlight) return RT_OK;                      // its slot has been reused since: it is complete
    if (int r = bind(ctx)) return r;
    const FrameSlot& sl = ctx->slots[(ctx->frameCounter - 1 - (unsigned long long)framesAgo) % (unsigned long long)ctx->framesInFlight];
    if (!sl.issued || !sl.done) return RT_OK;
    RT_HIP(ctx, hipStreamWaitEvent(hipStream ? reinterpret_cast<hipStream_t>(hipStream) : ctx->stream, sl.done, 0));
    return RT_OK;
}

int rt_set_trace_waves(rt_ctx* ctx, int wavesPerCu) {
    if (!ctx) return RT_ERR_INVALID;
    if (wavesPerCu < 0 || wavesPerCu > 32) return fail(ctx, RT_ERR_INVALID, "waves per CU must be in 0..32");
    ctx->maxWavesPerCu = wavesPerCu;
    return RT_OK;
}

int rt_set_sample_seed(rt_ctx* ctx, uint32_t seed) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->sampleSeed = seed;
    return RT_OK;
}

int rt_set_tile_list(rt_ctx* ctx, const int32_t* tiles, int count) {
    if (!ctx) return RT_ERR_INVALID;
    if (count < 0 || (count > 0 && !tiles)) return fail(ctx, RT_ERR_INVALID, "bad tile list");
    if (int r = bind(ctx)) return r;
    // validate and upload the new list first: a rejected call leaves the current list in place
    void* dev = nullptr;
    int32_t largest = -1;
    if (count > 0) {
        std::vector<int32_t> sorted(tiles, tiles + count);
        std::sort(sorted.begin(), sorted.end());
        if (sorted.front() < 0) return fail(ctx, RT_ERR_INVALID, "negative tile index in the tile list");
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return fail(ctx, RT_ERR_INVALID, "tile %d appears twice in the tile list", *std::adjacent_find(sorted.begin(), sorted.end()));
        largest = sorted.back();
        RT_HIP(ctx, hipMalloc(&dev, (size_t)count * sizeof(int32_t)));
        hipError_t e = uploadNow(dev, tiles, (size_t)count * sizeof(int32_t), ctx->stream);
        if (e != hipSuccess) {
            hipFree(dev);
            return fail(ctx, RT_ERR_DEVICE, "tile list upload: %s", hipGetErrorString(e));
        }
    }
    if (int r = syncFrames(ctx)) {  // frames in flight may still read the old list
        if (dev) hipFree(dev);
        return r;
    }
    if (ctx->tileList) hipFree(ctx->tileList);
    ctx->tileList = static_cast<int32_t*>(dev);
    ctx->tileListCnt = count;
    ctx->tileListMax = largest;
    return RT_OK;
}

int rt_set_tile_cost_buffer(rt_ctx* ctx, uint32_t* devCost) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->tileCost = devCost;
    return RT_OK;
}

int rt_set_accumulation(rt_ctx* ctx, void* devSum, size_t sumPitch, uint32_t samplesSoFar) {
    if (!ctx) return RT_ERR_INVALID;
    if (devSum && (sumPitch % 16 != 0 || (reinterpret_cast<uintptr_t>(devSum) & 15))) return fail(ctx, RT_ERR_INVALID, "accumulation surface must be 16-byte aligned with a pitch that is a multiple of 16");
    if (samplesSoFar > (1u << 24)) return fail(ctx, RT_ERR_LIMIT, "accumulation holds at most 2^24 samples per pixel");
    ctx->accum = static_cast<unsigned char*>(devSum);
    ctx->accumPitch = devSum ? sumPitch : 0;
    ctx->accumSamples = devSum ? samplesSoFar : 0;
    ctx->accumFresh = devSum && samplesSoFar == 0;
    return RT_OK;
}

int rt_set_tile_order(rt_ctx* ctx, int mode) {
    if (!ctx) return RT_ERR_INVALID;
    if (mode != RT_TILE_ORDER_ROW_MAJOR && mode != RT_TILE_ORDER_CENTER_OUT) return fail(ctx, RT_ERR_INVALID, "unknown tile order %d", mode);
    ctx->tileOrderMode = mode;
    return RT_OK;
}

int rt_set_lds_stack(rt_ctx* ctx, int entries) {
    if (!ctx) return RT_ERR_INVALID;
    if (entries < 1 || entries > RT_STACK_ENTRIES) return fail(ctx, RT_ERR_INVALID, "LDS stack entries must be in 1..%d", RT_STACK_ENTRIES);
    ctx->ldsStack = entries;
    return RT_OK;
}

int rt_set_box_test_split(rt_ctx* ctx, int minRays) {
    if (!ctx) return RT_ERR_INVALID;
    if (minRays < 0) return fail(ctx, RT_ERR_INVALID, "the threshold is a number of rays");
    ctx->splitSlots = minRays;
    return RT_OK;
}

int rt_set_unlit_shadow_skip(rt_ctx* ctx, int on) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->skipUnlit = on != 0;
    return RT_OK;
}

int rt_set_workspace_guards(rt_ctx* ctx, size_t bytes) {
    if (!ctx) return RT_ERR_INVALID;
    if (bytes > (1u << 20) || bytes % 256 != 0) return fail(ctx, RT_ERR_INVALID, "guard size must be a multiple of 256 bytes, at most 1 MiB");
    ctx->guardBytes = bytes;
    return RT_OK;
}

int rt_check_workspace_guards(rt_ctx* ctx) {
    if (!ctx) return RT_ERR_INVALID;
    if (int r = bind(ctx)) return r;
    if (int r = syncFrames(ctx)) return r;
    std::vector<unsigned char> host;
    for (int si = 0; si < RT_MAX_FRAMES_IN_FLIGHT; ++si) {
        Workspace& w = ctx->slots[si].ws;
        if (!w.ctl || !w.guardBytes) continue;
        std::vector<WsArray> arrays = workspaceArrays((size_t)w.nodeCap, (size_t)w.secCap, (size_t)w.lightCap);
        arrays.push_back({&Workspace::stackSpill, w.spillBytes, "stackSpill"});
        host.resize(w.guardBytes);
        for (const WsArray& a : arrays) {
            RT_HIP(ctx, hipMemcpy(host.data(), static_cast<unsigned char*>(w.*a.member) + a.bytes, w.guardBytes, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < host.size(); ++i)
                if (host[i] != kGuardByte)
                    return fail(ctx, RT_ERR_DEVICE, "workspace array '%s' of frame slot %d was written %zu bytes past its end", a.name, si, i);
        }
    }
    return RT_OK;
}

int rt_set_path_budget(rt_ctx* ctx, int nodesPerPixel) {
    if (!ctx) return RT_ERR_INVALID;
    if (nodesPerPixel < 1 || nodesPerPixel > 65536) return fail(ctx, RT_ERR_INVALID, "path budget must be in 1..65536");
    ctx->pathBudget = nodesPerPixel;
    return RT_OK;
}

int rt_scene_pack(rt_ctx* ctx, const RtSceneDesc* desc, void** blob, size_t* bytes) {
    if (!blob || !bytes) return fail(ctx, RT_ERR_INVALID, "null output argument");
    std::vector<unsigned char> host;
    std::string err;
    int r = packScene(desc, host, &err, nullptr, nullptr, ctx && ctx->uploadRefit);
    if (r != RT_OK) return fail(ctx, r, "%s", err.c_str());
    void* p = std::malloc(host.size());
    if (!p) return fail(ctx, RT_ERR_INVALID, "out of host memory");
    std::memcpy(p, host.data(), host.size());
    *blob = p;
    *bytes = host.size();
    return RT_OK;
}

void rt_blob_free(void* blob) { std::free(blob); }

namespace {
// Builds the scene's quad records from the pair records of the blob on the device (scene BVH first, then every mesh
// BVH) and the table of the meshes' root quads behind them.  The blob itself is only read.
int buildQuads(rt_ctx* ctx, rt_scene* sc) {
    const BlobHeader& hd = sc->header;
    std::vector<DevMesh> meshes(std::max(hd.meshCnt, 0));
    if (hd.meshCnt > 0)
        RT_HIP(ctx, hipMemcpy(meshes.data(), sc->blob + hd.offMeshes, meshes.size() * sizeof(DevMesh), hipMemcpyDeviceToHost));
    // a tree of n nodes has (n - 1) / 2 interior nodes, each with one pair, plus the root's pseudo pair
    auto pairsOf = [](int nodeCnt) { return (size_t)(nodeCnt + 1) / 2; };
    size_t total = pairsOf(hd.sceneNodeCnt);
    std::vector<uint32_t> rootQuad(meshes.size());
    for (size_t mi = 0; mi < meshes.size(); ++mi) {
        rootQuad[mi] = (RT_REF_ROOT_AXIS << 29) | (uint32_t)total;
        total += pairsOf(meshes[mi].nodeCnt);
    }
    if (total >= (1u << 29)) return fail(ctx, RT_ERR_LIMIT, "scene has too many BVH nodes for the packed node reference");
    const size_t quadBytes = total * 128, tableBytes = std::max<size_t>(meshes.size(), 1) * sizeof(uint32_t);
    RT_HIP(ctx, hipMalloc((void**)&sc->quads, quadBytes + tableBytes));
    sc->quadCnt = total;
    uint32_t at = 0;
    RT_HIP(ctx, (hipError_t)rtk_build_quads(sc->blob + hd.offSceneNodes, (uint32_t)(hd.offSceneNodes / 64), (uint32_t)pairsOf(hd.sceneNodeCnt), sc->quads, at,
                                             ctx->stream));
    at += (uint32_t)pairsOf(hd.sceneNodeCnt);
    for (size_t mi = 0; mi < meshes.size(); ++mi) {
        const uint32_t n = (uint32_t)pairsOf(meshes[mi].nodeCnt);
        RT_HIP(ctx, (hipError_t)rtk_build_quads(sc->blob + meshes[mi].offNodes, (uint32_t)(meshes[mi].offNodes / 64), n, sc->quads, at, ctx->stream));
        at += n;
    }
    if (!meshes.empty()) RT_HIP(ctx, hipMemcpyAsync(sc->quads + quadBytes, rootQuad.data(), rootQuad.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// rt_set_upload_refit: recomputes the boxes of the trees the scene arrived with, in the blob on the device (meshes whose
// BVH was built on the device already have the boxes rt_set_device_bvh asked for).  `hostBlob` is the packed blob as
// uploaded: its tables say where the trees are.
int refitOnDevice(rt_ctx* ctx, rt_scene* sc, const unsigned char* hostBlob, const std::vector<DeviceBuild>& builds, const RtSceneDesc* desc) {
    const BlobHeader& hd = sc->header;
    auto pairsOf = [](int nodeCnt) { return (uint32_t)((nodeCnt + 1) / 2); };
    const DevMesh* meshes = reinterpret_cast<const DevMesh*>(hostBlob + hd.offMeshes);
    for (int mi = 0; mi < hd.meshCnt; ++mi) {
        bool builtHere = false;
        for (const DeviceBuild& b : builds) builtHere = builtHere || b.mesh == mi;
        if (builtHere) continue;
        const DevMesh& m = meshes[mi];
        RT_HIP(ctx, (hipError_t)rtk_refit_tree(sc->blob + m.offNodes, (uint32_t)(m.offNodes / 64), pairsOf(m.nodeCnt), m.depth, sc->blob + hd.offRecords,
                                                sc->blob + m.offVertices, nullptr, ctx->stream));
    }
    // the scene BVH's leaves: the shapes' own boxes, which the blob does not carry
    std::vector<float> boxes(6 * (size_t)desc->shapeCnt);
    for (int si = 0; si < desc->shapeCnt; ++si) {
        std::memcpy(&boxes[6 * (size_t)si], desc->shapes[si].bbmin, 12);
        std::memcpy(&boxes[6 * (size_t)si + 3], desc->shapes[si].bbmax, 12);
    }
    float* devBoxes = nullptr;
    RT_HIP(ctx, hipMalloc((void**)&devBoxes, boxes.size() * sizeof(float)));
    hipError_t e = uploadNow(devBoxes, boxes.data(), boxes.size() * sizeof(float), ctx->stream);
    if (e == hipSuccess)
        e = (hipError_t)rtk_refit_tree(sc->blob + hd.offSceneNodes, (uint32_t)(hd.offSceneNodes / 64), pairsOf(hd.sceneNodeCnt), hd.sceneDepth, nullptr, nullptr,
                                       devBoxes, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(devBoxes);
    if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "bounds refit on the device: %s", hipGetErrorString(e));
    return RT_OK;
}
}  // namespace

namespace {
// The mesh BVH of rt_mesh_bvh_build(RT_BVH_BINNED_SAH, tight bounds) with the cuts decided on the device
// (rt_set_device_bvh_rule): same node array, same leaf order.  The host keeps the tree - its shape depends on the data -
// and drives the device level by level (kernels/rt_bvh_build.hip, "Surface-area cuts"); boxes are formed here, by the
// host builder's own functions, from the order the device returns.
int deviceSahBuild(rt_ctx* ctx, const RtMesh& m, int threshold, std::vector<RtLinearNode>& nodes, std::vector<RtTriangle>& ordered, std::string* err) {
    using rt::BoundingBox;
    const int n = m.triangleCnt;
    const bool verbose = std::getenv("RT_TRACE_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    double tBoxes = 0, tLevels = 0;
    int levels = 0;
    if (n < 1 || !m.triangles || m.vertexCnt < 1 || !m.vertices) {
        *err = "mesh with missing triangles or vertices";
        return RT_ERR_INVALID;
    }
    for (int t = 0; t < n; ++t)
        for (int k = 0; k < 3; ++k)
            if (m.triangles[t].i[k] < 0 || m.triangles[t].i[k] >= m.vertexCnt) {
                *err = "triangle vertex index out of range";
                return RT_ERR_INVALID;
            }
    // primitive boxes as csrc/host/BVH.h triangleBox(TIGHT_BOUNDS) forms them, on as many threads as the host builder takes
    std::vector<BoundingBox> boxes((size_t)n);
    onThreads((size_t)n, [&](int k, int parts) {
        const size_t from = (size_t)n * k / parts, to = (size_t)n * (k + 1) / parts;
        for (size_t t = from; t < to; ++t) {
            BoundingBox b = rt::BVH::emptyBox(rt::BVH::TIGHT_BOUNDS);
            for (int c = 0; c < 3; ++c) {
                const float* p = m.vertices[m.triangles[t].i[c]].position;
                b = BoundingBox::unionn(b, rt::Vec3f(p[0], p[1], p[2]));
            }
            boxes[t] = b;
        }
    });
    tBoxes = now() - t0;
    struct Node {
        int start, end, orderedBase, axis, first, second;  // children: indices into `tree`, -1 for a leaf
    };
    std::vector<Node> tree;
    tree.push_back(Node{0, n, 0, 0, -1, -1});
    std::vector<uint32_t> order((size_t)n);
    if (n > threshold) {
        std::vector<float> lo((size_t)n * 4), hi((size_t)n * 4);
        for (int t = 0; t < n; ++t)
            for (int k = 0; k < 3; ++k) {
                lo[(size_t)t * 4 + k] = boxes[t].bbmin[k];
                hi[(size_t)t * 4 + k] = boxes[t].bbmax[k];
            }
        RtSahBuild* build = nullptr;
        hipError_t e = (hipError_t)rtk_sah_begin(n, threshold, lo.data(), hi.data(), &build, ctx->stream);
        std::vector<int> active{0};  // nodes of `tree` still to be cut, in ascending order of their ranges
        std::vector<int32_t> starts, ends, result;
        while (e == hipSuccess && !active.empty()) {
            starts.clear();
            ends.clear();
            for (int k : active) {
                starts.push_back(tree[k].start);
                ends.push_back(tree[k].end);
            }
            result.assign(active.size() * 2, 0);
            e = (hipError_t)rtk_sah_level(build, (int)active.size(), threshold, starts.data(), ends.data(), result.data(), ctx->stream);
            if (e != hipSuccess) break;
            std::vector<int> next;
            for (size_t a = 0; a < active.size(); ++a) {
                const int k = active[a], code = result[2 * a], mid = result[2 * a + 1];
                if (code < 0) continue;  // the centroids coincide: a leaf, whatever it holds
                const Node nd = tree[k];
                if (mid <= nd.start || mid >= nd.end) {
                    *err = "BVH build on the device: a cut outside its range";
                    rtk_sah_end(build);
                    return RT_ERR_DEVICE;
                }
                // csrc/host/BVH.h halves: the second half's leaves take the front of the range's ordered slots
                const int first = (int)tree.size();
                tree.push_back(Node{nd.start, mid, nd.orderedBase + (nd.end - mid), 0, -1, -1});
                tree.push_back(Node{mid, nd.end, nd.orderedBase, 0, -1, -1});
                tree[k].axis = code & 3;
                tree[k].first = first;
                tree[k].second = first + 1;
                if (mid - nd.start > threshold) next.push_back(first);
                if (nd.end - mid > threshold) next.push_back(first + 1);
            }
            active.swap(next);
            ++levels;
        }
        tLevels = now() - t0 - tBoxes;
        if (e == hipSuccess) e = (hipError_t)rtk_sah_order(build, order.data(), ctx->stream);
        rtk_sah_end(build);
        if (e != hipSuccess) {
            *err = std::string("BVH build on the device: ") + hipGetErrorString(e);
            return RT_ERR_DEVICE;
        }
    } else {
        for (int t = 0; t < n; ++t) order[t] = (uint32_t)t;
    }
    // depth-first node array (csrc/host/BVH.h flattenBvhTree): node, first subtree, second subtree; an interior node names
    // its second child.  Children are laid out before their parent's box is formed, so one sweep from the back is bottom-up.
    nodes.assign(tree.size(), RtLinearNode{});
    ordered.assign((size_t)n, RtTriangle{});
    std::vector<int> slotOf(tree.size(), -1);
    {
        std::vector<int> stack{0};
        int at = 0;
        while (!stack.empty()) {
            const int k = stack.back();
            stack.pop_back();
            slotOf[k] = at++;
            if (tree[k].first >= 0) {
                stack.push_back(tree[k].second);
                stack.push_back(tree[k].first);
            }
        }
    }
    std::vector<int> bySlot(tree.size());
    for (size_t k = 0; k < tree.size(); ++k) bySlot[slotOf[k]] = (int)k;
    for (const Node& nd : tree)
        if (nd.first < 0 && nd.end - nd.start > 65535) {
            *err = "BvhFactory: leaf exceeds the 16-bit primitive count of LinearNode";
            return RT_ERR_LIMIT;
        }
    {
        // the leaves: boxes and ordered triangles, each leaf on its own - on several threads for a large mesh
        onThreads((size_t)n, [&](int k, int parts) {
            const size_t from = tree.size() * k / parts, to = tree.size() * (k + 1) / parts;
            for (size_t t = from; t < to; ++t) {
                const Node& nd = tree[t];
                if (nd.first >= 0) continue;
                BoundingBox bb = rt::BVH::emptyBox(rt::BVH::TIGHT_BOUNDS);
                for (int i = nd.start; i < nd.end; ++i) {
                    bb = BoundingBox::unionn(bb, boxes[order[i]]);
                    ordered[(size_t)nd.orderedBase + (i - nd.start)] = m.triangles[order[i]];
                }
                RtLinearNode& out = nodes[slotOf[t]];
                out.offset = nd.orderedBase;
                out.primCnt = (uint16_t)(nd.end - nd.start);
                for (int c = 0; c < 3; ++c) {
                    out.bbmin[c] = bb.bbmin[c];
                    out.bbmax[c] = bb.bbmax[c];
                }
            }
        });
    }
    for (int slot = (int)tree.size() - 1; slot >= 0; --slot) {
        const Node& nd = tree[bySlot[slot]];
        RtLinearNode& out = nodes[slot];
        BoundingBox bb = rt::BVH::emptyBox(rt::BVH::TIGHT_BOUNDS);
        if (nd.first < 0) {
            continue;
        } else {
            const RtLinearNode &a = nodes[slotOf[nd.first]], &b = nodes[slotOf[nd.second]];
            BoundingBox ba, bc;
            for (int c = 0; c < 3; ++c) {
                ba.bbmin[c] = a.bbmin[c];
                ba.bbmax[c] = a.bbmax[c];
                bc.bbmin[c] = b.bbmin[c];
                bc.bbmax[c] = b.bbmax[c];
            }
            bb = BoundingBox::unionn(ba, bc);
            out.offset = slotOf[nd.second];
            out.primCnt = 0;
            out.axis = (uint8_t)nd.axis;
        }
        for (int c = 0; c < 3; ++c) {
            out.bbmin[c] = bb.bbmin[c];
            out.bbmax[c] = bb.bbmax[c];
        }
    }
    if (verbose)
        std::fprintf(stderr, "  surface-area build: boxes %.3f s, %d levels on the device %.3f s, nodes and leaf order %.3f s\n", tBoxes, levels, tLevels,
                     now() - t0 - tBoxes - tLevels);
    return RT_OK;
}
}  // namespace

rt_scene* rt_scene_upload(rt_ctx* ctx, const RtSceneDesc* desc) {
    if (!ctx) return nullptr;
    std::vector<unsigned char> host;
    std::string err;
    std::vector<DeviceBuild> builds;
    const bool verbose = std::getenv("RT_TRACE_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    // with rt_set_upload_refit the blob is packed as it came and its boxes are recomputed on the device, below
    const bool refitHere = ctx->uploadRefit || ctx->bvhBuild.rebuildAll;  // (a rebuild gives the scene BVH true boxes too)
    // Surface-area cuts: the meshes that get their BVH on the device get it BEFORE the blob is laid out (the size of
    // such a tree is not known beforehand) and go into the blob like meshes that came with nodes.
    RtSceneDesc local;
    std::vector<RtMesh> localMeshes;
    std::vector<std::vector<RtLinearNode>> builtNodes;
    std::vector<std::vector<RtTriangle>> builtTriangles;
    BuildParams packBuild = ctx->bvhBuild;
    if (desc && ctx->bvhBuild.splitRule == RT_BVH_BINNED_SAH && desc->meshCnt > 0 && desc->meshes) {
        if (!ctx->bvhBuild.tightBounds) {
            fail(ctx, RT_ERR_INVALID, "surface-area cuts on the device are made on true bounds: rt_set_device_bvh(ctx, threshold, 1)");
            return nullptr;
        }
        if (bind(ctx)) return nullptr;
        localMeshes.assign(desc->meshes, desc->meshes + desc->meshCnt);
        builtNodes.resize(desc->meshCnt);
        builtTriangles.resize(desc->meshCnt);
        for (int mi = 0; mi < desc->meshCnt; ++mi) {
            if (!builtOnDevice(desc->meshes[mi], &ctx->bvhBuild)) continue;
            const double t1 = now();
            int rr;
            try {
                rr = deviceSahBuild(ctx, desc->meshes[mi], ctx->bvhBuild.threshold, builtNodes[mi], builtTriangles[mi], &err);
            } catch (const std::exception& e) {  // (out of host memory)
                err = e.what();
                rr = RT_ERR_LIMIT;
            }
            if (rr != RT_OK) {
                fail(ctx, rr, "mesh %d: %s", mi, err.c_str());
                return nullptr;
            }
            if (verbose)
                std::fprintf(stderr, "rt_scene_upload: BVH of mesh %d (%d triangles, %zu nodes) cut by surface area on the device in %.3f s\n", mi,
                             desc->meshes[mi].triangleCnt, builtNodes[mi].size(), now() - t1);
            localMeshes[mi].nodes = builtNodes[mi].data();
            localMeshes[mi].nodeCnt = (int)builtNodes[mi].size();
            localMeshes[mi].triangles = builtTriangles[mi].data();
        }
        local = *desc;
        local.meshes = localMeshes.data();
        desc = &local;
        packBuild.rebuildAll = false;  // every mesh carries its nodes now
    }
    int r = packScene(desc, host, &err, &packBuild, &builds, false, /*boxesWillBeRefit=*/refitHere);
    if (verbose) std::fprintf(stderr, "rt_scene_upload: packed %zu bytes in %.3f s\n", host.size(), now() - t0);
    if (r != RT_OK) {
        fail(ctx, r, "%s", err.c_str());
        return nullptr;
    }
    if (bind(ctx)) return nullptr;
    unsigned char* dev = nullptr;
    hipError_t e = hipMalloc((void**)&dev, host.size());
    if (e == hipSuccess) e = uploadNow(dev, host.data(), host.size(), ctx->stream);
    if (e != hipSuccess) {
        fail(ctx, RT_ERR_DEVICE, "scene upload (%zu bytes): %s", host.size(), hipGetErrorString(e));
        if (dev) hipFree(dev);
        return nullptr;
    }
    // meshes that came without BVH nodes: order their triangles and fill boxes, axes and records on the device
    for (DeviceBuild& b : builds) {
        const RtMesh& m = desc->meshes[b.mesh];
        int32_t* devTris = nullptr;
        e = hipMalloc((void**)&devTris, (size_t)m.triangleCnt * sizeof(RtTriangle));
        if (e == hipSuccess) e = uploadNow(devTris, m.triangles, (size_t)m.triangleCnt * sizeof(RtTriangle), ctx->stream);
        if (e == hipSuccess) {
            b.job.nodes = b.table.data();
            b.job.devTriangles = devTris;
            b.job.devVertices = dev + b.offVertices;
            b.job.devNodeRecords = dev + b.offNodes;
            b.job.devTriRecords = dev + b.offRecords;
            t0 = now();
            e = (hipError_t)rtk_build_mesh_bvh(&b.job, ctx->stream);
            if (verbose) std::fprintf(stderr, "rt_scene_upload: BVH of mesh %d (%d triangles) built on the device in %.3f s\n", b.mesh, m.triangleCnt, now() - t0);
        }
        if (devTris) hipFree(devTris);
        if (e != hipSuccess) {
            fail(ctx, RT_ERR_DEVICE, "BVH build of mesh %d on the device: %s", b.mesh, hipGetErrorString(e));
            hipFree(dev);
            return nullptr;
        }
    }
    rt_scene* sc = new rt_scene();
    sc->ctx = ctx;
    sc->blob = dev;
    sc->ownsBlob = true;
    sc->bytes = host.size();
    std::memcpy(&sc->header, host.data(), sizeof(BlobHeader));
    if (refitHere) {
        t0 = now();
        const int rr = refitOnDevice(ctx, sc, host.data(), builds, desc);
        if (verbose) std::fprintf(stderr, "rt_scene_upload: bounds refit on the device in %.3f s\n", now() - t0);
        if (rr != RT_OK) {
            rt_scene_free(sc);
            return nullptr;
        }
    }
    if (buildQuads(ctx, sc) != RT_OK) {
        rt_scene_free(sc);
        return nullptr;
    }
    return sc;
}

int rt_mesh_bvh_build(const RtVertex* vertices, int vertexCnt, const RtTriangle* triangles, int triangleCnt, int splitRule,
                      int leafThreshold, int tightBounds, RtLinearNode* nodes, int nodeCapacity, int* nodeCnt, int32_t* order) {
    auto bad = [](const char* what) {
        g_createError = what;
        return RT_ERR_INVALID;
    };
    if (!vertices || !triangles || !nodes || !nodeCnt || !order || vertexCnt < 1 || triangleCnt < 1) return bad("rt_mesh_bvh_build: null or empty argument");
    if (splitRule != RT_BVH_MEDIAN_CUT && splitRule != RT_BVH_BINNED_SAH) return bad("rt_mesh_bvh_build: unknown split rule");
    if (leafThreshold < 1 || leafThreshold > RT_REF_MAX_LEAF_PRIMS) return bad("rt_mesh_bvh_build: leaf threshold outside 1..RT_REF_MAX_LEAF_PRIMS");
    for (int t = 0; t < triangleCnt; ++t)
        for (int c = 0; c < 3; ++c)
            if (triangles[t].i[c] < 0 || triangles[t].i[c] >= vertexCnt) return bad("rt_mesh_bvh_build: vertex index out of range");
    try {
        static_assert(sizeof(rt::Vertex) == sizeof(RtVertex) && sizeof(rt::Triangle) == sizeof(RtTriangle), "host mirrors of the C structs");
        const std::vector<rt::Vertex> v(reinterpret_cast<const rt::Vertex*>(vertices), reinterpret_cast<const rt::Vertex*>(vertices) + vertexCnt);
        const std::vector<rt::Triangle> t(reinterpret_cast<const rt::Triangle*>(triangles), reinterpret_cast<const rt::Triangle*>(triangles) + triangleCnt);
        rt::BVH::BvhFactory<rt::Triangle> bvh(v, t, leafThreshold, tightBounds ? rt::BVH::TIGHT_BOUNDS : rt::BVH::REFERENCE_BOUNDS,
                                               splitRule == RT_BVH_BINNED_SAH ? rt::BVH::BINNED_SAH : rt::BVH::MEDIAN_CUT);
        *nodeCnt = (int)bvh.nodes.size();
        if ((int)bvh.nodes.size() > nodeCapacity) {
            g_createError = "rt_mesh_bvh_build: node array too small (2 * triangleCnt - 1 always fits)";
            return RT_ERR_LIMIT;
        }
        std::memcpy(nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(RtLinearNode));
        std::memcpy(order, bvh.orderedIdx.data(), bvh.orderedIdx.size() * sizeof(int32_t));
        return RT_OK;
    } catch (const std::exception& e) {
        g_createError = std::string("rt_mesh_bvh_build: ") + e.what();
        return RT_ERR_INVALID;
    }
}

int rt_set_device_bvh(rt_ctx* ctx, int threshold, int tightBounds) {
    if (!ctx) return RT_ERR_INVALID;
    if (threshold < 1 || threshold > RT_REF_MAX_LEAF_PRIMS) return fail(ctx, RT_ERR_INVALID, "leaf threshold must be in 1..%d", RT_REF_MAX_LEAF_PRIMS);
    ctx->bvhBuild.threshold = threshold;
    ctx->bvhBuild.tightBounds = tightBounds ? 1 : 0;
    return RT_OK;
}

int rt_set_device_bvh_rule(rt_ctx* ctx, int splitRule) {
    if (!ctx) return RT_ERR_INVALID;
    if (splitRule != RT_BVH_MEDIAN_CUT && splitRule != RT_BVH_BINNED_SAH) return fail(ctx, RT_ERR_INVALID, "unknown split rule");
    ctx->bvhBuild.splitRule = splitRule;
    return RT_OK;
}

int rt_set_upload_refit(rt_ctx* ctx, int on) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->uploadRefit = on != 0;
    return RT_OK;
}

int rt_set_upload_rebuild(rt_ctx* ctx, int on) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->bvhBuild.rebuildAll = on != 0;
    return RT_OK;
}

int rt_bvh_refit(RtLinearNode* nodes, int nodeCnt, const RtVertex* vertices, int vertexCnt, const RtTriangle* triangles, int triangleCnt) {
    auto bad = [](const char* what) {
        g_createError = what;
        return RT_ERR_INVALID;
    };
    if (!nodes || nodeCnt < 1 || !vertices || !triangles || vertexCnt < 1 || triangleCnt < 1) return bad("rt_bvh_refit: null or empty argument");
    std::string err;
    if (treeDepth(nodes, nodeCnt, triangleCnt, nullptr, &err, false) < 0) {
        g_createError = "rt_bvh_refit: " + err;
        return RT_ERR_INVALID;
    }
    for (int t = 0; t < triangleCnt; ++t)
        for (int c = 0; c < 3; ++c)
            if (triangles[t].i[c] < 0 || triangles[t].i[c] >= vertexCnt) return bad("rt_bvh_refit: vertex index out of range");
    refitNodes(nodes, nodeCnt, vertices, triangles, nullptr);
    return RT_OK;
}

int rt_scene_bvh_refit(RtLinearNode* nodes, int nodeCnt, const RtShape* shapes, int shapeCnt) {
    if (!nodes || nodeCnt < 1 || !shapes || shapeCnt < 1) {
        g_createError = "rt_scene_bvh_refit: null or empty argument";
        return RT_ERR_INVALID;
    }
    std::string err;
    if (treeDepth(nodes, nodeCnt, shapeCnt, nullptr, &err, false) < 0) {
        g_createError = "rt_scene_bvh_refit: " + err;
        return RT_ERR_INVALID;
    }
    refitNodes(nodes, nodeCnt, nullptr, nullptr, shapes);
    return RT_OK;
}

rt_scene* rt_scene_adopt_blob(rt_ctx* ctx, const void* devBlob, size_t bytes) {
    if (!ctx) return nullptr;
    if (!devBlob || bytes < sizeof(BlobHeader)) {
        fail(ctx, RT_ERR_INVALID, "null or short scene blob");
        return nullptr;
    }
    if (bind(ctx)) return nullptr;
    BlobHeader hd;
    hipError_t e = hipMemcpy(&hd, devBlob, sizeof hd, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        fail(ctx, RT_ERR_DEVICE, "reading blob header: %s", hipGetErrorString(e));
        return nullptr;
    }
    if (checkHeader(ctx, hd, bytes) != RT_OK) return nullptr;
    rt_scene* sc = new rt_scene();
    sc->ctx = ctx;
    sc->blob = const_cast<unsigned char*>(static_cast<const unsigned char*>(devBlob));
    sc->ownsBlob = false;
    sc->bytes = bytes;
    sc->header = hd;
    if (buildQuads(ctx, sc) != RT_OK) {
        rt_scene_free(sc);
        return nullptr;
    }
    return sc;
}

int rt_scene_blob(rt_scene* scene, const void** devBlob, size_t* bytes) {
    if (!scene || !devBlob || !bytes) return RT_ERR_INVALID;
    *devBlob = scene->blob;
    *bytes = scene->bytes;
    return RT_OK;
}

int rt_scene_set_lights(rt_scene* scene, const RtLight* lights, int lightCnt) {
    if (!scene) return RT_ERR_INVALID;
    rt_ctx* ctx = scene->ctx;
    if (!lights || lightCnt != scene->header.lightCnt)
        return fail(ctx, RT_ERR_INVALID, "light count must stay %d", scene->header.lightCnt);
    if (int r = bind(ctx)) return r;
    // frames already issued (on the context's stream or on the in-flight slots) keep the lights they were issued with
    if (int r = syncFrames(ctx)) return r;
    RT_HIP(ctx, uploadNow(scene->blob + scene->header.offLights, lights, (size_t)lightCnt * sizeof(RtLight), ctx->stream));
    return RT_OK;
}

void rt_scene_free(rt_scene* scene) {
    if (!scene) return;
    if (scene->ownsBlob || scene->quads) {
        hipSetDevice(scene->ctx->device);
        (void)syncFrames(scene->ctx);  // frames in flight may still read the blob and the quads
        if (scene->ownsBlob) hipFree(scene->blob);
        if (scene->quads) hipFree(scene->quads);
    }
    delete scene;
}

int rt_surface_create(rt_ctx* ctx, int width, int height, void** devSurface, size_t* pitch) {
    if (!ctx) return RT_ERR_INVALID;
    if (width < 1 || height < 1 || !devSurface || !pitch) return fail(ctx, RT_ERR_INVALID, "bad surface arguments");
    if (int r = bind(ctx)) return r;
    const size_t p = alignUp((size_t)width * 16, 256);
    void* d = nullptr;
    RT_HIP(ctx, hipMalloc(&d, p * height));
    RT_HIP(ctx, fillNow(d, 0x01, p * height, ctx->stream));  // the reference fills new surfaces with 0x01 bytes
    *devSurface = d;
    *pitch = p;
    return RT_OK;
}

int rt_surface_download(rt_ctx* ctx, const void* devSurface, size_t pitch, int width, int height, float* hostRgba) {
    if (!ctx) return RT_ERR_INVALID;
    if (!devSurface || !hostRgba || width < 1 || height < 1 || pitch < (size_t)width * 16) return fail(ctx, RT_ERR_INVALID, "bad download arguments");
    if (int r = bind(ctx)) return r;
    if (int r = syncFrames(ctx)) return r;
    // The caller's buffer is ordinary pageable memory, which the copy engine cannot reach directly: rows travel in
    // chunks through two pinned buffers, the next chunk leaving the device while the previous one is copied out.
    constexpr size_t kChunk = 4u << 20;
    if (!ctx->copyStream) {
        RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->copyStream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            RT_HIP(ctx, hipHostMalloc(&ctx->staging[k], kChunk, hipHostMallocDefault));
            RT_HIP(ctx, hipEventCreateWithFlags(&ctx->stagingDone[k], hipEventDisableTiming));
        }
    }
    const size_t rowBytes = (size_t)width * 16;
    if (rowBytes > kChunk) {  // a single row larger than a chunk: plain copy
        RT_HIP(ctx, hipMemcpy2D(hostRgba, rowBytes, devSurface, pitch, rowBytes, height, hipMemcpyDeviceToHost));
        return RT_OK;
    }
    const int rowsPerChunk = (int)(kChunk / rowBytes);
    const int chunks = (height + rowsPerChunk - 1) / rowsPerChunk;
    auto rowsOf = [&](int c) { return std::min(rowsPerChunk, height - c * rowsPerChunk); };
    for (int c = 0; c <= chunks; ++c) {
        if (c < chunks) {
            const unsigned char* src = static_cast<const unsigned char*>(devSurface) + (size_t)c * rowsPerChunk * pitch;
            RT_HIP(ctx, hipMemcpy2DAsync(ctx->staging[c & 1], rowBytes, src, pitch, rowBytes, rowsOf(c), hipMemcpyDeviceToHost, ctx->copyStream));
            RT_HIP(ctx, hipEventRecord(ctx->stagingDone[c & 1], ctx->copyStream));
        }
        if (c > 0) {
            const int p = c - 1;
            RT_HIP(ctx, hipEventSynchronize(ctx->stagingDone[p & 1]));
            std::memcpy(reinterpret_cast<unsigned char*>(hostRgba) + (size_t)p * rowsPerChunk * rowBytes, ctx->staging[p & 1], (size_t)rowsOf(p) * rowBytes);
        }
    }
    return RT_OK;
}

void rt_surface_free(rt_ctx* ctx, void* devSurface) { rt_device_free(ctx, devSurface); }

int rt_device_alloc(rt_ctx* ctx, size_t bytes, void** devPtr) {
    if (!ctx) return RT_ERR_INVALID;
    if (!devPtr || bytes == 0) return fail(ctx, RT_ERR_INVALID, "bad allocation arguments");
    if (int r = bind(ctx)) return r;
    void* d = nullptr;
    RT_HIP(ctx, hipMalloc(&d, bytes));
    RT_HIP(ctx, fillNow(d, 0, bytes, ctx->stream));
    *devPtr = d;
    return RT_OK;
}

int rt_device_download(rt_ctx* ctx, const void* devPtr, size_t bytes, void* host) {
    if (!ctx) return RT_ERR_INVALID;
    if (!devPtr || !host) return fail(ctx, RT_ERR_INVALID, "bad download arguments");
    if (int r = bind(ctx)) return r;
    if (int r = syncFrames(ctx)) return r;
    RT_HIP(ctx, hipMemcpy(host, devPtr, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_device_zero(rt_ctx* ctx, void* devPtr, size_t bytes) {
    if (!ctx) return RT_ERR_INVALID;
    if (!devPtr) return fail(ctx, RT_ERR_INVALID, "null pointer");
    if (int r = bind(ctx)) return r;
    RT_HIP(ctx, hipMemsetAsync(devPtr, 0, bytes, ctx->stream));
    return RT_OK;
}

void rt_device_free(rt_ctx* ctx, void* devPtr) {
    if (!ctx || !devPtr) return;
    hipSetDevice(ctx->device);
    (void)syncFrames(ctx);  // frames in flight may still write the buffer
    hipFree(devPtr);
}

}  // extern "C"

namespace {

// A new accumulation (rt_set_accumulation with samplesSoFar = 0) starts from a cleared running-sum surface: every
// pixel's sum and sample count are zero.  Enqueued once, in front of the first frame that accumulates (a memset
// allocates nothing).
int startAccumulation(rt_ctx* ctx, const RtDeviceOpts* opts, hipStream_t stream) {
    if (!ctx->accum || !ctx->accumFresh) return RT_OK;
    RT_HIP(ctx, hipMemset2DAsync(ctx->accum, ctx->accumPitch, 0, (size_t)opts->width * 16, (size_t)opts->height, stream));
    ctx->accumFresh = false;
    return RT_OK;
}

// Body of rt_render and rt_render_noalloc.  `noAlloc`: the frame's workspace is carved from the device's static
// scratch array, the frame runs on the context's stream, in order, with row-major tile order and without timing
// events, and nothing on the way allocates or frees.
int renderFrame(rt_ctx* ctx, rt_scene* scene, const RtDeviceOpts* opts, const RtCamera* cam, void* devSurface, size_t pitch,
                const RtFrameAux* aux, bool noAlloc) {
    if (!ctx) return RT_ERR_INVALID;
    if (!scene || scene->ctx != ctx) return fail(ctx, RT_ERR_INVALID, "scene does not belong to this context");
    if (!opts || !cam || !devSurface) return fail(ctx, RT_ERR_INVALID, "null argument");
    if (opts->width < 1 || opts->height < 1) return fail(ctx, RT_ERR_INVALID, "bad frame size %dx%d", opts->width, opts->height);
    if (pitch < (size_t)opts->width * 16 || pitch % 16 != 0 || (reinterpret_cast<uintptr_t>(devSurface) & 15))
        return fail(ctx, RT_ERR_INVALID, "surface pitch/alignment must hold %d float4 pixels per row, 16-byte aligned", opts->width);
    if (opts->samples < 1 || opts->samples > RT_MAX_SAMPLES) return fail(ctx, RT_ERR_INVALID, "samples %d outside 1..%d", opts->samples, RT_MAX_SAMPLES);
    if (opts->maxDepth < 0 || opts->maxDepth > RT_MAX_DEPTH)
        return fail(ctx, RT_ERR_LIMIT, "maxDepth %d outside 0..%d", opts->maxDepth, RT_MAX_DEPTH);
    const BlobHeader& hd = scene->header;

    FrameParams p;
    std::memset(&p, 0, sizeof p);
    p.blob = scene->blob;
    p.offLights = hd.offLights;
    p.offShapes = hd.offShapes;
    p.offSceneNodes = hd.offSceneNodes;
    p.offMeshes = hd.offMeshes;
    p.lightCnt = hd.lightCnt;
    p.shapeCnt = hd.shapeCnt;
    p.opts = *opts;
    std::memcpy(p.orig, cam->orig, sizeof p.orig);
    std::memcpy(p.camera, cam->camera, sizeof p.camera);
    p.surface = static_cast<unsigned char*>(devSurface);
    p.pitch = pitch;
    p.hitKeys = aux ? aux->hitKeys : nullptr;
    p.counters = aux ? reinterpret_cast<unsigned long long*>(aux->counters) : nullptr;
    p.tileCost = ctx->tileCost;
    p.tilesX = (opts->width + RT_TILE - 1) / RT_TILE;
    p.tilesY = (opts->height + RT_TILE - 1) / RT_TILE;
    p.tilesXMagic = (uint32_t)(0xFFFFFFFFull / (unsigned long long)p.tilesX);
    p.invWidth = 1.0f / (float)opts->width;
    p.invHeight = 1.0f / (float)opts->height;
    const long long tiles = (long long)p.tilesX * p.tilesY;
    const int tileFirst = aux ? aux->tileFirst : 0;
    const int tileStride = aux && aux->tileStride > 0 ? aux->tileStride : 1;
    if (tileFirst < 0 || tileFirst >= tileStride) return fail(ctx, RT_ERR_INVALID, "tileFirst must be in [0, tileStride)");
    if (ctx->tileList) {
        if (tileFirst != 0 || tileStride != 1) return fail(ctx, RT_ERR_INVALID, "a call takes its tiles either from rt_set_tile_list or from RtFrameAux.tileFirst / tileStride, not both");
        if (ctx->tileListMax >= tiles) return fail(ctx, RT_ERR_INVALID, "tile list names tile %d, the frame has %lld tiles", ctx->tileListMax, tiles);
    }
    const long long myTiles = ctx->tileList ? ctx->tileListCnt : (tileFirst < tiles ? (tiles - tileFirst + tileStride - 1) / tileStride : 0);
    const int secondary = hd.hasSecondary ? 1 : 0;
    const int budget = secondary && opts->maxDepth > 0 ? ctx->pathBudget : 1;
    const long long nodeCapWanted = myTiles * RT_TILE * RT_TILE * budget * opts->samples;
    if (nodeCapWanted > 0x7fffffffLL / 4) return fail(ctx, RT_ERR_LIMIT, "frame too large for 32-bit node indices");
    // shadow work items are (diffuse hit, light) pairs held in 32-bit signed indices
    if (nodeCapWanted * std::max(hd.lightCnt, 1) > 0x7fffffffLL)
        return fail(ctx, RT_ERR_LIMIT, "frame too large for 32-bit shadow-ray indices (%lld path nodes x %d lights)", nodeCapWanted, hd.lightCnt);
    // The walk over pairs holds one waiting sibling per level; the walk over quads up to three per two levels.
    p.stackEntries = std::max(hd.sceneDepth + hd.maxMeshDepth + 2, 3 * ((hd.sceneDepth + 2) / 2) + 3 * ((hd.maxMeshDepth + 2) / 2) + 2);
    p.sceneRootRef = hd.sceneRootRef;
    p.sceneRootQuad = RT_REF_ROOT_AXIS << 29;  // the scene BVH's quads come first
    p.quads = reinterpret_cast<const float4*>(scene->quads);
    p.meshRootQuad = reinterpret_cast<const uint32_t*>(scene->quads + scene->quadCnt * 128);
    p.offRecords = hd.offRecords;
    // the LDS part of the stack is a ring: a power of two
    p.stackLds = 2;
    while (p.stackLds * 2 <= ctx->ldsStack && p.stackLds < p.stackEntries) p.stackLds *= 2;

    if (int r = bind(ctx)) return r;
    if (ctx->wavesForLds != p.stackLds) {
        ctx->wavesPerCu = rtk_trace_waves_per_cu(p.stackLds);
        ctx->wavesForLds = p.stackLds;
    }
    p.traceWaves = (ctx->maxWavesPerCu > 0 ? std::min(ctx->wavesPerCu, ctx->maxWavesPerCu) : ctx->wavesPerCu) * ctx->cuCount;

    // measured on whole frames, on small reflection levels and on one GPU's share of a multi-GPU frame: 2 and 8 are slower;
    // launches with more than 6000 chunks (a half or a quarter of a 1080p frame) do 2-5 % better with 6 when frames overlap
    p.waveChunks = 4;
    p.waveChunksLarge = ctx->framesInFlight > 1 ? 6 : 4;  // alone on the machine, a launch is shortest with the most wavefronts
    p.splitSlots = ctx->splitSlots;
    p.skipUnlit = ctx->skipUnlit ? 1 : 0;
    p.tileFirst = tileFirst;
    p.tileStride = tileStride;
    p.tileCnt = (int)myTiles;
    p.slots = p.tileCnt * RT_TILE * RT_TILE * opts->samples;
    p.sampleSeed = ctx->sampleSeed;
    if (ctx->accum) {
        if (ctx->framesInFlight > 1) return fail(ctx, RT_ERR_INVALID, "accumulation over frames needs frames in order (rt_set_frames_in_flight(1)): every frame continues the previous one's sums");
        if (ctx->accumPitch < (size_t)opts->width * 16) return fail(ctx, RT_ERR_INVALID, "accumulation surface pitch holds fewer than %d float4 pixels", opts->width);
        if ((unsigned long long)ctx->accumSamples + (unsigned)opts->samples > (1u << 24))
            return fail(ctx, RT_ERR_LIMIT, "accumulation holds at most 2^24 samples per pixel (the count is kept in fp32)");
        p.accum = ctx->accum;
        p.accumPitch = ctx->accumPitch;
    }
    p.nodeCap = p.slots * budget;

    if (noAlloc) {
        if (!ctx->scratch) return fail(ctx, RT_ERR_DEVICE, "the static scratch array is not available on device %d", ctx->device);
        if (myTiles == 0) return RT_OK;
        // Carve the workspace from the scratch array.  The traversal-stack spill area takes at most a quarter of it:
        // fewer persistent wavefronts run when the scene's trees are deep.
        const size_t perWave = (size_t)(p.stackEntries + RT_MESH_ENTRY_SLOTS) * 64 * 8;
        const size_t spillMax = ctx->scratchBytes / 4;
        if ((size_t)p.traceWaves * perWave > spillMax) p.traceWaves = (int)(spillMax / perWave);
        if (p.traceWaves < 1) return fail(ctx, RT_ERR_LIMIT, "scratch array too small for one wavefront's traversal stack");
        Workspace w;
        size_t cursor = 0;
        auto carve = [&](size_t bytes) {
            unsigned char* at = ctx->scratch + cursor;
            cursor = alignUp(cursor + bytes);
            return static_cast<void*>(at);
        };
        for (const WsArray& a : workspaceArrays((size_t)p.nodeCap, (size_t)(p.nodeCap - p.slots), (size_t)std::max(hd.lightCnt, 1)))
            w.*a.member = carve(a.bytes);
        w.stackSpill = carve((size_t)p.traceWaves * perWave);
        if (cursor > ctx->scratchBytes)
            return fail(ctx, RT_ERR_LIMIT,
                        "rt_render_noalloc: this call needs %zu bytes of workspace, the static scratch array holds %zu "
                        "(%d path nodes, %d lights); render fewer tiles per call (RtFrameAux.tileStride) or use rt_render",
                        cursor, ctx->scratchBytes, p.nodeCap, hd.lightCnt);
        p.ctl = static_cast<uint32_t*>(w.ctl);
        p.hit = static_cast<float4*>(w.hit);
        p.hitTri = static_cast<int32_t*>(w.hitTri);
        p.color = static_cast<float4*>(w.color);
        p.secRay = static_cast<float4*>(w.secRay);
        p.child = static_cast<int32_t*>(w.child);
        p.kr = static_cast<float*>(w.kr);
        p.kind = static_cast<int32_t*>(w.kind);
        p.ctx = static_cast<float4*>(w.ctx);
        p.ctxNode = static_cast<int32_t*>(w.ctxNode);
        p.vis = static_cast<unsigned char*>(w.vis);
        p.stackSpill = static_cast<uint2*>(w.stackSpill);
        p.tileOrder = ctx->tileList;  // else row-major: an order table would have to be uploaded into fresh device memory
        if (int r = startAccumulation(ctx, opts, ctx->stream)) return r;
        hipError_t e = (hipError_t)rtk_render_frame(&p, secondary, p.counters ? 1 : 0, ctx->cuCount, ctx->stream, nullptr);
        if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "kernel launch: %s", hipGetErrorString(e));
        ctx->scratchMayOverflow = ctx->scratchMayOverflow || (secondary && opts->maxDepth > 0);
        if (ctx->accum) ctx->accumSamples += (uint32_t)opts->samples;
        return RT_OK;
    }

    // The frame's slot: with one frame in flight it runs on the context's stream, in order.  With
    // several, it runs on the slot's own stream behind everything enqueued on the context's
    // stream so far (and behind the slot's previous frame), and overlaps with the other slots.
    FrameSlot& sl = ctx->slots[ctx->frameCounter % (unsigned long long)ctx->framesInFlight];
    const bool forked = ctx->framesInFlight > 1;
    if (!sl.done) RT_HIP(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (forked && !sl.stream) {
        RT_HIP(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        RT_HIP(ctx, hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
    }
    hipStream_t exec = forked ? sl.stream : ctx->stream;
    if (sl.issued && sl.lastStream && sl.lastStream != exec) RT_HIP(ctx, hipStreamWaitEvent(exec, sl.done, 0));

    if (myTiles == 0) {
        // A shard without tiles (more ranks than tiles): nothing to launch, but the call still counts as a frame, so
        // that rt_stream_wait_frame(framesAgo) keeps meaning "that many rt_render calls ago" on every rank.
        if (forked) {
            RT_HIP(ctx, hipEventRecord(sl.ready, ctx->stream));
            RT_HIP(ctx, hipStreamWaitEvent(exec, sl.ready, 0));
        }
        RT_HIP(ctx, hipEventRecord(sl.done, exec));
        sl.issued = true;
        sl.lastStream = exec;
        ctx->frameCounter++;
        return RT_OK;
    }

    const size_t spillBytes = (size_t)p.traceWaves * (size_t)(p.stackEntries + RT_MESH_ENTRY_SLOTS) * 64 * 8;
    if (int r = ensureWorkspace(ctx, sl.ws, sl.lastStream ? sl.lastStream : exec, p.nodeCap, p.slots, hd.lightCnt, std::max<size_t>(spillBytes, 256))) return r;
    const Workspace& w = sl.ws;
    p.ctl = static_cast<uint32_t*>(w.ctl);
    p.hit = static_cast<float4*>(w.hit);
    p.hitTri = static_cast<int32_t*>(w.hitTri);
    p.color = static_cast<float4*>(w.color);
    p.secRay = static_cast<float4*>(w.secRay);
    p.child = static_cast<int32_t*>(w.child);
    p.kr = static_cast<float*>(w.kr);
    p.kind = static_cast<int32_t*>(w.kind);
    p.ctx = static_cast<float4*>(w.ctx);
    p.ctxNode = static_cast<int32_t*>(w.ctxNode);
    p.vis = static_cast<unsigned char*>(w.vis);
    p.stackSpill = static_cast<uint2*>(w.stackSpill);

    // Tile order.  The persistent trace kernels drain their queue front to back; when the image
    // regions that take longest come first, the last rays in flight are short ones and the
    // wavefronts run dry together.  Geometry sits around the image centre far more often than in
    // the corners, so the tiles are handed out centre first.  Which tiles a call renders does not
    // change, only their order in the queue.
    if (ctx->tileList) {
        p.tileOrder = ctx->tileList;  // the caller's tiles, in the caller's order
    } else if (ctx->tileOrderMode == RT_TILE_ORDER_CENTER_OUT) {
        const long long key[6] = {p.tilesX, p.tilesY, tileFirst, tileStride, opts->width, opts->height};
        const void* table = nullptr;
        for (const auto& t : ctx->tileOrders)
            if (std::memcmp(key, t.key, sizeof key) == 0) table = t.dev;
        if (!table) {
            if (ctx->tileOrders.size() >= 64) {  // a caller cycling through many shardings: start over
                if (int r = syncFrames(ctx)) return r;
                for (auto& t : ctx->tileOrders) hipFree(t.dev);
                ctx->tileOrders.clear();
            }
            std::vector<std::pair<long long, int32_t>> all((size_t)myTiles);
            const long long cx = opts->width, cy = opts->height;  // twice the image centre
            for (long long k = 0; k < myTiles; ++k) {
                const int32_t t = (int32_t)(tileFirst + k * tileStride);
                const long long dx = (long long)(t % p.tilesX) * 2 * RT_TILE + RT_TILE - cx;
                const long long dy = (long long)(t / p.tilesX) * 2 * RT_TILE + RT_TILE - cy;
                all[(size_t)k] = {dx * dx + dy * dy, t};
            }
            std::sort(all.begin(), all.end());
            std::vector<int32_t> order((size_t)myTiles);
            for (long long k = 0; k < myTiles; ++k) order[(size_t)k] = all[(size_t)k].second;
            void* dev = nullptr;
            RT_HIP(ctx, hipMalloc(&dev, (size_t)myTiles * sizeof(int32_t)));
            hipError_t ce = uploadNow(dev, order.data(), (size_t)myTiles * sizeof(int32_t), ctx->stream);
            if (ce != hipSuccess) {
                hipFree(dev);
                return fail(ctx, RT_ERR_DEVICE, "tile order upload: %s", hipGetErrorString(ce));
            }
            rt_ctx::TileOrder entry;
            std::memcpy(entry.key, key, sizeof key);
            entry.dev = dev;
            ctx->tileOrders.push_back(entry);
            table = dev;
        }
        p.tileOrder = static_cast<const int32_t*>(table);
    }

    if (ctx->launchesUsed >= ctx->launches.size() && ctx->launches.size() < rt_ctx::kMaxTimedLaunches) {
        TimedLaunch l;
        for (hipEvent_t& e : l.ev) RT_HIP(ctx, hipEventCreate(&e));
        ctx->launches.push_back(l);
    }
    TimedLaunch& l = ctx->launches[ctx->launchesUsed % rt_ctx::kMaxTimedLaunches];
    if (forked) {
        RT_HIP(ctx, hipEventRecord(sl.ready, ctx->stream));
        RT_HIP(ctx, hipStreamWaitEvent(exec, sl.ready, 0));
    }
    if (int r = startAccumulation(ctx, opts, exec)) return r;
    hipError_t e = (hipError_t)rtk_render_frame(&p, secondary, p.counters ? 1 : 0, ctx->cuCount, exec, reinterpret_cast<void**>(l.ev));
    if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, "kernel launch: %s", hipGetErrorString(e));
    RT_HIP(ctx, hipEventRecord(sl.done, exec));
    sl.issued = true;
    sl.mayOverflow = sl.mayOverflow || (secondary && opts->maxDepth > 0);
    sl.lastStream = exec;
    ctx->frameCounter++;
    ctx->launchesUsed++;
    if (ctx->accum) ctx->accumSamples += (uint32_t)opts->samples;
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_render(rt_ctx* ctx, rt_scene* scene, const RtDeviceOpts* opts, const RtCamera* cam, void* devSurface, size_t pitch,
              int blockDim, const RtFrameAux* aux) {
    (void)blockDim;
    return renderFrame(ctx, scene, opts, cam, devSurface, pitch, aux, false);
}

int rt_render_noalloc(rt_ctx* ctx, rt_scene* scene, const RtDeviceOpts* opts, const RtCamera* cam, void* devSurface, size_t pitch,
                      int blockDim, const RtFrameAux* aux) {
    (void)blockDim;
    return renderFrame(ctx, scene, opts, cam, devSurface, pitch, aux, true);
}

int rt_tiles_pack(rt_ctx* ctx, const void* devSurface, size_t pitch, int width, int height, const int32_t* devTiles, int tileCnt, void* devOut,
                  int quantize) {
    if (!ctx) return RT_ERR_INVALID;
    if (!devSurface || !devOut || width < 1 || height < 1 || tileCnt < 0 || (tileCnt > 0 && !devTiles) || pitch < (size_t)width * 16 || pitch % 16 != 0)
        return fail(ctx, RT_ERR_INVALID, "bad tile pack arguments");
    if (int r = bind(ctx)) return r;
    RT_HIP(ctx, (hipError_t)rtk_pack_tiles(devSurface, pitch, width, height, devTiles, tileCnt, devOut, quantize ? 1 : 0, ctx->stream));
    return RT_OK;
}

int rt_tiles_scatter(rt_ctx* ctx, const void* devIn, const int32_t* devTiles, int tileCnt, void* devSurface, size_t pitch, int width, int height,
                     int quantized) {
    if (!ctx) return RT_ERR_INVALID;
    const size_t px = quantized ? 4 : 16;
    if (!devSurface || !devIn || width < 1 || height < 1 || tileCnt < 0 || (tileCnt > 0 && !devTiles) || pitch < (size_t)width * px || pitch % px != 0)
        return fail(ctx, RT_ERR_INVALID, "bad tile scatter arguments");
    if (int r = bind(ctx)) return r;
    RT_HIP(ctx, (hipError_t)rtk_scatter_tiles(devIn, devTiles, tileCnt, devSurface, pitch, width, height, quantized ? 1 : 0, ctx->stream));
    return RT_OK;
}

int rt_timing_reset(rt_ctx* ctx) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->launchesUsed = 0;
    return RT_OK;
}

int rt_timing_read(rt_ctx* ctx, double* totalMs, int* launches) { return rt_timing_read_stages(ctx, totalMs, nullptr, launches); }

int rt_timing_read_stages(rt_ctx* ctx, double* totalMs, double* stageMs, int* launches) {
    if (!ctx) return RT_ERR_INVALID;
    if (int r = bind(ctx)) return r;
    double total = 0, stage[4] = {0, 0, 0, 0};
    const size_t timed = std::min(ctx->launchesUsed, ctx->launches.size());
    for (size_t i = 0; i < timed; ++i) {
        TimedLaunch& l = ctx->launches[i];
        RT_HIP(ctx, hipEventSynchronize(l.ev[4]));
        float ms = 0;
        RT_HIP(ctx, hipEventElapsedTime(&ms, l.ev[0], l.ev[4]));
        total += ms;
        for (int k = 0; k < 4; ++k) {
            RT_HIP(ctx, hipEventElapsedTime(&ms, l.ev[k], l.ev[k + 1]));
            stage[k] += ms;
        }
    }
    if (totalMs) *totalMs = total;
    if (stageMs)
        for (int k = 0; k < 4; ++k) stageMs[k] = stage[k];
    if (launches) *launches = (int)timed;
    return RT_OK;
}

}  // extern "C"
