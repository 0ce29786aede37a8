// rt_multi.cpp - several GPUs driven by ONE host process, on top of the per-GPU C-ABI (include/rt_capi.h).
//
// The reference's caller is a single C++ process that calls cudamain once per frame (cudaTracer/RayTracer.cpp:188-208,
// cudaTracer/main.cuh:7-12) on the one GPU its window lives on.  A host that adopts this library keeps that shape and
// gets the other GPUs of the node through the rt_multi_* entry points below: one context per device, the scene blob
// copied device 0 -> peers once (each peer over its own xGMI link), every device rendering its own list of 8x8 tiles
// of the same frame, and the finished tiles gathered into device 0's surface: packed on the peer (rt_tiles.hip), one
// peer copy, scattered on device 0.  No collective, no second process; cudaraytracer_amd/multi.py is the same
// scheme with one process per GPU over torch.distributed.
//
// Everything here goes through the public entry points; the only thing shared with rt_capi.cpp is the pair of tile
// kernels.  A device may be named more than once (two contexts on one GPU): the peer copies then are plain
// device-to-device copies, which is how a one-GPU machine exercises the whole path.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_capi.h"
#include "rt_device_format.h"

namespace {

struct MultiDev {
    int device = 0;
    rt_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    rt_scene* scene = nullptr;
    void* blob = nullptr;          // this device's copy of the scene blob (peers only; device 0's belongs to its scene)
    void* surface = nullptr;       // own float4 surface (peers; device 0 only when the output is 8-bit)
    size_t pitch = 0;
    std::vector<int32_t> tiles;    // this device's tiles, in queue order
    int32_t* devTiles = nullptr;   // the list on this device (pack) ...
    int32_t* tilesOn0 = nullptr;   // ... and on device 0 (scatter)
    void* packed = nullptr;        // packed tiles on this device
    void* recv = nullptr;          // where they arrive on device 0
    size_t listCap = 0;            // tiles the four buffers above hold
    hipEvent_t sent = nullptr;     // recorded behind this device's peer copy
};

}  // namespace

struct rt_multi {
    std::vector<MultiDev> dev;
    std::string error;
    int width = 0, height = 0;  // frame size the lists and surfaces were made for
    bool balanced = false;      // the lists come from rt_multi_balance (kept until the frame size changes)
    bool quantize = false;      // 8-bit output surface
    hipEvent_t scattered = nullptr;  // device 0 has consumed the receive buffers of the latest frame
    bool scatteredValid = false;
};

namespace {

thread_local std::string g_multiCreateError;

int failm(rt_multi* m, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (m)
        m->error = buf;
    else
        g_multiCreateError = buf;
    return code;
}

#define RTM_HIP(m, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return failm(m, RT_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define RTM_RT(m, i, call)                                                                                             \
    do {                                                                                                               \
        const int r_ = (call);                                                                                         \
        if (r_ != RT_OK) return failm(m, r_, "device %d: %s: %s", (m)->dev[i].device, #call, rt_last_error((m)->dev[i].ctx)); \
    } while (0)

void freeFrameBuffers(MultiDev& d, int device0) {
    if (d.surface) {
        hipSetDevice(d.device);
        hipFree(d.surface);
    }
    hipSetDevice(d.device);
    if (d.devTiles) hipFree(d.devTiles);
    if (d.packed) hipFree(d.packed);
    hipSetDevice(device0);
    if (d.tilesOn0) hipFree(d.tilesOn0);
    if (d.recv) hipFree(d.recv);
    d.surface = d.devTiles = nullptr;
    d.packed = d.recv = nullptr;
    d.tilesOn0 = nullptr;
    d.listCap = 0;
    d.pitch = 0;
}

// Same order as the library's own centre-out tile order: by the distance of the tile's centre from the image centre.
void centerOut(std::vector<int32_t>& tiles, int width, int height) {
    const int tx = (width + RT_TILE - 1) / RT_TILE;
    auto key = [&](int32_t t) {
        const long long dx = (long long)(t % tx) * 2 * RT_TILE + RT_TILE - width, dy = (long long)(t / tx) * 2 * RT_TILE + RT_TILE - height;
        return std::make_pair(dx * dx + dy * dy, t);
    };
    std::sort(tiles.begin(), tiles.end(), [&](int32_t a, int32_t b) { return key(a) < key(b); });
}

// Hands the lists to the contexts and puts copies where the pack and scatter kernels read them.
int installLists(rt_multi* m) {
    const int dev0 = m->dev[0].device;
    const size_t pixelBytes = m->quantize ? 4 : 16;
    for (size_t i = 0; i < m->dev.size(); ++i) {
        MultiDev& d = m->dev[i];
        RTM_RT(m, i, rt_sync(d.ctx));
        const size_t n = d.tiles.size();
        if (n > d.listCap) {
            RTM_HIP(m, hipSetDevice(d.device));
            if (d.devTiles) hipFree(d.devTiles);
            if (d.packed) hipFree(d.packed);
            d.devTiles = nullptr;
            d.packed = nullptr;
            RTM_HIP(m, hipMalloc((void**)&d.devTiles, n * sizeof(int32_t)));
            RTM_HIP(m, hipMalloc(&d.packed, n * 64 * 16));  // sized for float4 pixels: the output format may change
            RTM_HIP(m, hipSetDevice(dev0));
            if (d.tilesOn0) hipFree(d.tilesOn0);
            if (d.recv) hipFree(d.recv);
            d.tilesOn0 = nullptr;
            d.recv = nullptr;
            RTM_HIP(m, hipMalloc((void**)&d.tilesOn0, n * sizeof(int32_t)));
            RTM_HIP(m, hipMalloc(&d.recv, n * 64 * 16));
            d.listCap = n;
        }
        (void)pixelBytes;
        if (n) {
            RTM_HIP(m, hipSetDevice(d.device));
            RTM_HIP(m, hipMemcpy(d.devTiles, d.tiles.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
            RTM_HIP(m, hipSetDevice(dev0));
            RTM_HIP(m, hipMemcpy(d.tilesOn0, d.tiles.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
            RTM_RT(m, i, rt_set_tile_list(d.ctx, d.tiles.data(), (int)n));
        }
    }
    return RT_OK;
}

// Surfaces and (unless balanced lists for this size exist) interleaved tile lists for a w x h frame.
int ensureFrame(rt_multi* m, int width, int height) {
    const int n = (int)m->dev.size();
    const bool sizeChanged = width != m->width || height != m->height;
    if (sizeChanged) {
        for (int i = 0; i < n; ++i) RTM_RT(m, i, rt_sync(m->dev[i].ctx));
        for (int i = 0; i < n; ++i) {
            MultiDev& d = m->dev[i];
            if (d.surface) {
                rt_surface_free(d.ctx, d.surface);
                d.surface = nullptr;
            }
        }
        m->width = width;
        m->height = height;
        m->balanced = false;
    }
    for (int i = 0; i < n; ++i) {
        MultiDev& d = m->dev[i];
        const bool needsOwn = i > 0 || m->quantize;
        if (needsOwn && !d.surface) RTM_RT(m, i, rt_surface_create(d.ctx, width, height, &d.surface, &d.pitch));
    }
    if (sizeChanged || m->dev[0].tiles.empty()) {
        const int tiles = ((width + RT_TILE - 1) / RT_TILE) * ((height + RT_TILE - 1) / RT_TILE);
        for (int i = 0; i < n; ++i) {
            m->dev[i].tiles.clear();
            for (int t = i; t < tiles; t += n) m->dev[i].tiles.push_back(t);  // tile t on device t % n: every device covers the image thinly
            centerOut(m->dev[i].tiles, width, height);
        }
        if (int r = installLists(m)) return r;
    }
    return RT_OK;
}

}  // namespace

extern "C" {

rt_multi* rt_multi_create(int ndev, const int* devices) {
    if (ndev < 1 || ndev > 64 || !devices) {
        failm(nullptr, RT_ERR_INVALID, "rt_multi_create: 1..64 devices");
        return nullptr;
    }
    rt_multi* m = new rt_multi();
    m->dev.resize(ndev);
    for (int i = 0; i < ndev; ++i) {
        MultiDev& d = m->dev[i];
        d.device = devices[i];
        d.ctx = rt_create(devices[i]);
        if (!d.ctx) {
            failm(nullptr, RT_ERR_DEVICE, "rt_multi_create: device %d: %s", devices[i], rt_last_error(nullptr));
            rt_multi_destroy(m);
            return nullptr;
        }
        // one stream per device carries the device's frames, its pack kernel and its copy to device 0
        if (hipSetDevice(d.device) != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&d.sent, hipEventDisableTiming) != hipSuccess || rt_set_stream(d.ctx, d.stream) != RT_OK) {
            failm(nullptr, RT_ERR_DEVICE, "rt_multi_create: cannot set up device %d", devices[i]);
            rt_multi_destroy(m);
            return nullptr;
        }
    }
    // direct copies between device 0 and its peers where the hardware allows them (without, the copies are staged)
    for (int i = 1; i < ndev; ++i) {
        if (devices[i] == devices[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[0], devices[i]) == hipSuccess && can) {
            (void)hipSetDevice(devices[0]);
            (void)hipDeviceEnablePeerAccess(devices[i], 0);
            (void)hipSetDevice(devices[i]);
            (void)hipDeviceEnablePeerAccess(devices[0], 0);
        }
        (void)hipGetLastError();  // "already enabled" is fine
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreateWithFlags(&m->scattered, hipEventDisableTiming) != hipSuccess) {
        failm(nullptr, RT_ERR_DEVICE, "rt_multi_create: cannot set up device %d", devices[0]);
        rt_multi_destroy(m);
        return nullptr;
    }
    return m;
}

void rt_multi_destroy(rt_multi* m) {
    if (!m) return;
    const int dev0 = m->dev.empty() ? 0 : m->dev[0].device;
    for (MultiDev& d : m->dev)
        if (d.ctx) (void)rt_sync(d.ctx);
    for (size_t i = 0; i < m->dev.size(); ++i) {
        MultiDev& d = m->dev[i];
        if (!d.ctx) continue;
        if (d.scene) rt_scene_free(d.scene);
        if (d.surface) rt_surface_free(d.ctx, d.surface);
        d.surface = nullptr;
        freeFrameBuffers(d, dev0);
        hipSetDevice(d.device);
        if (d.blob) hipFree(d.blob);
        if (d.sent) hipEventDestroy(d.sent);
        rt_destroy(d.ctx);  // synchronises the stream it was given
        if (d.stream) hipStreamDestroy(d.stream);
    }
    if (m->scattered) {
        hipSetDevice(dev0);
        hipEventDestroy(m->scattered);
    }
    delete m;
}

const char* rt_multi_last_error(const rt_multi* m) { return m ? m->error.c_str() : g_multiCreateError.c_str(); }

int rt_multi_device_count(const rt_multi* m) { return m ? (int)m->dev.size() : 0; }

rt_ctx* rt_multi_context(rt_multi* m, int i) { return (m && i >= 0 && i < (int)m->dev.size()) ? m->dev[i].ctx : nullptr; }

int rt_multi_scene_upload(rt_multi* m, const RtSceneDesc* desc) {
    if (!m) return RT_ERR_INVALID;
    const int n = (int)m->dev.size();
    for (int i = 0; i < n; ++i) {
        MultiDev& d = m->dev[i];
        RTM_RT(m, i, rt_sync(d.ctx));
        if (d.scene) rt_scene_free(d.scene);
        d.scene = nullptr;
        if (d.blob) {
            RTM_HIP(m, hipSetDevice(d.device));
            hipFree(d.blob);
            d.blob = nullptr;
        }
    }
    // flattened (and, if device 0's context says so, refitted / built) once, on device 0
    m->dev[0].scene = rt_scene_upload(m->dev[0].ctx, desc);
    if (!m->dev[0].scene) return failm(m, RT_ERR_INVALID, "device %d: rt_scene_upload: %s", m->dev[0].device, rt_last_error(m->dev[0].ctx));
    const void* blob0 = nullptr;
    size_t bytes = 0;
    RTM_RT(m, 0, rt_scene_blob(m->dev[0].scene, &blob0, &bytes));
    // fan-out: every peer pulls the blob over its own link to device 0, all copies in flight together
    for (int i = 1; i < n; ++i) {
        MultiDev& d = m->dev[i];
        RTM_HIP(m, hipSetDevice(d.device));
        RTM_HIP(m, hipMalloc(&d.blob, bytes));
        RTM_HIP(m, hipMemcpyPeerAsync(d.blob, d.device, blob0, m->dev[0].device, bytes, d.stream));
    }
    for (int i = 1; i < n; ++i) {
        MultiDev& d = m->dev[i];
        RTM_HIP(m, hipSetDevice(d.device));
        RTM_HIP(m, hipStreamSynchronize(d.stream));
        d.scene = rt_scene_adopt_blob(d.ctx, d.blob, bytes);
        if (!d.scene) return failm(m, RT_ERR_INVALID, "device %d: rt_scene_adopt_blob: %s", d.device, rt_last_error(d.ctx));
    }
    return RT_OK;
}

int rt_multi_scene_set_lights(rt_multi* m, const RtLight* lights, int lightCnt) {
    if (!m) return RT_ERR_INVALID;
    for (size_t i = 0; i < m->dev.size(); ++i) {
        if (!m->dev[i].scene) return failm(m, RT_ERR_INVALID, "rt_multi_scene_set_lights: no scene");
        RTM_RT(m, i, rt_scene_set_lights(m->dev[i].scene, lights, lightCnt));
    }
    return RT_OK;
}

int rt_multi_set_quantize(rt_multi* m, int on) {
    if (!m) return RT_ERR_INVALID;
    for (size_t i = 0; i < m->dev.size(); ++i) RTM_RT(m, i, rt_sync(m->dev[i].ctx));
    m->quantize = on != 0;
    return RT_OK;
}

int rt_multi_balance(rt_multi* m, const RtDeviceOpts* opts, const RtCamera* cam) {
    if (!m || !opts || !cam) return RT_ERR_INVALID;
    if (!m->dev[0].scene) return failm(m, RT_ERR_INVALID, "rt_multi_balance: no scene");
    if (opts->width < 1 || opts->height < 1) return failm(m, RT_ERR_INVALID, "rt_multi_balance: bad frame size");
    if (int r = ensureFrame(m, opts->width, opts->height)) return r;
    const int n = (int)m->dev.size();
    const int tiles = ((opts->width + RT_TILE - 1) / RT_TILE) * ((opts->height + RT_TILE - 1) / RT_TILE);
    // a counter frame: every device books the box + triangle tests of its current tiles
    std::vector<uint64_t> cost((size_t)tiles, 0);
    std::vector<uint32_t> part((size_t)tiles);
    for (int i = 0; i < n; ++i) {
        MultiDev& d = m->dev[i];
        if (d.tiles.empty()) continue;
        void *devCost = nullptr, *devCnt = nullptr, *scratchSurface = nullptr;
        size_t scratchPitch = 0;
        RTM_RT(m, i, rt_device_alloc(d.ctx, (size_t)tiles * sizeof(uint32_t), &devCost));
        RTM_RT(m, i, rt_device_alloc(d.ctx, RT_COUNTER_CNT * sizeof(uint64_t), &devCnt));
        void* target = d.surface;
        size_t pitch = d.pitch;
        if (!target) {
            RTM_RT(m, i, rt_surface_create(d.ctx, opts->width, opts->height, &scratchSurface, &scratchPitch));
            target = scratchSurface;
            pitch = scratchPitch;
        }
        RtFrameAux aux = {nullptr, static_cast<uint64_t*>(devCnt), 0, 1};
        int r = rt_set_tile_cost_buffer(d.ctx, static_cast<uint32_t*>(devCost));
        if (r == RT_OK) r = rt_render(d.ctx, d.scene, opts, cam, target, pitch, opts->blockSize, &aux);
        if (r == RT_OK) r = rt_sync(d.ctx);
        if (r == RT_OK) r = rt_device_download(d.ctx, devCost, (size_t)tiles * sizeof(uint32_t), part.data());
        (void)rt_set_tile_cost_buffer(d.ctx, nullptr);
        rt_device_free(d.ctx, devCost);
        rt_device_free(d.ctx, devCnt);
        if (scratchSurface) rt_surface_free(d.ctx, scratchSurface);
        if (r != RT_OK) return failm(m, r, "device %d: counter frame: %s", d.device, rt_last_error(d.ctx));
        for (int t = 0; t < tiles; ++t) cost[(size_t)t] += part[(size_t)t];
    }
    // Tiles by falling cost, dealt out in rounds of n: within a round the device that is behind gets the most
    // expensive tile (longest-processing-time rule with equal counts, so that the gather buffers keep their size).
    std::vector<int32_t> order((size_t)tiles);
    for (int t = 0; t < tiles; ++t) order[(size_t)t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cost[(size_t)a] > cost[(size_t)b]; });
    std::vector<uint64_t> load((size_t)n, 0);
    for (MultiDev& d : m->dev) d.tiles.clear();
    for (int start = 0; start < tiles; start += n) {
        std::vector<int> ranks((size_t)n);
        for (int i = 0; i < n; ++i) ranks[(size_t)i] = i;
        std::stable_sort(ranks.begin(), ranks.end(), [&](int a, int b) {
            if (m->dev[a].tiles.size() != m->dev[b].tiles.size()) return m->dev[a].tiles.size() < m->dev[b].tiles.size();
            return load[(size_t)a] < load[(size_t)b];
        });
        for (int k = 0; k < n && start + k < tiles; ++k) {
            const int32_t t = order[(size_t)(start + k)];
            m->dev[ranks[(size_t)k]].tiles.push_back(t);
            load[(size_t)ranks[(size_t)k]] += cost[(size_t)t];
        }
    }
    for (MultiDev& d : m->dev) centerOut(d.tiles, opts->width, opts->height);  // which tiles balances the load; their order keeps neighbours together
    m->balanced = true;
    return installLists(m);
}

int rt_multi_tile_counts(const rt_multi* m, int* counts) {
    if (!m || !counts) return RT_ERR_INVALID;
    for (size_t i = 0; i < m->dev.size(); ++i) counts[i] = (int)m->dev[i].tiles.size();
    return RT_OK;
}

int rt_multi_render(rt_multi* m, const RtDeviceOpts* opts, const RtCamera* cam, void* devSurface, size_t pitch, int blockDim) {
    if (!m || !opts || !cam || !devSurface) return RT_ERR_INVALID;
    if (!m->dev[0].scene) return failm(m, RT_ERR_INVALID, "rt_multi_render: no scene");
    if (opts->width < 1 || opts->height < 1) return failm(m, RT_ERR_INVALID, "rt_multi_render: bad frame size");
    const size_t pixelBytes = m->quantize ? 4 : 16;
    if (pitch < (size_t)opts->width * pixelBytes || pitch % pixelBytes != 0 || (reinterpret_cast<uintptr_t>(devSurface) & (pixelBytes - 1)))
        return failm(m, RT_ERR_INVALID, "rt_multi_render: surface pitch / alignment must hold %d pixels of %zu bytes per row", opts->width, pixelBytes);
    if (int r = ensureFrame(m, opts->width, opts->height)) return r;
    const int n = (int)m->dev.size();
    const int dev0 = m->dev[0].device;
    // every device starts on its tiles ...
    for (int i = 0; i < n; ++i) {
        MultiDev& d = m->dev[i];
        if (d.tiles.empty()) continue;
        const bool direct = i == 0 && !m->quantize;  // device 0 renders its own tiles straight into the caller's surface
        RTM_RT(m, i, rt_render(d.ctx, d.scene, opts, cam, direct ? devSurface : d.surface, direct ? pitch : d.pitch, blockDim, nullptr));
    }
    // ... then packs them and sends them to device 0 (one copy per device, each over its own link) ...
    for (int i = 0; i < n; ++i) {
        MultiDev& d = m->dev[i];
        if (d.tiles.empty() || (i == 0 && !m->quantize)) continue;
        RTM_HIP(m, hipSetDevice(d.device));
        if (m->scatteredValid) RTM_HIP(m, hipStreamWaitEvent(d.stream, m->scattered, 0));  // the previous frame has left the receive buffer
        RTM_HIP(m, (hipError_t)rtk_pack_tiles(d.surface, d.pitch, opts->width, opts->height, d.devTiles, (int)d.tiles.size(), d.packed, m->quantize ? 1 : 0, d.stream));
        if (i > 0) RTM_HIP(m, hipMemcpyPeerAsync(d.recv, dev0, d.packed, d.device, d.tiles.size() * 64 * pixelBytes, d.stream));
        RTM_HIP(m, hipEventRecord(d.sent, d.stream));
    }
    // ... where they are written to their places in the frame
    RTM_HIP(m, hipSetDevice(dev0));
    bool any = false;
    for (int i = 0; i < n; ++i) {
        MultiDev& d = m->dev[i];
        if (d.tiles.empty() || (i == 0 && !m->quantize)) continue;
        if (i > 0) RTM_HIP(m, hipStreamWaitEvent(m->dev[0].stream, d.sent, 0));
        RTM_HIP(m, (hipError_t)rtk_scatter_tiles(i == 0 ? d.packed : d.recv, d.tilesOn0, (int)d.tiles.size(), devSurface, pitch, opts->width, opts->height,
                                                  m->quantize ? 1 : 0, m->dev[0].stream));
        any = true;
    }
    if (any) {
        RTM_HIP(m, hipEventRecord(m->scattered, m->dev[0].stream));
        m->scatteredValid = true;
    }
    return RT_OK;
}

int rt_multi_sync(rt_multi* m) {
    if (!m) return RT_ERR_INVALID;
    int first = RT_OK;
    // peers first: device 0's stream waits for their copies
    for (int i = (int)m->dev.size() - 1; i >= 0; --i) {
        const int r = rt_sync(m->dev[(size_t)i].ctx);
        if (r != RT_OK && first == RT_OK) first = failm(m, r, "device %d: rt_sync: %s", m->dev[(size_t)i].device, rt_last_error(m->dev[(size_t)i].ctx));
    }
    return first;
}

}  // extern "C"
