// Render options of the host API (cudaTracer/Options.cuh:11-44).  DeviceOpts is
// the per-frame block the reference copies to its kernel; it is bit-identical to
// RtDeviceOpts and crosses the C-ABI as is.  loadDefaults() reproduces the
// reference's start-up values (cudaTracer/main.cpp:44-64).
#ifndef RT_HOST_OPTIONS_H
#define RT_HOST_OPTIONS_H

#include <cmath>
#include <cstdint>
#include <string>

#include "../../../include/rt_capi.h"
#include "Mat.h"
#include "Vec.h"

namespace rt {

struct ModelDesc {
    std::string folder;
    std::string fname;
    Mat44f transform = Mat44f::identity();  // the reference leaves this unset until a JSON fills it
};

struct HostOpts {
    ModelDesc model;
    int lightCnt = 0;
    float fov = 90;
};

struct DeviceOpts {
    float imageAspectRatio = 1;
    float scale = 1;
    int width = 0;
    int height = 0;
    Vec3f backgroundColor;
    float shadowBias = 0;
    int blockSize = 4;
    int maxDepth = 0;
    int samples = 1;
    bool gammaCorrection = false;
    bool useSceneBvh = true;
    bool useMeshBvhs = true;
    const RtDeviceOpts* abi() const { return reinterpret_cast<const RtDeviceOpts*>(this); }
};
static_assert(sizeof(DeviceOpts) == sizeof(RtDeviceOpts) && sizeof(DeviceOpts) == 48, "DeviceOpts layout");

struct Options {
    HostOpts host;
    DeviceOpts device;
};

inline float deg2rad(const float& deg) { return deg * PI / 180; }

// scale = tan(fov / 2), aspect = width / height, as the reference derives them.
inline void setView(Options& o, int width, int height, float fovDeg) {
    o.host.fov = fovDeg;
    o.device.width = width;
    o.device.height = height;
    o.device.scale = tan(deg2rad(o.host.fov * 0.5f));
    o.device.imageAspectRatio = o.device.width / (float)o.device.height;
}

inline Options loadDefaults() {
    Options o;
    o.host.lightCnt = 10;
    o.device.maxDepth = 5;
    o.device.samples = 1;
    o.device.backgroundColor = Vec3f(0.2f, 0.7f, 0.8f);
    o.device.shadowBias = 0.0001f;
    o.device.blockSize = 4;
    o.device.useMeshBvhs = true;
    o.device.useSceneBvh = true;
    setView(o, 1024, 768, 90);
    o.device.gammaCorrection = true;
    return o;
}

}  // namespace rt
#endif
