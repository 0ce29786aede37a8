// rt_tiles.hip - the two ends of the multi-GPU tile gather.
//
// Every GPU renders a list of 8x8-pixel tiles of the same frame into its own surface.  packTilesKernel copies the
// pixels of those tiles out of the surface into one contiguous buffer (64 pixels per tile, tile after tile, in list
// order), which travels to GPU 0 as ONE device-to-device copy; scatterTilesKernel there writes the tiles to their
// places in the frame.  With `quantize` the packed pixels are 8-bit RGBA made by the reference's dump rule
// clamp((int)(v * 256), 0, 255), alpha included (cudaTracer/PicDumper.h:43-55) - a quarter of the bytes for a
// consumer that only wants the picture - and the scatter writes an 8-bit surface.
//
// Both are pure data movement: one thread per pixel, a tile's 64 pixels are 8 runs of 128 contiguous bytes in the
// surface and 1 KiB contiguous in the packed buffer.  The reference has no counterpart (one GPU, one surface).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../rt_device_format.h"

namespace {

__device__ __forceinline__ unsigned char dumpByte(float v) {
    // (int)(v * 256) clamped to 0..255; a NaN gives 0 (both comparisons fail), like the host's conversion in the dump
    const float s = v * 256.0f;
    return s >= 255.0f ? (unsigned char)255 : (s > 0.0f ? (unsigned char)(int)s : (unsigned char)0);
}

__global__ void __launch_bounds__(256) packTilesKernel(const unsigned char* __restrict__ surface, size_t pitch, int width, int height, int tilesX,
                                                       const int32_t* __restrict__ tiles, int tileCnt, void* __restrict__ out, int quantize) {
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if (p >= (unsigned)tileCnt * 64u) return;
    const int tile = tiles[p >> 6];
    const int x = (tile % tilesX) * RT_TILE + (int)(p & 7u), y = (tile / tilesX) * RT_TILE + (int)((p >> 3) & 7u);
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // pixels of an edge tile outside the frame
    if (x < width && y < height) v = *reinterpret_cast<const float4*>(surface + (size_t)y * pitch + (size_t)x * 16);
    if (quantize)
        reinterpret_cast<uchar4*>(out)[p] = make_uchar4(dumpByte(v.x), dumpByte(v.y), dumpByte(v.z), dumpByte(v.w));
    else
        reinterpret_cast<float4*>(out)[p] = v;
}

__global__ void __launch_bounds__(256) scatterTilesKernel(const void* __restrict__ in, const int32_t* __restrict__ tiles, int tileCnt, int tilesX,
                                                          unsigned char* __restrict__ surface, size_t pitch, int width, int height, int quantized) {
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    if (p >= (unsigned)tileCnt * 64u) return;
    const int tile = tiles[p >> 6];
    const int x = (tile % tilesX) * RT_TILE + (int)(p & 7u), y = (tile / tilesX) * RT_TILE + (int)((p >> 3) & 7u);
    if (x >= width || y >= height) return;
    if (quantized)
        *reinterpret_cast<uchar4*>(surface + (size_t)y * pitch + (size_t)x * 4) = reinterpret_cast<const uchar4*>(in)[p];
    else
        *reinterpret_cast<float4*>(surface + (size_t)y * pitch + (size_t)x * 16) = reinterpret_cast<const float4*>(in)[p];
}

}  // namespace

int rtk_pack_tiles(const void* surface, size_t pitch, int width, int height, const int32_t* tiles, int tileCnt, void* out, int quantize, void* stream) {
    if (tileCnt <= 0) return 0;
    const int tilesX = (width + RT_TILE - 1) / RT_TILE;
    hipLaunchKernelGGL(packTilesKernel, dim3(((unsigned)tileCnt * 64u + 255u) / 256u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const unsigned char*>(surface), pitch, width, height, tilesX, tiles, tileCnt, out, quantize);
    return (int)hipGetLastError();
}

int rtk_scatter_tiles(const void* in, const int32_t* tiles, int tileCnt, void* surface, size_t pitch, int width, int height, int quantized, void* stream) {
    if (tileCnt <= 0) return 0;
    const int tilesX = (width + RT_TILE - 1) / RT_TILE;
    hipLaunchKernelGGL(scatterTilesKernel, dim3(((unsigned)tileCnt * 64u + 255u) / 256u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, tiles,
                       tileCnt, tilesX, reinterpret_cast<unsigned char*>(surface), pitch, width, height, quantized);
    return (int)hipGetLastError();
}
