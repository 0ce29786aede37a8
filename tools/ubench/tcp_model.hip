// Microbenchmark (development aid): cost of divergent 16-byte loads from an L2-resident table
// as a function of active lanes and of how many lanes share a 64-byte line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

template <int MODE>
__global__ void k(const float4* __restrict__ tab, const unsigned* __restrict__ idx, int iters, float* out, int activeLanes, unsigned mask) {
    const int lane = threadIdx.x & 63;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0;
    unsigned j = idx[gid];
    if (lane < activeLanes) {
        for (int i = 0; i < iters; ++i) {
            if (MODE == 0) {  // each lane: 4 x 16B of its own 64B line (node-pair pattern)
                const float4* p = tab + 4 * (size_t)j;
                float4 a = p[0], b = p[1], c = p[2], d = p[3];
                acc += a.x + b.y + c.z + d.w;
                j = (j * 1664525u + 1013904223u + __float_as_uint(a.x)) & mask;
            } else if (MODE == 1) {  // each lane: 1 x 16B of its own line
                const float4* p = tab + 4 * (size_t)j;
                float4 a = p[0];
                acc += a.x;
                j = (j * 1664525u + 1013904223u + __float_as_uint(a.x)) & mask;
            } else {  // groups of 4 lanes fetch one 64B line cooperatively (1 x 16B each, consecutive)
                unsigned jj = __shfl(j, lane & ~3, 64);
                const float4* p = tab + 4 * (size_t)jj + (lane & 3);
                float4 a = p[0];
                acc += a.x;
                j = (j * 1664525u + 1013904223u + __float_as_uint(a.x)) & mask;
            }
        }
    }
    out[gid] = acc;
}

int main() {
    const size_t lines = 1u << 20;  // 1M x 64B = 64 MB table (L2 + Infinity Cache resident)
    float4* tab; unsigned* idx; float* out;
    const int blocks = 256 * 16, threads = 64;
    hipMalloc(&tab, lines * 64); hipMemset(tab, 0, lines * 64);
    hipMalloc(&idx, blocks * threads * 4); hipMalloc(&out, blocks * threads * 4);
    std::vector<unsigned> h(blocks * threads);
    for (auto& v : h) v = (unsigned)rand() & 0xffu;
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    for (unsigned mask : {0xffu, 0x7fffu, 0xfffffu})
    for (int mode = 0; mode < 3; ++mode)
        for (int lanes : {64, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) k<0><<<blocks, threads>>>(tab, idx, iters, out, lanes, mask);
                if (mode == 1) k<1><<<blocks, threads>>>(tab, idx, iters, out, lanes, mask);
                if (mode == 2) k<2><<<blocks, threads>>>(tab, idx, iters, out, lanes, mask);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            double steps = (double)blocks * iters;  // wave-steps
            printf("table %7u lines mode %d lanes %2d: %.3f ms, %.1f ns per wave-step per CU-slot, lane-loads/s %.2f G\n", mask + 1, mode, lanes, ms,
                   ms * 1e6 / (steps / 256.0), steps * lanes * (mode == 0 ? 4 : 1) / ms / 1e6);
        }
    return 0;
}
