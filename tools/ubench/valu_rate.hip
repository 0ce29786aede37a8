// Microbenchmark (development aid): cycles a SIMD needs per wavefront-wide vector instruction, by instruction
// kind and by resident wavefronts per SIMD.  Every kind is a block of 8 independent instructions in inline
// assembly, so that the compiler cannot fuse, pack or drop them.  Cycles assume the clock GRBM reports (2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
#define BLOCK(name, body)                                                                   \
    __global__ void __launch_bounds__(64) name(float* out, int iters, float seed) {         \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = 1.0000001f, b1 = 1e-9f;                                                  \
        for (int i = 0; i < iters; ++i) {                                                   \
            REP8(asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "vcc");) \
        }                                                                                   \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;         \
    }

BLOCK(kFma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
BLOCK(kMul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
            "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
BLOCK(kAdd, "v_add_f32 %0, %0, %9\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %9\n v_add_f32 %3, %3, %9\n"
            "v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n")
BLOCK(kMov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
            "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n")
BLOCK(kAddU, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
BLOCK(kCnd, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
            "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
BLOCK(kCmp, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
            "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n")
BLOCK(kCmpCnd, "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
               "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n")
BLOCK(kMinMax, "v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
               "v_max3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n")
BLOCK(kRcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
            "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
BLOCK(kMbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0\n v_mbcnt_hi_u32_b32 %1, -1, %1\n v_mbcnt_lo_u32_b32 %2, -1, %2\n v_mbcnt_hi_u32_b32 %3, -1, %3\n"
              "v_lshlrev_b32 %4, 3, %4\n v_and_b32 %5, %5, %8\n v_lshl_add_u32 %6, %6, 3, %8\n v_and_or_b32 %7, %7, %8, %9\n")

// v_pk_mul_f32 needs 64-bit operands: its own kernel
__global__ void __launch_bounds__(64) kPk(float* out, int iters, float seed) {
    typedef float F2 __attribute__((ext_vector_type(2)));
    F2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    F2 b = {1.0000001f, 1.0000002f};
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                          "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b));)
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y;
}

typedef void (*Kern)(float*, int, float);

int main() {
    float* out;
    hipMalloc(&out, 256 * 4 * 16 * 64 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 4000;
    struct {
        const char* name;
        Kern k;
    } kinds[] = {{"v_fma_f32", kFma}, {"v_mul_f32", kMul}, {"v_add_f32", kAdd}, {"v_mov_b32", kMov}, {"v_add_u32", kAddU},
                 {"v_cndmask_b32", kCnd}, {"v_cmp_lt_f32", kCmp}, {"v_cmp + v_cndmask", kCmpCnd}, {"v_min/max/min3/max3_f32", kMinMax},
                 {"v_pk_mul_f32", kPk}, {"v_rcp_f32", kRcp}, {"mbcnt/shift/and mix", kMbcnt}};
    for (auto& kind : kinds) {
        printf("%-26s", kind.name);
        for (int wavesPerSimd : {1, 2, 4, 8}) {
            const int blocks = 256 * 4 * wavesPerSimd;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                kind.k<<<blocks, 64>>>(out, iters, 1.0f);
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            const double insts = (double)wavesPerSimd * iters * 64;  // per SIMD
            printf("  %d w/SIMD: %5.2f cyc", wavesPerSimd, ms * 1e-3 * 2.4e9 / insts);
        }
        printf("\n");
    }
    return 0;
}
