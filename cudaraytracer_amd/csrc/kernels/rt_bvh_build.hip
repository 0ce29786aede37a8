// rt_bvh_build.hip - mesh BVH construction on the GPU.
//
// Builds the same tree as the host builder (csrc/host/BVH.h, which follows
// cudaTracer/BVH.h:104-181): a range of primitives becomes a leaf when it holds
// <= threshold primitives, otherwise it is cut at mid = (start + end) / 2 after
// ordering it by box centroid along the widest axis of the centroid bounds
// (ties Z over Y over X, cudaTracer/BoundingBox.cuh:59-68).  Because the cut is
// always the median position, the SHAPE of the tree depends only on the
// primitive count: the host lays out the node records (references, leaf ranges)
// beforehand, and the device only has to order the primitives and fill in
// boxes and split axes.
//
// Level by level, every primitive walks the implicit tree to the range that
// holds its position, the ranges' centroid bounds are reduced with atomics, and
// ONE global radix sort per level orders all ranges at once (key = range id in
// the high word, centroid coordinate in the low word; the range ids grow with
// the position, so ranges stay where they are).  The sort is this file's own
// (radixSortPairs below: least significant digit first, 8 bits per pass, stable).  The reference selects the
// median with nth_element; a full sort puts the same primitives on either side
// whenever the median value is unique, so node boxes and axes are those of the
// host builder, while the order inside a leaf may differ.
//
// Equal centroid coordinates: the sort is stable, so primitives whose keys tie keep the order they had (at first,
// the caller's order); the cut then falls between two equal keys, as it may with any nth_element.  Not reproduced:
// the reference turns a range whose centroids ALL coincide into a leaf of any size (cudaTracer/BVH.h:126-128).  Here
// such a range keeps its order and is cut at the median position like any other (split axis Z by the tie rule), so
// the tree keeps the shape the count dictates; the boxes are right either way, and only ties between primitives
// with identical box centroids can resolve differently.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include <utility>

#include "../rt_device_format.h"

namespace {

constexpr int kBlock = 256;

// float <-> unsigned with the same ordering
__device__ __forceinline__ uint32_t orderedBits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fromOrderedBits(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Walks `levels` cuts down from the whole range towards position i.  Returns true if the range reached is
// one that is cut at this level (more than `threshold` primitives).  `id` is the range's heap index (root 1);
// a range that stopped being cut earlier keeps a unique id by continuing to the left.
__device__ __forceinline__ bool descend(int i, int n, int threshold, int levels, int& s, int& e, uint32_t& id) {
    s = 0;
    e = n;
    id = 1u;
    for (int l = 0; l < levels; ++l) {
        if (e - s <= threshold) {
            id <<= (levels - l);
            return false;
        }
        const int mid = (s + e) / 2;
        if (i < mid) {
            e = mid;
            id = 2u * id;
        } else {
            s = mid;
            id = 2u * id + 1u;
        }
    }
    return e - s > threshold;
}

// csrc/host/BVH.h triangleBox: the reference box starts at (FLT_MAX, FLT_MIN), the tight one at (FLT_MAX, -FLT_MAX)
__global__ void __launch_bounds__(kBlock) primBoundsKernel(int n, const int32_t* __restrict__ tris, const unsigned char* __restrict__ vertices,
                                                           int tight, float4* __restrict__ bmin, float4* __restrict__ bmax) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    const float top = tight ? -FLT_MAX : FLT_MIN;
    float hi[3] = {top, top, top};
    for (int c = 0; c < 3; ++c) {
        const float* pos = reinterpret_cast<const float*>(vertices + (size_t)tris[3 * i + c] * sizeof(RtVertex) + 44);
        for (int a = 0; a < 3; ++a) {
            const float p = pos[a];
            if (p > hi[a]) hi[a] = p;
            if (p < lo[a]) lo[a] = p;
        }
    }
    bmin[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
    bmax[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
}

__global__ void __launch_bounds__(kBlock) iotaKernel(int n, uint32_t* order) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) order[i] = (uint32_t)i;
}

// centroid bounds of an empty range: {min x, y, z, max x, y, z} as ordered bits
__global__ void __launch_bounds__(kBlock) boundsInitKernel(uint32_t first, uint32_t count, int tight, uint32_t* cb) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    uint32_t* b = cb + 6 * (size_t)(first + k);
    const uint32_t lo = orderedBits(FLT_MAX), hi = orderedBits(tight ? -FLT_MAX : FLT_MIN);
    b[0] = b[1] = b[2] = lo;
    b[3] = b[4] = b[5] = hi;
}

// BoundingBox::centroid(): (bbmin + bbmax) / 2
__device__ __forceinline__ void centroidOf(const float4 lo, const float4 hi, float c[3]) {
    c[0] = (lo.x + hi.x) / 2.0f;
    c[1] = (lo.y + hi.y) / 2.0f;
    c[2] = (lo.z + hi.z) / 2.0f;
}

__global__ void __launch_bounds__(kBlock) centroidBoundsKernel(int n, int threshold, int level, const uint32_t* __restrict__ order,
                                                               const float4* __restrict__ bmin, const float4* __restrict__ bmax, uint32_t* cb) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    int s = 0, e = 0;
    uint32_t id = 0;
    const bool active = i < n && descend(i, n, threshold, level, s, e, id);
    uint32_t lo[3], hi[3];
    if (active) {
        const uint32_t prim = order[i];
        float c[3];
        centroidOf(bmin[prim], bmax[prim], c);
        for (int a = 0; a < 3; ++a) lo[a] = hi[a] = orderedBits(c[a]);
    } else {
        for (int a = 0; a < 3; ++a) {
            lo[a] = 0xffffffffu;
            hi[a] = 0u;
        }
    }
    // Ranges are runs of consecutive positions, so near the root a whole wavefront sits in one range: reduce
    // there first and let one lane touch memory, instead of 64 atomics on the same six words.
    const unsigned long long act = __ballot(active);
    if (act == 0ull) return;
    const uint32_t firstId = __shfl(id, __ffsll((long long)act) - 1, 64);
    const bool uniform = __ballot(active && id != firstId) == 0ull;
    if (uniform) {
        for (int off = 32; off > 0; off >>= 1)
            for (int a = 0; a < 3; ++a) {
                const uint32_t l = __shfl_xor(lo[a], off, 64), h = __shfl_xor(hi[a], off, 64);
                lo[a] = l < lo[a] ? l : lo[a];
                hi[a] = h > hi[a] ? h : hi[a];
            }
        if ((threadIdx.x & 63) == 0) {
            uint32_t* b = cb + 6 * (size_t)(firstId - 1u);
            for (int a = 0; a < 3; ++a) {
                atomicMin(b + a, lo[a]);
                atomicMax(b + 3 + a, hi[a]);
            }
        }
    } else if (active) {
        uint32_t* b = cb + 6 * (size_t)(id - 1u);
        for (int a = 0; a < 3; ++a) {
            atomicMin(b + a, lo[a]);
            atomicMax(b + 3 + a, hi[a]);
        }
    }
}

__global__ void __launch_bounds__(kBlock) sortKeyKernel(int n, int threshold, int level, const uint32_t* __restrict__ order,
                                                        const float4* __restrict__ bmin, const float4* __restrict__ bmax,
                                                        const uint32_t* __restrict__ cb, unsigned char* __restrict__ axisOut,
                                                        uint64_t* __restrict__ keys) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int s, e;
    uint32_t id;
    if (!descend(i, n, threshold, level, s, e, id)) {
        keys[i] = (uint64_t)id << 32 | (uint32_t)(i - s);  // a finished range keeps its order
        return;
    }
    const uint32_t* b = cb + 6 * (size_t)(id - 1u);
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = fromOrderedBits(b[a]);
        hi[a] = fromOrderedBits(b[3 + a]);
    }
    // BoundingBox::maxExtent
    const float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
    const int dim = (d0 > d1 && d0 > d2) ? 0 : (d1 > d2 ? 1 : 2);
    // (hi[dim] == lo[dim]: all centroids coincide, every key of the range ties and the range keeps its order)
    if (i == s) axisOut[id - 1u] = (unsigned char)dim;
    const uint32_t prim = order[i];
    float c[3];
    centroidOf(bmin[prim], bmax[prim], c);
    keys[i] = (uint64_t)id << 32 | orderedBits(c[dim]);
}

// Per-node work table built by the host from the primitive count alone.
struct BuildNode {
    int32_t start, end;   // primitive positions of the node's range
    int32_t rec;          // index of the node's 32-byte record in the mesh's node array
    int32_t left, right;  // records of the children (-1 for a leaf)
    uint32_t pathId;      // heap index of the range (root 1)
};

__global__ void __launch_bounds__(kBlock) leafBoxKernel(int count, const BuildNode* __restrict__ nodes, int tight, const uint32_t* __restrict__ order,
                                                        const float4* __restrict__ bmin, const float4* __restrict__ bmax, float* __restrict__ records) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const BuildNode nd = nodes[k];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    const float top = tight ? -FLT_MAX : FLT_MIN;
    float hi[3] = {top, top, top};
    for (int i = nd.start; i < nd.end; ++i) {
        const uint32_t prim = order[i];
        const float4 a = bmin[prim], b = bmax[prim];
        lo[0] = fminf(lo[0], a.x); lo[1] = fminf(lo[1], a.y); lo[2] = fminf(lo[2], a.z);
        hi[0] = fmaxf(hi[0], b.x); hi[1] = fmaxf(hi[1], b.y); hi[2] = fmaxf(hi[2], b.z);
    }
    float* r = records + 8 * (size_t)nd.rec;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2];
    r[3] = hi[0]; r[4] = hi[1]; r[5] = hi[2];
}

// One level of interior nodes, deepest level first: box = union of the children, axis into the reference word.
__global__ void __launch_bounds__(kBlock) interiorBoxKernel(int count, const BuildNode* __restrict__ nodes, const unsigned char* __restrict__ axisOf,
                                                            float* __restrict__ records) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const BuildNode nd = nodes[k];
    const float* a = records + 8 * (size_t)nd.left;
    const float* b = records + 8 * (size_t)nd.right;
    float* r = records + 8 * (size_t)nd.rec;
    for (int c = 0; c < 3; ++c) {
        r[c] = fminf(a[c], b[c]);
        r[3 + c] = fmaxf(a[3 + c], b[3 + c]);
    }
    uint32_t ref = __float_as_uint(r[6]);
    ref = (ref & ~(3u << 29)) | ((uint32_t)axisOf[nd.pathId - 1u] << 29);
    r[6] = __uint_as_float(ref);
}

// Triangle records in leaf order.  The host builder visits the second half of every range first
// (cudaTracer/BVH.h:146-149), so the leaf over positions [s, e) owns ordered slots n - e ...
__global__ void __launch_bounds__(kBlock) recordKernel(int n, int threshold, const uint32_t* __restrict__ order, const int32_t* __restrict__ tris,
                                                       const unsigned char* __restrict__ vertices, TriRecord* __restrict__ records) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int s = 0, e = n;
    while (e - s > threshold) {
        const int mid = (s + e) / 2;
        if (i < mid)
            e = mid;
        else
            s = mid;
    }
    const uint32_t prim = order[i];
    const int32_t i0 = tris[3 * prim], i1 = tris[3 * prim + 1], i2 = tris[3 * prim + 2];
    const float* p0 = reinterpret_cast<const float*>(vertices + (size_t)i0 * sizeof(RtVertex) + 44);
    const float* p1 = reinterpret_cast<const float*>(vertices + (size_t)i1 * sizeof(RtVertex) + 44);
    const float* p2 = reinterpret_cast<const float*>(vertices + (size_t)i2 * sizeof(RtVertex) + 44);
    TriRecord rec;
    for (int k = 0; k < 3; ++k) {
        rec.v0[k] = p0[k];
        rec.e1[k] = p1[k] - p0[k];
        rec.e2[k] = p2[k] - p0[k];
    }
    rec.idx[0] = i0;
    rec.idx[1] = i1;
    rec.idx[2] = i2;
    records[(n - e) + (i - s)] = rec;
}

inline int blocksFor(long long n) { return (int)((n + kBlock - 1) / kBlock); }

// ---------------------------------------------------------------------------
// Stable radix sort of (64-bit key, 32-bit value) pairs, least significant digit first, 8 bits per pass.
// A pass is three launches: per-tile digit histograms, their exclusive prefix sums (digit-major, so that one sum
// runs over all tiles of digit 0, then digit 1, ...), and the scatter, in which every tile ranks its keys again and
// writes them to their places.  A tile is kSortTile consecutive keys; it is walked in rounds of one key per thread,
// so position order is (round, thread), and a key's rank among the tile's keys of the same digit is
//   keys of that digit in earlier rounds  +  in lower wavefronts of this round  +  in lower lanes of this wavefront,
// the last from eight ballots (the lanes whose digit agrees bit by bit), the others from small LDS tables.  Keys of one
// digit keep their order: the sort is stable, which the build relies on where centroid coordinates tie.
constexpr int kSortRounds = 16;
constexpr int kSortTile = kBlock * kSortRounds;
constexpr int kSortWaves = kBlock / 64;
static_assert(kBlock == 256, "one thread per digit value: the histogram and place tables are filled and scanned by thread index");

// lanes of this wavefront (among `valid` ones) whose digit equals this lane's
__device__ __forceinline__ unsigned long long sameDigit(unsigned digit, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = ((digit >> b) & 1u) != 0u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

__global__ void __launch_bounds__(kBlock) sortHistKernel(const uint64_t* __restrict__ keys, int n, int shift, uint32_t* __restrict__ hist, int tiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kSortTile;
    for (int r = 0; r < kSortRounds; ++r) {
        const size_t i = base + (size_t)r * kBlock + threadIdx.x;
        const bool valid = i < (size_t)n;
        const unsigned digit = valid ? (unsigned)(keys[i] >> shift) & 255u : 0u;
        const unsigned long long peers = sameDigit(digit, valid);
        // the lowest lane of every group of equal digits adds the group's size: a tile often holds ONE digit (range ids)
        if (valid && (peers & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0ull) atomicAdd(&h[digit], (uint32_t)__popcll(peers));
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * tiles + blockIdx.x] = h[threadIdx.x];
}

// One block per digit: exclusive prefix sums over the digit's tile counts, in place; the digit's total goes to total[digit].
__global__ void __launch_bounds__(kBlock) sortScanTilesKernel(uint32_t* __restrict__ hist, int tiles, uint32_t* __restrict__ total) {
    __shared__ uint32_t waveSum[kSortWaves];
    __shared__ uint32_t carry;
    uint32_t* row = hist + (size_t)blockIdx.x * tiles;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (int at = 0; at < tiles; at += kBlock) {
        const int i = at + threadIdx.x;
        const uint32_t v = i < tiles ? row[i] : 0u;
        uint32_t inc = v;  // inclusive prefix within the wavefront
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(inc, off, 64);
            if ((int)(threadIdx.x & 63) >= off) inc += u;
        }
        if ((threadIdx.x & 63) == 63) waveSum[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t before = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += waveSum[w];
        if (i < tiles) row[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[blockIdx.x] = carry;
}

// exclusive prefix sums over the 256 digit totals (one block)
__global__ void __launch_bounds__(256) sortScanDigitsKernel(uint32_t* __restrict__ total) {
    __shared__ uint32_t t[256];
    t[threadIdx.x] = total[threadIdx.x];
    __syncthreads();
    uint32_t sum = 0u;
    for (int d = 0; d < (int)threadIdx.x; ++d) sum += t[d];
    total[threadIdx.x] = sum;
}

__global__ void __launch_bounds__(kBlock) sortScatterKernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, int n, int shift,
                                                            const uint32_t* __restrict__ hist, int tiles, const uint32_t* __restrict__ digitBase,
                                                            uint64_t* __restrict__ keysOut, uint32_t* __restrict__ valsOut) {
    __shared__ uint32_t place[256];               // where the tile's next key of each digit goes
    __shared__ uint32_t inWave[kSortWaves][256];  // this round's keys of each digit, per wavefront
    place[threadIdx.x] = digitBase[threadIdx.x] + hist[(size_t)threadIdx.x * tiles + blockIdx.x];
    for (int w = 0; w < kSortWaves; ++w) inWave[w][threadIdx.x] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kSortTile;
    const unsigned wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
    for (int r = 0; r < kSortRounds; ++r) {
        const size_t i = base + (size_t)r * kBlock + threadIdx.x;
        if (base + (size_t)r * kBlock >= (size_t)n) break;  // (the same for the whole block)
        const bool valid = i < (size_t)n;
        uint64_t key = 0;
        uint32_t val = 0;
        if (valid) {
            key = keys[i];
            val = vals[i];
        }
        const unsigned digit = valid ? (unsigned)(key >> shift) & 255u : 0u;
        const unsigned long long peers = sameDigit(digit, valid);
        if (valid && (peers & below) == 0ull) inWave[wave][digit] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t at = place[digit] + (uint32_t)__popcll(peers & below);
            for (unsigned w = 0; w < wave; ++w) at += inWave[w][digit];
            keysOut[at] = key;
            valsOut[at] = val;
        }
        __syncthreads();
        {
            // thread d moves digit d's place past this round's keys and clears the round's counts
            uint32_t add = 0u;
            for (int w = 0; w < kSortWaves; ++w) {
                add += inWave[w][threadIdx.x];
                inWave[w][threadIdx.x] = 0u;
            }
            place[threadIdx.x] += add;
        }
        __syncthreads();
    }
}

// Bytes of scratch radixSortPairs needs for n pairs.
inline size_t radixSortTempBytes(int n) {
    const size_t tiles = ((size_t)n + kSortTile - 1) / kSortTile;
    return (256 * tiles + 256) * sizeof(uint32_t);
}

// Sorts n pairs by the key bits [0, endBit).  The pairs start in (keysA, valsA) and move between the two buffer pairs
// pass by pass; returns 0 if the sorted pairs are in (keysA, valsA), 1 if in (keysB, valsB).
int radixSortPairs(void* temp, uint64_t* keysA, uint64_t* keysB, uint32_t* valsA, uint32_t* valsB, int n, int endBit, hipStream_t stream) {
    const int tiles = (n + kSortTile - 1) / kSortTile;
    uint32_t* hist = static_cast<uint32_t*>(temp);
    uint32_t* total = hist + (size_t)256 * tiles;
    int in = 0;
    for (int shift = 0; shift < endBit; shift += 8) {
        const uint64_t* kin = in ? keysB : keysA;
        const uint32_t* vin = in ? valsB : valsA;
        uint64_t* kout = in ? keysA : keysB;
        uint32_t* vout = in ? valsA : valsB;
        hipLaunchKernelGGL(sortHistKernel, dim3(tiles), dim3(kBlock), 0, stream, kin, n, shift, hist, tiles);
        hipLaunchKernelGGL(sortScanTilesKernel, dim3(256), dim3(kBlock), 0, stream, hist, tiles, total);
        hipLaunchKernelGGL(sortScanDigitsKernel, dim3(1), dim3(256), 0, stream, total);
        hipLaunchKernelGGL(sortScatterKernel, dim3(tiles), dim3(kBlock), 0, stream, kin, vin, n, shift, hist, tiles, total, kout, vout);
        in ^= 1;
    }
    return in;
}

}  // namespace

static_assert(sizeof(BuildNode) == sizeof(RtBuildNode), "host and device agree on the work table");

#define RTB_HIP(call)                      \
    do {                                   \
        hipError_t e_ = (call);            \
        if (e_ != hipSuccess) {            \
            cleanup();                     \
            return (int)e_;                \
        }                                  \
    } while (0)

int rtk_build_mesh_bvh(const RtBvhBuildJob* job, void* streamPtr) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(streamPtr);
    const int n = job->triangleCnt, thr = job->threshold, tight = job->tightBounds;
    float4 *bmin = nullptr, *bmax = nullptr;
    uint32_t *orderA = nullptr, *orderB = nullptr, *cb = nullptr;
    uint64_t *keysA = nullptr, *keysB = nullptr;
    unsigned char* axisOf = nullptr;
    void* temp = nullptr;
    BuildNode* nodes = nullptr;
    auto cleanup = [&]() {
        void* all[] = {bmin, bmax, orderA, orderB, cb, keysA, keysB, axisOf, temp, nodes};
        for (void* p : all)
            if (p) (void)hipFree(p);
    };
    const size_t heap = (size_t)1 << (job->depth + 1);  // heap indices of all ranges, all levels
    RTB_HIP(hipMalloc(&bmin, (size_t)n * sizeof(float4)));
    RTB_HIP(hipMalloc(&bmax, (size_t)n * sizeof(float4)));
    RTB_HIP(hipMalloc(&orderA, (size_t)n * 4));
    RTB_HIP(hipMalloc(&orderB, (size_t)n * 4));
    RTB_HIP(hipMalloc(&keysA, (size_t)n * 8));
    RTB_HIP(hipMalloc(&keysB, (size_t)n * 8));
    RTB_HIP(hipMalloc(&cb, heap * 6 * 4));
    RTB_HIP(hipMalloc(&axisOf, heap));
    RTB_HIP(hipMalloc(&nodes, (size_t)job->nodeCnt * sizeof(BuildNode)));
    RTB_HIP(hipMemsetAsync(axisOf, 0, heap, stream));
    RTB_HIP(hipMemcpyAsync(nodes, job->nodes, (size_t)job->nodeCnt * sizeof(BuildNode), hipMemcpyHostToDevice, stream));
    RTB_HIP(hipMalloc(&temp, radixSortTempBytes(n)));

    const int grid = blocksFor(n);
    hipLaunchKernelGGL(primBoundsKernel, dim3(grid), dim3(kBlock), 0, stream, n, job->devTriangles, job->devVertices, tight, bmin, bmax);
    hipLaunchKernelGGL(iotaKernel, dim3(grid), dim3(kBlock), 0, stream, n, orderA);
    for (int level = 0; level < job->depth; ++level) {
        const uint32_t first = (1u << level) - 1u, count = 1u << level;
        hipLaunchKernelGGL(boundsInitKernel, dim3(blocksFor(count)), dim3(kBlock), 0, stream, first, count, tight, cb);
        hipLaunchKernelGGL(centroidBoundsKernel, dim3(grid), dim3(kBlock), 0, stream, n, thr, level, orderA, bmin, bmax, cb);
        hipLaunchKernelGGL(sortKeyKernel, dim3(grid), dim3(kBlock), 0, stream, n, thr, level, orderA, bmin, bmax, cb, axisOf, keysA);
        // range ids of this level need level + 1 bits above the 32-bit coordinate
        if (radixSortPairs(temp, keysA, keysB, orderA, orderB, n, 32 + level + 1, stream)) {
            std::swap(orderA, orderB);  // the next level reads the order from orderA (and writes its keys over keysA)
            std::swap(keysA, keysB);
        }
    }
    float* recs = reinterpret_cast<float*>(job->devNodeRecords);
    hipLaunchKernelGGL(leafBoxKernel, dim3(blocksFor(job->leafCnt)), dim3(kBlock), 0, stream, job->leafCnt, nodes, tight, orderA, bmin, bmax, recs);
    // the table holds the leaves first, then the interior nodes by level, deepest first
    int at = job->leafCnt;
    for (int level = job->depth - 1; level >= 0; --level) {
        const int cnt = job->interiorPerLevel[level];
        if (cnt > 0) hipLaunchKernelGGL(interiorBoxKernel, dim3(blocksFor(cnt)), dim3(kBlock), 0, stream, cnt, nodes + at, axisOf, recs);
        at += cnt;
    }
    hipLaunchKernelGGL(recordKernel, dim3(grid), dim3(kBlock), 0, stream, n, thr, orderA, job->devTriangles, job->devVertices,
                       reinterpret_cast<TriRecord*>(job->devTriRecords));
    RTB_HIP(hipStreamSynchronize(stream));
    RTB_HIP(hipGetLastError());
    cleanup();
    return 0;
}

// ---------------------------------------------------------------------------
// Surface-area cuts on the device (rt_set_device_bvh_rule, RT_BVH_BINNED_SAH): the tree of csrc/host/BVH.h's
// sahPartition - every range cut where the heuristic over 32 centroid bins per axis is lowest - decided level by level
// on the device.  The tree's shape depends on the data, so the HOST keeps it: per level it hands over the ranges that
// are still to be cut (ascending, disjoint) and gets back, for each, the axis and the position of the cut, or "no cut"
// when the centroids coincide along the widest axis.  Per level: range of every position (binary search), centroid
// bounds and bins with atomics (whole-block ranges are reduced in LDS first), one thread per range walks the 3 x 31
// candidate cuts in the host builder's order with its arithmetic (same cut, same ties), and one stable radix sort by
// (range, side) moves every range's left side to its front - std::stable_partition for all ranges at once.
namespace {

constexpr int kSahBins = 32;
constexpr int kBinWords = 7;  // count, min x y z, max x y z (ordered bits)

__global__ void __launch_bounds__(kBlock) sahRangeOfKernel(int n, int ranges, const int32_t* __restrict__ rangeStart, const int32_t* __restrict__ rangeEnd,
                                                           int32_t* __restrict__ rangeOf) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = ranges;  // the last range that starts at or before i
    while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (rangeStart[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    rangeOf[i] = (ranges > 0 && rangeStart[lo] <= i && i < rangeEnd[lo]) ? lo : -1;
}

__global__ void __launch_bounds__(kBlock) sahInitKernel(int ranges, uint32_t* __restrict__ cb, uint32_t* __restrict__ bins) {
    const size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lo = orderedBits(FLT_MAX), hi = orderedBits(-FLT_MAX);
    if (k < (size_t)ranges * 6) cb[k] = (k % 6) < 3 ? lo : hi;
    if (k < (size_t)ranges * 3 * kSahBins * kBinWords) {
        const int w = (int)(k % kBinWords);
        bins[k] = w == 0 ? 0u : (w < 4 ? lo : hi);
    }
}

__global__ void __launch_bounds__(kBlock) sahCentroidBoundsKernel(int n, const int32_t* __restrict__ rangeOf, const uint32_t* __restrict__ order,
                                                                  const float4* __restrict__ bmin, const float4* __restrict__ bmax, uint32_t* cb) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int r = i < n ? rangeOf[i] : -1;
    const bool active = r >= 0;
    uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    if (active) {
        const uint32_t prim = order[i];
        float c[3];
        centroidOf(bmin[prim], bmax[prim], c);
        for (int a = 0; a < 3; ++a) lo[a] = hi[a] = orderedBits(c[a]);
    }
    if (__ballot(active) == 0ull) return;
    // Positions of one range are consecutive, so inside a wavefront they form a run of lanes: a segmented scan leaves the
    // run's bounds in its last lane, and only that lane touches memory (one lane per wavefront near the root, one per
    // ten or twenty positions at the bottom of the tree, instead of every lane).
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const int rr = __shfl_up(r, off, 64);
        const bool same = lane >= off && rr == r;
        for (int a = 0; a < 3; ++a) {
            const uint32_t l = __shfl_up(lo[a], off, 64), h = __shfl_up(hi[a], off, 64);
            if (same) {
                lo[a] = l < lo[a] ? l : lo[a];
                hi[a] = h > hi[a] ? h : hi[a];
            }
        }
    }
    const int next = __shfl_down(r, 1, 64);
    if (active && (lane == 63 || next != r))
        for (int a = 0; a < 3; ++a) {
            atomicMin(cb + 6 * (size_t)r + a, lo[a]);
            atomicMax(cb + 6 * (size_t)r + 3 + a, hi[a]);
        }
}

// csrc/host/BVH.h binOf
__device__ __forceinline__ int sahBinOf(float c, float lo, float extent) {
    const int b = (int)((c - lo) / extent * (float)kSahBins);
    return b < 0 ? 0 : (b >= kSahBins ? kSahBins - 1 : b);
}

__global__ void __launch_bounds__(kBlock) sahBinKernel(int n, const int32_t* __restrict__ rangeOf, const uint32_t* __restrict__ order,
                                                       const float4* __restrict__ bmin, const float4* __restrict__ bmax, const uint32_t* __restrict__ cb,
                                                       uint32_t* bins) {
    __shared__ uint32_t sh[3 * kSahBins * kBinWords];
    __shared__ int shRange, shMixed;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int r = i < n ? rangeOf[i] : -1;
    if (threadIdx.x == 0) {
        shRange = -2;
        shMixed = 0;
    }
    __syncthreads();
    if (r >= 0) {
        const int seen = atomicCAS(&shRange, -2, r);
        if (seen != -2 && seen != r) shMixed = 1;
    }
    __syncthreads();
    if (shRange == -2) return;  // nothing to cut in this block
    const bool local = shMixed == 0;  // every active position of the block lies in one range: bins in LDS first
    if (local) {
        const uint32_t lo = orderedBits(FLT_MAX), hi = orderedBits(-FLT_MAX);
        for (int k = threadIdx.x; k < 3 * kSahBins * kBinWords; k += kBlock) {
            const int w = k % kBinWords;
            sh[k] = w == 0 ? 0u : (w < 4 ? lo : hi);
        }
        __syncthreads();
    }
    if (r >= 0) {
        const uint32_t prim = order[i];
        const float4 a = bmin[prim], b = bmax[prim];
        float c[3];
        centroidOf(a, b, c);
        const uint32_t box[6] = {orderedBits(a.x), orderedBits(a.y), orderedBits(a.z), orderedBits(b.x), orderedBits(b.y), orderedBits(b.z)};
        const uint32_t* bounds = cb + 6 * (size_t)r;
        for (int dim = 0; dim < 3; ++dim) {
            const float lo = fromOrderedBits(bounds[dim]), extent = fromOrderedBits(bounds[3 + dim]) - lo;
            if (!(extent > 0.0f)) continue;
            const int bin = sahBinOf(c[dim], lo, extent);
            uint32_t* at = (local ? sh : bins + (size_t)r * 3 * kSahBins * kBinWords) + (dim * kSahBins + bin) * kBinWords;
            atomicAdd(at, 1u);
            for (int k = 0; k < 3; ++k) {
                atomicMin(at + 1 + k, box[k]);
                atomicMax(at + 4 + k, box[3 + k]);
            }
        }
    }
    if (local) {
        __syncthreads();
        uint32_t* out = bins + (size_t)shRange * 3 * kSahBins * kBinWords;
        for (int k = threadIdx.x; k < 3 * kSahBins * kBinWords; k += kBlock) {
            const int w = k % kBinWords;
            const uint32_t v = sh[k];
            if (sh[k - w] == 0u) continue;  // empty bin
            if (w == 0)
                atomicAdd(out + k, v);
            else if (w < 4)
                atomicMin(out + k, v);
            else
                atomicMax(out + k, v);
        }
    }
}

struct SahBox {
    float lo[3], hi[3];
};
__device__ __forceinline__ SahBox sahEmpty() { return SahBox{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}}; }
__device__ __forceinline__ void sahUnion(SahBox& acc, const uint32_t* bin) {
    for (int k = 0; k < 3; ++k) {
        const float l = fromOrderedBits(bin[1 + k]), h = fromOrderedBits(bin[4 + k]);
        acc.lo[k] = l < acc.lo[k] ? l : acc.lo[k];
        acc.hi[k] = h > acc.hi[k] ? h : acc.hi[k];
    }
}
// csrc/host/BVH.h halfArea over BoundingBox::diagonal
__device__ __forceinline__ float sahHalfArea(const SahBox& b) {
    const float d0 = b.hi[0] - b.lo[0], d1 = b.hi[1] - b.lo[1], d2 = b.hi[2] - b.lo[2];
    return d0 * d1 + d1 * d2 + d2 * d0;
}

// result[2 r] = axis | bin << 8, or -1: the centroids coincide along the widest axis (a leaf whatever its size);
// result[2 r + 1] = position of the cut.  csrc/host/BVH.h recursive + sahPartition, candidate for candidate.
__global__ void __launch_bounds__(kBlock) sahEvalKernel(int ranges, int threshold, const int32_t* __restrict__ rangeStart, const int32_t* __restrict__ rangeEnd,
                                                        const uint32_t* __restrict__ cb, const uint32_t* __restrict__ bins, int32_t* __restrict__ result) {
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= ranges) return;
    const int start = rangeStart[r], cnt = rangeEnd[r] - start;
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = fromOrderedBits(cb[6 * (size_t)r + a]);
        hi[a] = fromOrderedBits(cb[6 * (size_t)r + 3 + a]);
    }
    const float d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
    const int widest = (d0 > d1 && d0 > d2) ? 0 : (d1 > d2 ? 1 : 2);  // BoundingBox::maxExtent
    if (hi[widest] == lo[widest]) {
        result[2 * r] = -1;
        result[2 * r + 1] = start;
        return;
    }
    float bestCost = FLT_MAX;
    int bestDim = -1, bestBin = -1, bestLeft = 0;
    const bool wantTwoLeaves = cnt <= 2 * threshold;
    bool haveTwoLeaves = false;
    for (int dim = 0; dim < 3; ++dim) {
        if (!(hi[dim] - lo[dim] > 0.0f)) continue;
        const uint32_t* row = bins + ((size_t)r * 3 + dim) * kSahBins * kBinWords;
        float rightArea[kSahBins];
        int rightCount[kSahBins];
        SahBox acc = sahEmpty();
        int k = 0;
        for (int b = kSahBins - 1; b > 0; --b) {
            const uint32_t* bin = row + b * kBinWords;
            if (bin[0]) sahUnion(acc, bin);
            k += (int)bin[0];
            rightArea[b] = k ? sahHalfArea(acc) : 0.0f;
            rightCount[b] = k;
        }
        acc = sahEmpty();
        k = 0;
        for (int b = 1; b < kSahBins; ++b) {
            const uint32_t* bin = row + (b - 1) * kBinWords;
            if (bin[0]) sahUnion(acc, bin);
            k += (int)bin[0];
            if (k == 0 || rightCount[b] == 0) continue;
            const bool twoLeaves = k <= threshold && rightCount[b] <= threshold;
            if (wantTwoLeaves && !twoLeaves && haveTwoLeaves) continue;
            const float cost = wantTwoLeaves && twoLeaves ? sahHalfArea(acc) + rightArea[b] : sahHalfArea(acc) * (float)k + rightArea[b] * (float)rightCount[b];
            if (cost < bestCost || (wantTwoLeaves && twoLeaves && !haveTwoLeaves)) {
                bestCost = cost;
                bestDim = dim;
                bestBin = b;
                bestLeft = k;
                haveTwoLeaves = wantTwoLeaves && twoLeaves;
            }
        }
    }
    // (bestDim >= 0: along the widest axis bins 0 and 31 are both occupied)
    result[2 * r] = bestDim < 0 ? -1 : (bestDim | (bestBin << 8));
    result[2 * r + 1] = start + bestLeft;
}

// Sort keys: 2 * (start of the range) + side for positions of ranges that are cut, 2 * position for all others - ranges
// stay where they are, and inside a range the left side moves to the front in its old order.
__global__ void __launch_bounds__(kBlock) sahKeyKernel(int n, const int32_t* __restrict__ rangeOf, const int32_t* __restrict__ rangeStart,
                                                       const uint32_t* __restrict__ order, const float4* __restrict__ bmin, const float4* __restrict__ bmax,
                                                       const uint32_t* __restrict__ cb, const int32_t* __restrict__ result, uint64_t* __restrict__ keys) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int r = rangeOf[i];
    const int code = r >= 0 ? result[2 * r] : -1;
    if (code < 0) {
        keys[i] = 2ull * (uint64_t)i;
        return;
    }
    const int dim = code & 3, bin = code >> 8;
    const uint32_t prim = order[i];
    float c[3];
    centroidOf(bmin[prim], bmax[prim], c);
    const float lo = fromOrderedBits(cb[6 * (size_t)r + dim]), extent = fromOrderedBits(cb[6 * (size_t)r + 3 + dim]) - lo;
    keys[i] = 2ull * (uint64_t)rangeStart[r] + (sahBinOf(c[dim], lo, extent) < bin ? 0ull : 1ull);
}

}  // namespace

struct RtSahBuild {
    int n = 0, maxRanges = 0;
    float4 *bmin = nullptr, *bmax = nullptr;
    uint32_t *orderA = nullptr, *orderB = nullptr, *cb = nullptr, *bins = nullptr;
    uint64_t *keysA = nullptr, *keysB = nullptr;
    int32_t *rangeOf = nullptr, *rangeStart = nullptr, *rangeEnd = nullptr, *result = nullptr;
    void* temp = nullptr;
};

void rtk_sah_end(RtSahBuild* b) {
    if (!b) return;
    void* all[] = {b->bmin, b->bmax, b->orderA, b->orderB, b->cb, b->bins, b->keysA, b->keysB, b->rangeOf, b->rangeStart, b->rangeEnd, b->result, b->temp};
    for (void* p : all)
        if (p) (void)hipFree(p);
    delete b;
}

int rtk_sah_begin(int n, int threshold, const float* hostBmin, const float* hostBmax, RtSahBuild** out, void* streamPtr) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(streamPtr);
    RtSahBuild* b = new RtSahBuild();
    b->n = n;
    b->maxRanges = n / (threshold + 1) + 1;  // a range that is cut holds more than `threshold` primitives
    auto cleanup = [&]() { rtk_sah_end(b); };
    RTB_HIP(hipMalloc(&b->bmin, (size_t)n * sizeof(float4)));
    RTB_HIP(hipMalloc(&b->bmax, (size_t)n * sizeof(float4)));
    RTB_HIP(hipMalloc(&b->orderA, (size_t)n * 4));
    RTB_HIP(hipMalloc(&b->orderB, (size_t)n * 4));
    RTB_HIP(hipMalloc(&b->keysA, (size_t)n * 8));
    RTB_HIP(hipMalloc(&b->keysB, (size_t)n * 8));
    RTB_HIP(hipMalloc(&b->rangeOf, (size_t)n * 4));
    RTB_HIP(hipMalloc(&b->rangeStart, (size_t)b->maxRanges * 4));
    RTB_HIP(hipMalloc(&b->rangeEnd, (size_t)b->maxRanges * 4));
    RTB_HIP(hipMalloc(&b->result, (size_t)b->maxRanges * 8));
    RTB_HIP(hipMalloc(&b->cb, (size_t)b->maxRanges * 6 * 4));
    RTB_HIP(hipMalloc(&b->bins, (size_t)b->maxRanges * 3 * kSahBins * kBinWords * 4));
    RTB_HIP(hipMalloc(&b->temp, radixSortTempBytes(n)));
    RTB_HIP(hipMemcpyAsync(b->bmin, hostBmin, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, stream));
    RTB_HIP(hipMemcpyAsync(b->bmax, hostBmax, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(iotaKernel, dim3(blocksFor(n)), dim3(kBlock), 0, stream, n, b->orderA);
    RTB_HIP(hipStreamSynchronize(stream));
    *out = b;
    return 0;
}

// One level: `ranges` ranges [hostStart[r], hostEnd[r]) in ascending order, each with more than `threshold` primitives.
// hostResult[2 r] = axis | bin << 8 or -1 (no cut: all centroids coincide along the widest axis), hostResult[2 r + 1] =
// position of the cut; the primitives of every range that is cut are in their new order when the call returns.
int rtk_sah_level(RtSahBuild* b, int ranges, int threshold, const int32_t* hostStart, const int32_t* hostEnd, int32_t* hostResult, void* streamPtr) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(streamPtr);
    if (ranges < 1) return 0;
    if (ranges > b->maxRanges) return (int)hipErrorInvalidValue;
    auto cleanup = [] {};
    const int n = b->n, grid = blocksFor(n);
    RTB_HIP(hipMemcpyAsync(b->rangeStart, hostStart, (size_t)ranges * 4, hipMemcpyHostToDevice, stream));
    RTB_HIP(hipMemcpyAsync(b->rangeEnd, hostEnd, (size_t)ranges * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(sahRangeOfKernel, dim3(grid), dim3(kBlock), 0, stream, n, ranges, b->rangeStart, b->rangeEnd, b->rangeOf);
    hipLaunchKernelGGL(sahInitKernel, dim3(blocksFor((long long)ranges * 3 * kSahBins * kBinWords)), dim3(kBlock), 0, stream, ranges, b->cb, b->bins);
    hipLaunchKernelGGL(sahCentroidBoundsKernel, dim3(grid), dim3(kBlock), 0, stream, n, b->rangeOf, b->orderA, b->bmin, b->bmax, b->cb);
    hipLaunchKernelGGL(sahBinKernel, dim3(grid), dim3(kBlock), 0, stream, n, b->rangeOf, b->orderA, b->bmin, b->bmax, b->cb, b->bins);
    hipLaunchKernelGGL(sahEvalKernel, dim3(blocksFor(ranges)), dim3(kBlock), 0, stream, ranges, threshold, b->rangeStart, b->rangeEnd, b->cb, b->bins, b->result);
    hipLaunchKernelGGL(sahKeyKernel, dim3(grid), dim3(kBlock), 0, stream, n, b->rangeOf, b->rangeStart, b->orderA, b->bmin, b->bmax, b->cb, b->result, b->keysA);
    int bits = 1;
    while ((2ull * (uint64_t)n) >> bits) ++bits;
    if (radixSortPairs(b->temp, b->keysA, b->keysB, b->orderA, b->orderB, n, bits, stream)) {
        std::swap(b->orderA, b->orderB);
        std::swap(b->keysA, b->keysB);
    }
    RTB_HIP(hipMemcpyAsync(hostResult, b->result, (size_t)ranges * 8, hipMemcpyDeviceToHost, stream));
    RTB_HIP(hipStreamSynchronize(stream));
    RTB_HIP(hipGetLastError());
    return 0;
}

// The primitive at every position, after the last level.
int rtk_sah_order(RtSahBuild* b, uint32_t* hostOrder, void* streamPtr) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(streamPtr);
    auto cleanup = [] {};
    RTB_HIP(hipMemcpyAsync(hostOrder, b->orderA, (size_t)b->n * 4, hipMemcpyDeviceToHost, stream));
    RTB_HIP(hipStreamSynchronize(stream));
    return 0;
}

// ---------------------------------------------------------------------------
// Bounds refit of a tree that is already in the blob (rt_set_upload_refit).  The reference's builder starts every box
// union from an "empty" box whose max is FLT_MIN (cudaTracer/BoundingBox.cuh:25-27, cudaTracer/BVH.h:90-96), so every
// box over negative coordinates reaches up to 0.  Topology, split axes and leaf contents stay as the caller built them;
// only the boxes change: a mesh leaf gets the exact bounds of its triangles' vertex positions, an interior node the
// exact union of its two children.  One thread per 32-byte record of the tree's pair array.
namespace {
// Written as comparisons, like the host side (rt_capi.cpp refitNodes), so that both keep the value seen first when
// +0 meets -0 and the two refits agree bit for bit (fmin / fmax may pick either zero).
__device__ __forceinline__ float refitMin(float acc, float x) { return x < acc ? x : acc; }
__device__ __forceinline__ float refitMax(float acc, float x) { return x > acc ? x : acc; }

__global__ void __launch_bounds__(kBlock) refitLeafKernel(uint32_t* __restrict__ pairs, uint32_t recCnt, const TriRecord* __restrict__ records,
                                                          const unsigned char* __restrict__ vertices) {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= recCnt || r == 1u) return;  // record 1 is the padding beside the root
    uint32_t* rec = pairs + (size_t)r * 8;
    const uint32_t ref = rec[6];
    if (!RT_REF_IS_LEAF(ref)) return;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const uint32_t first = RT_REF_LEAF_FIRST(ref), cnt = RT_REF_LEAF_COUNT(ref);
    for (uint32_t t = first; t < first + cnt; ++t)
        for (int c = 0; c < 3; ++c) {
            const float* pos = reinterpret_cast<const float*>(vertices + (size_t)records[t].idx[c] * sizeof(RtVertex) + 44);
            for (int a = 0; a < 3; ++a) {
                lo[a] = refitMin(lo[a], pos[a]);
                hi[a] = refitMax(hi[a], pos[a]);
            }
        }
    for (int a = 0; a < 3; ++a) {
        rec[a] = __float_as_uint(lo[a]);
        rec[3 + a] = __float_as_uint(hi[a]);
    }
}

// The same for a BVH over shapes: a leaf gets the exact union of its shapes' own boxes (six floats per shape, in leaf
// order), without the FLT_MIN the reference's builder unions into every leaf.
__global__ void __launch_bounds__(kBlock) refitShapeLeafKernel(uint32_t* __restrict__ pairs, uint32_t recCnt, const float* __restrict__ shapeBoxes) {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= recCnt || r == 1u) return;
    uint32_t* rec = pairs + (size_t)r * 8;
    const uint32_t ref = rec[6];
    if (!RT_REF_IS_LEAF(ref)) return;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const uint32_t first = RT_REF_LEAF_FIRST(ref), cnt = RT_REF_LEAF_COUNT(ref);
    for (uint32_t k = first; k < first + cnt; ++k)
        for (int a = 0; a < 3; ++a) {
            lo[a] = refitMin(lo[a], shapeBoxes[6 * (size_t)k + a]);
            hi[a] = refitMax(hi[a], shapeBoxes[6 * (size_t)k + 3 + a]);
        }
    for (int a = 0; a < 3; ++a) {
        rec[a] = __float_as_uint(lo[a]);
        rec[3 + a] = __float_as_uint(hi[a]);
    }
}

// One pass: every interior record takes the union of its children's CURRENT boxes.  After pass k every node of height
// <= k is final (its children were final before the pass, and a child rewritten during the pass gets the same bits),
// so `depth` passes settle the whole tree without level tables.
__global__ void __launch_bounds__(kBlock) refitInteriorKernel(uint32_t* __restrict__ pairs, uint32_t pairBase, uint32_t recCnt) {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= recCnt || r == 1u) return;
    uint32_t* rec = pairs + (size_t)r * 8;
    const uint32_t ref = rec[6];
    if (RT_REF_IS_LEAF(ref)) return;
    const uint32_t pair = RT_REF_PAIR_UNITS(ref) - pairBase;
    if (2u * pair + 1u >= recCnt) return;  // a reference that leaves the tree (the packer lets none through)
    const float* c = reinterpret_cast<const float*>(pairs + (size_t)pair * 16);
    for (int a = 0; a < 3; ++a) {
        rec[a] = __float_as_uint(refitMin(c[a], c[8 + a]));
        rec[3 + a] = __float_as_uint(refitMax(c[3 + a], c[8 + 3 + a]));
    }
}
}  // namespace

int rtk_refit_tree(void* pairs, uint32_t pairBase, uint32_t pairCnt, int depth, const void* records, const void* vertices, const float* shapeBoxes,
                   void* stream) {
    if (pairCnt == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const uint32_t recCnt = 2u * pairCnt;
    uint32_t* p = reinterpret_cast<uint32_t*>(pairs);
    if (records && vertices)
        hipLaunchKernelGGL(refitLeafKernel, dim3(blocksFor(recCnt)), dim3(kBlock), 0, s, p, recCnt, reinterpret_cast<const TriRecord*>(records),
                           reinterpret_cast<const unsigned char*>(vertices));
    else if (shapeBoxes)
        hipLaunchKernelGGL(refitShapeLeafKernel, dim3(blocksFor(recCnt)), dim3(kBlock), 0, s, p, recCnt, shapeBoxes);
    for (int pass = 0; pass < depth; ++pass) hipLaunchKernelGGL(refitInteriorKernel, dim3(blocksFor(recCnt)), dim3(kBlock), 0, s, p, pairBase, recCnt);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// Quad records (rt_device_format.h): one thread per half of a quad copies the child pair of one record of the pair,
// or the record itself when it is a leaf, and turns the interior references it copies into references to quads.
namespace {
__global__ void quadKernel(const uint32_t* __restrict__ pairs, uint32_t pairBase, uint32_t pairCnt, uint32_t* __restrict__ quads, uint32_t quadBase) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2u * pairCnt) return;
    const uint32_t i = t >> 1, h = t & 1u;
    const uint32_t* rec = pairs + (size_t)i * 16 + h * 8;
    const uint32_t ref = rec[6];
    // records that are not there: a box of NaNs passes no box test, whatever the ray
    uint32_t out[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) out[k] = (k & 7) < 6 ? 0x7fc00000u : 0u;
    uint32_t axisWord = 0u;  // one-hot split axis of the record whose children the half holds; 0: fewer than two records
    if (i == 0u && h == 1u) {
        // the padding beside the tree's root record: an empty half
    } else if (RT_REF_IS_LEAF(ref)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = rec[k];
    } else if (RT_REF_PAIR_UNITS(ref) - pairBase >= pairCnt) {
        // a reference that leaves the tree (the packer lets none through): nothing to visit
    } else {
        const uint32_t* child = pairs + (size_t)(RT_REF_PAIR_UNITS(ref) - pairBase) * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) out[k] = child[k];
#pragma unroll
        for (int k = 6; k < 16; k += 8) {
            const uint32_t r = out[k];
            if (!RT_REF_IS_LEAF(r)) out[k] = (r & 0x60000000u) | (RT_REF_PAIR_UNITS(r) - pairBase + quadBase);
        }
        axisWord = 1u << RT_REF_AXIS(ref);
    }
    out[7] = axisWord;
    uint4* dst = reinterpret_cast<uint4*>(quads + (size_t)(quadBase + i) * 32 + h * 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
}
}  // namespace

int rtk_build_quads(const void* pairs, uint32_t pairBase, uint32_t pairCnt, void* quads, uint32_t quadBase, void* stream) {
    if (pairCnt == 0) return 0;
    const uint32_t threads = 2u * pairCnt;
    hipLaunchKernelGGL(quadKernel, dim3((threads + 255u) / 256u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint32_t*>(pairs), pairBase, pairCnt, reinterpret_cast<uint32_t*>(quads), quadBase);
    return (int)hipGetLastError();
}
