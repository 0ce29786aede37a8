// Host BVH builder of the API: BVH::BvhFactory<T> over triangles or shapes,
// producing the flattened LinearNode array and the primitives in leaf order.
// Same construction rule as the reference (cudaTracer/BVH.h:63-181):
//   - a range becomes a leaf when it holds <= threshold primitives or when the
//     centroids do not spread along the chosen axis;
//   - the axis is the widest extent of the centroid bounds (empty-box start and
//     tie rules of BoundingBox apply);
//   - the range is cut in two equal halves with std::nth_element on the box
//     centroid, so the tree is balanced and its depth is ceil(log2(n/threshold));
//   - nodes are laid out depth first: first child at self + 1, second child by
//     index.
// The reference builds both children as arguments of one call, which leaves the
// order to the compiler (BVH.h:146-149); MSVC and g++ both evaluate right to
// left, so the second half is numbered first here, explicitly.
//
// Threads.  The two halves of a range share nothing: each orders its own part of primitiveInfo, and where its leaves
// go in orderedPrims follows from the sizes of the ranges alone (the second half's primitives come first).  So large
// ranges hand their second half to another thread (at most 16, or RT_BVH_THREADS), and the arrays that come out are
// the same for any number of threads.
//
// Bounds mode.  REFERENCE_BOUNDS reproduces the reference builder bit for bit,
// including its empty box whose max corner starts at FLT_MIN: every box is then
// stretched up to ~0 on each axis, so geometry at negative coordinates gets
// heavily overlapping nodes (correct images, but hundreds of extra node and
// triangle tests per ray).  TIGHT_BOUNDS starts every union from a true empty
// box (+FLT_MAX / -FLT_MAX) and is otherwise the same algorithm; hits are the
// same except for equal-distance ties, which depend on visiting order.
// What the C++ driver (RayTracer.h) gets by default is a third thing: trees BUILT with REFERENCE_BOUNDS - the
// reference's shape, leaf order and ties, node for node - whose boxes the library recomputes on the device when the
// scene is uploaded (RayTracer::refitAtUpload, rt_set_upload_refit), so that rays walk true bounds.
#ifndef RT_HOST_BVH_H
#define RT_HOST_BVH_H

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <system_error>
#include <thread>
#include <vector>

#include "BoundingBox.h"
#include "Scene.h"

namespace rt {
namespace BVH {

enum BoundsMode { REFERENCE_BOUNDS = 0, TIGHT_BOUNDS = 1 };

// Where a range is cut.  MEDIAN_CUT is the reference's rule (above).  BINNED_SAH is not in the reference: it cuts where
// the surface-area heuristic, evaluated on 32 bins of the box centroids along each of the three axes, is lowest
// (cost = area(left) * count(left) + area(right) * count(right)), and falls back to the median cut when no bin
// boundary separates the range.  The tree is no longer balanced, rays visit fewer nodes, and hits are the same
// except for equal-distance ties, which depend on visiting order.
enum SplitRule { MEDIAN_CUT = 0, BINNED_SAH = 1 };

// How a mesh gets its BVH: built here on the host, or left to the device (rt_scene_upload builds the BVH of a
// mesh that arrives without nodes; same algorithm, see include/rt_capi.h rt_set_device_bvh).
struct BuildOptions {
    BoundsMode bounds = REFERENCE_BOUNDS;
    bool onDevice = false;
    SplitRule split = MEDIAN_CUT;
    int leafThreshold = 10;  // cudaTracer/CudaMesh.h:58
    BuildOptions() = default;
    BuildOptions(BoundsMode b, bool device = false, SplitRule rule = MEDIAN_CUT, int threshold = 10)
        : bounds(b), onDevice(device), split(rule), leafThreshold(threshold) {}
};

// Depth of the deepest leaf of the median-cut tree over n primitives (the tree's shape depends on n alone).
inline int medianTreeDepth(int n, int threshold) {
    int depth = 0;
    // the larger half decides: it holds n - n / 2 primitives
    while (n > threshold) {
        n = n - n / 2;
        ++depth;
    }
    return depth;
}

inline BoundingBox emptyBox(BoundsMode mode) {
    return mode == TIGHT_BOUNDS ? BoundingBox(Vec3f(FLT_MAX, FLT_MAX, FLT_MAX), Vec3f(-FLT_MAX, -FLT_MAX, -FLT_MAX)) : BoundingBox();
}
inline BoundingBox triangleBox(const Triangle& t, const std::vector<Vertex>& v, BoundsMode mode) {
    if (mode == REFERENCE_BOUNDS) return BoundingBox(t, v);
    BoundingBox b = emptyBox(mode);
    for (int corner = 0; corner < 3; ++corner) b = BoundingBox::unionn(b, v[t[corner]].position);
    return b;
}

struct PreNode {
    int primIdx;
    BoundingBox bb;
    Vec3f centroid;
    PreNode() : primIdx(-1), bb(Vec3f(-10.0f, -20.0f, -30.0f), Vec3f(-1.0f, -2.0f, -3.0f)), centroid(-1.0f, -2.0f, -3.0f) {}
    PreNode(int i, BoundingBox bb) : primIdx(i), bb(bb), centroid(bb.centroid()) {}
};

struct InterNode {
    BoundingBox bb;
    InterNode* children[2] = {nullptr, nullptr};
    int splitAxis = -1;
    int firstPrimOffset = -1;
    int primitiveCnt = -1;
    int subtreeNodes = 1;  // this node and everything below it
    int subtreeDepth = 0;  // levels below this node
    enum NodeTypes { NOT_SET = -1, LEAF, INTERIOR };
    NodeTypes nodeType = NOT_SET;

    void leaf(int firstPrim, int primCnt, BoundingBox box) {
        firstPrimOffset = firstPrim;
        primitiveCnt = primCnt;
        bb = box;
        nodeType = LEAF;
    }
    void interior(int axis, int primCnt, InterNode* c0, InterNode* c1) {
        children[0] = c0;
        children[1] = c1;
        bb = BoundingBox::unionn(c0->bb, c1->bb);
        splitAxis = axis;
        primitiveCnt = primCnt;
        subtreeNodes = 1 + c0->subtreeNodes + c1->subtreeNodes;
        subtreeDepth = 1 + (std::max)(c0->subtreeDepth, c1->subtreeDepth);
        nodeType = INTERIOR;
    }
};

template <typename T>
class BvhFactory {
   public:
    std::vector<T> orderedPrims;
    std::vector<int> orderedIdx;  // index into `primitives` of each ordered slot
    std::vector<T> primitives;
    std::vector<PreNode> primitiveInfo;
    std::vector<LinearNode> nodes;
    int threshold;
    int depth = 0;  // levels below the root
    BoundsMode boundsMode;
    SplitRule splitRule = MEDIAN_CUT;

    BvhFactory(const std::vector<Vertex>& vertices, const std::vector<Triangle>& prims, int threshold,
               BoundsMode mode = REFERENCE_BOUNDS, SplitRule rule = MEDIAN_CUT)
        : primitives(prims), threshold(threshold), boundsMode(mode), splitRule(rule) {
        primitiveInfo.resize(prims.size());
        for (int i = 0; i < (int)prims.size(); ++i) primitiveInfo[i] = PreNode(i, triangleBox(prims[i], vertices, mode));
        build();
    }
    BvhFactory(const std::vector<Shape>& prims, BoundsMode mode = REFERENCE_BOUNDS)
        : primitives(prims), threshold(1), boundsMode(mode) {
        primitiveInfo.resize(prims.size());
        for (int i = 0; i < (int)prims.size(); ++i) primitiveInfo[i] = PreNode(i, prims[i].bb);
        build();
    }

   private:
    // Nodes of the tree under construction.  Every thread allocates from an arena of its own.
    typedef std::deque<InterNode> Arena;
    std::vector<std::unique_ptr<Arena>> arenas;
    std::mutex arenasMutex;
    std::atomic<int> spareThreads{0};
    std::exception_ptr failure;  // first exception of a helper thread (guarded by arenasMutex)
    static constexpr int kThreadGrain = 1 << 15;  // smallest range worth a thread of its own

    Arena& newArena() {
        std::lock_guard<std::mutex> lock(arenasMutex);
        arenas.emplace_back(new Arena());
        return *arenas.back();
    }
    static int buildThreads() {
        if (const char* e = std::getenv("RT_BVH_THREADS")) {
            const int n = std::atoi(e);
            if (n >= 1) return (std::min)(n, 64);
        }
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)(std::min)(hw ? hw : 1u, 16u);
    }

    void build() {
        if (primitives.empty()) throw std::invalid_argument("BvhFactory: no primitives");
        orderedPrims.resize(primitives.size());
        orderedIdx.resize(primitives.size());
        spareThreads = buildThreads() - 1;
        InterNode* root = recursive(0, (int)primitives.size(), 0, newArena());
        if (failure) std::rethrow_exception(failure);
        depth = root->subtreeDepth;
        nodes = std::vector<LinearNode>(root->subtreeNodes);
        int offset = 0;
        flattenBvhTree(root, &offset);
        arenas.clear();
    }

    // Builds both halves of [start, end) cut at mid: the second half first (see the head of this file), on another
    // thread when the range is large and a thread is to spare.  The second half's leaves take the front of the
    // range's slots in orderedPrims.
    void halves(const int start, const int mid, const int end, const int orderedBase, Arena& arena, InterNode*& first, InterNode*& second) {
        const int firstBase = orderedBase + (end - mid);
        const bool claimed = end - start >= kThreadGrain && spareThreads.fetch_sub(1) > 0;
        std::thread helper;
        if (claimed) {
            try {
                helper = std::thread([&]() {
                    try {
                        second = recursive(mid, end, orderedBase, newArena());
                    } catch (...) {
                        std::lock_guard<std::mutex> lock(arenasMutex);
                        if (!failure) failure = std::current_exception();
                        second = nullptr;
                    }
                });
            } catch (const std::system_error&) {
                // no thread to be had (a process limit): this range is built by the calling thread
            }
        }
        if (helper.joinable()) {
            try {
                first = recursive(start, mid, firstBase, arena);
            } catch (...) {
                helper.join();
                spareThreads.fetch_add(1);
                throw;
            }
            helper.join();
            spareThreads.fetch_add(1);
            if (!second) std::rethrow_exception(failure);
            return;
        }
        if (end - start >= kThreadGrain) spareThreads.fetch_add(1);  // none to spare or none started: give the claim back
        second = recursive(mid, end, orderedBase, arena);
        first = recursive(start, mid, firstBase, arena);
    }

    InterNode* recursive(const int start, const int end, const int orderedBase, Arena& arena) {
        arena.emplace_back();
        InterNode* node = &arena.back();
        BoundingBox bb = emptyBox(boundsMode);
        for (int i = start; i < end; ++i) bb = BoundingBox::unionn(bb, primitiveInfo[i].bb);

        const int primCnt = end - start;
        if (primCnt <= threshold) {
            leaf(bb, primCnt, start, end, orderedBase, node);
            return node;
        }
        BoundingBox centroidBounds = emptyBox(boundsMode);
        for (int i = start; i < end; ++i)
            centroidBounds = BoundingBox::unionn(centroidBounds, primitiveInfo[i].bb.centroid());
        const int dim = centroidBounds.maxExtent();
        if (centroidBounds.bbmax[dim] == centroidBounds.bbmin[dim]) {
            leaf(bb, primCnt, start, end, orderedBase, node);
            return node;
        }
        InterNode *first = nullptr, *second = nullptr;
        if (splitRule == BINNED_SAH) {
            int sahDim = dim;
            const int mid = sahPartition(start, end, centroidBounds, &sahDim);
            if (mid > start && mid < end) {
                halves(start, mid, end, orderedBase, arena, first, second);
                node->interior(sahDim, primCnt, first, second);
                return node;
            }
        }
        const int mid = (start + end) / 2;
        std::nth_element(primitiveInfo.begin() + start, primitiveInfo.begin() + mid, primitiveInfo.begin() + end,
                         [dim](const PreNode& a, const PreNode& b) { return a.bb.centroid()[dim] < b.bb.centroid()[dim]; });
        halves(start, mid, end, orderedBase, arena, first, second);
        node->interior(dim, primCnt, first, second);
        return node;
    }

    static float halfArea(const BoundingBox& b) {
        const Vec3f d = b.diagonal();
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    }

    // Orders [start, end) so that the primitives left of the cheapest bin boundary come first; returns the position
    // of the cut (start or end: no boundary separates the range) and the axis it is on.
    int sahPartition(const int start, const int end, const BoundingBox& centroidBounds, int* dimOut) {
        constexpr int kBins = 32;
        float bestCost = FLT_MAX;
        int bestDim = -1, bestBin = -1;
        const bool wantTwoLeaves = end - start <= 2 * threshold;
        bool haveTwoLeaves = false;
        for (int dim = 0; dim < 3; ++dim) {
            const float lo = centroidBounds.bbmin[dim], extent = centroidBounds.bbmax[dim] - lo;
            if (!(extent > 0.0f)) continue;
            int count[kBins] = {0};
            BoundingBox box[kBins];
            for (int b = 0; b < kBins; ++b) box[b] = emptyBox(TIGHT_BOUNDS);
            for (int i = start; i < end; ++i) {
                const int b = binOf(primitiveInfo[i].centroid[dim], lo, extent, kBins);
                ++count[b];
                box[b] = BoundingBox::unionn(box[b], primitiveInfo[i].bb);
            }
            float rightArea[kBins];
            int rightCount[kBins];
            BoundingBox acc = emptyBox(TIGHT_BOUNDS);
            int n = 0;
            for (int b = kBins - 1; b > 0; --b) {
                if (count[b]) acc = BoundingBox::unionn(acc, box[b]);
                n += count[b];
                rightArea[b] = n ? halfArea(acc) : 0.0f;
                rightCount[b] = n;
            }
            acc = emptyBox(TIGHT_BOUNDS);
            n = 0;
            for (int b = 1; b < kBins; ++b) {  // cut between bins b - 1 and b
                if (count[b - 1]) acc = BoundingBox::unionn(acc, box[b - 1]);
                n += count[b - 1];
                if (n == 0 || rightCount[b] == 0) continue;
                // A leaf costs the same whatever it holds (its triangles are tested side by side), so a range that
                // can become two leaves does so by the cut with the smallest surface, not the smallest area x count.
                const bool twoLeaves = n <= threshold && rightCount[b] <= threshold;
                if (wantTwoLeaves && !twoLeaves && haveTwoLeaves) continue;
                const float cost = wantTwoLeaves && twoLeaves ? halfArea(acc) + rightArea[b] : halfArea(acc) * (float)n + rightArea[b] * (float)rightCount[b];
                if (cost < bestCost || (wantTwoLeaves && twoLeaves && !haveTwoLeaves)) {
                    bestCost = cost;
                    bestDim = dim;
                    bestBin = b;
                    haveTwoLeaves = wantTwoLeaves && twoLeaves;
                }
            }
        }
        if (bestDim < 0) return start;
        const float lo = centroidBounds.bbmin[bestDim], extent = centroidBounds.bbmax[bestDim] - lo;
        auto midIt = std::stable_partition(primitiveInfo.begin() + start, primitiveInfo.begin() + end, [&](const PreNode& p) {
            return binOf(p.centroid[bestDim], lo, extent, kBins) < bestBin;
        });
        *dimOut = bestDim;
        return (int)(midIt - primitiveInfo.begin());
    }
    static int binOf(float c, float lo, float extent, int bins) {
        const int b = (int)((c - lo) / extent * (float)bins);
        return b < 0 ? 0 : (b >= bins ? bins - 1 : b);
    }

    void leaf(const BoundingBox& bb, const int primCnt, const int start, const int end, const int orderedBase, InterNode* node) {
        if (primCnt > 65535) throw std::length_error("BvhFactory: leaf exceeds the 16-bit primitive count of LinearNode");
        for (int i = start; i < end; ++i) {
            const int idx = primitiveInfo[i].primIdx;
            orderedPrims[orderedBase + (i - start)] = primitives[idx];
            orderedIdx[orderedBase + (i - start)] = idx;
        }
        node->leaf(orderedBase, primCnt, bb);
    }

    int flattenBvhTree(InterNode* node, int* offset) {
        LinearNode* linearNode = &nodes[*offset];
        linearNode->bb = node->bb;
        const int myOffset = (*offset)++;
        if (node->nodeType == InterNode::LEAF) {
            linearNode->primitivesOffset = node->firstPrimOffset;
            linearNode->primtiveCnt = (uint16_t)node->primitiveCnt;
        } else if (node->nodeType == InterNode::INTERIOR) {
            linearNode->axis = (uint8_t)node->splitAxis;
            linearNode->primtiveCnt = 0;
            flattenBvhTree(node->children[0], offset);
            linearNode->secondChildOffset = flattenBvhTree(node->children[1], offset);
        } else {
            throw std::logic_error("BvhFactory: unclassified node");
        }
        return myOffset;
    }
};

}  // namespace BVH
}  // namespace rt
#endif
