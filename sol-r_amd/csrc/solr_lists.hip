/*
 * solr_lists.hip - the order-free node lists (solr_hip.hip, buildFreeOrderLists) built on the device.
 *
 * What is built is the engine's own hierarchy over the reference's leaves, not anything of the reference's: a
 * binary surface-area tree (binned SAH, sixteen bins, over the leaf boxes' centres), the inner nodes that hardly
 * cull left out (pruneInnerNodes), flattened depth-first once per sign octant of a ray's direction.  The host
 * builds it top-down with a stack in 40-65 ms for the 100k-primitive scenes (profiles/r2/upload_time.txt) - three to
 * four times what the device needs for the reference's whole tree (solr_tree.hip) - and a scene waits for it before
 * its first fast frame.  Here the same tree is built level by level:
 *   per level   k_bounds      bounds and centre bounds of every open node (atomic min / max from its leaves)
 *               k_bins        the 3 x 16 bins of every open node: counts and bounds (atomics)
 *               k_split       one thread per open node: the host's cost loop, word for word (double precision,
 *                             axes and bins in the same order, strict <), children allocated by a scan
 *               k_flags + hipcub scan + k_scatter   a STABLE partition of every node's leaves (the host uses
 *                             std::stable_partition: the order inside a node is the order of the leaf list, which
 *                             decides the one split that looks at it - leaves whose centres all coincide are
 *                             halved by position)
 *   then        k_prune       top-down per level, one workgroup per inner node: the share of the nearest kept
 *                             ancestor's leaves whose centre lies in the node (sampled like the host: every
 *                             stride-th leaf of that ancestor in octant-0 order, which is the order the partition
 *                             leaves the leaves in), the surface ratio, the decision - all in double
 *               k_sizes       bottom-up per level; k_places top-down per level, eight octants at once
 *               k_emit        rows, start indices and origins of the eight lists
 * Every quantity is a min, a max, a count or a double-precision expression of those: none depends on the order in
 * which the atomics land, so the lists are the host's BIT FOR BIT (tests/test_lists_gpu.py holds them to the host
 * builder on the BASELINE scenes; SOLR_HIP_LISTS_ON_HOST=1 keeps the host path).  Zeros are canonical (+0) in both
 * builders' inner bounds - std::min keeps whichever zero came first, an atomic cannot.
 *
 * Declined (returns -1, the caller builds on the host): fewer than two leaves, a tree deeper than 64 levels, an
 * allocation that fails.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lists_device.h"

namespace
{
const int BINS = 16;
const int MAX_DEPTH = 64;

struct Node
{
    float lo[3], hi[3];   /* bounds of the leaves below */
    float clo[3], chi[3]; /* bounds of their centres */
    int from, to;         /* its leaves: positions [from, to) of the leaf order */
    int left, right, axis, bin;
    int leaf;             /* position in the leaf arrays of a one-leaf node, else -1 */
    int keep;             /* inner node: stays in the lists */
    int keptAncestor;     /* nearest ancestor that stays, -1: none */
    int size;             /* nodes of its subtree in a list */
    int mid;              /* first position of the right child */
    int split;            /* 1: splits this level */
};

struct BinSet
{
    int count[3][BINS];
    float lo[3][BINS][3], hi[3][BINS][3];
};

template <class T>
struct Dev
{
    T *p = nullptr;
    size_t n = 0;
    bool alloc(size_t count)
    {
        n = count;
        return hipMalloc((void **)&p, std::max(count, (size_t)1) * sizeof(T)) == hipSuccess;
    }
    ~Dev()
    {
        if (p)
            (void)hipFree(p);
    }
};

__device__ inline void atomicMinF(float *addr, float v)
{
    v += 0.f; /* -0 -> +0: the integer orderings below disagree about the sign of zero */
    if (v >= 0.f)
        atomicMin((int *)addr, __float_as_int(v));
    else
        atomicMax((unsigned *)addr, __float_as_uint(v));
}
__device__ inline void atomicMaxF(float *addr, float v)
{
    v += 0.f;
    if (v >= 0.f)
        atomicMax((int *)addr, __float_as_int(v));
    else
        atomicMin((unsigned *)addr, __float_as_uint(v));
}

/* open nodes of the level [first, first + count): bounds start empty */
__global__ void k_open(Node *nodes, int first, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    Node &t = nodes[first + i];
    for (int k = 0; k < 3; ++k)
    {
        t.lo[k] = t.clo[k] = 1e30f;
        t.hi[k] = t.chi[k] = -1e30f;
    }
    t.left = t.right = -1;
    t.axis = 0;
    t.bin = 0;
    t.leaf = -1;
    t.keep = 1;
    t.split = 0;
    t.mid = t.from;
}

__global__ void k_bounds(Node *nodes, const int *nodeOf, const int *order, const float *llo, const float *lhi, int nbLeaves,
                         int levelFirst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbLeaves)
        return;
    const int t = nodeOf[i];
    if (t < levelFirst)
        return; /* its node closed on a level above */
    Node &node = nodes[t];
    const int leaf = order[i];
    if (node.to - node.from == 1)
    {
        for (int k = 0; k < 3; ++k)
        {
            node.lo[k] = llo[3 * leaf + k];
            node.hi[k] = lhi[3 * leaf + k];
        }
        node.leaf = leaf;
        return;
    }
    for (int k = 0; k < 3; ++k)
    {
        const float lo = llo[3 * leaf + k], hi = lhi[3 * leaf + k];
        const float c = 0.5f * (lo + hi);
        atomicMinF(&node.lo[k], lo);
        atomicMaxF(&node.hi[k], hi);
        atomicMinF(&node.clo[k], c);
        atomicMaxF(&node.chi[k], c);
    }
}

__global__ void k_clearBins(BinSet *bins, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * 3 * BINS)
        return;
    BinSet &b = bins[i / (3 * BINS)];
    const int axis = (i / BINS) % 3, bin = i % BINS;
    b.count[axis][bin] = 0;
    for (int k = 0; k < 3; ++k)
    {
        b.lo[axis][bin][k] = 1e30f;
        b.hi[axis][bin][k] = -1e30f;
    }
}

__device__ inline int binOf(float c, float origin, float scale)
{
    return min(BINS - 1, max(0, (int)((c - origin) * scale)));
}

__global__ void k_bins(const Node *nodes, BinSet *bins, const int *nodeOf, const int *order, const float *llo, const float *lhi,
                       int nbLeaves, int levelFirst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbLeaves)
        return;
    const int t = nodeOf[i];
    if (t < levelFirst)
        return;
    const Node &node = nodes[t];
    if (node.to - node.from < 2)
        return;
    BinSet &b = bins[t - levelFirst];
    const int leaf = order[i];
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k)
    {
        lo[k] = llo[3 * leaf + k];
        hi[k] = lhi[3 * leaf + k];
    }
    for (int axis = 0; axis < 3; ++axis)
    {
        const float extent = node.chi[axis] - node.clo[axis];
        const float scale = extent > 0.f ? BINS / extent : 0.f;
        if (!(scale > 0.f))
            continue;
        const float c = 0.5f * (lo[axis] + hi[axis]);
        const int bin = binOf(c, node.clo[axis], scale);
        atomicAdd(&b.count[axis][bin], 1);
        for (int k = 0; k < 3; ++k)
        {
            atomicMinF(&b.lo[axis][bin][k], lo[k]);
            atomicMaxF(&b.hi[axis][bin][k], hi[k]);
        }
    }
}

__device__ inline double areaOf(const float *lo, const float *hi)
{
    const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
    return x * y + y * z + z * x;
}

/* the host's cost loop (solr_hip.hip buildFreeOrderLists), one thread per open node */
__global__ void k_split(Node *nodes, const BinSet *bins, int levelFirst, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    Node &t = nodes[levelFirst + i];
    const int n = t.to - t.from;
    if (n < 2)
        return;
    const BinSet &b = bins[i];
    int bestAxis = -1, bestBin = 0;
    double bestCost = 1e300;
    for (int axis = 0; axis < 3; ++axis)
    {
        const float extent = t.chi[axis] - t.clo[axis];
        const float scale = extent > 0.f ? BINS / extent : 0.f;
        if (!(scale > 0.f))
            continue;
        double rightArea[BINS];
        int rightCount[BINS];
        float rlo[3] = {1e30f, 1e30f, 1e30f}, rhi[3] = {-1e30f, -1e30f, -1e30f};
        int rc = 0;
        for (int bin = BINS - 1; bin > 0; --bin)
        {
            rc += b.count[axis][bin];
            for (int k = 0; k < 3; ++k)
            {
                rlo[k] = fminf(rlo[k], b.lo[axis][bin][k]);
                rhi[k] = fmaxf(rhi[k], b.hi[axis][bin][k]);
            }
            rightCount[bin] = rc;
            rightArea[bin] = rc ? areaOf(rlo, rhi) : 0.0;
        }
        float llo[3] = {1e30f, 1e30f, 1e30f}, lhi[3] = {-1e30f, -1e30f, -1e30f};
        int lc = 0;
        for (int bin = 0; bin + 1 < BINS; ++bin)
        {
            lc += b.count[axis][bin];
            for (int k = 0; k < 3; ++k)
            {
                llo[k] = fminf(llo[k], b.lo[axis][bin][k]);
                lhi[k] = fmaxf(lhi[k], b.hi[axis][bin][k]);
            }
            if (lc == 0 || rightCount[bin + 1] == 0)
                continue;
            const double cost = areaOf(llo, lhi) * lc + rightArea[bin + 1] * rightCount[bin + 1];
            if (cost < bestCost)
            {
                bestCost = cost;
                bestAxis = axis;
                bestBin = bin;
            }
        }
    }
    t.split = 1;
    if (bestAxis < 0)
    {
        t.axis = 0;
        t.bin = -1; /* all centres coincide: halved by position */
        t.mid = t.from + n / 2;
    }
    else
    {
        int lc = 0;
        for (int bin = 0; bin <= bestBin; ++bin)
            lc += b.count[bestAxis][bin];
        t.axis = bestAxis;
        t.bin = bestBin;
        t.mid = t.from + lc;
    }
}

/* children of the nodes that split: two new nodes each, in the order of their parents */
__global__ void k_children(Node *nodes, const int *rank, int levelFirst, int count, int nextFree)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    Node &t = nodes[levelFirst + i];
    if (!t.split)
        return;
    const int left = nextFree + 2 * rank[i];
    t.left = left;
    t.right = left + 1;
    nodes[left].from = t.from;
    nodes[left].to = t.mid;
    nodes[left + 1].from = t.mid;
    nodes[left + 1].to = t.to;
}

__global__ void k_splitFlags(const Node *nodes, int *flags, int levelFirst, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count)
        flags[i] = nodes[levelFirst + i].split;
}

/* 1 for a leaf that goes to the left child of its node */
__global__ void k_flags(const Node *nodes, const int *nodeOf, const int *order, const float *llo, const float *lhi, int *flags,
                        int nbLeaves, int levelFirst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbLeaves)
        return;
    const int t = nodeOf[i];
    int f = 0;
    if (t >= levelFirst && nodes[t].split)
    {
        const Node &node = nodes[t];
        if (node.bin < 0)
            f = i < node.mid;
        else
        {
            const int leaf = order[i];
            const int axis = node.axis;
            const float scale = BINS / (node.chi[axis] - node.clo[axis]);
            const float c = 0.5f * (llo[3 * leaf + axis] + lhi[3 * leaf + axis]);
            f = binOf(c, node.clo[axis], scale) <= node.bin;
        }
    }
    flags[i] = f;
}

__global__ void k_scatter(const Node *nodes, const int *nodeOf, const int *order, const int *flags, const int *before,
                          int *nodeOfOut, int *orderOut, int nbLeaves, int levelFirst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbLeaves)
        return;
    const int t = nodeOf[i];
    if (t < levelFirst || !nodes[t].split)
    {
        nodeOfOut[i] = t;
        orderOut[i] = order[i];
        return;
    }
    const Node &node = nodes[t];
    const int leftRank = before[i] - before[node.from];
    const int to = flags[i] ? node.from + leftRank : node.mid + (i - node.from - leftRank);
    nodeOfOut[to] = flags[i] ? node.left : node.right;
    orderOut[to] = order[i];
}

/* Which inner nodes stay (solr_hip.hip pruneInnerNodes on the octant-0 flattening): one workgroup per node of the
 * level, top-down, so that the nearest kept ancestor is known.  The leaves of a node in octant-0 order are the
 * positions [from, to) of the final leaf order. */
__global__ __launch_bounds__(256) void k_prune(Node *nodes, const int *order, const float *llo, const float *lhi, int levelFirst,
                                                int count, double threshold, int nbLeaves, int *nbPruned)
{
    const int t = levelFirst + blockIdx.x;
    if ((int)blockIdx.x >= count)
        return;
    Node &node = nodes[t];
    __shared__ int inside, sampled;
    if (node.leaf >= 0)
        return;
    const int ancestor = node.keptAncestor;
    const int parentFrom = ancestor < 0 ? 0 : nodes[ancestor].from, parentTo = ancestor < 0 ? nbLeaves : nodes[ancestor].to;
    if (threadIdx.x == 0)
        inside = sampled = 0;
    __syncthreads();
    const int stride = max(1, (parentTo - parentFrom) / 4096);
    int mine = 0, seen = 0;
    for (int q = parentFrom + (int)threadIdx.x * stride; q < parentTo; q += (int)blockDim.x * stride)
    {
        const int leaf = order[q];
        bool in = true;
        for (int k = 0; k < 3 && in; ++k)
        {
            const double c = 0.5 * ((double)llo[3 * leaf + k] + lhi[3 * leaf + k]);
            in = c >= node.lo[k] && c <= node.hi[k];
        }
        ++seen;
        mine += in ? 1 : 0;
    }
    atomicAdd(&inside, mine);
    atomicAdd(&sampled, seen);
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    /* (the root of the scene: the union of the top-level nodes of the list, here the tree's root) */
    const double parentArea = ancestor < 0 ? areaOf(nodes[0].lo, nodes[0].hi) : areaOf(nodes[ancestor].lo, nodes[ancestor].hi);
    const double bySurface = parentArea > 0.0 ? fmin(1.0, areaOf(node.lo, node.hi) / parentArea) : 1.0;
    const double byOrigin = sampled ? (double)inside / sampled : 1.0;
    const int below = 2 * (node.to - node.from) - 2; /* nodes under it in the unpruned list */
    const bool prune = (1.0 - fmax(bySurface, byOrigin)) * below < threshold;
    node.keep = prune ? 0 : 1;
    if (prune)
        atomicAdd(nbPruned, 1);
    const int mineOrAbove = prune ? ancestor : t;
    nodes[node.left].keptAncestor = mineOrAbove;
    nodes[node.right].keptAncestor = mineOrAbove;
}

__global__ void k_sizes(Node *nodes, int levelFirst, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    Node &t = nodes[levelFirst + i];
    t.size = t.leaf >= 0 ? 1 : (t.keep ? 1 : 0) + nodes[t.left].size + nodes[t.right].size;
}

__global__ void k_places(const Node *nodes, int *place, int levelFirst, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * 8)
        return;
    const int t = levelFirst + i / 8, octant = i % 8;
    const Node &node = nodes[t];
    if (node.leaf >= 0)
        return;
    const int at = place[8 * t + octant] + (node.keep ? 1 : 0);
    const bool highFirst = (octant >> node.axis) & 1; /* direction negative along the split axis */
    const int first = highFirst ? node.right : node.left, second = highFirst ? node.left : node.right;
    place[8 * first + octant] = at;
    place[8 * second + octant] = at + nodes[first].size;
}

__global__ void k_emit(const Node *nodes, const int *place, const float4 *leafRows, const int *leafStart, const int *leafOrigin,
                       float4 *outRows, int *outStart, int *outOrigin, int nbNodes, int listLength)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbNodes * 8)
        return;
    const int t = i / 8, octant = i % 8;
    const Node &node = nodes[t];
    const int at = place[8 * t + octant];
    float4 *rows = outRows + 2 * (size_t)octant * listLength;
    if (node.leaf >= 0)
    {
        rows[2 * at] = leafRows[2 * node.leaf];
        float4 second = leafRows[2 * node.leaf + 1];
        second.w = __int_as_float(1);
        rows[2 * at + 1] = second;
        outStart[(size_t)octant * listLength + at] = leafStart[node.leaf];
        outOrigin[(size_t)octant * listLength + at] = leafOrigin[node.leaf];
        return;
    }
    if (!node.keep)
        return;
    rows[2 * at] = make_float4(node.lo[0], node.lo[1], node.lo[2], node.hi[2]);
    rows[2 * at + 1] = make_float4(node.hi[0], node.hi[1], __int_as_float(0), __int_as_float(node.size));
    outStart[(size_t)octant * listLength + at] = 0;
    outOrigin[(size_t)octant * listLength + at] = -1;
}

__global__ void k_iota(int *a, int n, int value, bool counting)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        a[i] = counting ? i : value;
}

inline dim3 blocksFor(size_t n, int block = 256)
{
    return dim3((unsigned)((n + block - 1) / block));
}
} // namespace

int solrBuildOrderFreeListsOnDevice(const float4 *rows, const int *start, const int *origin, int n, double threshold,
                                    std::vector<float4> &outRows, std::vector<int> &outStart, std::vector<int> &outOrigin,
                                    int *nbPruned, hipStream_t stream)
{
    *nbPruned = 0;
    /* the leaves: every node with primitives */
    std::vector<float> llo, lhi;
    std::vector<float4> leafRows;
    std::vector<int> leafStart, leafOrigin;
    for (int i = 0; i < n; ++i)
    {
        int count;
        memcpy(&count, &rows[2 * i + 1].z, 4);
        if (count <= 0)
            continue;
        llo.insert(llo.end(), {rows[2 * i].x, rows[2 * i].y, rows[2 * i].z});
        lhi.insert(lhi.end(), {rows[2 * i + 1].x, rows[2 * i + 1].y, rows[2 * i].w});
        leafRows.push_back(rows[2 * i]);
        leafRows.push_back(rows[2 * i + 1]);
        leafStart.push_back(start[i]);
        leafOrigin.push_back(origin[i]);
    }
    const int L = (int)leafStart.size();
    if (L < 2)
        return -1;
    const int maxNodes = 2 * L - 1;
    Dev<float> dLo, dHi;
    Dev<float4> dLeafRows, dOutRows;
    Dev<int> dLeafStart, dLeafOrigin, dOrder[2], dNodeOf[2], dFlags, dBefore, dRank, dPlace, dOutStart, dOutOrigin, dPruned;
    Dev<Node> dNodes;
    Dev<BinSet> dBins;
    Dev<char> dTemp;
    if (!dLo.alloc(3 * (size_t)L) || !dHi.alloc(3 * (size_t)L) || !dLeafRows.alloc(2 * (size_t)L) || !dLeafStart.alloc(L) ||
        !dLeafOrigin.alloc(L) || !dOrder[0].alloc(L) || !dOrder[1].alloc(L) || !dNodeOf[0].alloc(L) || !dNodeOf[1].alloc(L) ||
        !dFlags.alloc((size_t)L + 1) || !dBefore.alloc((size_t)L + 1) || !dRank.alloc((size_t)L + 1) || !dNodes.alloc(maxNodes) ||
        !dBins.alloc(L) || !dPlace.alloc(8 * (size_t)maxNodes) || !dPruned.alloc(1))
        return -1;
    size_t tempBytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tempBytes, dFlags.p, dBefore.p, L + 1, stream);
    if (!dTemp.alloc(tempBytes + 256))
        return -1;
#define LISTS_CHECK(call)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        if ((call) != hipSuccess)                                                                                      \
        {                                                                                                              \
            fprintf(stderr, "solr_lists: %s failed\n", #call);                                                         \
            return -1;                                                                                                 \
        }                                                                                                              \
    } while (0)
    LISTS_CHECK(hipMemcpyAsync(dLo.p, llo.data(), llo.size() * 4, hipMemcpyHostToDevice, stream));
    LISTS_CHECK(hipMemcpyAsync(dHi.p, lhi.data(), lhi.size() * 4, hipMemcpyHostToDevice, stream));
    LISTS_CHECK(hipMemcpyAsync(dLeafRows.p, leafRows.data(), leafRows.size() * 16, hipMemcpyHostToDevice, stream));
    LISTS_CHECK(hipMemcpyAsync(dLeafStart.p, leafStart.data(), (size_t)L * 4, hipMemcpyHostToDevice, stream));
    LISTS_CHECK(hipMemcpyAsync(dLeafOrigin.p, leafOrigin.data(), (size_t)L * 4, hipMemcpyHostToDevice, stream));
    LISTS_CHECK(hipMemsetAsync(dPruned.p, 0, 4, stream));
    hipLaunchKernelGGL(k_iota, blocksFor(L), dim3(256), 0, stream, dOrder[0].p, L, 0, true);
    hipLaunchKernelGGL(k_iota, blocksFor(L), dim3(256), 0, stream, dNodeOf[0].p, L, 0, false);
    /* the root */
    Node root;
    memset(&root, 0, sizeof(root));
    root.from = 0;
    root.to = L;
    root.keptAncestor = -1;
    LISTS_CHECK(hipMemcpyAsync(dNodes.p, &root, sizeof(root), hipMemcpyHostToDevice, stream));

    std::vector<int> levelFirst, levelCount;
    int first = 0, count = 1, nextFree = 1, cur = 0;
    for (int depth = 0; count > 0; ++depth)
    {
        if (depth >= MAX_DEPTH)
            return -1; /* a degenerate scene: the host's stack does not mind */
        levelFirst.push_back(first);
        levelCount.push_back(count);
        hipLaunchKernelGGL(k_open, blocksFor(count), dim3(256), 0, stream, dNodes.p, first, count);
        hipLaunchKernelGGL(k_bounds, blocksFor(L), dim3(256), 0, stream, dNodes.p, dNodeOf[cur].p, dOrder[cur].p, dLo.p, dHi.p, L, first);
        hipLaunchKernelGGL(k_clearBins, blocksFor((size_t)count * 3 * BINS), dim3(256), 0, stream, dBins.p, count);
        hipLaunchKernelGGL(k_bins, blocksFor(L), dim3(256), 0, stream, dNodes.p, dBins.p, dNodeOf[cur].p, dOrder[cur].p, dLo.p, dHi.p, L, first);
        hipLaunchKernelGGL(k_split, blocksFor(count, 64), dim3(64), 0, stream, dNodes.p, dBins.p, first, count);
        /* children: two per node that splits */
        hipLaunchKernelGGL(k_splitFlags, blocksFor(count), dim3(256), 0, stream, dNodes.p, dFlags.p, first, count);
        LISTS_CHECK(hipcub::DeviceScan::ExclusiveSum(dTemp.p, tempBytes, dFlags.p, dRank.p, count + 1, stream));
        int splitting = 0;
        LISTS_CHECK(hipMemcpyAsync(&splitting, dRank.p + count, 4, hipMemcpyDeviceToHost, stream));
        LISTS_CHECK(hipStreamSynchronize(stream));
        if (splitting == 0)
            break;
        if (nextFree + 2 * splitting > maxNodes)
            return -1;
        hipLaunchKernelGGL(k_children, blocksFor(count), dim3(256), 0, stream, dNodes.p, dRank.p, first, count, nextFree);
        /* the stable partition of every splitting node's leaves */
        hipLaunchKernelGGL(k_flags, blocksFor(L), dim3(256), 0, stream, dNodes.p, dNodeOf[cur].p, dOrder[cur].p, dLo.p, dHi.p, dFlags.p, L, first);
        LISTS_CHECK(hipcub::DeviceScan::ExclusiveSum(dTemp.p, tempBytes, dFlags.p, dBefore.p, L + 1, stream));
        hipLaunchKernelGGL(k_scatter, blocksFor(L), dim3(256), 0, stream, dNodes.p, dNodeOf[cur].p, dOrder[cur].p, dFlags.p, dBefore.p,
                           dNodeOf[cur ^ 1].p, dOrder[cur ^ 1].p, L, first);
        cur ^= 1;
        first = nextFree;
        count = 2 * splitting;
        nextFree += count;
    }
    const int nbNodes = nextFree;
    const int levels = (int)levelFirst.size();
    /* which inner nodes stay: top-down */
    for (int d = 0; d < levels; ++d)
        hipLaunchKernelGGL(k_prune, dim3((unsigned)levelCount[d]), dim3(256), 0, stream, dNodes.p, dOrder[cur].p, dLo.p, dHi.p, levelFirst[d],
                           levelCount[d], threshold, L, dPruned.p);
    for (int d = levels - 1; d >= 0; --d)
        hipLaunchKernelGGL(k_sizes, blocksFor(levelCount[d]), dim3(256), 0, stream, dNodes.p, levelFirst[d], levelCount[d]);
    LISTS_CHECK(hipMemsetAsync(dPlace.p, 0, 8 * sizeof(int), stream)); /* the root sits at 0 in every list */
    for (int d = 0; d < levels; ++d)
        hipLaunchKernelGGL(k_places, blocksFor((size_t)levelCount[d] * 8), dim3(256), 0, stream, dNodes.p, dPlace.p, levelFirst[d], levelCount[d]);
    Node rootNow;
    LISTS_CHECK(hipMemcpyAsync(&rootNow, dNodes.p, sizeof(Node), hipMemcpyDeviceToHost, stream));
    LISTS_CHECK(hipMemcpyAsync(nbPruned, dPruned.p, 4, hipMemcpyDeviceToHost, stream));
    LISTS_CHECK(hipStreamSynchronize(stream));
    const int listLength = rootNow.size;
    if (listLength < 2)
        return -1;
    if (!dOutRows.alloc(16 * (size_t)listLength) || !dOutStart.alloc(8 * (size_t)listLength) || !dOutOrigin.alloc(8 * (size_t)listLength))
        return -1;
    hipLaunchKernelGGL(k_emit, blocksFor((size_t)nbNodes * 8), dim3(256), 0, stream, dNodes.p, dPlace.p, dLeafRows.p, dLeafStart.p, dLeafOrigin.p,
                       dOutRows.p, dOutStart.p, dOutOrigin.p, nbNodes, listLength);
    LISTS_CHECK(hipGetLastError());
    outRows.resize(16 * (size_t)listLength);
    outStart.resize(8 * (size_t)listLength);
    outOrigin.resize(8 * (size_t)listLength);
    LISTS_CHECK(hipMemcpyAsync(outRows.data(), dOutRows.p, outRows.size() * 16, hipMemcpyDeviceToHost, stream));
    LISTS_CHECK(hipMemcpyAsync(outStart.data(), dOutStart.p, outStart.size() * 4, hipMemcpyDeviceToHost, stream));
    LISTS_CHECK(hipMemcpyAsync(outOrigin.data(), dOutOrigin.p, outOrigin.size() * 4, hipMemcpyDeviceToHost, stream));
    LISTS_CHECK(hipStreamSynchronize(stream));
#undef LISTS_CHECK
    return listLength;
}
