/*
 * solr_lists.hip - the order-free node lists (solr_scene.hip, buildFreeOrderLists) built on the device.
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
#include <mutex>
#include <vector>

#include <chrono>

#include "device_pool.h"
#include "lists_device.h"

namespace
{
struct Phase
{
    const bool on = getenv("SOLR_HIP_DEBUG_TIMING") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void mark(const char *what, int detail = -1)
    {
        if (!on)
            return;
        const auto now = std::chrono::steady_clock::now();
        if (detail >= 0)
            fprintf(stderr, "solr_lists: %-24s %8.2f ms (%d)\n", what, std::chrono::duration<double, std::milli>(now - last).count(), detail);
        else
            fprintf(stderr, "solr_lists: %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

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
    bool own = false;
    /* scratch of this build: from the pool (device_pool.h) while it lasts */
    bool alloc(size_t count)
    {
        n = count;
        const size_t bytes = std::max(count, (size_t)1) * sizeof(T);
        p = (T *)solrScratchPool().take(bytes);
        return p != nullptr || allocOwn(count);
    }
    /* memory that may outlive the build */
    bool allocOwn(size_t count)
    {
        n = count;
        own = true;
        return hipMalloc((void **)&p, std::max(count, (size_t)1) * sizeof(T)) == hipSuccess;
    }
    ~Dev()
    {
        if (p && own)
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
__global__ void k_open(Node *nodes, BinSet *bins, int first, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    Node &t = nodes[first + i];
    BinSet &b = bins[i];
    for (int axis = 0; axis < 3; ++axis)
        for (int bin = 0; bin < BINS; ++bin)
        {
            b.count[axis][bin] = 0;
            for (int k = 0; k < 3; ++k)
            {
                b.lo[axis][bin][k] = 1e30f;
                b.hi[axis][bin][k] = -1e30f;
            }
        }
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

__device__ inline float waveMin(float v)
{
    for (int off = 32; off > 0; off >>= 1)
        v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ inline float waveMax(float v)
{
    for (int off = 32; off > 0; off >>= 1)
        v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(256) void k_bounds(Node *nodes, const int *nodeOf, const int *order, const float *llo, const float *lhi,
                                                 int nbLeaves, int levelFirst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = i < nbLeaves ? nodeOf[i] : -1;
    const bool open = t >= levelFirst && nodes[t].to - nodes[t].from > 1;
    if (t >= levelFirst && !open)
    {
        Node &node = nodes[t]; /* a node of one leaf */
        const int leaf = order[i];
        for (int k = 0; k < 3; ++k)
        {
            node.lo[k] = llo[3 * leaf + k];
            node.hi[k] = lhi[3 * leaf + k];
        }
        node.leaf = leaf;
    }
    /* near the root a node holds thousands of consecutive leaves: a wave that lies inside one node reduces first and
     * sends one atomic per value instead of sixty-four to the same address */
    const int firstNode = __builtin_amdgcn_readfirstlane(t);
    const bool uniform = __builtin_amdgcn_ballot_w64(!open || t != firstNode) == 0ull;
    float lo[3], hi[3], c[3];
    if (open)
    {
        const int leaf = order[i];
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = llo[3 * leaf + k];
            hi[k] = lhi[3 * leaf + k];
            c[k] = 0.5f * (lo[k] + hi[k]);
        }
    }
    if (uniform)
    {
        Node &node = nodes[firstNode];
        for (int k = 0; k < 3; ++k)
        {
            const float a = waveMin(lo[k]), b = waveMax(hi[k]), d = waveMin(c[k]), e = waveMax(c[k]);
            if ((threadIdx.x & 63) == 0)
            {
                atomicMinF(&node.lo[k], a);
                atomicMaxF(&node.hi[k], b);
                atomicMinF(&node.clo[k], d);
                atomicMaxF(&node.chi[k], e);
            }
        }
    }
    else if (open)
    {
        Node &node = nodes[t];
        for (int k = 0; k < 3; ++k)
        {
            atomicMinF(&node.lo[k], lo[k]);
            atomicMaxF(&node.hi[k], hi[k]);
            atomicMinF(&node.clo[k], c[k]);
            atomicMaxF(&node.chi[k], c[k]);
        }
    }
}

__device__ inline int binOf(float c, float origin, float scale)
{
    return min(BINS - 1, max(0, (int)((c - origin) * scale)));
}

__device__ inline void ldsMinF(float *addr, float v)
{
    v += 0.f;
    if (v >= 0.f)
        atomicMin((int *)addr, __float_as_int(v));
    else
        atomicMax((unsigned *)addr, __float_as_uint(v));
}
__device__ inline void ldsMaxF(float *addr, float v)
{
    v += 0.f;
    if (v >= 0.f)
        atomicMax((int *)addr, __float_as_int(v));
    else
        atomicMin((unsigned *)addr, __float_as_uint(v));
}

__global__ __launch_bounds__(256) void k_bins(const Node *nodes, BinSet *bins, const int *nodeOf, const int *order, const float *llo,
                                               const float *lhi, int nbLeaves, int levelFirst)
{
    __shared__ BinSet local;
    __shared__ int sameNode;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = i < nbLeaves ? nodeOf[i] : -1;
    const bool open = t >= levelFirst && nodes[t].to - nodes[t].from > 1;
    /* does the whole workgroup lie in one open node?  (near the root: always) */
    if (threadIdx.x == 0)
        sameNode = t;
    __syncthreads();
    const int blockNode = sameNode;
    __syncthreads();
    if (!open || t != blockNode)
        sameNode = -1; /* benign race: every writer writes the same value */
    __syncthreads();
    const bool together = sameNode >= 0;
    if (together)
    {
        for (int q = threadIdx.x; q < 3 * BINS; q += blockDim.x)
        {
            const int axis = q / BINS, bin = q % BINS;
            local.count[axis][bin] = 0;
            for (int k = 0; k < 3; ++k)
            {
                local.lo[axis][bin][k] = 1e30f;
                local.hi[axis][bin][k] = -1e30f;
            }
        }
        __syncthreads();
    }
    if (open)
    {
        const Node &node = nodes[t];
        BinSet &b = together ? local : bins[t - levelFirst];
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
                if (together)
                {
                    ldsMinF(&b.lo[axis][bin][k], lo[k]);
                    ldsMaxF(&b.hi[axis][bin][k], hi[k]);
                }
                else
                {
                    atomicMinF(&b.lo[axis][bin][k], lo[k]);
                    atomicMaxF(&b.hi[axis][bin][k], hi[k]);
                }
            }
        }
    }
    if (together)
    {
        __syncthreads();
        BinSet &out = bins[blockNode - levelFirst];
        for (int q = threadIdx.x; q < 3 * BINS; q += blockDim.x)
        {
            const int axis = q / BINS, bin = q % BINS;
            if (local.count[axis][bin] == 0)
                continue;
            atomicAdd(&out.count[axis][bin], local.count[axis][bin]);
            for (int k = 0; k < 3; ++k)
            {
                atomicMinF(&out.lo[axis][bin][k], local.lo[axis][bin][k]);
                atomicMaxF(&out.hi[axis][bin][k], local.hi[axis][bin][k]);
            }
        }
    }
}

__device__ inline double areaOf(const float *lo, const float *hi)
{
    const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
    return x * y + y * z + z * x;
}

/* the host's cost loop (solr_scene.hip buildFreeOrderLists), one thread per open node */
__global__ void k_split(Node *nodes, const BinSet *bins, int levelFirst, int count, int nextFree, int *splitting)
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
    /* two children (which pair of the level's new nodes is no matter: the lists follow from the tree's shape) */
    const int left = nextFree + 2 * atomicAdd(splitting, 1);
    t.left = left;
    t.right = left + 1;
    nodes[left].from = t.from;
    nodes[left].to = t.mid;
    nodes[left + 1].from = t.mid;
    nodes[left + 1].to = t.to;
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

/* Which inner nodes stay (solr_scene.hip pruneInnerNodes on the octant-0 flattening): one workgroup per node of the
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

/* the leaves of a node list that is on the device already: which nodes, then their rows in list order */
__global__ void k_leafFlags(const float4 *rows, int n, int *flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n)
        flags[i] = i < n && __float_as_int(rows[2 * i + 1].z) > 0 ? 1 : 0;
}
__global__ void k_leafGather(const float4 *rows, const int *start, const int *before, int n, float *lo, float *hi, float4 *leafRows,
                             int *leafStart, int *leafOrigin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float4 a = rows[2 * i], b = rows[2 * i + 1];
    if (__float_as_int(b.z) <= 0)
        return;
    const int l = before[i];
    lo[3 * l] = a.x, lo[3 * l + 1] = a.y, lo[3 * l + 2] = a.z;
    hi[3 * l] = b.x, hi[3 * l + 1] = b.y, hi[3 * l + 2] = a.w;
    leafRows[2 * l] = a;
    leafRows[2 * l + 1] = b;
    leafStart[l] = start[i];
    leafOrigin[l] = i;
}

inline dim3 blocksFor(size_t n, int block = 256)
{
    return dim3((unsigned)((n + block - 1) / block));
}
} // namespace

int solrBuildOrderFreeListsOnDevice(const float4 *rows, const int *start, const int *origin, int n, double threshold,
                                    std::vector<float4> &outRows, std::vector<int> &outStart, std::vector<int> &outOrigin,
                                    int *nbPruned, hipStream_t stream, SolrDeviceLists *stay)
{
    *nbPruned = 0;
    Phase phase;
    /* scratch: per node of the source list two ints and the scan's; per leaf (a quarter of the nodes in the grid
     * builder's trees; what a scene with more asks for beyond the pool it gets from hipMalloc) the rows, bounds and
     * bookkeeping, two tree nodes and a set of bins */
    SolrScratchPool &pool = solrScratchPool();
    std::lock_guard<std::mutex> oneBuild(pool.busy);
    struct Scope
    {
        SolrScratchPool &pool;
        ~Scope() { pool.end(); }
    } scope{pool};
    pool.begin((size_t)n * 12 + (size_t)n / 4 * (160 + 2 * sizeof(Node) + sizeof(BinSet) + 64 * sizeof(int)) + ((size_t)4 << 20));
#define LISTS_CHECK(call)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        if ((call) != hipSuccess)                                                                                      \
        {                                                                                                              \
            fprintf(stderr, "solr_lists: %s failed\n", #call);                                                         \
            return -1;                                                                                                 \
        }                                                                                                              \
    } while (0)
    /* the leaves: every node with primitives */
    std::vector<float> llo, lhi;
    std::vector<float4> leafRows;
    std::vector<int> leafStart, leafOrigin;
    Dev<int> dLeafFlags, dLeafBefore;
    Dev<char> dLeafTemp;
    int L = 0;
    if (origin == nullptr)
    {
        /* rows and start are the arena's (device memory), every node its own origin */
        size_t bytes = 0;
        if (!dLeafFlags.alloc((size_t)n + 1) || !dLeafBefore.alloc((size_t)n + 1))
            return -1;
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, dLeafFlags.p, dLeafBefore.p, n + 1, stream);
        if (!dLeafTemp.alloc(bytes + 256))
            return -1;
        hipLaunchKernelGGL(k_leafFlags, blocksFor((size_t)n + 1), dim3(256), 0, stream, rows, n, dLeafFlags.p);
        LISTS_CHECK(hipcub::DeviceScan::ExclusiveSum(dLeafTemp.p, bytes, dLeafFlags.p, dLeafBefore.p, n + 1, stream));
        LISTS_CHECK(hipMemcpyAsync(&L, dLeafBefore.p + n, 4, hipMemcpyDeviceToHost, stream));
        LISTS_CHECK(hipStreamSynchronize(stream));
    }
    else
    {
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
        L = (int)leafStart.size();
    }
    if (L < 2)
        return -1;
    phase.mark("leaves gathered", L);
    const int maxNodes = 2 * L - 1;
    Dev<float> dLo, dHi;
    Dev<float4> dLeafRows, dOutRows;
    Dev<int> dLeafStart, dLeafOrigin, dOrder[2], dNodeOf[2], dFlags, dBefore, dPlace, dOutStart, dOutOrigin, dPruned;
    Dev<Node> dNodes;
    Dev<BinSet> dBins;
    Dev<char> dTemp;
    if (!dLo.alloc(3 * (size_t)L) || !dHi.alloc(3 * (size_t)L) || !dLeafRows.alloc(2 * (size_t)L) || !dLeafStart.alloc(L) ||
        !dLeafOrigin.alloc(L) || !dOrder[0].alloc(L) || !dOrder[1].alloc(L) || !dNodeOf[0].alloc(L) || !dNodeOf[1].alloc(L) ||
        !dFlags.alloc((size_t)L + 1) || !dBefore.alloc((size_t)L + 1) || !dNodes.alloc(maxNodes) ||
        !dBins.alloc(L) || !dPlace.alloc(8 * (size_t)maxNodes) || !dPruned.alloc(2))
        return -1;
    size_t tempBytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tempBytes, dFlags.p, dBefore.p, L + 1, stream);
    if (!dTemp.alloc(tempBytes + 256))
        return -1;
    phase.mark("allocations");
    if (origin == nullptr)
        hipLaunchKernelGGL(k_leafGather, blocksFor(n), dim3(256), 0, stream, rows, start, dLeafBefore.p, n, dLo.p, dHi.p, dLeafRows.p,
                           dLeafStart.p, dLeafOrigin.p);
    else
    {
        LISTS_CHECK(hipMemcpyAsync(dLo.p, llo.data(), llo.size() * 4, hipMemcpyHostToDevice, stream));
        LISTS_CHECK(hipMemcpyAsync(dHi.p, lhi.data(), lhi.size() * 4, hipMemcpyHostToDevice, stream));
        LISTS_CHECK(hipMemcpyAsync(dLeafRows.p, leafRows.data(), leafRows.size() * 16, hipMemcpyHostToDevice, stream));
        LISTS_CHECK(hipMemcpyAsync(dLeafStart.p, leafStart.data(), (size_t)L * 4, hipMemcpyHostToDevice, stream));
        LISTS_CHECK(hipMemcpyAsync(dLeafOrigin.p, leafOrigin.data(), (size_t)L * 4, hipMemcpyHostToDevice, stream));
    }
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
        hipLaunchKernelGGL(k_open, blocksFor(count, 64), dim3(64), 0, stream, dNodes.p, dBins.p, first, count);
        hipLaunchKernelGGL(k_bounds, blocksFor(L), dim3(256), 0, stream, dNodes.p, dNodeOf[cur].p, dOrder[cur].p, dLo.p, dHi.p, L, first);
        hipLaunchKernelGGL(k_bins, blocksFor(L), dim3(256), 0, stream, dNodes.p, dBins.p, dNodeOf[cur].p, dOrder[cur].p, dLo.p, dHi.p, L, first);
        LISTS_CHECK(hipMemsetAsync(dPruned.p + 1, 0, 4, stream));
        hipLaunchKernelGGL(k_split, blocksFor(count, 64), dim3(64), 0, stream, dNodes.p, dBins.p, first, count, nextFree, dPruned.p + 1);
        int splitting = 0;
        LISTS_CHECK(hipMemcpyAsync(&splitting, dPruned.p + 1, 4, hipMemcpyDeviceToHost, stream));
        LISTS_CHECK(hipStreamSynchronize(stream));
        if (splitting == 0)
            break;
        if (nextFree + 2 * splitting > maxNodes)
            return -1;
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
    phase.mark("tree, level by level", levels);
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
    phase.mark("pruning, sizes, places", listLength);
    if (listLength < 2)
        return -1;
    if (stay ? !dOutRows.allocOwn(16 * (size_t)listLength) || !dOutStart.allocOwn(8 * (size_t)listLength) ||
                   !dOutOrigin.allocOwn(8 * (size_t)listLength)
             : !dOutRows.alloc(16 * (size_t)listLength) || !dOutStart.alloc(8 * (size_t)listLength) || !dOutOrigin.alloc(8 * (size_t)listLength))
        return -1;
    hipLaunchKernelGGL(k_emit, blocksFor((size_t)nbNodes * 8), dim3(256), 0, stream, dNodes.p, dPlace.p, dLeafRows.p, dLeafStart.p, dLeafOrigin.p,
                       dOutRows.p, dOutStart.p, dOutOrigin.p, nbNodes, listLength);
    LISTS_CHECK(hipGetLastError());
    if (stay)
    {
        LISTS_CHECK(hipStreamSynchronize(stream));
        stay->rows = dOutRows.p, stay->start = dOutStart.p, stay->origin = dOutOrigin.p;
        dOutRows.p = nullptr, dOutStart.p = nullptr, dOutOrigin.p = nullptr;
        phase.mark("lists written");
        return listLength;
    }
    outRows.resize(16 * (size_t)listLength);
    outStart.resize(8 * (size_t)listLength);
    outOrigin.resize(8 * (size_t)listLength);
    LISTS_CHECK(hipMemcpyAsync(outRows.data(), dOutRows.p, outRows.size() * 16, hipMemcpyDeviceToHost, stream));
    LISTS_CHECK(hipMemcpyAsync(outStart.data(), dOutStart.p, outStart.size() * 4, hipMemcpyDeviceToHost, stream));
    LISTS_CHECK(hipMemcpyAsync(outOrigin.data(), dOutOrigin.p, outOrigin.size() * 4, hipMemcpyDeviceToHost, stream));
    LISTS_CHECK(hipStreamSynchronize(stream));
#undef LISTS_CHECK
    phase.mark("lists written and copied back");
    return listLength;
}


/* ---- pruneInnerNodes' decisions for a node list with skip pointers (solr_scene.hip), on the device ------------------------
 * The host walks the list once and, for every inner node, samples up to 8 000 leaf centres of its nearest kept
 * ancestor: 13 + 8 ms of h2d_scene for the 100k-primitive scenes.  The decisions of one depth of the list are
 * independent of each other once the depths above are decided, so: parents and depths on the host (one pass with a
 * stack), then one launch per depth, one workgroup per inner node, the host's arithmetic in double. */
namespace
{
/* the centres of the leaves, in list order, as the decisions compare them: 0.5 * ((double)lo + hi) */
__global__ void k_leafCentres(const float4 *rows, const int *leaves, int nbLeaves, double *centres)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nbLeaves)
        return;
    const int leaf = leaves[q];
    const float4 la = rows[2 * leaf], lb = rows[2 * leaf + 1];
    centres[3 * (size_t)q] = 0.5 * ((double)la.x + lb.x);
    centres[3 * (size_t)q + 1] = 0.5 * ((double)la.y + lb.y);
    centres[3 * (size_t)q + 2] = 0.5 * ((double)la.z + la.w);
}

/* a node that does not lie within its parent marks the parent: such a parent is never left out */
__global__ void k_strayChildren(const float4 *rows, const int *parent, int n, char *stray)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || parent[j] < 0)
        return;
    const int i = parent[j];
    const float4 a = rows[2 * i], b = rows[2 * i + 1], ca = rows[2 * j], cb = rows[2 * j + 1];
    if (!(ca.x >= a.x && ca.y >= a.y && ca.z >= a.z && cb.x <= b.x && cb.y <= b.y && ca.w <= a.w))
        stray[i] = 1;
}

__global__ __launch_bounds__(256) void k_pruneList(const float4 *rows, const int *parent, const int *nodesOfDepth, int count,
                                                    const int *leavesBefore, const int *leaves, int n, double threshold,
                                                    double sceneArea, char *keep, int *keptAncestor, const char *stray, const double *centres)
{
    if ((int)blockIdx.x >= count)
        return;
    const int i = nodesOfDepth[blockIdx.x];
    __shared__ int inside, sampled, outside;
    const float4 a = rows[2 * i], b = rows[2 * i + 1];
    const int skip = max(__float_as_int(b.w), 1);
    const int end = min(i + skip, n);
    const int up = parent[i];
    const int ancestor = up < 0 ? -1 : (keep[up] ? up : keptAncestor[up]);
    if (threadIdx.x == 0)
    {
        keptAncestor[i] = ancestor;
        inside = sampled = outside = 0;
    }
    if (__float_as_int(b.z) != 0 || end <= i + 1)
        return; /* a leaf (or an inner node without children): stays */
    __syncthreads();
    const float lo[3] = {a.x, a.y, a.z}, hi[3] = {b.x, b.y, a.w};
    int parentFrom = 0, parentTo = n;
    double parentArea = sceneArea;
    if (ancestor >= 0)
    {
        const float4 pa = rows[2 * ancestor], pb = rows[2 * ancestor + 1];
        parentFrom = ancestor;
        parentTo = min(ancestor + max(__float_as_int(pb.w), 1), n);
        const float plo[3] = {pa.x, pa.y, pa.z}, phi[3] = {pb.x, pb.y, pa.w};
        parentArea = areaOf(plo, phi);
    }
    const int firstLeaf = leavesBefore[parentFrom], lastLeaf = leavesBefore[parentTo];
    const int stride = max(1, (lastLeaf - firstLeaf) / 4096);
    int mine = 0, seen = 0;
    for (int q = firstLeaf + (int)threadIdx.x * stride; q < lastLeaf; q += (int)blockDim.x * stride)
    {
        const double cx = centres[3 * (size_t)q], cy = centres[3 * (size_t)q + 1], cz = centres[3 * (size_t)q + 2];
        const bool in = cx >= lo[0] && cx <= hi[0] && cy >= lo[1] && cy <= hi[1] && cz >= lo[2] && cz <= hi[2];
        ++seen;
        mine += in ? 1 : 0;
    }
    const int bad = stray[i]; /* a child outside the node (k_strayChildren) */
    atomicAdd(&inside, mine);
    atomicAdd(&sampled, seen);
    if (bad && threadIdx.x == 0)
        atomicAdd(&outside, 1);
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    const double bySurface = parentArea > 0.0 ? fmin(1.0, areaOf(lo, hi) / parentArea) : 1.0;
    const double byOrigin = sampled ? (double)inside / sampled : 1.0;
    if (!outside && (1.0 - fmax(bySurface, byOrigin)) * (end - i - 1) < threshold)
        keep[i] = 0;
}
} // namespace

int solrPruneDecisionsOnDevice(const float4 *rows, int n, double threshold, std::vector<char> &keepOut, hipStream_t stream)
{
    if (n < 2)
        return -1;
    Phase phase;
    SolrScratchPool &pool = solrScratchPool();
    std::lock_guard<std::mutex> oneBuild(pool.busy);
    struct Scope
    {
        SolrScratchPool &pool;
        ~Scope() { pool.end(); }
    } scope{pool};
    pool.begin((size_t)n * 80 + ((size_t)1 << 20));
    /* parents, depths, the leaves in list order, the scene's extent: one pass */
    std::vector<int> parent(n, -1), depth(n, 0), leaves, leavesBefore((size_t)n + 1, 0), stack;
    int deepest = 0;
    double sceneLo[3] = {1e300, 1e300, 1e300}, sceneHi[3] = {-1e300, -1e300, -1e300};
    auto skipOf = [&](int i) {
        int s;
        memcpy(&s, &rows[2 * i + 1].w, 4);
        return std::max(s, 1);
    };
    for (int i = 0; i < n; ++i)
    {
        while (!stack.empty() && i >= stack.back() + skipOf(stack.back()))
            stack.pop_back();
        parent[i] = stack.empty() ? -1 : stack.back();
        depth[i] = stack.empty() ? 0 : depth[stack.back()] + 1;
        deepest = std::max(deepest, depth[i]);
        stack.push_back(i);
        int count;
        memcpy(&count, &rows[2 * i + 1].z, 4);
        leavesBefore[(size_t)i + 1] = leavesBefore[i] + (count > 0 ? 1 : 0);
        if (count > 0)
            leaves.push_back(i);
        if (parent[i] < 0)
        {
            const float lo[3] = {rows[2 * i].x, rows[2 * i].y, rows[2 * i].z}, hi[3] = {rows[2 * i + 1].x, rows[2 * i + 1].y, rows[2 * i].w};
            for (int k = 0; k < 3; ++k)
            {
                sceneLo[k] = std::min(sceneLo[k], (double)lo[k]);
                sceneHi[k] = std::max(sceneHi[k], (double)hi[k]);
            }
        }
    }
    if (deepest > 256 || leaves.empty())
        return -1;
    const double sceneArea = (sceneHi[0] - sceneLo[0]) * (sceneHi[1] - sceneLo[1]) + (sceneHi[1] - sceneLo[1]) * (sceneHi[2] - sceneLo[2]) +
                             (sceneHi[2] - sceneLo[2]) * (sceneHi[0] - sceneLo[0]);
    /* nodes by depth */
    std::vector<int> firstOfDepth((size_t)deepest + 2, 0), byDepth(n);
    for (int i = 0; i < n; ++i)
        ++firstOfDepth[(size_t)depth[i] + 1];
    for (int d = 0; d <= deepest; ++d)
        firstOfDepth[(size_t)d + 1] += firstOfDepth[d];
    {
        std::vector<int> at(firstOfDepth.begin(), firstOfDepth.end() - 1);
        for (int i = 0; i < n; ++i)
            byDepth[at[depth[i]]++] = i;
    }
    phase.mark("prune: parents and depths", deepest + 1);
    Dev<float4> dRows;
    Dev<int> dParent, dByDepth, dLeavesBefore, dLeaves, dAncestor;
    Dev<char> dKeep, dStray;
    Dev<double> dCentres;
    if (!dCentres.alloc(3 * leaves.size()) || !dStray.alloc(n) || !dRows.alloc(2 * (size_t)n) || !dParent.alloc(n) || !dByDepth.alloc(n) || !dLeavesBefore.alloc((size_t)n + 1) ||
        !dLeaves.alloc(leaves.size()) || !dAncestor.alloc(n) || !dKeep.alloc(n))
        return -1;
    bool fine = hipMemcpyAsync(dRows.p, rows, 2 * (size_t)n * 16, hipMemcpyHostToDevice, stream) == hipSuccess &&
                hipMemcpyAsync(dParent.p, parent.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream) == hipSuccess &&
                hipMemcpyAsync(dByDepth.p, byDepth.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream) == hipSuccess &&
                hipMemcpyAsync(dLeavesBefore.p, leavesBefore.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, stream) == hipSuccess &&
                hipMemcpyAsync(dLeaves.p, leaves.data(), leaves.size() * 4, hipMemcpyHostToDevice, stream) == hipSuccess &&
                hipMemsetAsync(dKeep.p, 1, n, stream) == hipSuccess && hipMemsetAsync(dStray.p, 0, n, stream) == hipSuccess;
    if (!fine)
        return -1;
    hipLaunchKernelGGL(k_strayChildren, blocksFor(n), dim3(256), 0, stream, dRows.p, dParent.p, n, dStray.p);
    hipLaunchKernelGGL(k_leafCentres, blocksFor(leaves.size()), dim3(256), 0, stream, dRows.p, dLeaves.p, (int)leaves.size(), dCentres.p);
    for (int d = 0; d <= deepest; ++d)
    {
        const int count = firstOfDepth[(size_t)d + 1] - firstOfDepth[d];
        if (count > 0)
            hipLaunchKernelGGL(k_pruneList, dim3((unsigned)count), dim3(256), 0, stream, dRows.p, dParent.p, dByDepth.p + firstOfDepth[d], count,
                               dLeavesBefore.p, dLeaves.p, n, threshold, sceneArea, dKeep.p, dAncestor.p, dStray.p, dCentres.p);
    }
    keepOut.assign(n, 1);
    if (hipMemcpyAsync(keepOut.data(), dKeep.p, n, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return -1;
    phase.mark("prune: decisions");
    int pruned = 0;
    for (char k : keepOut)
        pruned += k ? 0 : 1;
    return pruned;
}
