/*
 * solr_tree.hip - the box-grid build of GPUKernel::compactBoxes(true) on the device.
 *
 * Reference: solr/engines/GPUKernel.cpp:917-992 (processBoxes: every primitive's p0 hashed into a 6400^3
 * grid, key 1 + 1000 (X 6400^2 + Y 6400 + Z) in unsigned arithmetic), :741-839 (updateBoundingBox),
 * :994-1039 (processOutterBoxes: the boxes of the level below re-hashed by their centre on a coarser grid,
 * key X n^2 + Y n + Z + 1 in int arithmetic held as unsigned), :841-890 (updateOutterBoundingBox),
 * :1041-1083 (compactBoxes: depth from the primitive count, grid n, n/4, n/16 ...), :1085-1281
 * (recursiveDataStreamToGPU / streamDataToGPU: depth-first flattening, skip pointer = subtree size, the lamp
 * box first).  The reference does all of it with one std::map per level on one thread: 1.4-1.6 s for 100k
 * primitives, 0.2 s with this repository's host containers.
 *
 * The tree that comes out decides results (its order is the order of the primitive tests), so this is not "a"
 * spatial index built on the GPU but THAT tree, bit for bit: the same float divisions and truncations for the
 * keys, the maps' order reproduced by stable radix sorts of (key, position below), box bounds by the same
 * comparisons in the same order (one thread per box walks its members: the sign of a zero and the first of two
 * equal values are decided by that order), the depth-first positions by subtree sizes summed bottom-up and
 * offsets handed down.  tests/test_tree_build_gpu.py holds it to the host builder node for node.
 *
 * Declined (return -2, the caller builds on the host): no emissive primitive (the reference then treats the
 * first ordinary top-level box as the lamp box and streams its child KEYS as primitive indices), a top-level
 * child hashing to key 0 (the lamp box's), a lamp's index equal to a key of the level below the top (the
 * reference recurses into the lamp box as if it held child keys, GPUKernel.cpp:1252: that subtree is then
 * emitted twice), more boxes than NB_MAX_BOXES.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <malloc.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/solr_hip.h"
#include "device_pool.h"

namespace
{
const int MAX_LEVELS = 16;
const unsigned AABB_MAGIC_NUMBER = 6400u; /* GPUKernel.cpp:69 */

struct f3
{
    float x, y, z;
};

#define TREE_CHECK(call)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
        {                                                                                                              \
            snprintf(message, sizeof(message), "%s: %s", #call, hipGetErrorString(e_));                                \
            return false;                                                                                              \
        }                                                                                                              \
    } while (0)

char message[256] = "";

/* SOLR_HIP_DEBUG_TIMING: where a build spends its time (the marks wait for the stream) */
struct TreePhase
{
    const bool on = getenv("SOLR_HIP_DEBUG_TIMING") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void mark(const char *what, hipStream_t stream)
    {
        if (!on)
            return;
        (void)hipStreamSynchronize(stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "solr_tree: %-27s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

/* arrays of one build: from the scratch pool (device_pool.h) while it lasts */
template <class T>
struct Buffer
{
    T *ptr = nullptr;
    size_t count = 0;
    bool own = false;
    bool reserve(size_t n)
    {
        if (n <= count && ptr)
            return true;
        if (ptr && own)
            (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
        own = false;
        const size_t bytes = (n ? n : 1) * sizeof(T);
        ptr = (T *)solrScratchPool().take(bytes);
        if (!ptr)
        {
            if (hipMalloc((void **)&ptr, bytes) != hipSuccess)
                return false;
            own = true;
        }
        count = n ? n : 1;
        return true;
    }
    ~Buffer()
    {
        if (ptr && own)
            (void)hipFree(ptr);
    }
};

/* one level of the tree on the device: boxes in key order */
struct Level
{
    int nbBoxes = 0;
    Buffer<unsigned> key;  /* per box */
    Buffer<int> first;     /* per box + 1: its members in `member` */
    Buffer<int> member;    /* level 0: primitive ids; above: box positions in the level below */
    Buffer<f3> lo, hi, centre;
    Buffer<int> nodes;     /* nodes its subtree emits (0: an empty cell) */
    Buffer<int> prims;     /* primitives its subtree streams */
    Buffer<int> position;  /* where its node goes */
    Buffer<int> primStart; /* first flattened primitive of its subtree */
    Buffer<int> count;     /* level 0: non-emissive members */
};

/* std::min / std::max as the host uses them: the FIRST argument wins a tie (and keeps its zero's sign) */
__device__ inline float minStd(float a, float b) { return (b < a) ? b : a; }
__device__ inline float maxStd(float a, float b) { return (a < b) ? b : a; }

/* GPUKernel.cpp:938-941 */
__global__ void k_level0Keys(const Primitive *__restrict__ prims, int n, f3 minPos, f3 steps, unsigned *__restrict__ key,
                             int *__restrict__ id)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const vec3f c = prims[i].p0;
    const unsigned X = (unsigned)(int)((c.x - minPos.x) / steps.x);
    const unsigned Y = (unsigned)(int)((c.y - minPos.y) / steps.y);
    const unsigned Z = (unsigned)(int)((c.z - minPos.z) / steps.z);
    key[i] = 1u + 1000u * (X * AABB_MAGIC_NUMBER * AABB_MAGIC_NUMBER + Y * AABB_MAGIC_NUMBER + Z);
    id[i] = i;
}

/* GPUKernel.cpp:1011-1017; the key is held as unsigned by the map */
__global__ void k_outerKeys(const f3 *__restrict__ centre, int n, f3 minPos, f3 steps, int boxSize, unsigned *__restrict__ key,
                            int *__restrict__ id)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const f3 c = centre[i];
    const int X = (int)((c.x - minPos.x) / steps.x);
    const int Y = (int)((c.y - minPos.y) / steps.y);
    const int Z = (int)((c.z - minPos.z) / steps.z);
    const unsigned u = (unsigned)X * (unsigned)boxSize * (unsigned)boxSize + (unsigned)Y * (unsigned)boxSize + (unsigned)Z;
    key[i] = u + 1u;
    id[i] = i;
}

/* 1 where a new key starts in the sorted list */
__global__ void k_heads(const unsigned *__restrict__ sortedKey, int n, int *__restrict__ head)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        head[i] = (i == 0 || sortedKey[i] != sortedKey[i - 1]) ? 1 : 0;
}

/* box b = rank of its key; first[b] = where its members start */
__global__ void k_segments(const unsigned *__restrict__ sortedKey, const int *__restrict__ head, const int *__restrict__ rank,
                           int n, unsigned *__restrict__ key, int *__restrict__ first, int nbBoxes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    if (head[i])
    {
        key[rank[i]] = sortedKey[i];
        first[rank[i]] = i;
    }
    if (i == n - 1)
        first[nbBoxes] = n;
}

/* updateBoundingBox (GPUKernel.cpp:741-839) for one level-0 cell: its non-emissive members in id order */
__global__ void k_level0Bounds(const Primitive *__restrict__ prims, const unsigned char *__restrict__ emissive,
                               const int *__restrict__ first, const int *__restrict__ member, int nbBoxes, f3 *__restrict__ lo,
                               f3 *__restrict__ hi, f3 *__restrict__ centre, int *__restrict__ count)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbBoxes)
        return;
    f3 mn = {1000000.f, 1000000.f, 1000000.f}, mx = {-1000000.f, -1000000.f, -1000000.f};
    int kept = 0;
    for (int j = first[b]; j < first[b + 1]; ++j)
    {
        const int id = member[j];
        if (emissive[id])
            continue;
        ++kept;
        const Primitive p = prims[id];
        vec3f c0, c1;
        if (p.type == ptTriangle)
        {
            c0.x = minStd(minStd(p.p0.x, p.p1.x), p.p2.x), c0.y = minStd(minStd(p.p0.y, p.p1.y), p.p2.y),
            c0.z = minStd(minStd(p.p0.z, p.p1.z), p.p2.z);
            c1.x = maxStd(maxStd(p.p0.x, p.p1.x), p.p2.x), c1.y = maxStd(maxStd(p.p0.y, p.p1.y), p.p2.y),
            c1.z = maxStd(maxStd(p.p0.z, p.p1.z), p.p2.z);
        }
        else if (p.type == ptCylinder)
        {
            c0.x = minStd(p.p0.x, p.p1.x), c0.y = minStd(p.p0.y, p.p1.y), c0.z = minStd(p.p0.z, p.p1.z);
            c1.x = maxStd(p.p0.x, p.p1.x), c1.y = maxStd(p.p0.y, p.p1.y), c1.z = maxStd(p.p0.z, p.p1.z);
        }
        else
            c0 = c1 = p.p0;
        f3 p0 = {minStd(c0.x, c1.x), minStd(c0.y, c1.y), minStd(c0.z, c1.z)};
        f3 p1 = {(c0.x > c1.x) ? c0.x : c1.x, (c0.y > c1.y) ? c0.y : c1.y, (c0.z > c1.z) ? c0.z : c1.z};
        const bool round = p.type == ptCylinder || p.type == ptSphere || p.type == ptCone;
        const float sx = p.size.x, sy = round ? p.size.x : p.size.y, sz = round ? p.size.x : p.size.z;
        p0.x -= sx, p0.y -= sy, p0.z -= sz;
        p1.x += sx, p1.y += sy, p1.z += sz;
        if (p0.x < mn.x) mn.x = p0.x;
        if (p0.y < mn.y) mn.y = p0.y;
        if (p0.z < mn.z) mn.z = p0.z;
        if (p1.x > mx.x) mx.x = p1.x;
        if (p1.y > mx.y) mx.y = p1.y;
        if (p1.z > mx.z) mx.z = p1.z;
    }
    lo[b] = mn;
    hi[b] = mx;
    centre[b] = f3{(mn.x + mx.x) / 2.f, (mn.y + mx.y) / 2.f, (mn.z + mx.z) / 2.f};
    count[b] = kept;
}

/* updateOutterBoundingBox (GPUKernel.cpp:841-890) and the subtree sums, one thread per box of the level */
__global__ void k_outerBounds(const int *__restrict__ first, const int *__restrict__ member, int nbBoxes, float vd,
                              const f3 *__restrict__ childLo, const f3 *__restrict__ childHi, const int *__restrict__ childNodes,
                              const int *__restrict__ childPrims, f3 *__restrict__ lo, f3 *__restrict__ hi,
                              f3 *__restrict__ centre, int *__restrict__ nodes, int *__restrict__ prims)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbBoxes)
        return;
    f3 mn = {vd, vd, vd}, mx = {-vd, -vd, -vd};
    int n = 1, p = 0;
    for (int j = first[b]; j < first[b + 1]; ++j)
    {
        const int c = member[j];
        const f3 cl = childLo[c], ch = childHi[c];
        if (mn.x > cl.x) mn.x = cl.x;
        if (mn.y > cl.y) mn.y = cl.y;
        if (mn.z > cl.z) mn.z = cl.z;
        if (mx.x < ch.x) mx.x = ch.x;
        if (mx.y < ch.y) mx.y = ch.y;
        if (mx.z < ch.z) mx.z = ch.z;
        n += childNodes[c];
        p += childPrims[c];
    }
    lo[b] = mn;
    hi[b] = mx;
    centre[b] = f3{(mn.x + mx.x) / 2.f, (mn.y + mx.y) / 2.f, (mn.z + mx.z) / 2.f};
    nodes[b] = n;
    prims[b] = p;
}

__global__ void k_leafSizes(const int *__restrict__ count, int nbBoxes, int *__restrict__ nodes, int *__restrict__ prims)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nbBoxes)
    {
        nodes[b] = count[b] > 0 ? 1 : 0; /* a cell without entries is not emitted (GPUKernel.cpp:1096) */
        prims[b] = count[b];
    }
}

/* top level: positions after the lamp box (node 0, primitives 0..nbLamps-1) */
__global__ void k_topOffsets(const int *__restrict__ nodesScan, const int *__restrict__ primsScan, int nbBoxes, int nbLamps,
                             int *__restrict__ position, int *__restrict__ primStart)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nbBoxes)
    {
        position[b] = 1 + nodesScan[b];
        primStart[b] = nbLamps + primsScan[b];
    }
}

/* hand a box's position down to its children: each child goes behind its parent and its earlier siblings' subtrees */
__global__ void k_childOffsets(const int *__restrict__ first, const int *__restrict__ member, int nbBoxes,
                               const int *__restrict__ position, const int *__restrict__ primStart,
                               const int *__restrict__ childNodes, const int *__restrict__ childPrims,
                               int *__restrict__ childPosition, int *__restrict__ childPrimStart)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbBoxes)
        return;
    int at = position[b] + 1, p = primStart[b];
    for (int j = first[b]; j < first[b + 1]; ++j)
    {
        const int c = member[j];
        childPosition[c] = at;
        childPrimStart[c] = p;
        at += childNodes[c];
        p += childPrims[c];
    }
}

/* write the nodes of one level (GPUKernel.cpp:1099-1103, 1143) */
__global__ void k_emitNodes(int depth, int nbBoxes, const f3 *__restrict__ lo, const f3 *__restrict__ hi,
                            const int *__restrict__ nodes, const int *__restrict__ position, const int *__restrict__ primStart,
                            const int *__restrict__ count, const int *__restrict__ first, const int *__restrict__ member,
                            const unsigned char *__restrict__ emissive, BoundingBox *__restrict__ out, int *__restrict__ order)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbBoxes || nodes[b] == 0)
        return;
    BoundingBox box;
    memset(&box, 0, sizeof(box));
    box.parameters[0].x = lo[b].x, box.parameters[0].y = lo[b].y, box.parameters[0].z = lo[b].z;
    box.parameters[1].x = hi[b].x, box.parameters[1].y = hi[b].y, box.parameters[1].z = hi[b].z;
    if (depth == 0)
    {
        box.nbPrimitives = count[b];
        box.startIndex = primStart[b];
        box.indexForNextBox.x = 1;
        int at = primStart[b];
        for (int j = first[b]; j < first[b + 1]; ++j)
            if (!emissive[member[j]])
                order[at++] = member[j];
    }
    else
    {
        box.nbPrimitives = 0;
        box.startIndex = depth;
        box.indexForNextBox.x = nodes[b];
    }
    out[position[b]] = box;
}

/* the lamp box: node 0, the emissive primitives in index order (GPUKernel.cpp:1181-1240) */
__global__ void k_lampFlags(const unsigned char *__restrict__ emissive, int n, int *__restrict__ flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        flag[i] = emissive[i] ? 1 : 0;
}
__global__ void k_lampOrder(const int *__restrict__ flag, const int *__restrict__ scan, int n, int *__restrict__ order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i])
        order[scan[i]] = i;
}

/* does any lamp's index equal a key of the level below the top (sorted)? */
__global__ void k_lampAliases(const int *__restrict__ order, int nbLamps, const unsigned *__restrict__ keys, int nbKeys,
                              int *__restrict__ found)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbLamps)
        return;
    const unsigned want = (unsigned)order[i];
    int a = 0, b = nbKeys;
    while (a < b)
    {
        const int m = (a + b) / 2;
        if (keys[m] < want)
            a = m + 1;
        else
            b = m;
    }
    if (a < nbKeys && keys[a] == want)
        atomicExch(found, 1);
}

inline dim3 blocks(int n) { return dim3((unsigned)((n + 255) / 256)); }

struct Sorter
{
    Buffer<unsigned char> temp;
    Buffer<unsigned> keyOut;
    Buffer<int> idOut, head, rank;
    /* (key, id) sorted by key, ties in input order; returns the number of distinct keys or -1 */
    int run(Buffer<unsigned> &key, Buffer<int> &id, int n, hipStream_t stream)
    {
        if (!keyOut.reserve(n) || !idOut.reserve(n) || !head.reserve(n) || !rank.reserve(n))
            return -1;
        size_t bytes = 0;
        if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, key.ptr, keyOut.ptr, id.ptr, idOut.ptr, n, 0, 32, stream) != hipSuccess)
            return -1;
        size_t scanBytes = 0;
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, head.ptr, rank.ptr, n, stream) != hipSuccess)
            return -1;
        if (!temp.reserve(bytes > scanBytes ? bytes : scanBytes))
            return -1;
        size_t tb = temp.count;
        if (hipcub::DeviceRadixSort::SortPairs(temp.ptr, tb, key.ptr, keyOut.ptr, id.ptr, idOut.ptr, n, 0, 32, stream) != hipSuccess)
            return -1;
        hipLaunchKernelGGL(k_heads, blocks(n), dim3(256), 0, stream, (const unsigned *)keyOut.ptr, n, head.ptr);
        tb = temp.count;
        if (hipcub::DeviceScan::ExclusiveSum(temp.ptr, tb, head.ptr, rank.ptr, n, stream) != hipSuccess)
            return -1;
        int lastRank = 0, lastHead = 0;
        if (hipMemcpyAsync(&lastRank, rank.ptr + (n - 1), sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipMemcpyAsync(&lastHead, head.ptr + (n - 1), sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess)
            return -1;
        return lastRank + lastHead;
    }
};

bool exclusiveScan(Buffer<unsigned char> &temp, const int *in, int *out, int n, hipStream_t stream)
{
    size_t bytes = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, n, stream) != hipSuccess || !temp.reserve(bytes))
        return false;
    size_t tb = temp.count;
    return hipcub::DeviceScan::ExclusiveSum(temp.ptr, tb, in, out, n, stream) == hipSuccess;
}
} // namespace

void solrTuneHostAllocator()
{
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("SOLR_HIP_MALLOC_DEFAULTS"))
            return;
        (void)mallopt(M_MMAP_THRESHOLD, 256 << 20);
        (void)mallopt(M_TRIM_THRESHOLD, 512 << 20);
        (void)mallopt(M_TOP_PAD, 64 << 20);
    });
}

SolrScratchPool &solrScratchPool()
{
    static SolrScratchPool pool;
    return pool;
}

extern "C" const char *solr_hip_build_tree_message(void)
{
    return message;
}

/* See include/solr_hip.h.  Returns the tree depth (>= 1), -1 on an error (solr_hip_build_tree_message), -2 when
 * the scene is one of the cases left to the host builder. */
extern "C" int solr_hip_build_tree(const Primitive *primitives, const unsigned char *emissive, int nbPrimitives,
                                   const float minPos[3], const float maxPos[3], float viewDistance, BoundingBox *boxes,
                                   int boxCapacity, int *order, int *nbBoxesOut, int *nbLampsOut)
{
    message[0] = 0;
    solrTuneHostAllocator();
    if (!primitives || !emissive || nbPrimitives <= 0 || !minPos || !maxPos || !boxes || !order || !nbBoxesOut || !nbLampsOut)
    {
        snprintf(message, sizeof(message), "solr_hip_build_tree: bad arguments");
        return -1;
    }
    const int n = nbPrimitives;
    hipStream_t stream = nullptr;
    /* on the engine's device, whichever thread calls (one process per GPU: a tree built on the calling thread's
     * current device - GPU 0 in a fresh thread - would put every rank's build, and a context, there); the caller's
     * current device is restored on the way out */
    struct DeviceScope
    {
        int before = -1;
        DeviceScope()
        {
            if (hipGetDevice(&before) != hipSuccess)
                before = -1;
            (void)hipSetDevice(solr_hip_get_device());
        }
        ~DeviceScope()
        {
            if (before >= 0)
                (void)hipSetDevice(before);
        }
    } deviceScope;

    /* depth and grid sizes, GPUKernel.cpp:1056-1076 */
    int depth = 0;
    std::vector<int> gridSize; /* gridSize[d] for d = 1..depth */
    gridSize.push_back(0);
    {
        int nb = n;
        do
        {
            ++depth;
            gridSize.push_back(nb);
            nb /= 4;
        } while (nb > 2);
    }
    if (depth >= MAX_LEVELS)
    {
        snprintf(message, sizeof(message), "solr_hip_build_tree: %d levels", depth);
        return -1;
    }
    auto stepsFor = [&](int boxSize) {
        f3 s;
        s.x = (maxPos[0] - minPos[0]) / boxSize;
        s.y = (maxPos[1] - minPos[1]) / boxSize;
        s.z = (maxPos[2] - minPos[2]) / boxSize;
        s.x = (s.x == 0.f) ? 1 : s.x;
        s.y = (s.y == 0.f) ? 1 : s.y;
        s.z = (s.z == 0.f) ? 1 : s.z;
        return s;
    };
    const f3 mn = {minPos[0], minPos[1], minPos[2]};

    auto body = [&]() -> bool {
        TreePhase phase;
        Buffer<Primitive> dPrims;
        Buffer<unsigned char> dEmissive, temp;
        Buffer<unsigned> key;
        Buffer<int> id, lampFlag, lampScan, scanA, scanB, dOrder, found;
        Buffer<BoundingBox> dBoxes;
        Sorter sorter;
        std::vector<Level> level(depth + 1);
        if (!dPrims.reserve(n) || !dEmissive.reserve(n) || !key.reserve(n) || !id.reserve(n) || !lampFlag.reserve(n) ||
            !lampScan.reserve(n) || !dOrder.reserve(n) || !found.reserve(1))
        {
            snprintf(message, sizeof(message), "solr_hip_build_tree: out of device memory");
            return false;
        }
        TREE_CHECK(hipMemcpyAsync(dPrims.ptr, primitives, (size_t)n * sizeof(Primitive), hipMemcpyHostToDevice, stream));
        TREE_CHECK(hipMemcpyAsync(dEmissive.ptr, emissive, (size_t)n, hipMemcpyHostToDevice, stream));
        phase.mark("primitives uploaded", stream);

        /* lamps, in index order */
        hipLaunchKernelGGL(k_lampFlags, blocks(n), dim3(256), 0, stream, (const unsigned char *)dEmissive.ptr, n, lampFlag.ptr);
        if (!exclusiveScan(temp, lampFlag.ptr, lampScan.ptr, n, stream))
            return false;
        int lastScan = 0, lastFlag = 0;
        TREE_CHECK(hipMemcpyAsync(&lastScan, lampScan.ptr + (n - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
        TREE_CHECK(hipMemcpyAsync(&lastFlag, lampFlag.ptr + (n - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
        TREE_CHECK(hipStreamSynchronize(stream));
        const int nbLamps = lastScan + lastFlag;
        *nbLampsOut = nbLamps;
        if (nbLamps == 0)
            return true; /* declined below */
        hipLaunchKernelGGL(k_lampOrder, blocks(n), dim3(256), 0, stream, (const int *)lampFlag.ptr, (const int *)lampScan.ptr, n,
                           dOrder.ptr);

        /* level 0 */
        hipLaunchKernelGGL(k_level0Keys, blocks(n), dim3(256), 0, stream, (const Primitive *)dPrims.ptr, n, mn,
                           stepsFor((int)AABB_MAGIC_NUMBER), key.ptr, id.ptr);
        {
            Level &L = level[0];
            const int nb = sorter.run(key, id, n, stream);
            if (nb <= 0)
            {
                snprintf(message, sizeof(message), "solr_hip_build_tree: sort failed");
                return false;
            }
            L.nbBoxes = nb;
            if (!L.key.reserve(nb) || !L.first.reserve(nb + 1) || !L.member.reserve(n) || !L.lo.reserve(nb) || !L.hi.reserve(nb) ||
                !L.centre.reserve(nb) || !L.nodes.reserve(nb) || !L.prims.reserve(nb) || !L.position.reserve(nb) ||
                !L.primStart.reserve(nb) || !L.count.reserve(nb))
                return false;
            hipLaunchKernelGGL(k_segments, blocks(n), dim3(256), 0, stream, (const unsigned *)sorter.keyOut.ptr,
                               (const int *)sorter.head.ptr, (const int *)sorter.rank.ptr, n, L.key.ptr, L.first.ptr, nb);
            TREE_CHECK(hipMemcpyAsync(L.member.ptr, sorter.idOut.ptr, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, stream));
            hipLaunchKernelGGL(k_level0Bounds, blocks(nb), dim3(256), 0, stream, (const Primitive *)dPrims.ptr,
                               (const unsigned char *)dEmissive.ptr, (const int *)L.first.ptr, (const int *)L.member.ptr, nb,
                               L.lo.ptr, L.hi.ptr, L.centre.ptr, L.count.ptr);
            hipLaunchKernelGGL(k_leafSizes, blocks(nb), dim3(256), 0, stream, (const int *)L.count.ptr, nb, L.nodes.ptr, L.prims.ptr);
        }
        phase.mark("lamps, level 0", stream);
        /* outer levels */
        for (int d = 1; d <= depth; ++d)
        {
            Level &B = level[d - 1], &L = level[d];
            const int m = B.nbBoxes;
            if (!key.reserve(m) || !id.reserve(m))
                return false;
            hipLaunchKernelGGL(k_outerKeys, blocks(m), dim3(256), 0, stream, (const f3 *)B.centre.ptr, m, mn, stepsFor(gridSize[d]),
                               gridSize[d], key.ptr, id.ptr);
            const int nb = sorter.run(key, id, m, stream);
            if (nb <= 0)
                return false;
            L.nbBoxes = nb;
            if (!L.key.reserve(nb) || !L.first.reserve(nb + 1) || !L.member.reserve(m) || !L.lo.reserve(nb) || !L.hi.reserve(nb) ||
                !L.centre.reserve(nb) || !L.nodes.reserve(nb) || !L.prims.reserve(nb) || !L.position.reserve(nb) ||
                !L.primStart.reserve(nb))
                return false;
            hipLaunchKernelGGL(k_segments, blocks(m), dim3(256), 0, stream, (const unsigned *)sorter.keyOut.ptr,
                               (const int *)sorter.head.ptr, (const int *)sorter.rank.ptr, m, L.key.ptr, L.first.ptr, nb);
            TREE_CHECK(hipMemcpyAsync(L.member.ptr, sorter.idOut.ptr, (size_t)m * sizeof(int), hipMemcpyDeviceToDevice, stream));
            hipLaunchKernelGGL(k_outerBounds, blocks(nb), dim3(256), 0, stream, (const int *)L.first.ptr, (const int *)L.member.ptr, nb,
                               viewDistance, (const f3 *)B.lo.ptr, (const f3 *)B.hi.ptr, (const int *)B.nodes.ptr,
                               (const int *)B.prims.ptr, L.lo.ptr, L.hi.ptr, L.centre.ptr, L.nodes.ptr, L.prims.ptr);
        }
        phase.mark("outer levels", stream);
        /* the cases left to the host */
        Level &T = level[depth];
        unsigned firstKey = 1;
        TREE_CHECK(hipMemcpyAsync(&firstKey, T.key.ptr, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
        TREE_CHECK(hipMemsetAsync(found.ptr, 0, sizeof(int), stream));
        hipLaunchKernelGGL(k_lampAliases, blocks(nbLamps), dim3(256), 0, stream, (const int *)dOrder.ptr, nbLamps,
                           (const unsigned *)level[depth - 1].key.ptr, level[depth - 1].nbBoxes, found.ptr);
        int aliased = 0;
        TREE_CHECK(hipMemcpyAsync(&aliased, found.ptr, sizeof(int), hipMemcpyDeviceToHost, stream));
        TREE_CHECK(hipStreamSynchronize(stream));
        if (firstKey == 0u || aliased)
        {
            *nbLampsOut = -1; /* declined */
            return true;
        }
        /* positions: the top level behind the lamp box, then down */
        if (!scanA.reserve(T.nbBoxes) || !scanB.reserve(T.nbBoxes))
            return false;
        if (!exclusiveScan(temp, T.nodes.ptr, scanA.ptr, T.nbBoxes, stream) || !exclusiveScan(temp, T.prims.ptr, scanB.ptr, T.nbBoxes, stream))
            return false;
        hipLaunchKernelGGL(k_topOffsets, blocks(T.nbBoxes), dim3(256), 0, stream, (const int *)scanA.ptr, (const int *)scanB.ptr,
                           T.nbBoxes, nbLamps, T.position.ptr, T.primStart.ptr);
        int lastNodes = 0, lastScanNodes = 0;
        TREE_CHECK(hipMemcpyAsync(&lastNodes, T.nodes.ptr + (T.nbBoxes - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
        TREE_CHECK(hipMemcpyAsync(&lastScanNodes, scanA.ptr + (T.nbBoxes - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
        TREE_CHECK(hipStreamSynchronize(stream));
        const int totalNodes = 1 + lastScanNodes + lastNodes;
        *nbBoxesOut = totalNodes;
        if (totalNodes > boxCapacity || totalNodes >= NB_MAX_BOXES)
        {
            *nbLampsOut = -1;
            return true;
        }
        if (!dBoxes.reserve(totalNodes))
            return false;
        for (int d = depth; d >= 1; --d)
        {
            Level &L = level[d], &B = level[d - 1];
            hipLaunchKernelGGL(k_childOffsets, blocks(L.nbBoxes), dim3(256), 0, stream, (const int *)L.first.ptr,
                               (const int *)L.member.ptr, L.nbBoxes, (const int *)L.position.ptr, (const int *)L.primStart.ptr,
                               (const int *)B.nodes.ptr, (const int *)B.prims.ptr, B.position.ptr, B.primStart.ptr);
        }
        for (int d = depth; d >= 0; --d)
        {
            Level &L = level[d];
            hipLaunchKernelGGL(k_emitNodes, blocks(L.nbBoxes), dim3(256), 0, stream, d, L.nbBoxes, (const f3 *)L.lo.ptr,
                               (const f3 *)L.hi.ptr, (const int *)L.nodes.ptr, (const int *)L.position.ptr,
                               (const int *)L.primStart.ptr, (const int *)(d == 0 ? L.count.ptr : nullptr),
                               (const int *)L.first.ptr, (const int *)L.member.ptr, (const unsigned char *)dEmissive.ptr,
                               dBoxes.ptr, dOrder.ptr);
        }
        TREE_CHECK(hipGetLastError());
        phase.mark("positions, nodes emitted", stream);
        TREE_CHECK(hipMemcpyAsync(boxes, dBoxes.ptr, (size_t)totalNodes * sizeof(BoundingBox), hipMemcpyDeviceToHost, stream));
        TREE_CHECK(hipMemcpyAsync(order, dOrder.ptr, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, stream));
        TREE_CHECK(hipStreamSynchronize(stream));
        phase.mark("nodes and order copied back", stream);
        /* node 0: the lamp box (GPUKernel.cpp:1181-1190) */
        memset(&boxes[0], 0, sizeof(BoundingBox));
        boxes[0].parameters[0].x = boxes[0].parameters[0].y = boxes[0].parameters[0].z = -viewDistance;
        boxes[0].parameters[1].x = boxes[0].parameters[1].y = boxes[0].parameters[1].z = viewDistance;
        boxes[0].nbPrimitives = nbLamps;
        boxes[0].startIndex = 0;
        boxes[0].indexForNextBox.x = 1;
        return true;
    };
    *nbLampsOut = 0;
    *nbBoxesOut = 0;
    /* the primitive records, five arrays per primitive, eleven per box of a level (the levels shrink by four or so
     * each), the sorter's: what a first build asks for; later ones know (device_pool.h) */
    SolrScratchPool &pool = solrScratchPool();
    std::lock_guard<std::mutex> oneBuild(pool.busy);
    {
        TreePhase phase;
        pool.begin((size_t)n * (sizeof(Primitive) + 160 + sizeof(BoundingBox) * 4) + ((size_t)8 << 20));
        phase.mark("scratch pool", stream);
    }
    const bool built = body();
    pool.end();
    if (!built)
    {
        if (!message[0])
            snprintf(message, sizeof(message), "solr_hip_build_tree: device allocation or launch failed");
        return -1;
    }
    if (*nbLampsOut <= 0)
        return -2;
    return depth;
}
