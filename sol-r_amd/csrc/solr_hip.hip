/*
 * solr_hip.hip - kernels and the C-ABI of the MI355X rendering engine
 * (include/solr_hip.h).  gfx950 only.
 *
 * Kernels
 *   k_standardRenderer   one wave = one 8x8 pixel tile; primary-ray setup
 *                        (CudaRayTracer.cu:437-563), the bounce loop, and - when
 *                        no neighbourhood post-process is requested - the
 *                        float->RGB8 conversion of k_default fused in
 *                        (CudaRayTracer.cu:1057-1073, GeometryShaders.cuh:132-165)
 *   k_default            stand-alone conversion (bound bitmap / non-fused path)
 *   k_ambientOcclusion   CudaRayTracer.cu:1128-1181
 *   k_depthOfField       CudaRayTracer.cu:1081-1120
 *
 * Host layer: device memory, AoS -> plane re-packing, launches, timing.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <functional>
#include <chrono>
#include <vector>

#include "../../include/solr_hip.h"
#include "rt_device.h"
#include "device_pool.h"
#include "lists_device.h"

using namespace solrdev;

/* ======================================================================= */
/* Kernels                                                                  */
/* ======================================================================= */

#include "renderer.h"

/* the renderer's instantiations live in the files under csrc/rows (one object per row of renderImpl's table) */
namespace solrrows
{
RendererFn renderer(int count, int features, bool volume)
{
    if (volume || (features & ~(F_DEEP | F_STACK)) == F_ALL)
        return (features & F_STACK) ? nullptr : everything(count, features, volume);
    RendererFn fn = nullptr;
    if (!(fn = spherePlane(count, features)) && !(fn = sphereTriangle(count, features)) && !(fn = sphereCylinder(count, features)) &&
        !(fn = untexturedMix(count, features)) && !(fn = textured(count, features)))
        fn = specialCameras(count, features);
    return fn;
}
WalkBoundFn walkBound(int row, int features)
{
    switch (row)
    {
    case 0: return walkBoundRow0(features);
    case 1: return walkBoundRow1(features);
    case 2: return walkBoundRow2(features);
    case 3: return walkBoundRow3(features);
    default: return nullptr;
    }
}
} // namespace solrrows

/* CRT:1057-1073 */
__global__ __launch_bounds__(256) void k_default(const SceneInfo si, int nbPixels,
                                                 const PixelRecord *__restrict__ pp,
                                                 unsigned char *__restrict__ bitmap)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= nbPixels)
        return;
    float4 c4 = pp[index].colorInfo;
    v3 c = V(c4.x, c4.y, c4.z);
    if (si.pathTracingIteration > NB_MAX_ITERATIONS)
    {
        float d = (float)(si.pathTracingIteration - NB_MAX_ITERATIONS + 1);
        c.x /= d;
        c.y /= d;
        c.z /= d;
    }
    makeColor(si, c, bitmap, index);
}

/* CRT:1189-1228; gathers stay inside this process's strip */
__global__ __launch_bounds__(256) void k_radiosity(const SceneInfo si, const PostProcessingInfo ppi, int nbRows,
                                                   const PixelRecord *__restrict__ pp, const int4 *__restrict__ ids,
                                                   const float *__restrict__ randoms, long nbRandoms,
                                                   unsigned char *__restrict__ bitmap)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = si.size.x;
    const int wh = W * nbRows;
    if (index >= wh)
        return;
    const int x = index % W;
    const int y = index / W;
    const int div = (si.pathTracingIteration > NB_MAX_ITERATIONS) ? (si.pathTracingIteration - NB_MAX_ITERATIONS + 1) : 1;
    const float4 own = pp[index].colorInfo;
    v3 local = V(0.f, 0.f, 0.f);
    for (int i = 0; i < ppi.param3; ++i)
    {
        const int ix = (i + si.pathTracingIteration) % wh;
        const int iy = (i + 100 + si.pathTracingIteration) % wh;
        const float rx = (ix >= 0 && ix < nbRandoms) ? randoms[ix] : 0.f;
        const float ry = (iy >= 0 && iy < nbRandoms) ? randoms[iy] : 0.f;
        const int xx = (int)((float)x + rx * ppi.param2);
        const int yy = (int)((float)y + ry * ppi.param2);
        local.x += own.x;
        local.y += own.y;
        local.z += own.z;
        if (xx >= 0 && xx < W && yy >= 0 && yy < nbRows)
        {
            const int localIndex = yy * W + xx;
            const float4 light = pp[localIndex].colorInfo;
            const float w = (float)ids[localIndex].z;
            local.x += light.x * w / 256.f;
            local.y += light.y * w / 256.f;
            local.z += light.z * w / 256.f;
        }
    }
    local.x /= (float)ppi.param3;
    local.y /= (float)ppi.param3;
    local.z /= (float)ppi.param3;
    local.x /= (float)div;
    local.y /= (float)div;
    local.z /= (float)div;
    saturate3(local);
    makeColor(si, local, bitmap, index);
}

/* CRT:1236-1333: six convolution filters selected by param3, wrapping around the strip */
__device__ const int FILTER_SIZE[6][2] = {{3, 3}, {5, 5}, {3, 3}, {3, 3}, {5, 5}, {5, 5}};
__device__ const float FILTER_FACTORS[6][2] = {{1.f, 128.f}, {1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}, {0.2f, 0.f}, {0.125f, 0.f}};
__device__ const float FILTER_INFO[6][5][5] = {
    {{-1.f, -1.f, 0.f, 0.f, 0.f}, {-1.f, 0.f, 1.f, 0.f, 0.f}, {0.f, 1.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}},
    {{0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {-1.f, -1.f, 2.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}},
    {{-1.f, -1.f, -1.f, 0.f, 0.f}, {-1.f, 9.f, -1.f, 0.f, 0.f}, {-1.f, -1.f, -1.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}},
    {{0.f, 0.2f, 0.f, 0.f, 0.f}, {0.2f, 0.2f, 0.2f, 0.f, 0.f}, {0.f, 0.2f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}},
    {{1.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 1.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 1.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 1.f}},
    {{-1.f, -1.f, -1.f, -1.f, -1.f}, {-1.f, 2.f, 2.f, 2.f, -1.f}, {-1.f, 2.f, 8.f, 2.f, -1.f}, {-1.f, 2.f, 2.f, 2.f, -1.f}, {-1.f, -1.f, -1.f, -1.f, -1.f}}};

__global__ __launch_bounds__(256) void k_filter(const SceneInfo si, const PostProcessingInfo ppi, int nbRows,
                                                const PixelRecord *__restrict__ pp, unsigned char *__restrict__ bitmap)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = si.size.x;
    if (index >= W * nbRows)
        return;
    const int x = index % W;
    const int y = index / W;
    v3 local = V(0.f, 0.f, 0.f);
    v3 color = V(0.f, 0.f, 0.f);
    const int f = ppi.param3;
    if (f >= 0 && f < 6)
    {
        for (int filterX = 0; filterX < FILTER_SIZE[f][0]; filterX++)
            for (int filterY = 0; filterY < FILTER_SIZE[f][1]; filterY++)
            {
                const int imageX = (x - FILTER_SIZE[f][0] / 2 + filterX + W) % W;
                const int imageY = (y - FILTER_SIZE[f][1] / 2 + filterY + nbRows) % nbRows;
                const float4 p = pp[imageY * W + imageX].colorInfo;
                v3 c = V(p.x, p.y, p.z);
                if (si.pathTracingIteration > NB_MAX_ITERATIONS)
                {
                    const float d = (float)(si.pathTracingIteration - NB_MAX_ITERATIONS + 1);
                    c.x /= d;
                    c.y /= d;
                    c.z /= d;
                }
                local.x += c.x * FILTER_INFO[f][filterX][filterY];
                local.y += c.y * FILTER_INFO[f][filterX][filterY];
                local.z += c.z * FILTER_INFO[f][filterX][filterY];
            }
        color.x += fminf(fmaxf(FILTER_FACTORS[f][0] * local.x + FILTER_FACTORS[f][1] / 255.f, 0.f), 1.f);
        color.y += fminf(fmaxf(FILTER_FACTORS[f][0] * local.y + FILTER_FACTORS[f][1] / 255.f, 0.f), 1.f);
        color.z += fminf(fmaxf(FILTER_FACTORS[f][0] * local.z + FILTER_FACTORS[f][1] / 255.f, 0.f), 1.f);
    }
    saturate3(color);
    makeColor(si, color, bitmap, index);
}

/* CRT:1341-1358: depth shown as grey */
__global__ __launch_bounds__(256) void k_cartoon(const SceneInfo si, const PostProcessingInfo ppi, int nbRows,
                                                 const PixelRecord *__restrict__ pp, unsigned char *__restrict__ bitmap)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= si.size.x * nbRows)
        return;
    const float depth = si.viewDistance / fabsf(pp[index].colorInfo.w - ppi.param1);
    v3 color = V(depth, depth, depth);
    saturate3(color);
    makeColor(si, color, bitmap, index);
}

/* TileScheduling.  A frame is tens of thousands of one-wave workgroups whose costs differ by an
 * order of magnitude (a tile of sky against a tile of mesh seen at a grazing angle) and the
 * dispatcher hands them out in launch order, so an expensive tile that happens to be launched late
 * runs on alone while the rest of the chip idles (profiles/r1/tile_timeline_*.txt: 23 % of the
 * 100k-triangle frame).  Consecutive frames of a renderer see nearly the same picture: every wave
 * records what its tile cost (one store); this kernel - one workgroup - reduces the costs to their
 * maximum and sum for the host (every sixteenth frame) and, when the
 * host has seen a heavy tail (max > 2 x mean), sorts the tiles by cost with a counting sort in LDS
 * (64 cost classes) so that the following frames are launched most-expensive-first; the order is
 * refreshed every sixteenth frame.  Only the order of work changes,
 * never a result.  (Per-wave atomics for max / sum were tried first: 32 400 same-address device-scope
 * atomics per frame serialise at the memory side and tripled the frame time.) */
__device__ unsigned orderSerial = 0u;

/* Frames in flight: the frame on the other stream may still be storing its tiles' costs while this kernel
 * runs.  Every cost is therefore read from `cost` exactly ONCE, into `snapshot` (private to the sort, written
 * and read by this workgroup only); maximum, histogram and scatter all work on that one stable copy, so the
 * histogram and the scatter agree and `order` is a permutation of 0..n-1 whatever is being stored meanwhile. */
/* The tiles that are the frame's critical path (criterion below), at most SPLIT_TILES_MAX of them, are launched
 * as four quadrant waves each, first of all: a frame is as long as its longest wave (the 100k-triangle mesh: one tile
 * seen at a grazing angle took the whole 0.78 ms of the frame), and a 4 x 4 quadrant of such a tile takes
 * about 0.6 of the tile's time.  `order` therefore holds n + 3 * SPLIT_TILES_MAX entries: 4 per split tile,
 * one per other tile, ORDER_NOTHING to the end. */
__global__ __launch_bounds__(1024) void k_orderTiles(const unsigned *cost, unsigned *__restrict__ snapshot,
                                                      unsigned *__restrict__ order, int n,
                                                      volatile unsigned *hostStats, int sort)
{
    __shared__ unsigned nbSplit;
    __shared__ unsigned splitClass;
    __shared__ unsigned bins[1024];
    __shared__ unsigned scan[1024];
    __shared__ unsigned maxCost;
    __shared__ unsigned long long sumCost;
    const int t = threadIdx.x;
    const int BATCH = 8; /* independent loads in flight per thread: the passes are latency bound */
    bins[t] = 0u;
    if (t == 0)
    {
        maxCost = 0u;
        sumCost = 0ull;
    }
    __syncthreads();
    unsigned m = 0u;
    unsigned long long sum = 0ull;
    for (int base = 0; base < n; base += BATCH * 1024)
    {
        unsigned c[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
        {
            const int i = base + k * 1024 + t;
            c[k] = (i < n) ? __builtin_nontemporal_load(&cost[i]) : 0u;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
        {
            const int i = base + k * 1024 + t;
            if (sort && i < n)
                snapshot[i] = c[k]; /* re-read below by the thread that wrote it */
            m = max(m, c[k]);
            sum += c[k];
        }
    }
    atomicMax(&maxCost, m);
    atomicAdd(&sumCost, sum);
    __syncthreads();
    if (t == 0) /* {max, sum lo, sum hi, tiles, serial}: the host reads them without synchronising */
    {
        hostStats[0] = maxCost;
        hostStats[1] = (unsigned)sumCost;
        hostStats[2] = (unsigned)(sumCost >> 32);
        hostStats[3] = (unsigned)n;
        hostStats[4] = ++orderSerial;
    }
    if (!sort)
        return;
    /* 64 cost classes x 16 sub-bins picked by the tile index: tiles of similar cost are the common
     * case and would otherwise all contend for one LDS counter */
    const float toClass = 64.f / ((float)maxCost + 1.f); /* the same expression in both passes */
    for (int base = 0; base < n; base += BATCH * 1024)
    {
        unsigned c[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
        {
            const int i = base + k * 1024 + t;
            c[k] = (i < n) ? snapshot[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
        {
            const int i = base + k * 1024 + t;
            if (i < n)
                atomicAdd(&bins[(min(63u, (unsigned)((float)c[k] * toClass)) << 4) | ((unsigned)i & 15u)], 1u);
        }
    }
    __syncthreads();
    /* exclusive prefix over bins in DESCENDING bin order (Hillis-Steele on the reversed array) */
    const unsigned mine = bins[1023 - t];
    scan[t] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1)
    {
        const unsigned add = (t >= off) ? scan[t - off] : 0u;
        __syncthreads();
        scan[t] += add;
        __syncthreads();
    }
    bins[1023 - t] = scan[t] - mine; /* first slot of this bin */
    /* which classes are split: those above the class of twice the mean cost, as far down as SPLIT_TILES_MAX
     * tiles go (whole classes only: the split tiles are then a prefix of the order) */
    if (t == 0)
    {
        nbSplit = 0u;
        splitClass = 64u;
    }
    __syncthreads();
    if (t < 64)
    {
        /* tiles in classes >= c = inclusive scan at the end of class c in descending order: bin (c << 4) is the
         * last of class c's sixteen sub-bins there, scan[1023 - (c << 4)] counts everything up to and including it */
        /* worth splitting: a tile that alone takes more than 0.8 of what the whole frame would take if its
         * work were spread evenly over the chip's 4096 wave slots (256 CUs x 16 resident waves of this
         * kernel) - such a tile IS the frame's critical path - and more than twice the mean.  A frame whose
         * longest tile is short against that (Cornell: 0.1 ms of 0.35; the molecule: 0.5 of 1.0) is bound by
         * throughput, and there the 3.2 x work of four quadrant waves is a loss.  Frames in flight (sort = how
         * many) hide a critical path behind the next frames: the bar is that many times higher - which no tile
         * of a whole 1080p frame passes, but the horizon tile of the mesh in a 1/8 strip does (a rank of an
         * eight-GPU frame: its strip is as slow as that one wave however many frames overlap). */
        const unsigned c = (unsigned)t;
        const float mean = (float)sumCost / (float)max(n, 1);
        const float critical = fmaxf(2.f * mean, (float)sort * (float)sumCost / 5120.f);
        const unsigned above = min(63u, (unsigned)(critical * toClass)) + 1u;
        const unsigned upTo = scan[1023 - (c << 4)];
        if (c >= above && c < 64u && upTo <= (unsigned)SPLIT_TILES_MAX)
            atomicMin(&splitClass, c);
    }
    __syncthreads();
    if (t == 0)
        nbSplit = splitClass < 64u ? scan[1023 - (splitClass << 4)] : 0u;
    __syncthreads();
    const unsigned split = nbSplit;
    for (int i = n + (SPLIT_PARTS - 1) * (int)split + t; i < n + (SPLIT_PARTS - 1) * SPLIT_TILES_MAX; i += 1024)
        order[i] = ORDER_NOTHING;
    for (int base = 0; base < n; base += BATCH * 1024)
    {
        unsigned c[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
        {
            const int i = base + k * 1024 + t;
            c[k] = (i < n) ? snapshot[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
        {
            const int i = base + k * 1024 + t;
            if (i < n)
            {
                const unsigned b = (min(63u, (unsigned)((float)c[k] * toClass)) << 4) | ((unsigned)i & 15u);
                const unsigned at = atomicAdd(&bins[b], 1u); /* position in descending order of cost */
                if (at < split)
                    for (unsigned q = 0; q < (unsigned)SPLIT_PARTS; ++q)
                        order[(unsigned)SPLIT_PARTS * at + q] = (unsigned)i | ((q + 1u) << ORDER_PART_SHIFT);
                else
                    order[at + (unsigned)(SPLIT_PARTS - 1) * split] = (unsigned)i;
            }
        }
    }
}

/* CRT:1128-1181; gathers stay inside this process's strip */
#define AO_TILE_W 32
#define AO_TILE_H 8
#define AO_WINDOW_FLOATS 8192 /* LDS window of a tile: (AO_TILE_W + 2 rx) x (AO_TILE_H + 2 ry) depths */
/* depths of the rows next to a strip that belong to the ranks above and below (multi-GPU frames: §6 of
 * DESIGN.md): `above` holds the nbAbove rows just above the strip, `below` the nbBelow rows just below it */
struct DepthHalo
{
    const float *above, *below;
    int nbAbove, nbBelow;
};
/* the depth a tap reads: the strip's own frame buffer, or a neighbour's row out of the halo; rows that are in
 * neither are outside the frame (occluded, CRT:1164-1165) - or beyond the halo, which the host sizes by the reach
 * of the taps */
__device__ __forceinline__ bool aoDepthAt(const PixelRecord *__restrict__ pp, const DepthHalo &halo, int W, int nbRows,
                                          int xx, int yy, float &depth)
{
    if (xx < 0 || xx >= W || yy < -halo.nbAbove || yy >= nbRows + halo.nbBelow)
        return false;
    if (yy < 0)
        depth = halo.above[(yy + halo.nbAbove) * W + xx];
    else if (yy >= nbRows)
        depth = halo.below[(yy - nbRows) * W + xx];
    else
        depth = pp[yy * W + xx].colorInfo.w;
    return true;
}
/* tiles a workgroup renders one after the other (a run along x): what does not depend on the tile - the tap pairs,
 * their reach and, inside one binade, the deduped offsets - is made once per run instead of once per 256 pixels */
#define AO_TILES_PER_GROUP 8
#define AO_AHEAD 4 /* window depths a thread holds for the next tile: windows of up to 256 x AO_AHEAD floats are asked for a tile ahead */
__global__ __launch_bounds__(256) void k_ambientOcclusion(const SceneInfo si, const PostProcessingInfo ppi, int nbRows,
                                                          const PixelRecord *__restrict__ pp,
                                                          const float *__restrict__ randoms, long nbRandoms,
                                                          unsigned char *__restrict__ bitmap, const DepthHalo halo,
                                                          int firstRow, int windowFloats)
{
    /* The 256 taps of a pixel sit at x + X * param2 * randoms[i % wh] / 10.f, y + Y * param2 * randoms[(i + 100)
     * % wh] / 10.f (CRT:1146-1153): the offsets depend on the tap, not on the pixel.  The workgroup's 256
     * threads evaluate one tap's pair each - the same expressions, the two correctly rounded divisions
     * included - and every pixel then adds them to its coordinates: 2 divisions per thread instead of 512.
     *
     * A tile is 32 x 8 pixels.  Every tap of every pixel of the tile lands within rx = max |tapX| + 1 columns and
     * ry = max |tapY| + 1 rows of the tile: that window of depths (colorInfo.w of the 32-byte frame-buffer records) is
     * read once into LDS - 2 772 four-byte reads for taps of up to 16 pixels (432 for cfg4's, which reach one) instead
     * of 65 536 - and the comparisons of a pixel read LDS, consecutive lanes consecutive words.  Same comparisons on the
     * same values, counted in integers: the order of the additions does not matter.  A window that does not fit (taps
     * that reach beyond about 40 pixels) is gathered from memory as before. */
    __shared__ float tapX[256], tapY[256];
    __shared__ int reach[2];
    /* (dynamic: the host sizes the window for the reach the random buffer and param2 allow - 432 floats for cfg4's
     * taps instead of 32 KB - so that eight workgroups share a CU instead of four: a tile's work is a chain of waits) */
    extern __shared__ float window[];
    __shared__ int tapOffset[256];
    __shared__ unsigned block[1024]; /* the hash table and the deduped offsets of a steady tile, or the four histograms of a tile in two binades */
    unsigned *const table = block;
    int *const distinctOffset = (int *)block + 512, *const distinctWeight = (int *)block + 768;
    __shared__ int nbDistinct;
    __shared__ int cls[8];       /* a tile in two binades: {smallest, largest exponent of its regular columns, a column of each; the same for rows} */
    __shared__ float tapRange[16]; /* per wave: min / max of tapX, min / max of tapY */
    const int W = si.size.x;
    const int wh = W * si.size.y; /* the frame's, also when this rank renders a strip of it */
    const int tilesX = (W + AO_TILE_W - 1) / AO_TILE_W;
    const int nbTiles = tilesX * ((nbRows + AO_TILE_H - 1) / AO_TILE_H);
    if (threadIdx.x < 2)
        reach[threadIdx.x] = 0;
    __syncthreads();
    {
        const int i = threadIdx.x; /* tap i: X = -16 + 2 * (i / 16), Y = -16 + 2 * (i % 16), in loop order */
        const int X = -16 + 2 * (i >> 4), Y = -16 + 2 * (i & 15);
        const int ix = i % wh;
        const int iy = (i + 100) % wh;
        const float rx = (ix < nbRandoms) ? randoms[ix] : 0.f;
        const float ry = (iy < nbRandoms) ? randoms[iy] : 0.f;
        const float tx = X * ppi.param2 * rx / 10.f;
        const float ty = Y * ppi.param2 * ry / 10.f;
        tapX[i] = tx;
        tapY[i] = ty;
        /* (int)(x + t) stays within ceil(|t|) + 1 of x for an integer x below 2^23; anything else (NaN, huge)
         * sends the tile down the gather path */
        const float ax = fabsf(tx), ay = fabsf(ty);
        int cx = (ax < 1.0e6f) ? (int)ax + 2 : (1 << 20);
        int cy = (ay < 1.0e6f) ? (int)ay + 2 : (1 << 20);
        float lowX = tx, highX = tx, lowY = ty, highY = ty;
        for (int off = 32; off > 0; off >>= 1) /* (a wave's maximum first: 8 atomics on one word instead of 512) */
        {
            cx = max(cx, __shfl_xor(cx, off, 64));
            cy = max(cy, __shfl_xor(cy, off, 64));
            lowX = fminf(lowX, __shfl_xor(lowX, off, 64)), highX = fmaxf(highX, __shfl_xor(highX, off, 64));
            lowY = fminf(lowY, __shfl_xor(lowY, off, 64)), highY = fmaxf(highY, __shfl_xor(highY, off, 64));
        }
        if ((i & 63) == 0)
        {
            atomicMax(&reach[0], cx);
            atomicMax(&reach[1], cy);
            tapRange[4 * (i >> 6)] = lowX, tapRange[4 * (i >> 6) + 1] = highX;
            tapRange[4 * (i >> 6) + 2] = lowY, tapRange[4 * (i >> 6) + 3] = highY;
        }
    }
    __syncthreads();
    const int rx = reach[0], ry = reach[1];
    const int ww = AO_TILE_W + 2 * rx, wrows = AO_TILE_H + 2 * ry;
    const bool tiled = rx < 4096 && ry < 4096 && ww * wrows <= windowFloats && ww * wrows <= AO_WINDOW_FLOATS;
    /* (a NaN among the taps makes fminf / fmaxf skip it; such a buffer has an enormous reach and is not tiled) */
    const float tapLowX = fminf(fminf(tapRange[0], tapRange[4]), fminf(tapRange[8], tapRange[12]));
    const float tapHighX = fmaxf(fmaxf(tapRange[1], tapRange[5]), fmaxf(tapRange[9], tapRange[13]));
    const float tapLowY = fminf(fminf(tapRange[2], tapRange[6]), fminf(tapRange[10], tapRange[14]));
    const float tapHighY = fmaxf(fmaxf(tapRange[3], tapRange[7]), fmaxf(tapRange[11], tapRange[15]));
    const int binsX = 2 * rx + 1, binsY = 2 * ry + 1;
    int tableKey = 0; /* the binades (of x and of the frame's y) the deduped offsets in LDS were made for; 0: none */
    const bool pipelined = tiled && ww * wrows <= 256 * AO_AHEAD;
    float aheadDepth[AO_AHEAD];
    float4 aheadLocal = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ahead = [&](int t) { /* this thread's share of tile t's window, and its own pixel's record */
        const int tx0 = (t % tilesX) * AO_TILE_W, ty0 = (t / tilesX) * AO_TILE_H;
#pragma unroll
        for (int k = 0; k < AO_AHEAD; ++k)
        {
            const int i = (int)threadIdx.x + 256 * k;
            float d = 0.f;
            if (i < ww * wrows)
                aoDepthAt(pp, halo, W, nbRows, tx0 - rx + i % ww, ty0 - ry + i / ww, d);
            aheadDepth[k] = d;
        }
        const int px = tx0 + (int)(threadIdx.x % AO_TILE_W), py = ty0 + (int)(threadIdx.x / AO_TILE_W);
        aheadLocal = pp[(px < W && py < nbRows) ? py * W + px : 0].colorInfo;
    };
    for (int run = 0; run < AO_TILES_PER_GROUP; ++run)
    {
        /* tile `run` of this workgroup: a stride of the grid apart, not side by side.  The tiles of the frame's first
         * tile row and column (x or y below the tile's size: regular columns of up to five binades) take the per-pixel
         * loop, 30 times the cost of a tile - side by side they were eight of them in one workgroup, and that
         * workgroup was the kernel: 0.52 ms whatever the other 4 000 did */
        const int tile = (int)blockIdx.x + run * (int)gridDim.x;
        if (tile >= nbTiles)
            break;
        const int x0 = (tile % tilesX) * AO_TILE_W;
        const int y0 = (tile / tilesX) * AO_TILE_H;
        const int wx0 = x0 - rx, wy0 = y0 - ry;
        const int x = x0 + (int)(threadIdx.x % AO_TILE_W);
        const int y = y0 + (int)(threadIdx.x / AO_TILE_W);
        const bool mine = x < W && y < nbRows;
        const int index = mine ? y * W + x : 0;
        /* A window of up to 1 024 depths (taps that reach 12 pixels) is asked for ONE TILE AHEAD, into registers, behind
         * the barrier below: the loads of tile n + 1 are in flight while tile n is compared and stored, and a tile is
         * no longer two memory latencies long. */
        if (pipelined && run == 0)
            ahead(tile);
        float4 local;
        if (pipelined)
        {
            local = aheadLocal;
            for (int k = 0; k < AO_AHEAD; ++k)
                if ((int)threadIdx.x + 256 * k < ww * wrows)
                    window[threadIdx.x + 256 * k] = aheadDepth[k];
        }
        else
        {
            local = pp[index].colorInfo; /* (asked for before the window: the two waits overlap) */
            if (tiled)
                for (int i = threadIdx.x; i < ww * wrows; i += 256)
                {
                    const int gx = wx0 + i % ww, gy = wy0 + i / ww;
                    float d = 0.f;
                    aoDepthAt(pp, halo, W, nbRows, gx, gy, d);
                    window[i] = d;
                }
        }
        /* Steady taps.  (int)(x + t) - x is the same for every x of the tile when x and all the sums x + t lie in one
         * binade: x is a multiple of that binade's ulp U (a power of two below 1, so x / U is even and ties round the
         * same way), hence RN(x + t) = x + RN_U(t), and the sums are positive, so the truncation is a floor.  Then a
         * tap is ONE integer offset into the window for the whole tile - evaluated once per tap, on the tile's first
         * column and row, with the reference's own expression - and the same for every tile of those two binades.
         * Tiles that straddle a power of two in x or in the frame's y, or whose window leaves the frame, take the
         * per-pixel evaluation below.
         *
         * ... and the taps that land on the same depth are one comparison.  cfg4's taps (param2 = 10, randoms of
         * +-0.005) reach one pixel: 256 taps, FOUR distinct offsets.  The count is an integer sum, so it is taken over
         * the distinct offsets with their multiplicities: the workgroup dedupes its 256 offsets (a 512-slot hash
         * table: key and count in one word, atomicCAS to claim, atomicAdd to count) and a pixel then makes one read
         * and one compare per DISTINCT offset - and never more than before: beyond 128 distinct offsets the plain loop
         * runs. */
        bool steady = false;
        if (tiled)
        {
            const int xlo = x0 - rx, xhi = x0 + AO_TILE_W - 1 + rx;
            const int ylo = y0 + firstRow - ry, yhi = y0 + firstRow + AO_TILE_H - 1 + ry;
            steady = wx0 >= 0 && wy0 >= -halo.nbAbove && wx0 + ww <= W && wy0 + wrows <= nbRows + halo.nbBelow && xlo >= 1 &&
                     ylo >= 1 && __clz(xlo) == __clz(xhi) && __clz(ylo) == __clz(yhi);
            const int key = steady ? (1 << 16) | (__clz(xlo) << 8) | __clz(ylo) : 0;
            if (steady && key != tableKey)
            {
                const int i = threadIdx.x;
                const int dx = (int)((float)x0 + tapX[i]) - x0;
                const int dy = (int)((float)(y0 + firstRow) + tapY[i]) - (y0 + firstRow);
                const int off = dy * ww + dx; /* |off| < ww * wrows <= AO_WINDOW_FLOATS: inside the window */
                tapOffset[i] = off;
                if (i == 0)
                    nbDistinct = 0;
                table[i] = 0u;
                table[i + 256] = 0u;
                __syncthreads();
                const unsigned tag = (unsigned)(off + AO_WINDOW_FLOATS) + 1u; /* 1 ... 2 x 8192: 0 is an empty slot */
                unsigned h = (tag * 2654435761u) >> 23;
                for (;;)
                {
                    const unsigned before = atomicCAS(&table[h], 0u, tag << 9);
                    if (before == 0u || (before >> 9) == tag)
                    {
                        atomicAdd(&table[h], 1u); /* at most 256 taps: the count stays below the key's bits */
                        break;
                    }
                    h = (h + 1u) & 511u;
                }
                __syncthreads();
                for (int slot = i; slot < 512; slot += 256)
                {
                    const unsigned entry = table[slot];
                    if (entry != 0u)
                    {
                        const int at = atomicAdd(&nbDistinct, 1);
                        distinctOffset[at] = (int)(entry >> 9) - 1 - AO_WINDOW_FLOATS;
                        distinctWeight[at] = (int)(entry & 511u);
                    }
                }
            }
            tableKey = steady ? key : tableKey;
        }
        /* A tile in TWO binades (it straddles a power of two in x, in the frame's y, or both: a fifth of a 4K frame's
         * tiles - and until this was here 80 % of the kernel's time, 256 float additions and conversions per pixel).  A
         * pixel's column is REGULAR when x, x + the smallest tap and x + the largest tap lie in one binade (the sums
         * are monotonic in the tap): for such columns of one binade (int)(x + t) - x is the same, by the argument
         * above; likewise rows.  A tile has regular columns of at most two binades and regular rows of at most two:
         * four histograms of tap offsets, made once per tile with the reference's own expression on one column and one
         * row of each class, serve every pixel whose column and row are regular - one read and one compare per bin
         * (25 for cfg4's taps) instead of 256 evaluations.  The pixels of the irregular columns and rows (cfg4: the one
         * column AT the power of two, whose sums with negative taps fall into the binade below) keep the per-pixel loop. */
        const int e0x = (int)(__float_as_uint((float)x) >> 23), e0y = (int)(__float_as_uint((float)(y + firstRow)) >> 23);
        const bool regularX = x >= 1 && (int)(__float_as_uint((float)x + tapLowX) >> 23) == e0x &&
                              (int)(__float_as_uint((float)x + tapHighX) >> 23) == e0x;
        const bool regularY = y + firstRow >= 1 && (int)(__float_as_uint((float)(y + firstRow) + tapLowY) >> 23) == e0y &&
                              (int)(__float_as_uint((float)(y + firstRow) + tapHighY) >> 23) == e0y;
        const bool windowInside = tiled && wx0 >= 0 && wy0 >= -halo.nbAbove && wx0 + ww <= W && wy0 + wrows <= nbRows + halo.nbBelow;
        bool classed = tiled && !steady && binsX * binsY <= 256;
        if (classed)
        {
            const int i = threadIdx.x;
            tableKey = 0; /* (the histograms take the place of the steady tiles' table) */
            if (i < 8)
                cls[i] = (i == 0 || i == 2 || i == 4 || i == 6) ? 0x7fffffff : -1;
            block[i] = block[i + 256] = block[i + 512] = block[i + 768] = 0u;
            if (i < binsX * binsY)
                tapOffset[i] = (i / binsX - ry) * ww + (i % binsX - rx);
            __syncthreads();
            /* (thread i of the first row of the tile speaks for column i, thread 32 r for row r) */
            if (i < AO_TILE_W && regularX)
            {
                atomicMin(&cls[0], e0x);
                atomicMax(&cls[1], e0x);
            }
            if (i % AO_TILE_W == 0 && regularY)
            {
                atomicMin(&cls[4], e0y);
                atomicMax(&cls[5], e0y);
            }
            __syncthreads();
            if (i < AO_TILE_W && regularX)
            {
                if (e0x == cls[0])
                    atomicMin(&cls[2], x);
                if (e0x == cls[1])
                    atomicMax(&cls[3], x);
            }
            if (i % AO_TILE_W == 0 && regularY)
            {
                if (e0y == cls[4])
                    atomicMin(&cls[6], y + firstRow);
                if (e0y == cls[5])
                    atomicMax(&cls[7], y + firstRow);
            }
            __syncthreads();
            classed = cls[1] >= 0 && cls[5] >= 0 && cls[1] - cls[0] <= 1 && cls[5] - cls[4] <= 1;
            if (classed)
            {
                int dx[2], dy[2];
                for (int c = 0; c < 2; ++c)
                {
                    const int xr = c ? cls[3] : cls[2], yr = c ? cls[7] : cls[6];
                    dx[c] = (int)((float)xr + tapX[i]) - xr;
                    dy[c] = (int)((float)yr + tapY[i]) - yr;
                }
                for (int c = 0; c < 4; ++c)
                    atomicAdd(&block[c * 256 + (dy[c & 1] + ry) * binsX + (dx[c >> 1] + rx)], 1u);
            }
        }
        __syncthreads(); /* the window is in LDS, and so are the offsets */
        if (pipelined && run + 1 < AO_TILES_PER_GROUP && tile + (int)gridDim.x < nbTiles)
            ahead(tile + (int)gridDim.x);
        if (mine)
        {
            float occ = 0.f;
            const float depth = local.w;
            float c = 0.f;
            if (tiled)
            {
                /* a tile whose window lies inside the frame needs no bounds test per tap */
                const bool inside = wx0 >= 0 && wy0 >= -halo.nbAbove && wx0 + ww <= W && wy0 + wrows <= nbRows + halo.nbBelow;
                /* counted in an integer (at most 256: the float sum of the reference is the same number) */
                /* a strip is rows [firstRow, firstRow + nbRows) of the frame: the tap's row is evaluated with the frame's y
                 * (the float addition rounds, and truncates towards zero, by the row's position in the frame) */
                const int origin = -((wy0 + firstRow) * ww + wx0);
                const float fx = (float)x, fy = (float)(y + firstRow);
                int count = 0;
                if (steady)
                {
                    const float *centre = window + ((y - wy0) * ww + (x - wx0));
                    const int distinct = nbDistinct;
                    if (distinct <= 128)
                    {
                        for (int i = 0; i < distinct; ++i)
                            count += (centre[distinctOffset[i]] >= depth) ? distinctWeight[i] : 0;
                    }
                    else
                    {
#pragma unroll 16
                        for (int i = 0; i < 256; ++i)
                            count += (centre[tapOffset[i]] >= depth) ? 1 : 0;
                    }
                }
                else if (classed && regularX && regularY)
                {
                    const float *centre = window + ((y - wy0) * ww + (x - wx0));
                    const unsigned *hist = block + 256 * ((e0x == cls[0] ? 0 : 2) + (e0y == cls[4] ? 0 : 1));
                    const int bins = binsX * binsY;
                    if (windowInside)
                        for (int b = 0; b < bins; ++b)
                        {
                            const int weight = (int)hist[b];
                            if (weight)
                                count += (centre[tapOffset[b]] >= depth) ? weight : 0;
                        }
                    else /* a tile at the frame's edge: a tap that lands outside the frame (or the strip's halo) counts, CRT:1164-1165 */
                        for (int b = 0; b < bins; ++b)
                        {
                            const int weight = (int)hist[b];
                            const int xx = x + b % binsX - rx, yy = y + b / binsX - ry;
                            const bool in = xx >= 0 && xx < W && yy >= -halo.nbAbove && yy < nbRows + halo.nbBelow;
                            if (weight)
                                count += (!in || centre[tapOffset[b]] >= depth) ? weight : 0;
                        }
                }
                else if (inside)
                {
#pragma unroll 8
                    for (int i = 0; i < 256; ++i)
                    {
                        const int xx = (int)(fx + tapX[i]);
                        const int yy = (int)(fy + tapY[i]);
                        count += (window[__mul24(yy, ww) + xx + origin] >= depth) ? 1 : 0;
                    }
                }
                else
                {
                    /* (branch-free, so that the loop unrolls and its LDS reads overlap: a tap outside the frame reads cell 0
                     * of the window and counts whatever it holds) */
#pragma unroll 8
                    for (int i = 0; i < 256; ++i)
                    {
                        const int xx = (int)(fx + tapX[i]);
                        const int yy = (int)(fy + tapY[i]);
                        const bool in = xx >= 0 && xx < W && yy - firstRow >= -halo.nbAbove && yy - firstRow < nbRows + halo.nbBelow;
                        const float tap = window[in ? __mul24(yy, ww) + xx + origin : 0];
                        count += (!in || tap >= depth) ? 1 : 0;
                    }
                }
                occ = (float)count;
                c = 256.f;
            }
            else
            {
                for (int i = 0; i < 256; ++i)
                {
                    c += 1.f;
                    int xx = (int)(x + tapX[i]);
                    int yy = (int)((y + firstRow) + tapY[i]) - firstRow;
                    float tap;
                    if (aoDepthAt(pp, halo, W, nbRows, xx, yy, tap))
                    {
                        if (tap >= depth)
                            occ += 1.f;
                    }
                    else
                        occ += 1.f;
                }
            }
            occ /= (float)c;
            occ += 0.3f;
            v3 col = V(local.x, local.y, local.z);
            if (occ < 1.f)
            {
                col.x *= occ;
                col.y *= occ;
                col.z *= occ;
            }
            if (si.pathTracingIteration > NB_MAX_ITERATIONS)
            {
                float d = (float)(si.pathTracingIteration - NB_MAX_ITERATIONS + 1);
                col.x /= d;
                col.y /= d;
                col.z /= d;
            }
            saturate3(col);
            makeColor(si, col, bitmap, index);
        }
        __syncthreads(); /* the next tile's window goes where this one's is still being read */
    }
}

/* CRT:1081-1120 */
/* the depths of rows [row0, row0 + n) of a strip, packed for the neighbour that needs them */
__global__ __launch_bounds__(256) void k_packDepthRows(const PixelRecord *__restrict__ pp, int W, int row0, int n,
                                                       float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W * n)
        out[i] = pp[row0 * W + i].colorInfo.w;
}

__global__ __launch_bounds__(256) void k_depthOfField(const SceneInfo si, const PostProcessingInfo ppi, int nbRows,
                                                      const PixelRecord *__restrict__ pp,
                                                      const float *__restrict__ randoms, long nbRandoms,
                                                      unsigned char *__restrict__ bitmap)
{
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = si.size.x;
    const int wh = W * nbRows;
    if (index >= wh)
        return;
    const int x = index % W;
    const int y = index / W;
    v3 local = V(0.f, 0.f, 0.f);
    const float4 own = pp[index].colorInfo;
    float depth = fabsf(own.w - ppi.param1) / si.viewDistance;
    for (int i = 0; i < ppi.param3; ++i)
    {
        int ix = i % wh;
        int iy = (i + 1000) % wh;
        float rx = (ix < nbRandoms) ? randoms[ix] : 0.f;
        float ry = (iy < nbRandoms) ? randoms[iy] : 0.f;
        int xx = (int)(x + depth * rx * ppi.param2);
        int yy = (int)(y + depth * ry * ppi.param2);
        if (xx >= 0 && xx < W && yy >= 0 && yy < nbRows)
        {
            int localIndex = yy * W + xx;
            if (localIndex >= 0 && localIndex < wh)
            {
                float4 o = pp[localIndex].colorInfo;
                local.x += o.x;
                local.y += o.y;
                local.z += o.z;
            }
        }
        else
        {
            local.x += own.x;
            local.y += own.y;
            local.z += own.z;
        }
    }
    local.x /= (float)ppi.param3;
    local.y /= (float)ppi.param3;
    local.z /= (float)ppi.param3;
    if (si.pathTracingIteration > NB_MAX_ITERATIONS)
    {
        float d = (float)(si.pathTracingIteration - NB_MAX_ITERATIONS + 1);
        local.x /= d;
        local.y /= d;
        local.z /= d;
    }
    makeColor(si, local, bitmap, index);
}

/* ======================================================================= */
/* Host layer                                                               */
/* ======================================================================= */

namespace
{
/* SOLR_HIP_DEBUG_TIMING=1: where the host side of an upload spends its time (stderr) */
/* SOLR_HIP_HOST_PROFILE=1 (diagnostics): what the HOST spends per call inside the entry points of a frame - at eight
 * GPUs a strip takes 0.04 ms and the host's own 0.04-0.05 ms per step is what bounds the frame rate.  Totals go to stderr
 * at finalize_scene. */
struct HostProfile
{
    const bool on = getenv("SOLR_HIP_HOST_PROFILE") != nullptr;
    struct Entry
    {
        const char *name;
        double seconds;
        long calls;
    } entries[16] = {};
    int used = 0;
    Entry *find(const char *name)
    {
        for (int i = 0; i < used; ++i)
            if (entries[i].name == name)
                return &entries[i];
        if (used < 16)
        {
            entries[used].name = name;
            return &entries[used++];
        }
        return nullptr;
    }
    void report()
    {
        if (!on)
            return;
        for (int i = 0; i < used; ++i)
            fprintf(stderr, "solr_hip host: %-34s %9.3f us per call over %ld calls\n", entries[i].name,
                    1e6 * entries[i].seconds / (entries[i].calls ? entries[i].calls : 1), entries[i].calls);
        used = 0;
    }
    ~HostProfile() { report(); } /* (a host that never finalizes: at exit) */
};
HostProfile gHostProfile;
struct HostSpan
{
    const char *name;
    std::chrono::steady_clock::time_point t0;
    explicit HostSpan(const char *n) : name(n)
    {
        if (gHostProfile.on)
            t0 = std::chrono::steady_clock::now();
    }
    ~HostSpan()
    {
        if (!gHostProfile.on)
            return;
        if (HostProfile::Entry *e = gHostProfile.find(name))
        {
            e->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            e->calls++;
        }
    }
};

struct PhaseTimer
{
    const bool on = getenv("SOLR_HIP_DEBUG_TIMING") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
        if (!on)
            return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "solr_hip: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};

/* ---- animated scenes: rotate + refit on the device ---------------------------------------------------
 * The reference animates a scene by GPUKernel::rotatePrimitives + compactBoxes(false) on the host and a
 * full upload, every frame (MoleculeScene.cpp:75-81; GPUKernel.cpp:1378-1460 rotates the primitives of
 * the level-0 boxes and refits every level, :1151-1281 flattens again).  The flattened tree keeps its
 * shape under that - only primitive coordinates and node bounds change - so the same arithmetic runs
 * here on the resident arena instead: the primitive rows in place, then the nodes bottom-up.  Every
 * expression below is the host builder's (sol-r_amd/host/GPUKernel.cpp rotateVector, updateBoundingBox,
 * updateOutterBoundingBox), in its order and with its comparisons, so that the arena afterwards holds
 * bit for bit what a host rotation followed by a fresh upload would have put there. */
struct RotationArgs
{
    float cx, cy, cz;
    float cosx, cosy, cosz;
    float sinx, siny, sinz;
};

__device__ inline void rotateRow(float4 &v, float cx, float cy, float cz, const RotationArgs &R)
{
    float vx = v.x - cx, vy = v.y - cy, vz = v.z - cz;
    float ry = vy * R.cosx - vz * R.sinx;
    float rz = vy * R.sinx + vz * R.cosx;
    vy = ry;
    vz = rz;
    rz = vz * R.cosy - vx * R.siny;
    float rx = vz * R.siny + vx * R.cosy;
    vz = rz;
    vx = rx;
    rx = vx * R.cosz - vy * R.sinz;
    ry = vx * R.sinz + vy * R.cosz;
    v.x = rx + cx;
    v.y = ry + cy;
    v.z = rz + cz;
}

/* Leaf records (scene_layout.h): for every leaf of a node list, the first primitive's test data and index in
 * one 64-byte line.  A function of the primitive records and the list's start indices alone: run after every
 * upload of the arena and after every device-side rotation of the primitives. */
__global__ __launch_bounds__(256) void k_buildLeafRecords(float4 *__restrict__ arena, unsigned offNodes, unsigned offStart,
                                                         unsigned offPrims, unsigned offLeaf, int nbNodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbNodes)
        return;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
    const int nb = __float_as_int(arena[offNodes + 2u * (unsigned)i + 1u].z);
    if (nb > 0)
    {
        const int start = ((const int *)arena)[offStart + (unsigned)i];
        const float4 *prim = arena + offPrims + 8u * (unsigned)start;
        r0 = prim[ROW_P0_TYPE];
        r1 = prim[ROW_SIZE_MAT];
        r2 = prim[ROW_P1_INDEX];
        r3 = prim[ROW_P2];
        if (planeClass(__float_as_int(r0.w) & PRIM_TYPE_MASK))
        {
            const float4 n0 = prim[ROW_N0];
            r2 = make_float4(n0.x, n0.y, n0.z, r2.w);
            r3 = make_float4(r3.w, 0.f, 0.f, 0.f);
        }
        r3.w = __int_as_float(start);
    }
    float4 *out = arena + offLeaf + 4u * (unsigned)i;
    out[0] = r0;
    out[1] = r1;
    out[2] = r2;
    out[3] = r3;
}

/* The thin copy of a node list (rt_device.h tightRay; scene_layout.h SceneArgs::tightLists): leaf by leaf.  A leaf
 * whose primitives are all plain axis planes becomes the union of their rectangles, `margin` thick and `margin` wider,
 * cut with the reference's box (never larger than it: a ray the thin box lets in, the reference's let in as well);
 * every other node is copied.  k_tightenInner then makes the inner nodes the unions of the leaves below them. */
__global__ __launch_bounds__(256) void k_tightenLeaves(float4 *__restrict__ arena, unsigned offNodes, unsigned offTight,
                                                      unsigned offStart, unsigned offPrims, int nbNodes, float margin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbNodes)
        return;
    float4 row0 = arena[offNodes + 2u * (unsigned)i], row1 = arena[offNodes + 2u * (unsigned)i + 1u];
    const int nb = __float_as_int(row1.z);
    if (nb > 0)
    {
        const int start = ((const int *)arena)[offStart + (unsigned)i];
        float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
        bool plain = true;
        for (int k = 0; k < nb && plain; ++k)
        {
            const float4 *prim = arena + offPrims + 8u * (unsigned)(start + k);
            const float4 p = prim[ROW_P0_TYPE], s = prim[ROW_SIZE_MAT];
            const int kind = (__float_as_int(p.w) >> PRIM_KIND_SHIFT) & 15;
            plain = kind == KIND_PLANE_XY || kind == KIND_PLANE_YZ || kind == KIND_PLANE_XZ;
            /* (a size is compared with a distance: its sign cannot make the rectangle larger than |size|) */
            const float ex = kind == KIND_PLANE_YZ ? margin : fabsf(s.x) + margin;
            const float ey = kind == KIND_PLANE_XZ ? margin : fabsf(s.y) + margin;
            const float ez = kind == KIND_PLANE_XY ? margin : fabsf(s.z) + margin;
            lx = fminf(lx, p.x - ex), hx = fmaxf(hx, p.x + ex);
            ly = fminf(ly, p.y - ey), hy = fmaxf(hy, p.y + ey);
            lz = fminf(lz, p.z - ez), hz = fmaxf(hz, p.z + ez);
        }
        /* (finite, ordered bounds only: anything else keeps the reference's box) */
        plain = plain && lx <= hx && ly <= hy && lz <= hz && fabsf(lx) < 3.0e38f && fabsf(hx) < 3.0e38f && fabsf(ly) < 3.0e38f &&
                fabsf(hy) < 3.0e38f && fabsf(lz) < 3.0e38f && fabsf(hz) < 3.0e38f;
        if (plain)
        {
            const float nlx = fmaxf(row0.x, lx), nly = fmaxf(row0.y, ly), nlz = fmaxf(row0.z, lz);
            const float nhx = fminf(row1.x, hx), nhy = fminf(row1.y, hy), nhz = fminf(row0.w, hz);
            if (nlx <= nhx && nly <= nhy && nlz <= nhz)
            {
                row0 = make_float4(nlx, nly, nlz, nhz);
                row1 = make_float4(nhx, nhy, row1.z, row1.w);
            }
        }
    }
    arena[offTight + 2u * (unsigned)i] = row0;
    arena[offTight + 2u * (unsigned)i + 1u] = row1;
}

/* inner node i of the thin copy: the union of the leaves of its subtree (nodes i + 1 ... i + skip - 1: skip pointers
 * are nested intervals), cut with its own box.  A group still passes whenever one of its members does. */
__global__ __launch_bounds__(256) void k_tightenInner(float4 *__restrict__ arena, unsigned offTight, int nbNodes, int listLength)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbNodes)
        return;
    const float4 row0 = arena[offTight + 2u * (unsigned)i], row1 = arena[offTight + 2u * (unsigned)i + 1u];
    const int nb = __float_as_int(row1.z), skip = __float_as_int(row1.w);
    if (nb > 0 || skip <= 1)
        return;
    const int listEnd = (i / listLength + 1) * listLength; /* (several lists one behind the other: stay in this one) */
    const int end = min(i + skip, listEnd);
    float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
    for (int j = i + 1; j < end; ++j)
    {
        const float4 b = arena[offTight + 2u * (unsigned)j + 1u];
        if (__float_as_int(b.z) <= 0)
            continue;
        const float4 a = arena[offTight + 2u * (unsigned)j];
        lx = fminf(lx, a.x), ly = fminf(ly, a.y), lz = fminf(lz, a.z);
        hx = fmaxf(hx, b.x), hy = fmaxf(hy, b.y), hz = fmaxf(hz, a.w);
    }
    const float nlx = fmaxf(row0.x, lx), nly = fmaxf(row0.y, ly), nlz = fmaxf(row0.z, lz);
    const float nhx = fminf(row1.x, hx), nhy = fminf(row1.y, hy), nhz = fminf(row0.w, hz);
    if (!(nlx <= nhx && nly <= nhy && nlz <= nhz))
        return; /* no leaf below it, or bounds that are not numbers: the reference's box stays */
    arena[offTight + 2u * (unsigned)i] = make_float4(nlx, nly, nlz, nhz);
    arena[offTight + 2u * (unsigned)i + 1u] = make_float4(nhx, nhy, row1.z, row1.w);
}

/* maybeBuildOrderFreeLists' precondition, for the exact list as the arena holds it: every inner node holds its
 * direct children, every leaf its primitives (the same float arithmetic as the host loop there, which stays as the
 * route for an arena that is not laid out).  *bad is raised for a node that does not. */
__global__ __launch_bounds__(256) void k_listEncloses(const float4 *__restrict__ arena, unsigned offNodes, unsigned offStart,
                                                      unsigned offPrims, int nbNodes, int nbPrims, int *bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbNodes)
        return;
    const float4 a = arena[offNodes + 2u * (unsigned)i], b = arena[offNodes + 2u * (unsigned)i + 1u];
    const int count = __float_as_int(b.z);
    const int end = min(i + max(__float_as_int(b.w), 1), nbNodes);
    bool encloses = true;
    if (count <= 0)
    {
        for (int j = i + 1; j < end && encloses;)
        {
            const float4 ca = arena[offNodes + 2u * (unsigned)j], cb = arena[offNodes + 2u * (unsigned)j + 1u];
            encloses = ca.x >= a.x && ca.y >= a.y && ca.z >= a.z && cb.x <= b.x && cb.y <= b.y && ca.w <= a.w;
            j += max(__float_as_int(cb.w), 1);
        }
    }
    else
    {
        const int start = ((const int *)arena)[offStart + (unsigned)i];
        for (int k = 0; k < count && encloses; ++k)
        {
            const long long pi = (long long)start + k;
            if (start < 0 || pi >= nbPrims)
            {
                encloses = false;
                break;
            }
            const float4 *r = arena + offPrims + (size_t)PRIM_ROWS * (size_t)pi;
            const float4 p0 = r[ROW_P0_TYPE], size = r[ROW_SIZE_MAT];
            const int type = __float_as_int(p0.w) & PRIM_TYPE_MASK;
            float lo[3] = {p0.x, p0.y, p0.z}, hi[3] = {p0.x, p0.y, p0.z};
            auto add = [&](const float4 &v) {
                lo[0] = v.x < lo[0] ? v.x : lo[0], lo[1] = v.y < lo[1] ? v.y : lo[1], lo[2] = v.z < lo[2] ? v.z : lo[2];
                hi[0] = hi[0] < v.x ? v.x : hi[0], hi[1] = hi[1] < v.y ? v.y : hi[1], hi[2] = hi[2] < v.z ? v.z : hi[2];
            };
            float grow[3] = {size.x, size.y, size.z};
            if (type == ptTriangle)
            {
                add(r[ROW_P1_INDEX]);
                add(r[ROW_P2]);
                grow[0] = grow[1] = grow[2] = 0.f;
            }
            else if (type == ptCylinder)
            {
                add(r[ROW_P1_INDEX]);
                grow[1] = grow[2] = grow[0];
            }
            else if (type == ptSphere)
                grow[1] = grow[2] = grow[0];
            auto larger = [](float x, float y) { return x < y ? y : x; }; /* std::max */
            auto slack = [&](int k) { return 4.f * 1.1920929e-7f * larger(larger(fabsf(lo[k]), fabsf(hi[k])), fabsf(grow[k])); };
            const float ex = slack(0), ey = slack(1), ez = slack(2);
            encloses = a.x <= lo[0] - fabsf(grow[0]) + ex && a.y <= lo[1] - fabsf(grow[1]) + ey && a.z <= lo[2] - fabsf(grow[2]) + ez &&
                       b.x >= hi[0] + fabsf(grow[0]) - ex && b.y >= hi[1] + fabsf(grow[1]) - ey && a.w >= hi[2] + fabsf(grow[2]) - ez;
        }
    }
    if (!encloses)
        *bad = 1;
}

__global__ __launch_bounds__(256) void k_rotatePrimitives(float4 *__restrict__ arena, unsigned offPrims, int nbPrimitives,
                                                          const unsigned char *__restrict__ movable,
                                                          const RotationArgs R)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbPrimitives || !movable[i])
        return;
    float4 *r = arena + offPrims + (size_t)PRIM_ROWS * i;
    float4 p0 = r[ROW_P0_TYPE];
    const int type = __float_as_int(p0.w) & PRIM_TYPE_MASK;
    rotateRow(p0, R.cx, R.cy, R.cz, R);
    r[ROW_P0_TYPE] = p0;
    if (type == ptCylinder || type == ptTriangle)
    {
        float4 p1 = r[ROW_P1_INDEX], p2 = r[ROW_P2], n0 = r[ROW_N0], n1 = r[ROW_N1], n2 = r[ROW_N2];
        rotateRow(p1, R.cx, R.cy, R.cz, R);
        rotateRow(p2, R.cx, R.cy, R.cz, R);
        rotateRow(n0, 0.f, 0.f, 0.f, R);
        rotateRow(n1, 0.f, 0.f, 0.f, R);
        rotateRow(n2, 0.f, 0.f, 0.f, R);
        if (type == ptCylinder)
        {
            float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;
            const float len = __builtin_sqrtf(ax * ax + ay * ay + az * az);
            if (len != 0)
            {
                ax /= len;
                ay /= len;
                az /= len;
            }
            n1.x = ax;
            n1.y = ay;
            n1.z = az;
        }
        r[ROW_P1_INDEX] = p1;
        r[ROW_P2] = p2;
        r[ROW_N0] = n0;
        r[ROW_N1] = n1;
        r[ROW_N2] = n2;
    }
}

/* One node per thread, the nodes of one height of the tree per launch (children first).  A node with
 * primitives is a level-0 box: updateBoundingBox; one without is the union of its children:
 * updateOutterBoundingBox, seeded like it (+-viewDistance; +-infinity for our own grouping nodes, which
 * the list marks with the sign bit). */
__global__ __launch_bounds__(256) void k_refitNodes(float4 *__restrict__ arena, unsigned offNodes, unsigned offStart,
                                                    unsigned offPrims, const int *__restrict__ list, int count,
                                                    float seed)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count)
        return;
    const int entry = list[t];
    const int node = entry & 0x7fffffff;
    if (entry < 0)
        seed = INFINITY; /* one of our own grouping nodes: the plain union */
    float4 *rows = arena + offNodes;
    const float4 row1 = rows[2 * node + 1];
    const int nb = __float_as_int(row1.z);
    const int skip = __float_as_int(row1.w);
    float lx, ly, lz, hx, hy, hz;
    if (nb > 0)
    {
        const int first = ((const int *)arena)[offStart + node];
        lx = ly = lz = 1000000.f;
        hx = hy = hz = -1000000.f;
        for (int k = 0; k < nb; ++k)
        {
            const float4 *r = arena + offPrims + (size_t)PRIM_ROWS * (first + k);
            const float4 p0 = r[ROW_P0_TYPE];
            const float4 size = r[ROW_SIZE_MAT];
            const int type = __float_as_int(p0.w) & PRIM_TYPE_MASK;
            /* std::min(a, b) is (b < a) ? b : a and std::max(a, b) is (a < b) ? b : a: kept as such, the
             * sign of a zero that ties depends on it */
            float c0x = p0.x, c0y = p0.y, c0z = p0.z, c1x = p0.x, c1y = p0.y, c1z = p0.z;
            if (type == ptTriangle || type == ptCylinder)
            {
                const float4 p1 = r[ROW_P1_INDEX];
                c0x = (p1.x < p0.x) ? p1.x : p0.x;
                c0y = (p1.y < p0.y) ? p1.y : p0.y;
                c0z = (p1.z < p0.z) ? p1.z : p0.z;
                c1x = (p0.x < p1.x) ? p1.x : p0.x;
                c1y = (p0.y < p1.y) ? p1.y : p0.y;
                c1z = (p0.z < p1.z) ? p1.z : p0.z;
                if (type == ptTriangle)
                {
                    const float4 p2 = r[ROW_P2];
                    c0x = (p2.x < c0x) ? p2.x : c0x;
                    c0y = (p2.y < c0y) ? p2.y : c0y;
                    c0z = (p2.z < c0z) ? p2.z : c0z;
                    c1x = (c1x < p2.x) ? p2.x : c1x;
                    c1y = (c1y < p2.y) ? p2.y : c1y;
                    c1z = (c1z < p2.z) ? p2.z : c1z;
                }
            }
            float ax = (c1x < c0x) ? c1x : c0x, ay = (c1y < c0y) ? c1y : c0y, az = (c1z < c0z) ? c1z : c0z;
            float bx = (c0x > c1x) ? c0x : c1x, by = (c0y > c1y) ? c0y : c1y, bz = (c0z > c1z) ? c0z : c1z;
            const bool round = type == ptCylinder || type == ptSphere || type == ptCone;
            const float sy = round ? size.x : size.y, sz = round ? size.x : size.z;
            ax -= size.x;
            ay -= sy;
            az -= sz;
            bx += size.x;
            by += sy;
            bz += sz;
            if (ax < lx) lx = ax;
            if (ay < ly) ly = ay;
            if (az < lz) lz = az;
            if (bx > hx) hx = bx;
            if (by > hy) hy = by;
            if (bz > hz) hz = bz;
        }
    }
    else
    {
        lx = ly = lz = seed;
        hx = hy = hz = -seed;
        for (int c = node + 1; c < node + skip;)
        {
            const float4 a = rows[2 * c], b = rows[2 * c + 1];
            if (lx > a.x) lx = a.x;
            if (ly > a.y) ly = a.y;
            if (lz > a.z) lz = a.z;
            if (hx < b.x) hx = b.x;
            if (hy < b.y) hy = b.y;
            if (hz < a.w) hz = a.w;
            const int s = __float_as_int(b.w);
            c += (s > 1) ? s : 1;
        }
    }
    rows[2 * node] = make_float4(lx, ly, lz, hz);
    rows[2 * node + 1] = make_float4(hx, hy, row1.z, row1.w);
}

struct DeviceBuffer
{
    void *ptr = nullptr;
    size_t bytes = 0;
};

/* frames in flight at most (per-pixel buffer sets and streams).  Whole 1080p frames gain nothing beyond three, a
 * 1/8 strip - one round of waves, as slow as its longest - up to four; six and eight were tried (the mesh's
 * slowest strip: 0.114 ms with three, 0.089 with four, 0.12 and 0.11 with six and eight). */
const int MAX_FLIGHTS = 4;
/* head of the shared segment of solr_hip_image_share; the images follow, page-aligned.  done[r][slot]: the serial of
 * the last copy of rank r into that slot that has landed; consumed: the last serial the root has handed to its host. */
struct SharedRing
{
    std::atomic<long> done[64][MAX_FLIGHTS + 2];
    std::atomic<long> consumed;
    long frameBytes, imageStride;
};
struct Engine
{
    bool initialized = false;
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    int errorCode = 0;
    std::string errorText;

    /* scene planes */
    /* two arenas (scene_layout.h) and their host images */
    DeviceBuffer geometry, materials, textures, randoms, lamps;
    std::vector<float4> hostBoxes, hostBoxesCompact, hostPrims, hostLights;
    std::vector<int> hostBoxStart, hostBoxStartCompact, hostOriginCompact;
    int freeCountdown = 0; /* renders until the order-free lists are built (0: not scheduled) */
    /* the order-free list: the leaves of the scene under a surface-area hierarchy of our own (buildFreeOrderList) */
    std::vector<float4> hostBoxesFree;
    std::vector<int> hostBoxStartFree;
    /* lists built on the device stay there: `freeRows` float4 rows (16 per node of a list) that go into the arena with
     * a device-to-device copy (freeStage, until the next flushGeometry); the host images above are filled from the
     * arena when somebody needs them (ensureHostFreeLists: the refit plan of a rotated scene, a second layout) */
    size_t freeRows = 0;
    bool freeHostValid = true;
    bool freeDirty = false;  /* the staged lists are to be added to an arena that is otherwise up to date */
    unsigned rowsFixed = 0;  /* rows of the arena in front of the order-free lists */
    SolrDeviceLists freeStage;
    unsigned offBoxesFree = 0, offBoxStartFree = 0, offLeafFree = 0;
    int nbBoxesFree = 0;        /* nodes per list; there are eight, one per direction octant */
    bool freeStale = false;     /* rotated on the device since it was built: not refitted, not walked */
    bool primsContained = false; /* every primitive lies inside its leaf's box (retagPrimitives) */
    bool opaqueShadows = false;  /* no transparent primitive, no textured plane (retagPrimitives) */
    /* the thin copies of the walk-order list and of the order-free lists (tightenList; rt_device.h tightRay) */
    bool plainPlanes = false;    /* the scene holds a plain axis plane: thin copies are worth making (retagPrimitives) */
    float sceneExtent = 1.f;     /* max |coordinate| + |size| over the primitives, at least 1 */
    bool tightCompact = false, tightFree = false; /* the copy behind that list is up to date */
    /* bounce rays on the order-free lists, checked (rt_device.h closestHitWalk): -1 the engine decides per frame
     * (shortRayListsChoice: with frames in flight), 0 / 1 forced */
    int shortRayListsMode = -1;
    std::vector<int> materialTags; /* PRIM_* bits per material id */
    /* texture tables of the textured materials and the size of the uploaded atlas: checked against each other
     * before the first frame that follows either upload (checkTextureTables) */
    struct TextureUse
    {
        int material;
        long texels;      /* bytes of the diffuse map: x * y * depth */
        long offsets[7];  /* diffuse, normal, bump, specular, reflection, transparency, ambient occlusion; -1 unused */
    };
    std::vector<TextureUse> textureUses;
    size_t atlasBytes = 0;
    bool textureTablesChecked = false;
    std::vector<float> materialAverage; /* (r + g + b) / 3.f per material id (plane colour key, GI:561) */
    int sceneFeatures = F_ALL & ~F_FULL; /* rt_device.h enum Feature, recomputed with the tags */
    unsigned offBoxes = 0, offBoxesCompact = 0, offBoxStart = 0, offBoxStartCompact = 0, offPrims = 0, offLights = 0;
    unsigned offLeaf = 0, offLeafCompact = 0; /* leaf records of the two node lists (scene_layout.h) */
    unsigned offMatCold = 0;
    bool geometryDirty = true;
    int nbBoxesCompact = 0;
    int orderedExact = 0, orderedCompact = 0; /* sign-free slab test allowed on that node list */
    int nbBoxes = 0, nbPrimitives = 0, nbLights = 0, nbLamps = 0, nbMaterials = 0;
    int nested = 1;
    long nbRandoms = 0;

    /* per-pixel buffers of the strip */
    DeviceBuffer pp, ids, bitmap, counters, tileClock, tileCost, tileCostSnapshot, tileOrder;
    /* ambient occlusion across strips: the depths of the neighbours' rows next to this rank's strip */
    DeviceBuffer haloAbove[MAX_FLIGHTS], haloBelow[MAX_FLIGHTS], haloSendTop[MAX_FLIGHTS], haloSendBottom[MAX_FLIGHTS]; /* per frame in flight */
    DeviceBuffer haloGivenAbove, haloGivenBelow; /* solr_hip_set_depth_halo */
    int haloSuppliedAbove = 0, haloSuppliedBelow = 0; /* rows handed over by solr_hip_set_depth_halo (0: none) */
    float randomsReach = 0.f;                          /* max |randoms[i]|, i < 356: what the 256 taps can read */
    int haloWanted = -1; /* rows beyond a strip the last frame's post-processing reached (0: none; -1: no frame here) */
    /* Frames in flight (solr_hip_set_frames_in_flight): with n > 1, consecutive first-pass frames rotate
     * over n streams and n sets of per-pixel buffers, so that the tail of one frame - a few long waves
     * on an otherwise idle chip - overlaps the start of the next.  Set 0 is the members above. */
    int flights = 1;
    hipStream_t extraStream[MAX_FLIGHTS - 1] = {}; /* streams of sets 1 .. MAX_FLIGHTS - 1 */
    bool callerStreams = false; /* the streams belong to the caller (solr_hip_set_flight_streams) */
    DeviceBuffer ppX[MAX_FLIGHTS - 1], idsX[MAX_FLIGHTS - 1], bitmapX[MAX_FLIGHTS - 1], tileOrder2;
    int current = 0;           /* set / stream of the last render */
    unsigned frameSerial = 0;
    hipEvent_t orderEvent = nullptr; /* completion of the last tile sort */
    bool orderWait[MAX_FLIGHTS] = {}; /* that stream has not yet waited for it */
    int orderBuffer = 0;       /* which of tileOrder / tileOrder2 holds the valid order */
    /* cost-ordered launch: 0 off, 1 automatic (default), 2 always */
    int tileScheduling = 1;
    unsigned *hostStats = nullptr;    /* mapped host memory, 8 words */
    unsigned *hostStatsDev = nullptr; /* its device address */
    long costKey[6] = {0, 0, 0, 0, 0, 0}; /* the frame geometry the recorded costs belong to */
    int costFrames = 0;               /* frames rendered with that geometry */
    bool reorder = false;             /* current decision of the automatic mode */
    bool orderValid = false;          /* tileOrder holds an order for the current geometry */
    unsigned lastSerial = 0;
    bool tileClocks = false; /* diagnostics, solr_hip_enable_tile_clocks */
    int nbTilesTimed = 0;
    void *boundBitmap = nullptr;
    int width = 0, height = 0;       /* full image */
    int firstRow = 0, nbRows = -1;   /* strip; nbRows < 0 -> full frame, 0 -> this process renders no row */
    int allocW = 0, allocRows = 0;

    /* timing */
    int timing = 0; /* 0 off, n: every n-th launch is bracketed with events */
    unsigned timingTick = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double timedMs = 0.0;
    int timedLaunches = 0;
    std::vector<float> kernelSamples, intervalSamples; /* per timed launch: its duration; end-to-end gap to the one before */

    /* pipelined read-back (solr_hip_d2h_image_async): a ring of page-locked host images, a copy stream, and per
     * slot the event that says its copy has landed */
    static const int IMAGE_RING = MAX_FLIGHTS + 2; /* MAX_FLIGHTS tickets outstanding, the image on show, one spare */
    hipStream_t copyStream = nullptr;
    BitmapBuffer *pinnedImage[IMAGE_RING] = {};
    size_t pinnedBytes = 0;
    hipEvent_t imageDone[IMAGE_RING] = {};
    hipEvent_t frameRendered = nullptr;
    /* a ticket is (serial mod TICKET_PERIOD) * IMAGE_RING + slot - a positive int whatever the age of the process (the
     * serial itself is 64 bits, counts every ticket this process ever handed out and is never reset or reduced: 0.04 ms
     * per frame of an eight-rank job is 2^31 / 6 tickets in four hours) - and the serial tells a ticket whose slot has
     * been handed out again (or whose ring was re-allocated for a larger frame, or shared / unshared since) from a live
     * one: two tickets of one process are alike only 357 million tickets apart */
    static const long TICKET_PERIOD = ((long)0x7fffffff / IMAGE_RING / IMAGE_RING - 1) * IMAGE_RING;
    static int ticketOf(long serial, int slot) { return (int)((serial % TICKET_PERIOD) * IMAGE_RING + slot); }
    long imageSerial = 0;
    long slotSerial[IMAGE_RING] = {};
    /* A ring the ranks of a job share (solr_hip_image_share) is addressed by a sequence number of its own, counted
     * from the share on every rank alike (the ranks run the same program): it picks the slot and is what `done` /
     * `consumed` of the segment's head hold; the ticket's generation stays this process's own serial */
    long shareSeq = 0;
    long slotShareSeq[IMAGE_RING] = {};
    long lastWaitedSeq = 0;               /* sequence number of the newest ticket solr_hip_image_wait was asked for */
    long sharePublished[IMAGE_RING] = {}; /* the sequence number this rank has reported as landed, per slot */
    /* the ring in memory that several processes share (solr_hip_image_share): every rank's strip lands, over that
     * rank's own PCIe link, at its rows of ONE host image */
    struct SharedRing *sharedRing = nullptr;
    size_t sharedBytes = 0;
    std::string sharedName;
    int shareRank = 0, shareWorld = 0;
    bool slotOfStrips[IMAGE_RING] = {}; /* that slot's ticket was for every rank's strip (not the root's gathered frame) */
    long lastHandedOut = 0;             /* root: the serial of the image its last solr_hip_image_wait returned */
    bool copyOnRenderStream = false;    /* solr_hip_set_copy_route */
    /* the reciprocal of tilesX that was verified for a frame geometry (renderImpl) */
    int tileCheckedX = 0, tileCheckedTiles = 0, tileCheckedShift = 0;
    unsigned tileCheckedMagic = 0;
    /* every buffer set has a second RGB image ("side") for the time a copy still reads the first: a refinement or
     * accumulation pass stays on the set of the pass before it, and would otherwise wait for that pass's copy */
    DeviceBuffer bitmapAlt[MAX_FLIGHTS];
    DeviceBuffer deepStack[MAX_FLIGHTS]; /* F_STACK frames: the colour-stack slots beyond the LDS ones, per buffer set */
    int bitmapSide[MAX_FLIGHTS] = {0, 0, 0, 0};
    int flightCopy[MAX_FLIGHTS][2] = {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}}; /* slot whose copy reads that image, or -1 */

    /* device-side rotation (solr_hip_rotate_primitives): what to refit, in which order */
    DeviceBuffer movable, refitPlan;
    int nbMovable = -1;                 /* flags uploaded for that many primitives, -1: none */
    std::vector<int> refitLevels;       /* exact list: [offset, count] per height, offsets into refitPlan (ints) */
    std::vector<int> refitWalkLevels;   /* walk-order list, same form */
    std::vector<int> refitFreeLevels;   /* the eight order-free lists as one forest, same form */
    bool refitReady = false;
    bool refitPlanPending = false;      /* the lists changed: the plan is made when the first rotation asks (ensureRefitPlan) */
    std::vector<int> hostOriginFree;    /* per node of the order-free lists: the node of the reference's list it is, -1: ours */
    bool exactStale = false;            /* the exact list has not been refitted since the last rotation */
    float exactStaleViewDistance = 0.f;
    bool deviceAhead = false;           /* the arena has moved on from the host images */
    int nbDeviceRotations = 0;

    int variant = 0;
    bool grouping = true; /* groupSiblings(); variant 5 turns it off for A/B measurements */

    /* the walk's own ceiling (solr_hip_walk_bound): the next frame records its walks; how that frame was launched */
    DeviceBuffer walkRecords, walkVisits;
    bool recordNext = false;
    bool recorded = false;
    unsigned recordGrid = 0;
    size_t recordLds = 0;
    int recordVariant = -1; /* row of renderImpl's table */
    bool recordDeep = false;
    SceneArgs recordScene;
};

/* One Engine per device this process renders on.  The reference drives occupancyParameters.x devices from ONE host
 * thread - per-device allocations and uploads (CudaRayTracer.cu:1404-1480, 1536-1625), one launch per device on an
 * equal row strip (:1694-1696, 1709-1815), every device's strip copied to its place in the host arrays (:1647-1672) -
 * and so does this library when initialize_scene is handed occupancyParameters.x > 1: the ten entry points of the
 * boundary then run once per engine (the wrappers at the end of the C ABI), each engine on its own device with its
 * own streams, buffers and error state, the scene replicated, the frame shared out in equal row strips.  Engine 0
 * always exists and is the engine of every one-device process (all the multi-process machinery: strips, RCCL).
 * `g` is the engine a function works on. */
Engine gFirst;
Engine *gEngines[SOLR_MAX_GPU_COUNT] = {&gFirst};
int gDevices = 1;   /* engines in use since initialize_scene: min(occupancyParameters.x, devices visible) */
int gRequested = 1; /* occupancyParameters.x as initialize_scene was given it */
Engine *gCurrent = &gFirst;
#define g (*gCurrent)
template <class F>
void onEveryDevice(F &&f)
{
    for (int d = 0; d < gDevices; ++d)
    {
        gCurrent = gEngines[d];
        if (gDevices > 1)
            (void)hipSetDevice(g.device); /* (allocations and launches go to the calling thread's device) */
        f(d);
    }
    gCurrent = &gFirst;
    if (gDevices > 1)
        (void)hipSetDevice(g.device);
}

/* how many frames may really be in flight: what was asked for, as far as streams exist */
int activeFlights()
{
    if (g.flights < 2 || !(g.ownStream || g.callerStreams))
        return 1;
    int n = 1;
    while (n < g.flights && n < MAX_FLIGHTS && g.extraStream[n - 1])
        ++n;
    return n;
}
bool twoFlights() { return activeFlights() > 1; }
hipStream_t flightStream(int f) { return f ? g.extraStream[f - 1] : g.stream; }
DeviceBuffer &flightPp(int f) { return f ? g.ppX[f - 1] : g.pp; }
DeviceBuffer &flightIds(int f) { return f ? g.idsX[f - 1] : g.ids; }
DeviceBuffer &flightBitmap(int f) { return g.bitmapSide[f] ? g.bitmapAlt[f] : (f ? g.bitmapX[f - 1] : g.bitmap); }
/* nothing may touch scene or frame buffers while a frame is still in flight on the other stream */
void quiesce()
{
    for (hipStream_t extra : g.extraStream)
        if (extra)
            (void)hipStreamSynchronize(extra);
    if (g.stream)
        (void)hipStreamSynchronize(g.stream);
    if (g.copyStream)
        (void)hipStreamSynchronize(g.copyStream);
}

void setError(int code, const char *what, const char *file, int line)
{
    if (g.errorCode != 0)
        return;
    g.errorCode = code;
    char buf[512];
    snprintf(buf, sizeof(buf), "%s (%s:%d)", what, file, line);
    g.errorText = buf;
    fprintf(stderr, "solr_hip: error %d: %s\n", code, buf);
    const char *fatal = getenv("SOLR_HIP_FATAL");
    if (fatal && fatal[0] == '1')
        exit(EXIT_FAILURE); /* the reference's behaviour, helper_cuda.h:749-763 */
}

#define HIPCHECK(expr)                                                                                           \
    do                                                                                                           \
    {                                                                                                            \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
        {                                                                                                        \
            std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e_);                                  \
            setError((int)e_, m_.c_str(), __FILE__, __LINE__);                                                   \
        }                                                                                                        \
    } while (0)

#define ARGCHECK(cond, msg)                                                                                      \
    do                                                                                                           \
    {                                                                                                            \
        if (!(cond))                                                                                             \
            setError(-1, msg, __FILE__, __LINE__);                                                               \
    } while (0)

bool ok()
{
    return g.errorCode == 0;
}

bool ready(const char *who)
{
    if (!ok())
        return false;
    if (!g.initialized)
    {
        setError(-1, (std::string(who) + ": initialize_scene has not been called").c_str(), __FILE__, __LINE__);
        return false;
    }
    return true;
}

void release(DeviceBuffer &b)
{
    if (b.ptr)
        (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.bytes = 0;
}

/* grow-only device allocation */
void reserve(DeviceBuffer &b, size_t bytes)
{
    if (bytes < 16)
        bytes = 16;
    if (b.ptr && b.bytes >= bytes)
        return;
    release(b);
    HIPCHECK(hipMalloc(&b.ptr, bytes));
    if (ok())
        b.bytes = bytes;
}

template <class T>
void upload(DeviceBuffer &b, const std::vector<T> &host)
{
    reserve(b, host.size() * sizeof(T));
    if (ok() && !host.empty())
    {
        /* pageable source: the copy is complete for the caller when this returns */
        HIPCHECK(hipMemcpyAsync(b.ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, g.stream));
        HIPCHECK(hipStreamSynchronize(g.stream));
    }
}

inline int bitsi(float v)
{
    int i;
    memcpy(&i, &v, sizeof(i));
    return i;
}
inline float bitsf(int v)
{
    float f;
    memcpy(&f, &v, 4);
    return f;
}

int stripRows()
{
    return g.nbRows >= 0 ? g.nbRows : g.height;
}

void allocateFrame()
{
    const int rows = stripRows();
    const size_t pixels = (size_t)std::max(g.width, 1) * (size_t)std::max(rows, 1);
    const bool grow = pixels * sizeof(PostProcessingBuffer) > g.pp.bytes;
    reserve(g.pp, pixels * sizeof(PostProcessingBuffer));
    reserve(g.ids, pixels * sizeof(PrimitiveXYIdBuffer));
    reserve(g.bitmap, pixels * SOLR_COLOR_DEPTH);
#ifdef SOLR_TIMING
    if (!g.counters.ptr)
    {
        reserve(g.counters, (16 + 16 * SOLR_TIMING_SLOTS) * sizeof(unsigned long long));
        if (ok())
            HIPCHECK(hipMemset(g.counters.ptr, 0, g.counters.bytes));
    }
#else
    reserve(g.counters, 16 * sizeof(unsigned long long));
#endif
    const bool fresh = grow || g.allocW != g.width || g.allocRows != rows;
    if (ok() && fresh)
    {
        HIPCHECK(hipMemsetAsync(g.pp.ptr, 0, g.pp.bytes, g.stream));
        HIPCHECK(hipMemsetAsync(g.ids.ptr, 0, g.ids.bytes, g.stream));
        HIPCHECK(hipMemsetAsync(g.bitmap.ptr, 0, g.bitmap.bytes, g.stream));
    }
    if (ok() && g.flights >= 2 && (g.ownStream || g.callerStreams))
        for (int f = 1; f < g.flights && f < MAX_FLIGHTS; ++f)
        {
            if (!g.extraStream[f - 1])
            {
                if (g.callerStreams)
                    break; /* the caller gave fewer streams */
                HIPCHECK(hipStreamCreate(&g.extraStream[f - 1]));
            }
            const bool growX = pixels * sizeof(PostProcessingBuffer) > g.ppX[f - 1].bytes;
            reserve(g.ppX[f - 1], pixels * sizeof(PostProcessingBuffer));
            reserve(g.idsX[f - 1], pixels * sizeof(PrimitiveXYIdBuffer));
            reserve(g.bitmapX[f - 1], pixels * SOLR_COLOR_DEPTH);
            if (ok() && (fresh || growX))
            {
                HIPCHECK(hipMemsetAsync(g.ppX[f - 1].ptr, 0, g.ppX[f - 1].bytes, g.extraStream[f - 1]));
                HIPCHECK(hipMemsetAsync(g.idsX[f - 1].ptr, 0, g.idsX[f - 1].bytes, g.extraStream[f - 1]));
                HIPCHECK(hipMemsetAsync(g.bitmapX[f - 1].ptr, 0, g.bitmapX[f - 1].bytes, g.extraStream[f - 1]));
            }
        }
    g.allocW = g.width;
    g.allocRows = rows;
}

/* skip pointers must describe nested intervals for the ballot-only walk */
int validateNesting(const BoundingBox *boxes, int n)
{
    std::vector<int> ends;
    for (int i = 0; i < n; ++i)
    {
        const int skip = boxes[i].indexForNextBox.x;
        if (skip < 1 || (long)i + skip > n)
            return 0;
        while (!ends.empty() && ends.back() <= i)
            ends.pop_back();
        const int end = i + skip;
        if (!ends.empty() && end > ends.back())
            return 0;
        ends.push_back(end);
    }
    return 1;
}

/* join the material facts the walks need into every primitive's tag (scene_layout.h) */
int materialTag(const Material &m)
{
    int tag = 0;
    if (m.attributes.x == 0)
        tag |= PRIM_FAST0;
    if (m.attributes.x == 1)
        tag |= PRIM_FAST1;
    if (m.attributes.y != 0)
        tag |= PRIM_PROCEDURAL;
    if (m.transparency != 0.f)
        tag |= PRIM_TRANSPARENT;
    if (m.attributes.z == 1)
        tag |= PRIM_WIRE1;
    if (m.attributes.z == 2)
        tag |= PRIM_WIRE2;
    if (m.innerIllumination.x != 0.f)
        tag |= PRIM_EMISSIVE;
    if (m.textureIds.x != TEXTURE_NONE)
        tag |= PRIM_TEXTURED;
    int w = m.attributes.w;
    w = w < -1 ? -1 : (w > 100 ? 100 : w); /* wireFrameMapping compares X % 100 <= width */
    tag |= (w + 1) << PRIM_WIDTH_SHIFT;
    return tag;
}

/* What solr_hip_rotate_primitives refits and in which order: the nodes of a list by height, children
 * before parents.  A frame walks the walk-order list, so that is the one refitted with every rotation; the
 * reference's own list (box-debug view, census, variant 3, read-back) follows when somebody needs it
 * (refreshExactList) - node bounds are a function of the primitives alone, so late is as good as at once.
 * Both give a node of the reference's tree the same bounds: min / max over the level-0 boxes below it,
 * clamped once or several times by the same +-viewDistance seed, first occurrence winning a tie in either
 * nesting.  Node 0, the light cell, keeps its +-viewDistance (GPUKernel.cpp:1189). */
static void buildRefitPlan(const std::vector<float4> &exact, const std::vector<float4> &walk, const std::vector<int> &origin,
                           const std::vector<float4> &free, const std::vector<int> &freeOrigin)
{
    g.refitReady = false;
    g.exactStale = false;
    g.refitLevels.clear();
    g.refitWalkLevels.clear();
    g.refitFreeLevels.clear();
    if (!g.nested)
        return;
    auto heights = [](const std::vector<float4> &rows, std::vector<int> &height) {
        const int n = (int)(rows.size() / 2);
        height.assign(n, 0);
        /* nested skip pointers: a node's subtree is the nodes after it up to its skip; going backwards
         * every child is finished before its parent reads it */
        std::vector<int> parent(n, -1), stack;
        for (int i = 0; i < n; ++i)
        {
            while (!stack.empty() && i >= stack.back() + std::max(bitsi(rows[2 * stack.back() + 1].w), 1))
                stack.pop_back();
            parent[i] = stack.empty() ? -1 : stack.back();
            stack.push_back(i);
        }
        int top = 0;
        for (int i = n - 1; i >= 0; --i)
        {
            if (parent[i] >= 0)
                height[parent[i]] = std::max(height[parent[i]], height[i] + 1);
            top = std::max(top, height[i]);
        }
        return n ? top + 1 : 0;
    };
    std::vector<int> plan;
    auto byHeight = [&](const std::vector<int> &height, int nbHeights, std::vector<int> &levels, auto entry) {
        std::vector<std::vector<int>> bucket((size_t)nbHeights);
        for (int i = 0; i < (int)height.size(); ++i)
        {
            const long e = entry(i);
            if (e != -1)
                bucket[(size_t)height[i]].push_back((int)e);
        }
        for (const std::vector<int> &b : bucket)
            if (!b.empty())
            {
                levels.push_back((int)plan.size());
                levels.push_back((int)b.size());
                plan.insert(plan.end(), b.begin(), b.end());
            }
    };
    std::vector<int> height;
    int nbHeights = heights(exact, height);
    byHeight(height, nbHeights, g.refitLevels, [](int i) { return i != 0 ? (long)i : -1L; });
    nbHeights = heights(walk, height);
    byHeight(height, nbHeights, g.refitWalkLevels, [&](int j) {
        if (origin[j] == 0)
            return -1L;                                   /* the light cell */
        return origin[j] < 0 ? (long)(j | (int)0x80000000) : (long)j; /* sign bit: a grouping node */
    });
    /* the eight order-free lists, one behind the other: a forest with the same kinds of node (leaves of the
     * reference's tree, unions above them) */
    if (!free.empty())
    {
        nbHeights = heights(free, height);
        byHeight(height, nbHeights, g.refitFreeLevels, [&](int j) {
            if (freeOrigin[j] == 0)
                return -1L;
            return freeOrigin[j] < 0 ? (long)(j | (int)0x80000000) : (long)j;
        });
    }
    if (plan.empty())
        plan.push_back(0);
    upload(g.refitPlan, plan);
    g.refitReady = ok();
}

static void refitList(const std::vector<int> &levels, unsigned offNodes, unsigned offStart, float viewDistance)
{
    float4 *arena = (float4 *)g.geometry.ptr;
    const int *plan = (const int *)g.refitPlan.ptr;
    for (size_t l = 0; l + 1 < levels.size(); l += 2)
        hipLaunchKernelGGL(k_refitNodes, dim3((unsigned)((levels[l + 1] + 255) / 256)), dim3(256), 0, g.stream, arena,
                           offNodes, offStart, g.offPrims, plan + levels[l], levels[l + 1], viewDistance);
}

/* the reference's node list is wanted: refit it from the primitives as they are now */
static void refreshExactList()
{
    if (!g.exactStale || !g.geometry.ptr)
        return;
    quiesce();
    refitList(g.refitLevels, g.offBoxes, g.offBoxStart, g.exactStaleViewDistance);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(g.stream));
    g.exactStale = false;
}

/* the buffers the device builder left its lists in: rows and start indices until they are in the arena, the origins
 * (only the refit plan reads them) until the host has them or the lists go */
static void dropFreeStage(bool originToo)
{
    if (g.freeStage.rows)
        (void)hipFree(g.freeStage.rows);
    if (g.freeStage.start)
        (void)hipFree(g.freeStage.start);
    g.freeStage.rows = nullptr;
    g.freeStage.start = nullptr;
    if (originToo && g.freeStage.origin)
    {
        (void)hipFree(g.freeStage.origin);
        g.freeStage.origin = nullptr;
    }
}

/* host images of order-free lists that were built on the device: from where they are now */
static void ensureHostFreeLists()
{
    if (g.freeHostValid || !ok())
        return;
    quiesce();
    g.hostBoxesFree.resize(g.freeRows);
    g.hostBoxStartFree.resize(g.freeRows / 2);
    g.hostOriginFree.resize(g.freeRows / 2);
    const bool staged = g.freeStage.rows != nullptr;
    const char *arena = (const char *)g.geometry.ptr;
    if (!staged && !arena)
    {
        setError(-1, "order-free lists neither staged nor in the arena", __FILE__, __LINE__);
        return;
    }
    HIPCHECK(hipMemcpy(g.hostBoxesFree.data(), staged ? (const void *)g.freeStage.rows : arena + (size_t)g.offBoxesFree * 16, g.freeRows * 16,
                       hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(g.hostBoxStartFree.data(), staged ? (const void *)g.freeStage.start : arena + (size_t)g.offBoxStartFree * 4,
                       g.freeRows / 2 * 4, hipMemcpyDeviceToHost));
    if (g.freeStage.origin)
        HIPCHECK(hipMemcpy(g.hostOriginFree.data(), g.freeStage.origin, g.freeRows / 2 * 4, hipMemcpyDeviceToHost));
    if (ok())
    {
        g.freeHostValid = true;
        dropFreeStage(true); /* the next layout takes them from the host images */
    }
}

/* the arena moved on (device-side rotations): bring the host images up to date before anything reads them */
static void pullGeometry()
{
    if (!g.deviceAhead || !g.geometry.ptr)
        return;
    refreshExactList();
    quiesce();
    auto get = [&](unsigned at, void *dst, size_t bytes) {
        if (bytes)
            HIPCHECK(hipMemcpy(dst, (const char *)g.geometry.ptr + (size_t)at * 16, bytes, hipMemcpyDeviceToHost));
    };
    get(g.offBoxes, g.hostBoxes.data(), g.hostBoxes.size() * 16);
    get(g.offBoxesCompact, g.hostBoxesCompact.data(), g.hostBoxesCompact.size() * 16);
    if (g.freeHostValid)
        get(g.offBoxesFree, g.hostBoxesFree.data(), g.hostBoxesFree.size() * 16);
    get(g.offPrims, g.hostPrims.data(), g.hostPrims.size() * 16);
    g.deviceAhead = false;
}

void retagPrimitives()
{
    pullGeometry();
    const size_t n = g.hostPrims.size() / PRIM_ROWS;
    const bool noKinds = getenv("SOLR_HIP_NO_KINDS") != nullptr; /* tests: every primitive through the general tests */
    int features = 0;
    bool contained = true, opaque = true, planes = false;
    float extent = 1.f;
    for (size_t i = 0; i < n; ++i)
    {
        float4 *r = &g.hostPrims[PRIM_ROWS * i];
        for (int row : {(int)ROW_P0_TYPE, (int)ROW_P1_INDEX, (int)ROW_P2})
            for (float c : {r[row].x, r[row].y, r[row].z})
                if (fabsf(c) < 3.0e38f) /* (a comparison with NaN is false: the extent stays a number) */
                    extent = std::max(extent, fabsf(c));
        int tag, mat;
        memcpy(&tag, &r[ROW_P0_TYPE].w, 4);
        memcpy(&mat, &r[ROW_SIZE_MAT].w, 4);
        const int type = tag & PRIM_TYPE_MASK;
        /* a material that was never uploaded reads as all zeros on the device */
        const int facts = (mat >= 0 && (size_t)mat < g.materialTags.size()) ? g.materialTags[mat] : (PRIM_FAST0 | (1 << PRIM_WIDTH_SHIFT));
        int kind = KIND_GENERAL;
        if (noKinds || !(facts & PRIM_FAST0))
            kind = KIND_GENERAL; /* the closest-hit walk lets every lane of the leaf test a primitive with a kind */
        else if (type == ptSphere && !(facts & PRIM_PROCEDURAL))
            kind = KIND_SPHERE;
        else if ((type == ptXYPlane || type == ptYZPlane || type == ptXZPlane) && !(facts & (PRIM_TEXTURED | PRIM_WIRE2)) &&
                 !(type == ptYZPlane && (facts & PRIM_EMISSIVE)))
            kind = type == ptXYPlane ? KIND_PLANE_XY : (type == ptYZPlane ? KIND_PLANE_YZ : KIND_PLANE_XZ);
        else if (type == ptTriangle)
            kind = KIND_TRIANGLE;
        else if (type == ptCylinder || type == ptCone)
            kind = KIND_CYLINDER;
        r[ROW_P0_TYPE].w = bitsf(type | facts | (kind << PRIM_KIND_SHIFT));
        planes = planes || kind == KIND_PLANE_XY || kind == KIND_PLANE_YZ || kind == KIND_PLANE_XZ;
        /* inside the box the reference's builder gives its leaf (GPUKernel.cpp:762-830: the vertices of a triangle,
         * p0 +- radius of a sphere, min / max (p0, p1) +- radius of a cylinder, p0 +- size of a plane; a cone's box
         * is built around p0 alone, a procedural sphere's surface is displaced, the others are not bounded by
         * their size) */
        /* every occluder saturates a shadow (GI:880: intensity 1 x sceneInfo.shadowIntensity) unless it is transparent
         * (GI:881-892 scales and tints) or a textured plane (its texel's alpha is the intensity, GI:553-558) */
        opaque = opaque && !(facts & PRIM_TRANSPARENT) && !(facts & PRIM_TEXTURED) && type != ptCamera;
        contained = contained && (type == ptTriangle || type == ptCylinder || (type == ptSphere && !(facts & PRIM_PROCEDURAL)) ||
                                  type == ptXYPlane || type == ptYZPlane || type == ptXZPlane);
        r[ROW_P2].w = (mat >= 0 && (size_t)mat < g.materialAverage.size()) ? g.materialAverage[mat] : 0.f;
        switch (type)
        {
        case ptSphere:
        case ptEnvironment:
            features |= (facts & PRIM_PROCEDURAL) ? F_PROC : F_SPHERE;
            break;
        case ptCylinder:
        case ptCone:
            features |= F_CYL;
            break;
        case ptEllipsoid:
            features |= F_ELL;
            break;
        case ptTriangle:
            features |= F_TRI;
            break;
        case ptCamera:
            features |= F_PLANE | F_TEX;
            break;
        default:
            features |= F_PLANE;
            break;
        }
        if (facts & PRIM_TEXTURED)
            features |= F_TEX;
    }
    /* |p0| + |size| of the largest primitive, at least: the scale the thin leaves' margin is a 2^-10 of */
    float reach = 0.f;
    for (size_t i = 0; i < n; ++i)
        for (float c : {g.hostPrims[PRIM_ROWS * i + ROW_SIZE_MAT].x, g.hostPrims[PRIM_ROWS * i + ROW_SIZE_MAT].y,
                        g.hostPrims[PRIM_ROWS * i + ROW_SIZE_MAT].z})
            if (fabsf(c) < 3.0e38f)
                reach = std::max(reach, fabsf(c));
    g.sceneExtent = extent + reach;
    g.plainPlanes = planes;
    g.sceneFeatures = features;
    g.primsContained = contained && n > 0;
    g.opaqueShadows = opaque && n > 0;
    g.geometryDirty = true;
}

/* The thin copy of a node list behind it (rows offNodes + 2 n + 2 ...; rt_device.h tightRay): made where the scene has
 * plain axis planes at all and the list is short enough for an inner node's thread to read its whole subtree (the
 * room of a 100 k-triangle model keeps the reference's boxes).  false: there is no copy to walk. */
bool tightenList(unsigned offNodes, unsigned offStart, int nbNodes, int listLength)
{
    static const bool off = getenv("SOLR_HIP_NO_TIGHT_LEAVES") != nullptr;
    if (off || !g.plainPlanes || nbNodes <= 0 || listLength <= 0 || listLength > 65536 || !ok())
        return false;
    float4 *arena = (float4 *)g.geometry.ptr;
    const unsigned offTight = offNodes + 2u * (unsigned)nbNodes + 2u;
    const float margin = g.sceneExtent * (1.f / 1024.f);
    const dim3 grid((unsigned)((nbNodes + 255) / 256));
    hipLaunchKernelGGL(k_tightenLeaves, grid, dim3(256), 0, g.stream, arena, offNodes, offTight, offStart, g.offPrims, nbNodes, margin);
    hipLaunchKernelGGL(k_tightenInner, grid, dim3(256), 0, g.stream, arena, offTight, nbNodes, listLength);
    HIPCHECK(hipGetLastError());
    return ok();
}

/* the leaf records of both node lists from the primitive records as the arena holds them now */
void buildLeafRecords()
{
    if (!ok() || !g.geometry.ptr)
        return;
    float4 *arena = (float4 *)g.geometry.ptr;
    const int n = (int)(g.hostBoxes.size() / 2), nc = (int)(g.hostBoxesCompact.size() / 2);
    if (n > 0)
        hipLaunchKernelGGL(k_buildLeafRecords, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, arena, g.offBoxes,
                           g.offBoxStart, g.offPrims, g.offLeaf, n);
    if (nc > 0)
        hipLaunchKernelGGL(k_buildLeafRecords, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, g.stream, arena,
                           g.offBoxesCompact, g.offBoxStartCompact, g.offPrims, g.offLeafCompact, nc);
    const int nf = (int)(g.freeRows / 2);
    if (nf > 0 && !g.freeStale)
        hipLaunchKernelGGL(k_buildLeafRecords, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, g.stream, arena,
                           g.offBoxesFree, g.offBoxStartFree, g.offPrims, g.offLeafFree, nf);
    HIPCHECK(hipGetLastError());
    /* the thin copies follow the bounds and the primitives they were made from (an upload, a rotation on the device) */
    g.tightCompact = tightenList(g.offBoxesCompact, g.offBoxStartCompact, nc, nc);
    g.tightFree = nf > 0 && !g.freeStale && tightenList(g.offBoxesFree, g.offBoxStartFree, nf, nf / 8);
    HIPCHECK(hipStreamSynchronize(g.stream));
}

/* where the order-free lists go: behind everything else, so that they can be added to an arena that is laid out */
static unsigned layoutFreeLists(unsigned row)
{
    g.offBoxesFree = row;
    row += 2u * ((unsigned)g.freeRows + 2u); /* (one pad record, as behind every node list; then the thin copy, padded alike) */
    g.offBoxStartFree = row * 4;
    row += (unsigned)((g.freeRows / 2 + 3) / 4);
    row = (row + 3u) & ~3u; /* leaf records: one 64-byte line per node */
    g.offLeafFree = row;
    row += 2u * (unsigned)g.freeRows + 4u;
    return row;
}

/* the lists the device builder has just left (g.freeStage) into an arena that holds everything else already: what
 * is there stays where it is (moved to a larger allocation when this one is too small), nothing is uploaded again */
static void appendFreeLists()
{
    const unsigned end = layoutFreeLists(g.rowsFixed);
    const size_t bytes = (size_t)end * 16, fixedBytes = (size_t)g.rowsFixed * 16;
    PhaseTimer phase;
    if (g.geometry.bytes < bytes)
    {
        DeviceBuffer larger;
        reserve(larger, bytes);
        if (!ok())
            return;
        HIPCHECK(hipMemcpyAsync(larger.ptr, g.geometry.ptr, fixedBytes, hipMemcpyDeviceToDevice, g.stream));
        HIPCHECK(hipStreamSynchronize(g.stream));
        release(g.geometry);
        g.geometry = larger;
    }
    char *arena = (char *)g.geometry.ptr;
    HIPCHECK(hipMemsetAsync(arena + fixedBytes, 0, bytes - fixedBytes, g.stream));
    HIPCHECK(hipMemcpyAsync(arena + (size_t)g.offBoxesFree * 16, g.freeStage.rows, g.freeRows * 16, hipMemcpyDeviceToDevice, g.stream));
    HIPCHECK(hipMemcpyAsync(arena + (size_t)g.offBoxStartFree * 4, g.freeStage.start, g.freeRows / 2 * 4, hipMemcpyDeviceToDevice, g.stream));
    const int nf = (int)(g.freeRows / 2);
    if (ok() && nf > 0)
        hipLaunchKernelGGL(k_buildLeafRecords, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, g.stream, (float4 *)g.geometry.ptr,
                           g.offBoxesFree, g.offBoxStartFree, g.offPrims, g.offLeafFree, nf);
    HIPCHECK(hipGetLastError());
    g.tightFree = nf > 0 && tightenList(g.offBoxesFree, g.offBoxStartFree, nf, nf / 8);
    HIPCHECK(hipStreamSynchronize(g.stream));
    phase.mark("geometry: lists appended");
    if (ok())
    {
        dropFreeStage(false);
        g.freeDirty = false;
    }
}

/* assemble and upload the geometry arena from its host images (scene_layout.h) */
void flushGeometry()
{
    if (!g.geometryDirty)
    {
        if (g.freeDirty && g.freeStage.rows && g.geometry.ptr && g.rowsFixed > 0)
            appendFreeLists();
        else if (g.freeDirty)
            g.geometryDirty = true; /* (not the case this shortcut is for: everything again) */
        if (!g.geometryDirty)
            return;
    }
    pullGeometry();
    if (!g.freeStage.rows)
        ensureHostFreeLists(); /* laid out again from the host images: the lists too, then */
    auto rowsOfInts = [](size_t n) { return (unsigned)((n + 3) / 4); };
    unsigned row = 0;
    /* each node list is followed by one pad record: the walk requests the record after the node it tests
     * (rt_device.h advanceTidy), after the last node too */
    g.offBoxes = row;
    row += (unsigned)g.hostBoxes.size() + 2u;
    g.offBoxesCompact = row;
    row += 2u * ((unsigned)g.hostBoxesCompact.size() + 2u); /* ... and its thin copy (tightenList), padded alike */
    row = (row + 3u) & ~3u; /* primitive records start on a 64-byte line */
    g.offPrims = row;
    row += (unsigned)g.hostPrims.size();
    g.offLights = row;
    row += (unsigned)g.hostLights.size();
    const unsigned startRow = row;
    row += rowsOfInts(g.hostBoxStart.size());
    const unsigned startRowCompact = row;
    row += rowsOfInts(g.hostBoxStartCompact.size());
    g.offBoxStart = startRow * 4;
    g.offBoxStartCompact = startRowCompact * 4;
    row = (row + 3u) & ~3u; /* leaf records: one 64-byte line per node */
    g.offLeaf = row;
    row += 2u * (unsigned)g.hostBoxes.size() + 4u;
    g.offLeafCompact = row;
    row += 2u * (unsigned)g.hostBoxesCompact.size() + 4u;
    g.rowsFixed = row;
    row = layoutFreeLists(row);
    PhaseTimer phase;
    /* the pieces go straight to their rows of the arena (a staged host copy of the whole arena, zero-filled first,
     * took 10-14 ms for 100 k primitives); pad records and the leaf-record area start as zeros */
    reserve(g.geometry, (size_t)std::max(row, 1u) * 16);
    if (!ok())
        return;
    HIPCHECK(hipMemsetAsync(g.geometry.ptr, 0, (size_t)std::max(row, 1u) * 16, g.stream));
    auto put = [&](unsigned at, const void *src, size_t bytes) {
        if (bytes && ok())
            HIPCHECK(hipMemcpyAsync((char *)g.geometry.ptr + (size_t)at * 16, src, bytes, hipMemcpyHostToDevice, g.stream));
    };
    put(g.offBoxes, g.hostBoxes.data(), g.hostBoxes.size() * 16);
    put(g.offBoxesCompact, g.hostBoxesCompact.data(), g.hostBoxesCompact.size() * 16);
    put(g.offPrims, g.hostPrims.data(), g.hostPrims.size() * 16);
    put(g.offLights, g.hostLights.data(), g.hostLights.size() * 16);
    put(startRow, g.hostBoxStart.data(), g.hostBoxStart.size() * 4);
    put(startRowCompact, g.hostBoxStartCompact.data(), g.hostBoxStartCompact.size() * 4);
    if (g.freeStage.rows && ok())
    {
        HIPCHECK(hipMemcpyAsync((char *)g.geometry.ptr + (size_t)g.offBoxesFree * 16, g.freeStage.rows, g.freeRows * 16,
                                hipMemcpyDeviceToDevice, g.stream));
        HIPCHECK(hipMemcpyAsync((char *)g.geometry.ptr + (size_t)g.offBoxStartFree * 4, g.freeStage.start, g.freeRows / 2 * 4,
                                hipMemcpyDeviceToDevice, g.stream));
    }
    else
    {
        put(g.offBoxesFree, g.hostBoxesFree.data(), g.hostBoxesFree.size() * 16);
        put(g.offBoxStartFree / 4, g.hostBoxStartFree.data(), g.hostBoxStartFree.size() * 4);
    }
    HIPCHECK(hipStreamSynchronize(g.stream)); /* pageable sources: complete for the caller when this returns */
    if (ok())
        dropFreeStage(false);
    phase.mark("geometry: upload");
    buildLeafRecords();
    phase.mark("geometry: leaf records");
    if (ok())
    {
        g.geometryDirty = false;
        g.freeDirty = false;
    }
}

/* the order-free lists exist for the resident scene and every condition of their use holds (rt_device.h closestHitWalk) */
bool orderFreeListsUsable()
{
    return g.nbBoxesFree > 0 && g.freeRows == 16 * (size_t)g.nbBoxesFree && g.primsContained && !g.freeStale &&
           g.nested && g.orderedCompact && g.variant != 6;
}

/* bounce rays on the order-free lists: the API's word, else SOLR_HIP_SHORT_RAY_LISTS=0|1 (experiments), else the engine's
 * own choice for this frame */
bool shortRayListsChoice()
{
    static const int fromEnv = getenv("SOLR_HIP_SHORT_RAY_LISTS") ? atoi(getenv("SOLR_HIP_SHORT_RAY_LISTS")) : -1;
    const int mode = g.shortRayListsMode >= 0 ? g.shortRayListsMode : fromEnv;
    /* The engine's own choice.  Bounce rays on the order-free lists save work in nearly every tile and add some to the
     * few whose lanes have to be walked again (the mesh's horizon tiles: + 13 %).  With frames in flight the next frame
     * fills the chip behind those tiles and the saving is what shows (the mesh delivered 0.368 -> 0.356 ms, a 136-row
     * frame of it 0.239 -> 0.222); one frame at a time is as long as its longest tile and gets longer (0.43 -> 0.48 ms). */
    return mode < 0 ? activeFlights() >= 2 : mode != 0;
}

SceneArgs makeScene(bool exactNodes)
{
    SceneArgs S;
    memset(&S, 0, sizeof(S));
    S.geometry = g.geometry.ptr;
    S.materials = g.materials.ptr;
    S.textures = g.textures.ptr;
    S.randoms = g.randoms.ptr;
    S.offBoxes = exactNodes ? g.offBoxes : g.offBoxesCompact;
    S.offBoxStart = exactNodes ? g.offBoxStart : g.offBoxStartCompact;
    S.offLeaf = exactNodes ? g.offLeaf : g.offLeafCompact;
    S.offPrims = g.offPrims;
    S.offLights = g.offLights;
    S.offMatCold = g.offMatCold;
    S.nbBoxes = exactNodes ? g.nbBoxes : g.nbBoxesCompact;
    S.nbPrimitives = g.nbPrimitives;
    S.nbLights = g.nbLights;
    S.nbLamps = g.nbLamps;
    S.nested = g.nested;
    S.orderedBoxes = exactNodes ? g.orderedExact : g.orderedCompact;
    S.nbRandoms = g.randoms.ptr ? g.nbRandoms : 0;
    if (!exactNodes && orderFreeListsUsable())
    {
        S.offBoxesFree = g.offBoxesFree;
        S.offLeafFree = g.offLeafFree;
        S.nbBoxesFree = g.nbBoxesFree; /* per list; the eight lists and their leaf records lie one behind the other */
        S.opaqueShadows = g.opaqueShadows ? 1 : 0;
        S.shortRayLists = shortRayListsChoice() ? 1 : 0;
    }
    /* the thin copies behind the lists this frame walks (set by tightListsFor: they also depend on the frame) */
    S.tightLists = 0;
    return S;
}

/* may the walks of a frame with this SceneInfo take the thin copies of the lists S names (rt_device.h tightRay)? */
int tightListsFor(const SceneArgs &S, const SceneInfo &sceneInfo, bool exactNodes)
{
    if (exactNodes || g.variant == 8 || !g.tightCompact || !sceneInfo.extendedGeometry)
        return 0;
    if (S.nbBoxesFree > 0 && !g.tightFree)
        return 0;
    return (sceneInfo.viewDistance > 0.f && sceneInfo.viewDistance <= 64.f * g.sceneExtent) ? 1 : 0;
}

/* The texel fetch (rt_device.h fetchTexel, skyboxMapping) indexes the atlas with textureOffset + index % texels
 * and reads three bytes, for the diffuse map and, at the same index, for every secondary map of the
 * material.  The reference reads whatever lies there when the tables and the atlas disagree; on this
 * device that is a memory fault which ends the process's use of the GPU.  So the tables are checked against
 * the atlas once after either was uploaded, and a frame with a material that points outside is refused. */
void checkTextureTables()
{
    if (g.textureTablesChecked)
        return;
    g.textureTablesChecked = true;
    for (const Engine::TextureUse &use : g.textureUses)
    {
        ARGCHECK(use.texels > 0, "cudaRender: a textured material with an empty or negative texture mapping");
        ARGCHECK(g.textures.ptr != nullptr && g.atlasBytes > 0,
                 "cudaRender: textured materials but no texture atlas was uploaded (h2d_textures)");
        for (int t = 0; ok() && t < 7; ++t)
            if (use.offsets[t] >= 0 || t == 0)
                ARGCHECK(use.offsets[t] >= 0 && (size_t)(use.offsets[t] + use.texels + 2) <= g.atlasBytes,
                         "cudaRender: a material's texture table points outside the uploaded atlas");
        if (!ok())
        {
            g.textureTablesChecked = false; /* checked again once the caller has uploaded something else */
            return;
        }
    }
}

/* (defined with the list builders further down) */
void maybeBuildOrderFreeLists();

/* (defined with the RCCL layer at the end of this file) */
void exchangeDepthHalo(int flight, hipStream_t stream, const PixelRecord *pp, int W, int firstRow, int nbRows, int frameRows,
                       int wanted, DepthHalo *halo);
int agreedHaloRows(const PostProcessingInfo &ppInfo);
bool haveCommunicator();

/* A frame with the ambient-occlusion post-process on a rank of a communicator owes its neighbours the boundary rows
 * of its strip, whatever becomes of the frame on this rank: when renderImpl leaves before it got there (an argument
 * check, an error state, a strip it holds no row of), the exchange is posted with zeros on the way out. */
struct HaloDebt
{
    bool owed = false;
    int wanted = 0, width = 0, frameRows = 0;
    ~HaloDebt()
    {
        if (owed)
            exchangeDepthHalo(g.current, flightStream(g.current), nullptr, width, 0, 0, frameRows, wanted, nullptr);
    }
};

/* the features a frame of the resident scene needs (rt_device.h, enum Feature): decides the kernel instantiation */
int neededFeatures(const SceneInfo &sceneInfo, bool full)
{
    int need = g.sceneFeatures;
    if (!sceneInfo.extendedGeometry)
        /* every primitive is tested as a triangle, GI:743-747 - and textured as one (GI:916-931) */
        need = F_TRI | (g.sceneFeatures & F_TEX);
    if (full)
        need |= F_FULL;
    /* SOLR_HIP_FORCE_FEATURES=mask (experiments, rt_device.h enum Feature): as if the scene had these features too */
    static const int forced = getenv("SOLR_HIP_FORCE_FEATURES") ? atoi(getenv("SOLR_HIP_FORCE_FEATURES")) & F_ALL : 0;
    need |= forced;
    if (sceneInfo.skyboxMaterialId >= 0 && sceneInfo.skyboxMaterialId < (int)g.materialTags.size() &&
        (g.materialTags[sceneInfo.skyboxMaterialId] & PRIM_TEXTURED))
        need |= F_TEX;
    return need;
}

/* The neighbourhood post-processing of a frame - the switch of cudaRender, CRT:1857-1890 - behind the renderer on `stream`,
 * over buffer set `flight`.  (Also what the test-only solr_hip_probe_postprocess runs over a frame buffer of the caller's.) */
void launchPostProcess(const SceneInfo &sceneInfo, const PostProcessingInfo &ppInfo, int flight, hipStream_t stream, int firstRow,
                       int nbRows, unsigned char *bitmap, HaloDebt &debt)
{
    const int nbPixels = sceneInfo.size.x * nbRows;
    const dim3 pgrid((nbPixels + 255) / 256), pblock(256);
    if (ppInfo.type == ppe_ambientOcclusion)
    {
        /* a strip's taps reach into the rows of the ranks above and below: their depths come from the host
         * (solr_hip_set_depth_halo) or, with a communicator, from the neighbours over RCCL, on this stream */
        DepthHalo halo = {nullptr, nullptr, 0, 0};
        if (g.nbRows >= 0 && nbRows > 0)
        {
            const float reach = 16.f * fabsf(ppInfo.param2) * g.randomsReach / 10.f;
            const int wanted = debt.owed ? debt.wanted : (reach < 4096.f ? (int)reach + 2 : 4096);
            g.haloWanted = wanted;
            if (g.haloSuppliedAbove || g.haloSuppliedBelow)
            {
                halo.above = (const float *)g.haloGivenAbove.ptr;
                halo.below = (const float *)g.haloGivenBelow.ptr;
                halo.nbAbove = g.haloSuppliedAbove;
                halo.nbBelow = g.haloSuppliedBelow;
            }
            else if (debt.owed)
            {
                debt.owed = false;
                exchangeDepthHalo(flight, stream, (const PixelRecord *)flightPp(flight).ptr, sceneInfo.size.x, firstRow,
                                  nbRows, sceneInfo.size.y, wanted, &halo);
            }
        }
        /* the window the taps of this random buffer and this param2 can need (the kernel takes its own, exact reach and
         * gathers from memory if this should ever be too small): |tap| <= 16 |param2| max|random| / 10 */
        const float aoReach = 16.f * fabsf(ppInfo.param2) * g.randomsReach / 10.f;
        const int aoR = aoReach < 4096.f ? (int)aoReach + 3 : 4096;
        const long aoCells = (long)(AO_TILE_W + 2 * aoR) * (AO_TILE_H + 2 * aoR);
        const int aoWindow = (int)std::min<long>(std::max<long>(aoCells, 64), AO_WINDOW_FLOATS);
        if (ok())
            hipLaunchKernelGGL(k_ambientOcclusion,
                               dim3((((sceneInfo.size.x + AO_TILE_W - 1) / AO_TILE_W) * ((nbRows + AO_TILE_H - 1) / AO_TILE_H) +
                                     AO_TILES_PER_GROUP - 1) / AO_TILES_PER_GROUP),
                               pblock, (size_t)aoWindow * sizeof(float), stream, sceneInfo, ppInfo, nbRows,
                               (const PixelRecord *)flightPp(flight).ptr, (const float *)g.randoms.ptr,
                               g.randoms.ptr ? g.nbRandoms : 0L, bitmap, halo, firstRow, aoWindow);
    }
    else if (ppInfo.type == ppe_depthOfField)
        hipLaunchKernelGGL(k_depthOfField, pgrid, pblock, 0, stream, sceneInfo, ppInfo, nbRows,
                           (const PixelRecord *)flightPp(flight).ptr, (const float *)g.randoms.ptr,
                           g.randoms.ptr ? g.nbRandoms : 0L, bitmap);
    else if (ppInfo.type == ppe_radiosity)
        hipLaunchKernelGGL(k_radiosity, pgrid, pblock, 0, stream, sceneInfo, ppInfo, nbRows,
                           (const PixelRecord *)flightPp(flight).ptr, (const int4 *)flightIds(flight).ptr,
                           (const float *)g.randoms.ptr, g.randoms.ptr ? g.nbRandoms : 0L, bitmap);
    else if (ppInfo.type == ppe_filter)
        hipLaunchKernelGGL(k_filter, pgrid, pblock, 0, stream, sceneInfo, ppInfo, nbRows,
                           (const PixelRecord *)flightPp(flight).ptr, bitmap);
    else
        hipLaunchKernelGGL(k_cartoon, pgrid, pblock, 0, stream, sceneInfo, ppInfo, nbRows,
                           (const PixelRecord *)flightPp(flight).ptr, bitmap);
    HIPCHECK(hipGetLastError());
}

void renderImpl(const SceneInfo &sceneInfo, const vec4i &objects, const PostProcessingInfo &ppInfo,
                const float origin[3], const float direction[3], const float angles[4], bool counting,
                unsigned long long counts[8])
{
    HostSpan whole("cudaRender (whole)");
    HaloDebt debt;
    if (ppInfo.type == ppe_ambientOcclusion && haveCommunicator())
    {
        /* (every rank, before anything rank-local can end the call: an all-reduce when the figure is stale) */
        debt.wanted = agreedHaloRows(ppInfo);
        debt.width = sceneInfo.size.x;
        debt.frameRows = sceneInfo.size.y;
        debt.owed = !(g.haloSuppliedAbove || g.haloSuppliedBelow);
    }
    if (!ready("cudaRender"))
        return;
    ARGCHECK(sceneInfo.size.x > 0 && sceneInfo.size.y > 0, "cudaRender: empty image");
    ARGCHECK(objects.x <= g.nbBoxes && objects.y <= g.nbPrimitives, "cudaRender: more objects than were uploaded");
    ARGCHECK(objects.w <= g.nbLights, "cudaRender: more lights than were uploaded");
    ARGCHECK(g.materials.ptr != nullptr, "cudaRender: no materials uploaded");
    ARGCHECK(sceneInfo.skyboxMaterialId <= NB_MAX_MATERIALS, "cudaRender: skybox material beyond the material table");
    if (!ok())
        return;
    checkTextureTables();
    if (!ok())
        return;
    HIPCHECK(hipSetDevice(g.device));
    if (sceneInfo.size.x != g.width || sceneInfo.size.y != g.height)
    {
        g.width = sceneInfo.size.x;
        g.height = sceneInfo.size.y;
    }
    allocateFrame();
    if (!ok())
        return;
    if (stripRows() == 0)
        return; /* an empty strip (more processes than rows to share out): nothing to render */
    /* which stream / buffer set: first-pass frames alternate when two frames may be in flight; a
     * refinement or accumulation pass reads what the previous pass wrote and stays where that is */
    int flight = g.current;
    /* (the 3D-vision camera reads a depth of the frame before: it stays on one buffer set) */
    if (twoFlights() && !counting && sceneInfo.pathTracingIteration == 0 && sceneInfo.cameraType != ctVR)
        flight = (int)(g.frameSerial++ % (unsigned)activeFlights());
    else if (!twoFlights())
        flight = 0;
    const hipStream_t stream = flightStream(flight);
    g.current = flight;
    if (!g.boundBitmap && g.flightCopy[flight][g.bitmapSide[flight]] >= 0)
    {
        /* an asynchronous read-back (solr_hip_d2h_image_async) may still be reading the image this set rendered
         * last: this frame goes to the set's other image; only the copy of the frame before last - long done - is
         * waited for */
        const int side = g.bitmapSide[flight] ^ 1;
        reserve(g.bitmapAlt[flight], flightBitmap(flight).bytes);
        if (!ok())
            return;
        g.bitmapSide[flight] = side;
        if (g.flightCopy[flight][side] >= 0)
        {
            HIPCHECK(hipStreamWaitEvent(stream, g.imageDone[g.flightCopy[flight][side]], 0));
            g.flightCopy[flight][side] = -1;
        }
    }

    /* the box-debug view and the census count every node of the original tree */
    const bool full = sceneInfo.renderBoxes != 0 || sceneInfo.advancedIllumination == aiBasic ||
                      sceneInfo.advancedIllumination == aiFull || sceneInfo.cameraType == ctAntialiazed ||
                      sceneInfo.cameraType == ctAnaglyph || sceneInfo.cameraType == ctPanoramic ||
                      sceneInfo.cameraType == ctVR || sceneInfo.cameraType == ctVolumeRendering;
    /* (the volume camera keeps every hit, nearest first, ties in the order it met them: the reference's list) */
    const bool exactNodes = counting || sceneInfo.renderBoxes != 0 || objects.x != g.nbBoxes || g.variant == 3 ||
                            sceneInfo.cameraType == ctVolumeRendering;
    maybeBuildOrderFreeLists();
    flushGeometry();
    if (exactNodes)
        refreshExactList();
    if (!ok())
        return;
    SceneArgs S = makeScene(exactNodes);
    S.tightLists = tightListsFor(S, sceneInfo, exactNodes);
    if (exactNodes)
        S.nbBoxes = objects.x;
    S.nbPrimitives = objects.y;
    S.nbLamps = objects.z;
    S.nbLights = objects.w;

    FrameArgs F;
    memset(&F, 0, sizeof(F));
    F.si = sceneInfo;
    F.ppi = ppInfo;
    F.ox = origin[0];
    F.oy = origin[1];
    F.oz = origin[2];
    F.dx = direction[0];
    F.dy = direction[1];
    F.dz = direction[2];
    F.ax = angles[0];
    F.ay = angles[1];
    F.az = angles[2];
    F.aw = angles[3];
    {
        const float ratio = (float)sceneInfo.size.x / (float)sceneInfo.size.y;
        F.stepx = ratio * F.aw / (float)sceneInfo.size.x;
        F.stepy = F.aw / (float)sceneInfo.size.y;
    }
    /* VectorUtils.cuh:108-114 evaluates these per pixel; they are uniform */
    F.trig.cx = cosf(angles[0]);
    F.trig.cy = cosf(angles[1]);
    F.trig.cz = cosf(angles[2]);
    F.trig.sx = sinf(angles[0]);
    F.trig.sy = sinf(angles[1]);
    F.trig.sz = sinf(angles[2]);
    F.firstRow = g.nbRows >= 0 ? g.firstRow : 0;
    F.nbRows = stripRows();
    F.tilesX = (sceneInfo.size.x + TILE - 1) / TILE;
    const int tilesY = (F.nbRows + TILE - 1) / TILE;
    {
        /* the reciprocal of tilesX for the kernel's tile -> (column, row): exact for every tile of this frame
         * (round-up multiplier of ceil(log2) + 16 extra bits; verified below, once per frame geometry) */
        const int tiles = F.tilesX * tilesY;
        if (g.tileCheckedX != F.tilesX || g.tileCheckedTiles < tiles)
        {
            /* shift = ceil(log2 tilesX) - 1: the multiplier ceil(2^(32 + shift) / tilesX) has 32 bits and is exact for
             * every index below 2^31; one tile per row (magic 0) needs no division */
            int shift = 0;
            while ((2 << shift) < F.tilesX)
                ++shift;
            const unsigned long long magic =
                F.tilesX == 1 ? 0ull : ((1ull << (32 + shift)) + (unsigned long long)F.tilesX - 1) / (unsigned long long)F.tilesX;
            bool exact = magic <= 0xffffffffull;
            for (int t = 0; t < tiles && exact && magic; ++t)
                exact = (int)(((unsigned long long)(unsigned)t * magic) >> (32 + shift)) == t / F.tilesX;
            ARGCHECK(exact, "cudaRender: no exact reciprocal for this frame width");
            if (!exact)
                return; /* (cannot happen below 2^31 tiles; nothing is cached, the next frame checks again) */
            g.tileCheckedX = F.tilesX;
            g.tileCheckedTiles = tiles;
            g.tileCheckedMagic = (unsigned)magic;
            g.tileCheckedShift = shift;
        }
        F.tileMagic = g.tileCheckedMagic;
        F.tileShift = g.tileCheckedShift;
    }
    const bool neighbourhood = (ppInfo.type == ppe_ambientOcclusion || ppInfo.type == ppe_depthOfField ||
                                ppInfo.type == ppe_radiosity || ppInfo.type == ppe_filter || ppInfo.type == ppe_cartoon);
    unsigned char *bitmap = (unsigned char *)(g.boundBitmap ? g.boundBitmap : flightBitmap(flight).ptr);
    F.fuseDefault = neighbourhood ? 0 : 1;

    int maxIt = (sceneInfo.graphicsLevel < glReflectionsAndRefractions)
                    ? 1
                    : sceneInfo.nbRayIterations + sceneInfo.pathTracingIteration;
    maxIt = maxIt > NB_MAX_ITERATIONS ? NB_MAX_ITERATIONS : maxIt;
    maxIt = maxIt < 1 ? 1 : maxIt;
    F.stackSlots = maxIt;
    if (sceneInfo.cameraType == ctVolumeRendering)
        F.stackSlots = 11; /* the ten layers of launchVolumeRendering and the element behind them */
    if (sceneInfo.cameraType == ctVR)
    {
        /* the focus pixel of k_3DVisionRenderer (CRT:973, integer expression as written there) as the frame
         * before left it; a strip that does not hold it reads 0 */
        const long focusIndex = (long)(sceneInfo.size.x / 2 * sceneInfo.size.y / 2);
        const long focusRow = focusIndex / sceneInfo.size.x - F.firstRow;
        if (focusRow >= 0 && focusRow < F.nbRows && flightPp(flight).ptr)
        {
            const PostProcessingBuffer *at = (const PostProcessingBuffer *)flightPp(flight).ptr +
                                             focusRow * sceneInfo.size.x + focusIndex % sceneInfo.size.x;
            HIPCHECK(hipMemcpyAsync(&F.focusDepth, &at->colorInfo.w, sizeof(float), hipMemcpyDeviceToHost, stream));
            HIPCHECK(hipStreamSynchronize(stream));
        }
    }
    /* SOLR_HIP_LDS_PAD (bytes, experiments): more LDS per wave = fewer waves per SIMD; what occupancy is worth */
    static const size_t ldsPad = getenv("SOLR_HIP_LDS_PAD") ? (size_t)atol(getenv("SOLR_HIP_LDS_PAD")) : 0;
    size_t ldsBytes = ((size_t)F.stackSlots * 4 + COLD_FIELDS) * WAVE * sizeof(float) + ldsPad;

    const dim3 grid(F.tilesX * tilesY), block(WAVE);
    if (g.tileClocks)
    {
        reserve(g.tileClock, (size_t)grid.x * 2 * sizeof(unsigned long long));
        if (!ok())
            return;
        F.tileClock = (unsigned long long *)g.tileClock.ptr;
        g.nbTilesTimed = (int)grid.x;
    }
    if (g.tileScheduling > 0 && !counting)
    {
        const long key[6] = {(long)grid.x, F.tilesX, F.firstRow, F.nbRows, sceneInfo.size.x, sceneInfo.size.y};
        if (!g.hostStats)
        {
            HIPCHECK(hipHostMalloc((void **)&g.hostStats, 8 * sizeof(unsigned), hipHostMallocMapped));
            if (ok())
            {
                memset(g.hostStats, 0, 8 * sizeof(unsigned));
                HIPCHECK(hipHostGetDevicePointer((void **)&g.hostStatsDev, g.hostStats, 0));
            }
        }
        if (memcmp(key, g.costKey, sizeof(key)) != 0 || !g.tileCost.ptr)
        {
            memcpy(g.costKey, key, sizeof(key));
            g.costFrames = 0;
            g.reorder = false;
            g.orderValid = false;
            /* none of them is read before a sort has written it; a fresh allocation still gets a defined
             * content (a buffer that is kept may be in use by a frame in flight and is left alone) */
            for (DeviceBuffer *b : {&g.tileCost, &g.tileCostSnapshot, &g.tileOrder, &g.tileOrder2})
            {
                const void *before = b->ptr;
                reserve(*b, ((size_t)grid.x + (SPLIT_PARTS - 1) * SPLIT_TILES_MAX) * sizeof(unsigned));
                if (ok() && b->ptr != before)
                    HIPCHECK(hipMemset(b->ptr, 0, b->bytes));
            }
            g.orderBuffer = 0;
            for (bool &w : g.orderWait)
                w = false;
        }
        if (!ok())
            return;
        /* decision of the automatic mode from the newest frame the host can see (no synchronisation:
         * the figures are one or two frames old, which is as good for a scheduling hint) */
        if (g.costFrames > 0 && g.hostStats[4] != 0 && g.hostStats[3] == grid.x)
        {
            const unsigned long long sum = (unsigned long long)g.hostStats[1] | ((unsigned long long)g.hostStats[2] << 32);
            const unsigned long long mx = g.hostStats[0];
            if (mx * grid.x > 2ull * sum)
                g.reorder = true;
            else if (2ull * mx * grid.x < 3ull * sum)
                g.reorder = false;
        }
        F.tileCost = (unsigned *)g.tileCost.ptr;
        /* statistics (and, in cost order, a fresh order) every sixteenth frame, and at once when the
         * decision has just changed; in between the last order is reused */
        const bool ordered = g.costFrames > 0 && (g.tileScheduling == 2 || g.reorder);
        const bool refresh = g.costFrames > 0 && (g.costFrames % 16 == 1 || (ordered && !g.orderValid));
        const bool sort = ordered && refresh;
        if (!ordered)
            g.orderValid = false;
        if (refresh)
        {
            /* a new order goes to the buffer no frame in flight is reading; the other stream waits for
             * the sort before its next frame picks that buffer up */
            const int target = sort ? (g.orderBuffer ^ 1) : g.orderBuffer;
            DeviceBuffer &orderOut = target ? g.tileOrder2 : g.tileOrder;
            hipLaunchKernelGGL(k_orderTiles, dim3(1), dim3(1024), 0, stream, (const unsigned *)g.tileCost.ptr,
                               (unsigned *)g.tileCostSnapshot.ptr, (unsigned *)orderOut.ptr, (int)grid.x, (volatile unsigned *)g.hostStatsDev,
                               sort ? activeFlights() : 0);
            HIPCHECK(hipGetLastError());
            if (sort)
            {
                g.orderValid = true;
                g.orderBuffer = target;
                if (twoFlights())
                {
                    if (!g.orderEvent)
                        HIPCHECK(hipEventCreateWithFlags(&g.orderEvent, hipEventDisableTiming));
                    if (ok())
                        HIPCHECK(hipEventRecord(g.orderEvent, stream));
                    for (int f = 0; f < MAX_FLIGHTS; ++f)
                        g.orderWait[f] = (f != flight);
                }
            }
        }
        if (g.orderWait[flight] && g.orderEvent)
        {
            HIPCHECK(hipStreamWaitEvent(stream, g.orderEvent, 0));
            g.orderWait[flight] = false;
        }
        if (ordered && g.orderValid)
            F.tileOrder = (const unsigned *)(g.orderBuffer ? g.tileOrder2.ptr : g.tileOrder.ptr);
        g.costFrames++;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g.timing > 0 && !counting && (g.timingTick++ % (unsigned)g.timing) == 0)
    {
        HIPCHECK(hipEventCreate(&e0));
        HIPCHECK(hipEventCreate(&e1));
        HIPCHECK(hipEventRecord(e0, stream));
    }
    PixelRecord *ppPtr = (PixelRecord *)flightPp(flight).ptr;
    int4 *idPtr = (int4 *)flightIds(flight).ptr;
    unsigned long long *cntPtr = (unsigned long long *)g.counters.ptr;
    /* smallest instantiation that covers the scene (rt_device.h, enum Feature) */
    const int need = neededFeatures(sceneInfo, full);
    typedef RendererFn KernelFn;
    /* the rows of the table, smallest first: feature masks of the instantiations csrc/rows/ holds (renderer.h); the
     * lean ones exist with the two-bank and the three-bank walk loop (rt_device.h advanceTidy), the others with the
     * three-bank loop only */
    static const struct
    {
        int features;
        bool bothLoops;
    } variants[] = {
        {F_SPHERE | F_PLANE, true},
        {F_SPHERE | F_TRI, true},
        {F_SPHERE | F_CYL, true},
        {F_SPHERE | F_PLANE | F_TRI | F_CYL, false},
        /* textured scenes of the usual primitives (OBJ meshes with their MTL images; a textured room): the texture tier
         * without the procedural spheres and the ellipsoids */
        {F_SPHERE | F_TRI | F_TEX, true},
        {F_SPHERE | F_PLANE | F_TRI | F_CYL | F_TEX, false},
        /* the special cameras, global illumination and the box-debug view over the usual untextured primitives (the
         * texture tier is what costs the registers: profiles/r3/generic_kernels.txt) */
        {F_SPHERE | F_PLANE | F_TRI | F_CYL | F_FULL, false},
        {F_ALL & ~F_FULL, false},
        {F_ALL, false},
    };
    /* a list of more than a thousand nodes does not live in the scalar cache: skips land on cold records */
    const bool deepList = S.nbBoxes > 1024;
    const bool volumeCamera = sceneInfo.cameraType == ctVolumeRendering;
    KernelFn fn = solrrows::renderer(1, F_ALL, volumeCamera);
    int deepSlots = 0; /* colour-stack slots of this frame kept in HBM (F_STACK) */
    if (!counting)
    {
        fn = solrrows::renderer(0, F_ALL | F_DEEP, volumeCamera);
        int row = 0, chosen = -1;
        for (const auto &v : variants)
        {
            if ((need & ~v.features) == 0 && g.variant != 4 && !volumeCamera)
            {
                const int mask = v.features | ((deepList || !v.bothLoops) ? F_DEEP : 0);
                fn = solrrows::renderer(0, mask, false);
                /* more bounces than colour-stack slots fit the LDS of 16 waves per CU: the lean rows have an
                 * instantiation that keeps the deeper slots in HBM (rt_device.h ColorStack, F_STACK) */
                if (maxIt > SOLR_LDS_STACK_SLOTS && !g.recordNext && g.variant != 7)
                    if (KernelFn spilling = solrrows::renderer(0, mask | F_STACK, false))
                    {
                        fn = spilling;
                        deepSlots = maxIt - SOLR_LDS_STACK_SLOTS;
                    }
                chosen = row;
                break;
            }
            ++row;
        }
        g.recordVariant = chosen;
    }
    ARGCHECK(fn != nullptr, "cudaRender: no instantiation of the renderer for this scene (csrc/rows)");
    if (!ok())
        return;
    else
        HIPCHECK(hipMemsetAsync(g.counters.ptr, 0, 8 * sizeof(unsigned long long), stream));
    /* the ordered launch has a fixed number of extra workgroups for the quadrant waves of split tiles
     * (k_orderTiles); the ones the order does not use return at once */
    F.nbTiles = (int)grid.x;
    const dim3 launchGrid(F.tileOrder ? grid.x + (unsigned)(SPLIT_PARTS - 1) * SPLIT_TILES_MAX : grid.x);
    if (g.recordNext && !counting)
    {
        /* this frame leaves a record of its walks (rt_device.h recordWalk; solr_hip_walk_bound): the same kernel with
         * COUNT == 2, launched exactly as it would have been - grid, order, LDS - with the record buffer in place of the
         * counters.  Only the lean rows of the table have such an instantiation. */
        g.recordNext = false;
        ARGCHECK(g.recordVariant >= 0 && g.recordVariant < 4,
                 "solr_hip_walk_bound: the kernel this scene needs has no recording instantiation (untextured spheres, "
                 "planes, triangles, cylinders only)");
        if (!ok())
            return;
        reserve(g.walkRecords, (size_t)launchGrid.x * SOLR_WALK_SLOT_BYTES);
        reserve(g.walkVisits, (size_t)launchGrid.x * WAVE * sizeof(unsigned) + 64);
        if (!ok())
            return;
        HIPCHECK(hipMemsetAsync(g.walkRecords.ptr, 0, (size_t)launchGrid.x * SOLR_WALK_SLOT_BYTES, stream));
        fn = solrrows::renderer(2, variants[g.recordVariant].features | ((deepList || g.recordVariant == 3) ? F_DEEP : 0), false);
        ARGCHECK(fn != nullptr, "solr_hip_walk_bound: no recording instantiation");
        if (!ok())
            return;
        cntPtr = (unsigned long long *)g.walkRecords.ptr;
        g.recordGrid = launchGrid.x;
        g.recordLds = ldsBytes;
        g.recordDeep = deepList || g.recordVariant == 3;
        g.recordScene = S;
        g.recorded = true;
    }
    if (deepSlots > 0)
    {
        /* an F_STACK instantiation: SOLR_LDS_STACK_SLOTS slots in LDS - 16 waves per CU whatever the bounce limit - and
         * the rest of this buffer set's frame in HBM, a plane of the strip per slot (3840 x 2160 x 7 slots: 0.9 GB of
         * the 288; touched only by the rays that go that deep) */
        F.stackSlots = SOLR_LDS_STACK_SLOTS;
        ldsBytes = ((size_t)F.stackSlots * 4 + COLD_FIELDS) * WAVE * sizeof(float) + ldsPad;
        F.deepStride = (long)sceneInfo.size.x * F.nbRows;
        reserve(g.deepStack[flight], (size_t)deepSlots * (size_t)F.deepStride * sizeof(float4));
        if (!ok())
            return;
        F.deepStack = (float4 *)g.deepStack[flight].ptr;
    }
    {
        HostSpan launch("  of which the kernel launch");
        hipLaunchKernelGGL(fn, launchGrid, block, ldsBytes, stream, S, F, ppPtr, idPtr, bitmap, cntPtr);
    }
    HIPCHECK(hipGetLastError());
    if (e0)
    {
        HIPCHECK(hipEventRecord(e1, stream));
        g.events.push_back(std::make_pair(e0, e1));
    }

    g.haloWanted = 0;
    if (neighbourhood)
        launchPostProcess(sceneInfo, ppInfo, flight, stream, F.firstRow, F.nbRows, bitmap, debt);

    if (counting && counts)
    {
        HIPCHECK(hipMemcpyAsync(counts, g.counters.ptr, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                stream));
        HIPCHECK(hipStreamSynchronize(stream));
    }
}

void releaseImageRing()
{
    if (g.copyStream)
        (void)hipStreamSynchronize(g.copyStream);
    for (int i = 0; i < Engine::IMAGE_RING; ++i)
    {
        if (g.pinnedImage[i] && !g.sharedRing)
            (void)hipHostFree(g.pinnedImage[i]);
        g.pinnedImage[i] = nullptr;
        if (g.imageDone[i])
            (void)hipEventDestroy(g.imageDone[i]);
        g.imageDone[i] = nullptr;
    }
    if (g.frameRendered)
        (void)hipEventDestroy(g.frameRendered);
    g.frameRendered = nullptr;
    if (g.copyStream)
        (void)hipStreamDestroy(g.copyStream);
    g.copyStream = nullptr;
    g.pinnedBytes = 0;
    for (long &serial : g.slotSerial)
        serial = -1;
    if (g.sharedRing)
    {
        (void)hipHostUnregister(g.sharedRing);
        (void)munmap(g.sharedRing, g.sharedBytes);
        if (g.shareRank == 0 && !g.sharedName.empty())
            (void)shm_unlink(g.sharedName.c_str());
        g.sharedName.clear();
        g.sharedRing = nullptr;
        g.sharedBytes = 0;
    }
    for (int f = 0; f < MAX_FLIGHTS; ++f)
    {
        g.flightCopy[f][0] = g.flightCopy[f][1] = -1;
        g.bitmapSide[f] = 0;
        release(g.bitmapAlt[f]);
    }
}

void collectEvents()
{
    hipEvent_t before = nullptr;
    for (auto &ev : g.events)
    {
        float ms = 0.f, gap = 0.f;
        if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess)
        {
            g.timedMs += ms;
            g.timedLaunches++;
            if (g.kernelSamples.size() < 65536)
            {
                g.kernelSamples.push_back(ms);
                /* end of the launch before to the end of this one: what a step of a pipelined loop takes */
                g.intervalSamples.push_back((before && hipEventElapsedTime(&gap, before, ev.second) == hipSuccess) ? gap : -1.f);
            }
        }
        if (before)
            (void)hipEventDestroy(before);
        (void)hipEventDestroy(ev.first);
        before = ev.second;
    }
    if (before)
        (void)hipEventDestroy(before);
    g.events.clear();
}
} // namespace

/* For csrc/solr_probes.hip (the test-only entry points of include/solr_hip_probes.h): the resident scene exactly as
 * renderImpl hands it to the renderer - pending uploads flushed, the order-free lists built when they are due - the
 * features a frame with this SceneInfo needs, whether the renderer would take the three-bank node loop, and the
 * engine's stream.  exactNodes: the reference's own node list instead of the walk-order list.  Returns 0, or -1 with
 * the engine's error set. */
namespace solrprobe
{
int residentScene(const SceneInfo &sceneInfo, bool exactNodes, SceneArgs *S, int *features, int *deepList, hipStream_t *stream)
{
    if (!ready("solr_hip_probe"))
        return -1;
    quiesce();
    HIPCHECK(hipSetDevice(g.device));
    ARGCHECK(g.materials.ptr != nullptr, "solr_hip_probe: no materials uploaded");
    if (!ok())
        return -1;
    checkTextureTables();
    maybeBuildOrderFreeLists();
    flushGeometry();
    if (exactNodes)
        refreshExactList();
    if (!ok())
        return -1;
    *S = makeScene(exactNodes);
    S->tightLists = tightListsFor(*S, sceneInfo, exactNodes);
    *features = neededFeatures(sceneInfo, false);
    *deepList = S->nbBoxes > 1024;
    *stream = flightStream(0);
    return 0;
}
void fail(int code, const char *what) { setError(code, what, __FILE__, __LINE__); }

/* the read-back ticket of the serial-th frame (no engine needed: plain arithmetic), and the serial counter itself, so
 * that a test can put a running engine a few frames before 2^31 / IMAGE_RING tickets and go across */
int ticketOfSerial(long long serial, int *slot, long long *period)
{
    const int s = (int)(serial % Engine::IMAGE_RING); /* (a ring of this process's own; a shared ring counts its slots itself) */
    if (slot)
        *slot = s;
    if (period)
        *period = Engine::TICKET_PERIOD;
    return Engine::ticketOf((long)serial, s);
}
long long imageSerial(long long setTo)
{
    if (setTo >= 0)
        gFirst.imageSerial = (long)setTo;
    return gFirst.imageSerial;
}

/* The post-processing stage of cudaRender (CRT:1857-1890) over a float frame buffer of the caller's: the buffer goes
 * into the engine's current buffer set, launchPostProcess - what renderImpl launches behind the renderer - runs over
 * it (type ppe_none: the stand-alone k_default, the conversion the renderer otherwise fuses into its epilogue), and
 * the RGB8 image comes back.  Whole frames only.  Returns 0, or -1 with the engine's error set. */
int postProcess(const SceneInfo &sceneInfo, const PostProcessingInfo &ppInfo, const PostProcessingBuffer *frame,
                unsigned char *bitmapOut)
{
    if (!ready("solr_hip_probe_postprocess"))
        return -1;
    quiesce();
    HIPCHECK(hipSetDevice(g.device));
    ARGCHECK(frame != nullptr && bitmapOut != nullptr, "solr_hip_probe_postprocess: no buffer");
    ARGCHECK(sceneInfo.size.x > 0 && sceneInfo.size.y > 0, "solr_hip_probe_postprocess: empty image");
    ARGCHECK(g.nbRows < 0 && gDevices == 1, "solr_hip_probe_postprocess: whole frames of one device only");
    if (!ok())
        return -1;
    g.width = sceneInfo.size.x;
    g.height = sceneInfo.size.y;
    allocateFrame();
    if (!ok())
        return -1;
    const int flight = g.current;
    const hipStream_t stream = flightStream(flight);
    const size_t pixels = (size_t)g.width * g.height;
    HIPCHECK(hipMemcpyAsync(flightPp(flight).ptr, frame, pixels * sizeof(PostProcessingBuffer), hipMemcpyHostToDevice, stream));
    unsigned char *bitmap = (unsigned char *)flightBitmap(flight).ptr;
    const bool neighbourhood = (ppInfo.type == ppe_ambientOcclusion || ppInfo.type == ppe_depthOfField ||
                                ppInfo.type == ppe_radiosity || ppInfo.type == ppe_filter || ppInfo.type == ppe_cartoon);
    if (neighbourhood)
    {
        HaloDebt nothingOwed;
        launchPostProcess(sceneInfo, ppInfo, flight, stream, 0, g.height, bitmap, nothingOwed);
    }
    else
    {
        hipLaunchKernelGGL(k_default, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, stream, sceneInfo, (int)pixels,
                           (const PixelRecord *)flightPp(flight).ptr, bitmap);
        HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipMemcpyAsync(bitmapOut, bitmap, pixels * SOLR_COLOR_DEPTH, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    return ok() ? 0 : -1;
}
} // namespace solrprobe

/* ======================================================================= */
/* C ABI                                                                    */
/* ======================================================================= */

extern "C" {

int solr_hip_last_error(char *buf, int len)
{
    /* the first engine in trouble speaks for the process (one engine unless occupancyParameters.x asked for more) */
    const Engine *bad = &gFirst;
    for (int d = 0; d < gDevices; ++d)
        if (gEngines[d] && gEngines[d]->errorCode != 0)
        {
            bad = gEngines[d];
            break;
        }
    if (buf && len > 0)
    {
        strncpy(buf, bad->errorText.c_str(), len - 1);
        buf[len - 1] = 0;
    }
    return bad->errorCode;
}

void solr_hip_clear_error(void)
{
    for (Engine *e : gEngines)
        if (e)
        {
            e->errorCode = 0;
            e->errorText.clear();
        }
}

int solr_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

/* (for the other translation units of the library: sol-r_amd/csrc/solr_tree.hip builds on the engine's device) */
int solr_hip_get_device(void)
{
    return g.device;
}

void solr_hip_set_device(int device)
{
    g.device = device;
}

void dropExtraStreams()
{
    for (hipStream_t &extra : g.extraStream)
    {
        if (extra && !g.callerStreams)
            (void)hipStreamDestroy(extra);
        extra = nullptr;
    }
    g.callerStreams = false;
}

void solr_hip_set_flight_streams(void *const *streams, int n)
{
    quiesce();
    g.current = 0;
    if (!streams || n < 1 || !streams[0])
        return;
    if (g.ownStream && g.stream)
        (void)hipStreamDestroy(g.stream);
    dropExtraStreams();
    g.ownStream = false;
    g.stream = (hipStream_t)streams[0];
    for (int f = 1; f < n && f < MAX_FLIGHTS; ++f)
        g.extraStream[f - 1] = (hipStream_t)streams[f];
    g.callerStreams = n > 1;
    if (g.initialized && g.width > 0)
        allocateFrame();
}

void solr_hip_set_stream(void *stream)
{
    quiesce();
    g.current = 0; /* a caller's stream is the only stream: one frame in flight */
    if (g.callerStreams)
        dropExtraStreams();
    if (g.ownStream && g.stream)
    {
        (void)hipStreamSynchronize(g.stream);
        (void)hipStreamDestroy(g.stream);
        g.ownStream = false;
    }
    g.stream = (hipStream_t)stream;
    if (!g.stream && g.initialized)
    {
        HIPCHECK(hipStreamCreate(&g.stream));
        g.ownStream = ok();
    }
}

static void synchronizeOne()
{
    if (!ready("solr_hip_synchronize"))
        return;
    HIPCHECK(hipStreamSynchronize(g.stream));
    for (hipStream_t extra : g.extraStream)
        if (extra)
            HIPCHECK(hipStreamSynchronize(extra));
}

void solr_hip_set_strip(int firstRow, int nbRows)
{
    if (gDevices > 1)
    {
        /* the frame is already shared out between this process's devices (occupancyParameters.x): a strip of the
         * multi-process split on top of that is a different program */
        setError(-1, "solr_hip_set_strip: this process renders on several devices (occupancyParameters.x > 1); strips "
                     "belong to the one-process-per-GPU model", __FILE__, __LINE__);
        return;
    }
    quiesce();
    g.firstRow = nbRows >= 0 ? firstRow : 0;
    g.nbRows = nbRows >= 0 ? nbRows : -1;
    if (g.initialized && g.width > 0)
        allocateFrame();
}

void *solr_hip_device_bitmap(void)
{
    return g.boundBitmap ? g.boundBitmap : flightBitmap(g.current).ptr;
}
void *solr_hip_device_primitive_ids(void)
{
    return flightIds(g.current).ptr;
}
void *solr_hip_device_postprocessing(void)
{
    return flightPp(g.current).ptr;
}
/* the strip this process renders now (solr_hip_set_strip, solr_hip_balance_strips): rows [*firstRow, *firstRow +
 * *nbRows) of the frame; the full frame reads as (0, height of the last frame or 0 before one) */
void solr_hip_get_strip(int *firstRow, int *nbRows)
{
    if (firstRow)
        *firstRow = g.nbRows >= 0 ? g.firstRow : 0;
    if (nbRows)
        *nbRows = stripRows();
}

void solr_hip_bind_device_bitmap(void *deviceBitmap)
{
    g.boundBitmap = deviceBitmap;
}

static void initializeOne(const SceneInfo &sceneInfo)
{
    if (!ok())
        return;
    solrTuneHostAllocator();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
    {
        setError(e != hipSuccess ? (int)e : -1,
                 "initialize_scene: no HIP device available (this engine has no CPU fallback)", __FILE__, __LINE__);
        return;
    }
    ARGCHECK(g.device >= 0 && g.device < n, "initialize_scene: device index out of range");
    if (!ok())
        return;
    HIPCHECK(hipSetDevice(g.device));
    if (!g.stream)
    {
        HIPCHECK(hipStreamCreate(&g.stream));
        g.ownStream = ok();
        /* the other streams right away: streams are dealt to the hardware queues in creation order, and
         * streams that share a hardware queue do not overlap (measured: created after a framework had
         * made its pool of 32, both engine streams sat on one queue and frames in flight gained nothing) */
        for (int f = 1; f < MAX_FLIGHTS && ok(); ++f)
            if (!g.extraStream[f - 1])
                HIPCHECK(hipStreamCreate(&g.extraStream[f - 1]));
    }
    g.initialized = ok();
    g.width = sceneInfo.size.x;
    g.height = sceneInfo.size.y;
}


static void finalizeOne()
{
    gHostProfile.report();
    if (!g.initialized)
        return;
    (void)hipSetDevice(g.device);
    if (g.stream)
        (void)hipStreamSynchronize(g.stream);
    for (hipStream_t extra : g.extraStream)
        if (extra)
            (void)hipStreamSynchronize(extra);
    collectEvents();
    DeviceBuffer *all[] = {&g.geometry, &g.materials, &g.textures, &g.randoms, &g.lamps,
                           &g.pp,       &g.ids,       &g.bitmap,   &g.counters, &g.tileClock,
                           &g.tileCost, &g.tileCostSnapshot, &g.tileOrder, &g.tileOrder2, &g.movable,  &g.refitPlan,
                           &g.walkRecords, &g.walkVisits};
    for (DeviceBuffer *b : all)
        release(*b);
    for (int f = 0; f < MAX_FLIGHTS; ++f)
    {
        release(g.deepStack[f]);
        release(g.haloAbove[f]);
        release(g.haloBelow[f]);
        release(g.haloSendTop[f]);
        release(g.haloSendBottom[f]);
    }
    release(g.haloGivenAbove);
    release(g.haloGivenBelow);
    g.haloSuppliedAbove = g.haloSuppliedBelow = 0;
    for (int f = 0; f < MAX_FLIGHTS - 1; ++f)
    {
        release(g.ppX[f]);
        release(g.idsX[f]);
        release(g.bitmapX[f]);
    }
    dropExtraStreams();
    releaseImageRing();
    if (g.orderEvent)
        (void)hipEventDestroy(g.orderEvent);
    g.orderEvent = nullptr;
    g.current = 0;
    g.orderBuffer = 0;
    for (bool &w : g.orderWait)
        w = false;
    if (g.hostStats)
        (void)hipHostFree(g.hostStats);
    g.hostStats = g.hostStatsDev = nullptr;
    g.costFrames = 0;
    g.reorder = false;
    g.orderValid = false;
    memset(g.costKey, 0, sizeof(g.costKey));
    if (g.ownStream && g.stream)
        (void)hipStreamDestroy(g.stream);
    g.stream = nullptr;
    g.ownStream = false;
    g.initialized = false;
    g.refitReady = false;
    g.refitPlanPending = false;
    g.hostOriginFree.clear();
    g.exactStale = false;
    g.deviceAhead = false;
    g.nbMovable = -1;
    g.nbDeviceRotations = 0;
    g.nbBoxes = g.nbPrimitives = g.nbLights = g.nbLamps = g.nbMaterials = 0;
    g.allocW = g.allocRows = 0;
    g.boundBitmap = nullptr;
    g.hostBoxes.clear();
    g.hostBoxesCompact.clear();
    g.hostPrims.clear();
    g.hostLights.clear();
    g.hostBoxStart.clear();
    g.hostBoxStartCompact.clear();
    g.hostBoxesFree.clear();
    g.hostBoxStartFree.clear();
    g.freeRows = 0;
    g.freeHostValid = true;
    g.freeDirty = false;
    dropFreeStage(true);
    {
        SolrScratchPool &pool = solrScratchPool(); /* the builders' scratch goes with the scene */
        std::lock_guard<std::mutex> nobodyBuilding(pool.busy);
        pool.release();
    }
    g.hostOriginCompact.clear();
    g.nbBoxesFree = 0;
    g.freeCountdown = 0;
    g.freeStale = false;
    g.materialTags.clear();
    g.materialAverage.clear();
    g.geometryDirty = true;
    /* no hipDeviceReset: the process may share the device with torch/RCCL */
}

static void reshapeOne(const SceneInfo &sceneInfo)
{
    if (!ready("reshape_scene"))
        return;
    quiesce();
    g.width = sceneInfo.size.x;
    g.height = sceneInfo.size.y;
    allocateFrame();
}


/* Inner nodes that hardly ever cull are left out of the walk list.  An inner node - one of the reference's tree
 * whose children all lie inside it, or a grouping node, which is the union of its members - passes whenever one
 * of its children would (the argument of groupSiblings below, read the other way: slab values are monotonic in
 * the bounds, the cut-off only shrinks along a walk), so testing the children without it reaches the same
 * leaves in the same order.  What the node buys is the tests of its subtree for the rays that miss it; what it
 * costs is one test for those that do not.  A ray that is in the parent enters the node
 *   - because it starts there: the rays of a frame start on the geometry (and at the camera, which is in the
 *     room it looks at), so about the share of the parent's leaves whose centre lies in the node;
 *   - otherwise with the surface-area probability area(node) / area(parent).
 * The node stays if (1 - the larger of the two) x (nodes below it) is at least `threshold` tests.  Cornell's
 * upper cells and the groups around its walls hold every leaf centre of the room: they go, the groups of small
 * spheres on the floor stay.  Works on the walk-order rows in place; returns the new node count. */
static int pruneInnerNodes(std::vector<float4> &rows, std::vector<int> &start, std::vector<int> &origin, int *nbPruned,
                           bool everyInnerNode = false)
{
    const int n = (int)start.size();
    const double threshold = everyInnerNode ? 1e300 : (getenv("SOLR_HIP_PRUNE") ? atof(getenv("SOLR_HIP_PRUNE")) : 1.0);
    *nbPruned = 0;
    if (n < 2 || !(threshold > 0.0))
        return n;
    auto skipOf = [&](int i) { return std::max(bitsi(rows[2 * i + 1].w), 1); };
    auto countOf = [&](int i) { return bitsi(rows[2 * i + 1].z); };
    auto lo = [&](int i, int k) { return k == 0 ? rows[2 * i].x : (k == 1 ? rows[2 * i].y : rows[2 * i].z); };
    auto hi = [&](int i, int k) { return k == 0 ? rows[2 * i + 1].x : (k == 1 ? rows[2 * i + 1].y : rows[2 * i].w); };
    auto areaOf = [&](int i) {
        const double x = (double)hi(i, 0) - lo(i, 0), y = (double)hi(i, 1) - lo(i, 1), z = (double)hi(i, 2) - lo(i, 2);
        return x * y + y * z + z * x;
    };
    std::vector<char> keep(n, 1);
    /* the decisions: on the device (solr_lists.hip, one launch per depth of the list; the same arithmetic, the same
     * decisions) unless told otherwise or declined */
    int decided = -1;
    if (!everyInnerNode && g.initialized && !getenv("SOLR_HIP_LISTS_ON_HOST"))
        decided = solrPruneDecisionsOnDevice(rows.data(), n, threshold, keep, g.stream);
    if (decided >= 0)
        *nbPruned = decided;
    else
    {
        keep.assign(n, 1);
        std::vector<int> leaves; /* node indices of the leaves, in walk order */
        std::vector<int> leavesBefore(n + 1, 0);
        for (int i = 0; i < n; ++i)
        {
            leavesBefore[i + 1] = leavesBefore[i] + (countOf(i) > 0 ? 1 : 0);
            if (countOf(i) > 0)
                leaves.push_back(i);
        }
        struct Open
        {
            int node, end;
        };
        std::vector<Open> open; /* kept ancestors of node i */
        double sceneLo[3] = {1e300, 1e300, 1e300}, sceneHi[3] = {-1e300, -1e300, -1e300};
        for (int j = 0; j < n; j += skipOf(j))
            for (int k = 0; k < 3; ++k)
            {
                sceneLo[k] = std::min(sceneLo[k], (double)lo(j, k));
                sceneHi[k] = std::max(sceneHi[k], (double)hi(j, k));
            }
        const double sceneArea = (sceneHi[0] - sceneLo[0]) * (sceneHi[1] - sceneLo[1]) + (sceneHi[1] - sceneLo[1]) * (sceneHi[2] - sceneLo[2]) +
                                 (sceneHi[2] - sceneLo[2]) * (sceneHi[0] - sceneLo[0]);
        for (int i = 0; i < n; ++i)
        {
            while (!open.empty() && open.back().end <= i)
                open.pop_back();
            const int end = std::min(i + skipOf(i), n);
            if (countOf(i) == 0 && end > i + 1)
            {
                const int parentFrom = open.empty() ? 0 : open.back().node, parentTo = open.empty() ? n : open.back().end;
                const double parentArea = open.empty() ? sceneArea : areaOf(open.back().node);
                const double bySurface = parentArea > 0.0 ? std::min(1.0, areaOf(i) / parentArea) : 1.0;
                /* share of the parent's leaves whose centre lies in the node (sampled beyond 4096 leaves) */
                const int firstLeaf = leavesBefore[parentFrom], lastLeaf = leavesBefore[parentTo];
                const int stride = std::max(1, (lastLeaf - firstLeaf) / 4096);
                int sampled = 0, inside = 0;
                for (int q = firstLeaf; q < lastLeaf; q += stride)
                {
                    const int leaf = leaves[q];
                    bool in = true;
                    for (int k = 0; k < 3 && in; ++k)
                    {
                        const double c = 0.5 * ((double)lo(leaf, k) + hi(leaf, k));
                        in = c >= lo(i, k) && c <= hi(i, k);
                    }
                    ++sampled;
                    inside += in ? 1 : 0;
                }
                const double byOrigin = sampled ? (double)inside / sampled : 1.0;
                bool encloses = true; /* every child within the node: what the argument above rests on */
                for (int j = i + 1; j < end && encloses; j += skipOf(j))
                    for (int k = 0; k < 3; ++k)
                        encloses = encloses && lo(j, k) >= lo(i, k) && hi(j, k) <= hi(i, k);
                if (encloses && (1.0 - std::max(bySurface, byOrigin)) * (end - i - 1) < threshold)
                {
                    keep[i] = 0;
                    ++*nbPruned;
                    continue;
                }
            }
            open.push_back({i, end});
        }
    }
    if (*nbPruned == 0)
        return n;
    std::vector<int> newIndex(n + 1, 0);
    for (int i = 0; i < n; ++i)
        newIndex[i + 1] = newIndex[i] + (keep[i] ? 1 : 0);
    const int m = newIndex[n];
    std::vector<float4> outRows(2 * (size_t)m);
    std::vector<int> outStart(m), outOrigin(m);
    for (int i = 0; i < n; ++i)
        if (keep[i])
        {
            const int j = newIndex[i];
            const int end = std::min(i + skipOf(i), n);
            outRows[2 * j] = rows[2 * i];
            outRows[2 * j + 1] = rows[2 * i + 1];
            outRows[2 * j + 1].w = bitsf(newIndex[end] - j);
            outStart[j] = start[i];
            outOrigin[j] = origin[i];
        }
    rows.swap(outRows);
    start.swap(outStart);
    origin.swap(outOrigin);
    return m;
}

/* The order-free lists: the leaves of the scene - every node with primitives, whatever the reference put above
 * it - under a binary surface-area hierarchy of our own (binned SAH over the leaf boxes' centres, sixteen bins),
 * flattened depth-first with skip pointers like the other lists, EIGHT TIMES: once per sign octant of a ray's
 * direction, the child on the near side of each split first.  Closest-hit walks whose result does not depend on
 * the order of the leaves (rt_device.h closestHitWalk: rays longer than 2, ties to the smaller flattened index)
 * walk the list of their octant instead of the reference's order - near boxes first, so that the first hits
 * shrink the cut-off and the far side of the scene is culled, which no fixed order can do for every direction.
 * Any of the eight is correct for any ray; the choice is only speed.  Inner nodes that hardly cull are left
 * out as in the other lists (decided once, on the first flattening).  Valid only when every primitive lies
 * inside its leaf's box and every inner node of the reference's list encloses its children (the caller checks
 * both).  `rows` / `start`: a nested list.  Output: 8 x count nodes, list after list. */
static int buildFreeOrderLists(const std::vector<float4> &rows, const std::vector<int> &start, const std::vector<int> &origin,
                               std::vector<float4> &outRows, std::vector<int> &outStart, std::vector<int> &outOrigin,
                               int *nbPruned)
{
    struct Leaf
    {
        float lo[3], hi[3];
        int node;
    };
    struct TreeNode
    {
        float lo[3], hi[3];
        int left, right, axis, leaf; /* leaf: node of the input list, -1 for an inner node */
        int depth;
        bool keep;
    };
    const int n = (int)start.size();
    std::vector<Leaf> leaves;
    for (int i = 0; i < n; ++i)
        if (bitsi(rows[2 * i + 1].z) > 0)
        {
            Leaf l;
            l.lo[0] = rows[2 * i].x, l.lo[1] = rows[2 * i].y, l.lo[2] = rows[2 * i].z;
            l.hi[0] = rows[2 * i + 1].x, l.hi[1] = rows[2 * i + 1].y, l.hi[2] = rows[2 * i].w;
            l.node = i;
            leaves.push_back(l);
        }
    outRows.clear();
    outStart.clear();
    outOrigin.clear();
    *nbPruned = 0;
    if (leaves.size() < 2)
        return 0;
    auto area = [](const float *lo, const float *hi) {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return x * y + y * z + z * x;
    };
    std::vector<TreeNode> tree;
    tree.reserve(2 * leaves.size());
    struct Range
    {
        int from, to, node;
    };
    std::vector<Range> todo;
    tree.push_back(TreeNode());
    tree[0].depth = 0;
    todo.push_back({0, (int)leaves.size(), 0});
    while (!todo.empty())
    {
        const Range r = todo.back();
        todo.pop_back();
        const int count = r.to - r.from;
        TreeNode t;
        t.left = t.right = -1;
        t.axis = 0;
        t.leaf = -1;
        t.depth = tree[r.node].depth;
        t.keep = true;
        if (count == 1)
        {
            for (int k = 0; k < 3; ++k)
                t.lo[k] = leaves[r.from].lo[k], t.hi[k] = leaves[r.from].hi[k];
            t.leaf = leaves[r.from].node;
            tree[r.node] = t;
            continue;
        }
        float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
        for (int k = 0; k < 3; ++k)
            t.lo[k] = 1e30f, t.hi[k] = -1e30f;
        for (int q = r.from; q < r.to; ++q)
            for (int k = 0; k < 3; ++k)
            {
                t.lo[k] = std::min(t.lo[k], leaves[q].lo[k]);
                t.hi[k] = std::max(t.hi[k], leaves[q].hi[k]);
                const float c = 0.5f * (leaves[q].lo[k] + leaves[q].hi[k]);
                clo[k] = std::min(clo[k], c);
                chi[k] = std::max(chi[k], c);
            }
        /* zeros are +0 (std::min keeps whichever zero it met first; the device builder of solr_lists.hip, whose
         * minima are atomics, could not tell which that was) */
        for (int k = 0; k < 3; ++k)
            t.lo[k] += 0.f, t.hi[k] += 0.f;
        /* binned surface-area split: one pass over the leaves fills the bins of all three axes */
        const int BINS = 16;
        int bestAxis = -1, bestBin = 0;
        double bestCost = 1e300;
        {
            int counts[3][BINS];
            float blo[3][BINS][3], bhi[3][BINS][3];
            float scale[3];
            for (int axis = 0; axis < 3; ++axis)
            {
                const float extent = chi[axis] - clo[axis];
                scale[axis] = extent > 0.f ? BINS / extent : 0.f;
                for (int b = 0; b < BINS; ++b)
                {
                    counts[axis][b] = 0;
                    for (int k = 0; k < 3; ++k)
                        blo[axis][b][k] = 1e30f, bhi[axis][b][k] = -1e30f;
                }
            }
            for (int q = r.from; q < r.to; ++q)
            {
                const Leaf &l = leaves[q];
                for (int axis = 0; axis < 3; ++axis)
                {
                    if (!(scale[axis] > 0.f))
                        continue;
                    const float c = 0.5f * (l.lo[axis] + l.hi[axis]);
                    const int b = std::min(BINS - 1, std::max(0, (int)((c - clo[axis]) * scale[axis])));
                    ++counts[axis][b];
                    float *lo3 = blo[axis][b], *hi3 = bhi[axis][b];
                    lo3[0] = std::min(lo3[0], l.lo[0]), lo3[1] = std::min(lo3[1], l.lo[1]), lo3[2] = std::min(lo3[2], l.lo[2]);
                    hi3[0] = std::max(hi3[0], l.hi[0]), hi3[1] = std::max(hi3[1], l.hi[1]), hi3[2] = std::max(hi3[2], l.hi[2]);
                }
            }
            for (int axis = 0; axis < 3; ++axis)
            {
                if (!(scale[axis] > 0.f))
                    continue;
                double rightArea[BINS];
                int rightCount[BINS];
                float rlo[3] = {1e30f, 1e30f, 1e30f}, rhi[3] = {-1e30f, -1e30f, -1e30f};
                int rc = 0;
                for (int b = BINS - 1; b > 0; --b)
                {
                    rc += counts[axis][b];
                    for (int k = 0; k < 3; ++k)
                    {
                        rlo[k] = std::min(rlo[k], blo[axis][b][k]);
                        rhi[k] = std::max(rhi[k], bhi[axis][b][k]);
                    }
                    rightCount[b] = rc;
                    rightArea[b] = rc ? area(rlo, rhi) : 0.0;
                }
                float llo[3] = {1e30f, 1e30f, 1e30f}, lhi[3] = {-1e30f, -1e30f, -1e30f};
                int lc = 0;
                for (int b = 0; b + 1 < BINS; ++b)
                {
                    lc += counts[axis][b];
                    for (int k = 0; k < 3; ++k)
                    {
                        llo[k] = std::min(llo[k], blo[axis][b][k]);
                        lhi[k] = std::max(lhi[k], bhi[axis][b][k]);
                    }
                    if (lc == 0 || rightCount[b + 1] == 0)
                        continue;
                    const double cost = area(llo, lhi) * lc + rightArea[b + 1] * rightCount[b + 1];
                    if (cost < bestCost)
                    {
                        bestCost = cost;
                        bestAxis = axis;
                        bestBin = b;
                    }
                }
            }
        }
        int mid;
        if (bestAxis < 0)
            mid = r.from + count / 2; /* all centres coincide */
        else
        {
            const float scale = BINS / (chi[bestAxis] - clo[bestAxis]);
            const float origin = clo[bestAxis];
            const int axis = bestAxis, bin = bestBin;
            /* stable: the order inside a node stays the order of the leaf list (it decides the halving by position
             * below, and the device builder partitions the same way) */
            mid = (int)(std::stable_partition(leaves.begin() + r.from, leaves.begin() + r.to,
                                       [&](const Leaf &l) {
                                           const float c = 0.5f * (l.lo[axis] + l.hi[axis]);
                                           return std::min(BINS - 1, std::max(0, (int)((c - origin) * scale))) <= bin;
                                       }) -
                        leaves.begin());
            if (mid == r.from || mid == r.to)
                mid = r.from + count / 2;
            t.axis = bestAxis;
        }
        t.left = (int)tree.size(); /* the low side of the split */
        t.right = t.left + 1;
        tree.push_back(TreeNode());
        tree.push_back(TreeNode());
        tree[t.left].depth = tree[t.right].depth = t.depth + 1;
        tree[r.node] = t;
        todo.push_back({r.from, mid, t.left});
        todo.push_back({mid, r.to, t.right});
    }

    /* one flattening: depth-first, the child on the near side of a ray of this octant first */
    auto flatten = [&](int octant, std::vector<float4> &fr, std::vector<int> &fs, std::vector<int> *which,
                       std::vector<int> *from) {
        struct Visit
        {
            int node, slot; /* slot >= 0: close the inner node written at `slot` */
        };
        std::vector<Visit> stack;
        stack.push_back({0, -1});
        while (!stack.empty())
        {
            const Visit v = stack.back();
            stack.pop_back();
            if (v.slot >= 0)
            {
                fr[2 * v.slot + 1].w = bitsf((int)fs.size() - v.slot);
                continue;
            }
            const TreeNode &t = tree[v.node];
            if (t.leaf >= 0)
            {
                fr.push_back(rows[2 * t.leaf]);
                float4 second = rows[2 * t.leaf + 1];
                second.w = bitsf(1);
                fr.push_back(second);
                fs.push_back(start[t.leaf]);
                if (which)
                    which->push_back(v.node);
                if (from)
                    from->push_back(origin[t.leaf]); /* the node of the reference's list this leaf is */
                continue;
            }
            if (t.keep)
            {
                const int slot = (int)fs.size();
                fr.push_back(make_float4(t.lo[0], t.lo[1], t.lo[2], t.hi[2]));
                fr.push_back(make_float4(t.hi[0], t.hi[1], bitsf(0), bitsf(1)));
                fs.push_back(0);
                if (which)
                    which->push_back(v.node);
                if (from)
                    from->push_back(-1);
                stack.push_back({0, slot});
            }
            const bool highFirst = (octant >> t.axis) & 1; /* direction negative along the split axis */
            stack.push_back({highFirst ? t.left : t.right, -1});
            stack.push_back({highFirst ? t.right : t.left, -1}); /* popped first */
        }
    };
    /* which inner nodes stay: decided on the first flattening */
    {
        std::vector<float4> fr;
        std::vector<int> fs, which;
        flatten(0, fr, fs, &which, nullptr);
        std::vector<int> survivors(which);
        pruneInnerNodes(fr, fs, survivors, nbPruned);
        std::vector<char> kept(tree.size(), 0);
        for (int t : survivors)
            kept[t] = 1;
        const int wide = getenv("SOLR_HIP_FREE_WIDE") ? atoi(getenv("SOLR_HIP_FREE_WIDE")) : 0;
        for (size_t t = 0; t < tree.size(); ++t)
            if (tree[t].leaf < 0)
            {
                tree[t].keep = kept[t] != 0;
                if (wide > 1 && tree[t].depth % wide != 0) /* experiment: only every wide-th level keeps its nodes */
                    tree[t].keep = false;
            }
    }
    /* the eight lists: every node's place follows from the sizes of the subtrees before it (children are stored
     * behind their parent in `tree`, so one backward pass gives the sizes); skip pointers are relative, each list
     * is self-contained */
    std::vector<int> size(tree.size(), 0);
    for (int t = (int)tree.size() - 1; t >= 0; --t)
        size[t] = tree[t].leaf >= 0 ? 1 : (tree[t].keep ? 1 : 0) + size[tree[t].left] + size[tree[t].right];
    const int count = size[0];
    outRows.assign(16 * (size_t)count, make_float4(0.f, 0.f, 0.f, 0.f));
    outStart.assign(8 * (size_t)count, 0);
    outOrigin.assign(8 * (size_t)count, -1);
    struct Place
    {
        int node, at;
    };
    std::vector<Place> stack;
    for (int octant = 0; octant < 8; ++octant)
    {
        float4 *fr = outRows.data() + 2 * (size_t)octant * count;
        int *fs = outStart.data() + (size_t)octant * count, *fo = outOrigin.data() + (size_t)octant * count;
        stack.clear();
        stack.push_back({0, 0});
        while (!stack.empty())
        {
            const Place v = stack.back();
            stack.pop_back();
            const TreeNode &t = tree[v.node];
            if (t.leaf >= 0)
            {
                fr[2 * v.at] = rows[2 * t.leaf];
                float4 second = rows[2 * t.leaf + 1];
                second.w = bitsf(1);
                fr[2 * v.at + 1] = second;
                fs[v.at] = start[t.leaf];
                fo[v.at] = origin[t.leaf]; /* the node of the reference's list this leaf is */
                continue;
            }
            int at = v.at;
            if (t.keep)
            {
                fr[2 * at] = make_float4(t.lo[0], t.lo[1], t.lo[2], t.hi[2]);
                fr[2 * at + 1] = make_float4(t.hi[0], t.hi[1], bitsf(0), bitsf(size[v.node]));
                ++at;
            }
            const bool highFirst = (octant >> t.axis) & 1; /* direction negative along the split axis */
            const int first = highFirst ? t.right : t.left, second = highFirst ? t.left : t.right;
            stack.push_back({second, at + size[first]});
            stack.push_back({first, at});
        }
    }
    return count;
}

/* The scene has been rendered `freeCountdown` times since its upload: build the order-free lists now, from the
 * host images of the reference's list and the primitives as they are (brought up to date first if rotations ran
 * on the device), after checking what their use rests on - every inner node encloses its children, every leaf
 * holds its primitives (the reference's builder makes it so, GPUKernel.cpp:741-830; another host's boxes are
 * taken at their word only after this check; the types whose extent is not what the builder adds around p0 -
 * cones, ellipsoids ... - are sorted out by retagPrimitives). */
namespace
{
void maybeBuildOrderFreeLists()
{
    if (g.freeCountdown <= 0 || --g.freeCountdown > 0)
        return;
    if (!g.primsContained)
    {
        g.freeCountdown = 1; /* no walk would take them (orderFreeListsUsable): asked again with the next frame */
        return;
    }
    PhaseTimer phase;
    quiesce();
    pullGeometry();
    if (!ok())
        return;
    phase.mark("order-free: host images");
    const std::vector<float4> &rows = g.hostBoxes;
    const std::vector<int> &start = g.hostBoxStart;
    const int n = (int)start.size();
    if (n < 2 || rows.size() != 2 * (size_t)n || n > 16000000) /* (beyond that the eight lists pass a dozen GB) */
        return;
    auto skipOf = [&](int i) { return std::max(bitsi(rows[2 * i + 1].w), 1); };
    bool encloses = true;
    /* with the arena laid out as the host images are (the usual case: the scene has been rendered once), the checks
     * and the builder read the exact list and the primitive records there */
    const bool fromArena = !g.geometryDirty && g.geometry.ptr != nullptr && !g.exactStale && !g.deviceAhead &&
                           !getenv("SOLR_HIP_LISTS_ON_HOST") && !getenv("SOLR_HIP_LISTS_VIA_HOST") && !getenv("SOLR_HIP_FREE_WIDE");
    const float4 *arena = (const float4 *)g.geometry.ptr;
    if (fromArena)
    {
        HIPCHECK(hipSetDevice(g.device));
        int *bad = nullptr, found = 1;
        HIPCHECK(hipMalloc((void **)&bad, sizeof(int)));
        if (ok())
        {
            HIPCHECK(hipMemsetAsync(bad, 0, sizeof(int), g.stream));
            hipLaunchKernelGGL(k_listEncloses, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, arena, g.offBoxes, g.offBoxStart,
                               g.offPrims, n, (int)(g.hostPrims.size() / PRIM_ROWS), bad);
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipMemcpyAsync(&found, bad, sizeof(int), hipMemcpyDeviceToHost, g.stream));
            HIPCHECK(hipStreamSynchronize(g.stream));
            (void)hipFree(bad);
        }
        if (!ok())
            return;
        encloses = found == 0;
    }
    for (int i = 0; i < n && encloses && !fromArena; ++i)
    {
        const int end = std::min(i + skipOf(i), n);
        if (bitsi(rows[2 * i + 1].z) > 0 || end <= i + 1)
            continue;
        for (int j = i + 1; j < end && encloses; j += skipOf(j))
            encloses = rows[2 * j].x >= rows[2 * i].x && rows[2 * j].y >= rows[2 * i].y && rows[2 * j].z >= rows[2 * i].z &&
                       rows[2 * j + 1].x <= rows[2 * i + 1].x && rows[2 * j + 1].y <= rows[2 * i + 1].y &&
                       rows[2 * j].w <= rows[2 * i].w;
    }
    const size_t nbPrims = g.hostPrims.size() / PRIM_ROWS;
    for (int i = 0; i < n && encloses && !fromArena; ++i)
    {
        const int count = bitsi(rows[2 * i + 1].z);
        for (int k = 0; k < count && encloses; ++k)
        {
            const size_t pi = (size_t)start[i] + k;
            if (start[i] < 0 || pi >= nbPrims)
            {
                encloses = false;
                break;
            }
            const float4 *r = &g.hostPrims[PRIM_ROWS * pi];
            const int type = bitsi(r[ROW_P0_TYPE].w) & PRIM_TYPE_MASK;
            float lo[3] = {r[ROW_P0_TYPE].x, r[ROW_P0_TYPE].y, r[ROW_P0_TYPE].z};
            float hi[3] = {lo[0], lo[1], lo[2]};
            auto add = [&](const float4 &v) {
                lo[0] = std::min(lo[0], v.x), lo[1] = std::min(lo[1], v.y), lo[2] = std::min(lo[2], v.z);
                hi[0] = std::max(hi[0], v.x), hi[1] = std::max(hi[1], v.y), hi[2] = std::max(hi[2], v.z);
            };
            float grow[3] = {r[ROW_SIZE_MAT].x, r[ROW_SIZE_MAT].y, r[ROW_SIZE_MAT].z};
            if (type == ptTriangle)
            {
                add(r[ROW_P1_INDEX]);
                add(r[ROW_P2]);
                grow[0] = grow[1] = grow[2] = 0.f;
            }
            else if (type == ptCylinder)
            {
                add(r[ROW_P1_INDEX]);
                grow[1] = grow[2] = grow[0];
            }
            else if (type == ptSphere)
                grow[1] = grow[2] = grow[0];
            /* the builder subtracts and adds in another order: four ulps of the coordinates' magnitude of slack, per
             * axis - relative, so that it stays far below the order-free walks' cut-off margin (2e-4 of the distance
             * + 1e-4 of the origin's coordinates, rt_device.h) whatever the scale of the scene */
            auto slack = [&](int k) { return 4.f * 1.1920929e-7f * std::max(std::max(fabsf(lo[k]), fabsf(hi[k])), fabsf(grow[k])); };
            const float ex = slack(0), ey = slack(1), ez = slack(2);
            encloses = rows[2 * i].x <= lo[0] - fabsf(grow[0]) + ex && rows[2 * i].y <= lo[1] - fabsf(grow[1]) + ey &&
                       rows[2 * i].z <= lo[2] - fabsf(grow[2]) + ez && rows[2 * i + 1].x >= hi[0] + fabsf(grow[0]) - ex &&
                       rows[2 * i + 1].y >= hi[1] + fabsf(grow[1]) - ey && rows[2 * i].w >= hi[2] + fabsf(grow[2]) - ez;
        }
    }
    if (!encloses)
    {
        if (getenv("SOLR_HIP_DEBUG_TREE"))
            fprintf(stderr, "solr_hip: no order-free lists: a node does not hold its children or primitives\n");
        return;
    }
    phase.mark("order-free: checks");
    std::vector<int> origin;
    if (!fromArena)
    {
        origin.resize(n);
        for (int i = 0; i < n; ++i)
            origin[i] = i;
    }
    std::vector<float4> boxesF;
    std::vector<int> startF, originF;
    int prunedFree = 0;
    /* on the device (solr_lists.hip: the same tree level by level, the same lists bit for bit) unless told otherwise
     * or declined */
    int count = -1;
    const bool onHost = getenv("SOLR_HIP_LISTS_ON_HOST") != nullptr || getenv("SOLR_HIP_FREE_WIDE") != nullptr;
    if (!onHost)
    {
        HIPCHECK(hipSetDevice(g.device));
        const double threshold = getenv("SOLR_HIP_PRUNE") ? atof(getenv("SOLR_HIP_PRUNE")) : 1.0;
        if (ok() && threshold > 0.0)
        {
            dropFreeStage(true);
            if (fromArena)
                count = solrBuildOrderFreeListsOnDevice(arena + g.offBoxes, (const int *)arena + g.offBoxStart, nullptr, n, threshold, boxesF,
                                                        startF, originF, &prunedFree, g.stream, &g.freeStage);
            else
                count = solrBuildOrderFreeListsOnDevice(rows.data(), start.data(), origin.data(), n, threshold, boxesF, startF, originF,
                                                        &prunedFree, g.stream, getenv("SOLR_HIP_LISTS_VIA_HOST") ? nullptr : &g.freeStage);
        }
    }
    const bool stayed = count > 0 && g.freeStage.rows != nullptr;
    if (count < 0)
    {
        if (origin.empty())
        {
            origin.resize(n);
            for (int i = 0; i < n; ++i)
                origin[i] = i;
        }
        count = buildFreeOrderLists(rows, start, origin, boxesF, startF, originF, &prunedFree);
    }
    if (getenv("SOLR_HIP_DEBUG_TREE"))
        fprintf(stderr, "solr_hip: order-free lists: 8 x %d nodes (%d inner nodes that hardly cull left out)\n", count, prunedFree);
    if (count <= 0)
        return;
    phase.mark("order-free: tree, pruning, eight flattenings");
    g.hostBoxesFree.swap(boxesF);
    g.hostBoxStartFree.swap(startF);
    g.nbBoxesFree = count;
    g.freeRows = 16 * (size_t)count;
    g.freeHostValid = !stayed;
    g.freeStale = false;
    g.hostOriginFree.swap(originF);
    g.refitReady = false;
    g.refitPlanPending = true; /* 8-12 ms for 100 k primitives: only scenes that are rotated on the device pay them */
    /* the lists join the arena: added behind what it holds when they are on the device and it is up to date, else
     * laid out and uploaded again */
    if (stayed && fromArena)
        g.freeDirty = true;
    else
        g.geometryDirty = true;
}
} // namespace

/* Grouping nodes.  The reference's grid builder produces wide levels - 31 sibling leaves under the root of
 * the Cornell scene, 134 top-level cells for the 100k-primitive molecule - and a walk tests every sibling
 * of every node it enters.  Here runs of CONSECUTIVE siblings are wrapped in nodes of our own whose bounds
 * are the union of the siblings' bounds (up to four parts per level, split points by the surface-area
 * heuristic, recursively while a part has more than four members; members about as large as their whole
 * run are left out).  No result can change:
 *   - the depth-first order of the original nodes, hence of every primitive test, is untouched (only
 *     consecutive runs are wrapped), so ties and the shadow accumulation resolve as before;
 *   - a walk reaches an original node only through nodes whose tests it passed, and a group passes
 *     whenever one of its members does: the slab values (b - o) * inv are monotonic in b under IEEE
 *     rounding, so the union's near values are <= and its far values >= the member's on every axis, and
 *     the member's three conditions tnear <= tfar, tnear < far, tfar > 0 carry over (for the sign-selected
 *     form with an infinite reciprocal as well: a member can only pass an axis whose slab contains the
 *     origin coordinate, and then so does the union); the closest-distance cut-off a group is tested
 *     with is never smaller than the one its members will see;
 *   - groups hold no primitives and have no side effects.
 * Requires nested skip pointers and ordered finite bounds (checked by the caller).  Rewrites the node
 * rows and the first-primitive plane in place; returns the new node count. */
static int groupSiblings(std::vector<float4> &rows, std::vector<int> &start, std::vector<int> &origin)
{
    const int n = (int)start.size();
    auto skipOf = [&](int i) { return bitsi(rows[2 * i + 1].w); };
    struct Bounds
    {
        float lo[3], hi[3];
    };
    auto boundsOf = [&](int i) {
        Bounds b;
        b.lo[0] = rows[2 * i].x, b.lo[1] = rows[2 * i].y, b.lo[2] = rows[2 * i].z;
        b.hi[0] = rows[2 * i + 1].x, b.hi[1] = rows[2 * i + 1].y, b.hi[2] = rows[2 * i].w;
        return b;
    };
    auto merge = [](Bounds a, const Bounds &b) {
        for (int k = 0; k < 3; ++k)
        {
            a.lo[k] = std::min(a.lo[k], b.lo[k]);
            a.hi[k] = std::max(a.hi[k], b.hi[k]);
        }
        return a;
    };
    auto area = [](const Bounds &b) {
        const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
        return x * y + y * z + z * x;
    };
    std::vector<float4> outRows;
    std::vector<int> outStart, outOrigin; /* origin: the caller's tag of each node, -1 for the nodes made here */
    outRows.reserve(rows.size() + rows.size() / 2);
    outStart.reserve(start.size() + start.size() / 2);
    outOrigin.reserve(start.size() + start.size() / 2);

    /* best split of sib[from, to) into two consecutive parts */
    std::vector<Bounds> suffix;
    auto splitPoint = [&](const std::vector<int> &sib, int from, int to) {
        const int count = to - from;
        suffix.resize((size_t)count);
        Bounds acc = boundsOf(sib[to - 1]);
        suffix[count - 1] = acc;
        for (int k = count - 2; k >= 0; --k)
        {
            acc = merge(acc, boundsOf(sib[from + k]));
            suffix[k] = acc;
        }
        Bounds left = boundsOf(sib[from]);
        double best = 1e300;
        int bestAt = from + count / 2;
        for (int k = 1; k < count; ++k)
        {
            const double cost = area(left) * k + area(suffix[k]) * (count - k);
            if (cost < best)
            {
                best = cost;
                bestAt = from + k;
            }
            left = merge(left, boundsOf(sib[from + k]));
        }
        return bestAt;
    };

    /* tuning knobs (tools/group_sweep.sh); parts[] / next[] below hold at most 2^4 parts */
    const int flatMax = std::max(1, getenv("SOLR_HIP_GROUP_FLAT") ? atoi(getenv("SOLR_HIP_GROUP_FLAT")) : 4);
    struct Emit
    {
        std::function<void(const std::vector<int> &, int, int)> siblings;
        std::function<void(int)> node;
    } emit;
    emit.node = [&](int i) {
        const size_t at = outStart.size();
        outRows.push_back(rows[2 * i]);
        outRows.push_back(rows[2 * i + 1]);
        outStart.push_back(start[i]);
        outOrigin.push_back(origin[i]);
        /* (most inner nodes have a handful of children, which siblings() would emit as they are: no list is made for
         * them - a vector per inner node was two thirds of this function's time for a 100k-primitive scene) */
        const int end = std::min(i + skipOf(i), n);
        int few = 0;
        for (int j = i + 1; j < end && few <= flatMax; j += std::max(skipOf(j), 1))
            ++few;
        if (few > flatMax)
        {
            std::vector<int> children;
            for (int j = i + 1; j < end; j += std::max(skipOf(j), 1))
                children.push_back(j);
            emit.siblings(children, 0, (int)children.size());
        }
        else
            for (int j = i + 1; j < end;)
            {
                const int next = j + std::max(skipOf(j), 1); /* (read before the node is emitted: rows are not touched, but so it stays) */
                emit.node(j);
                j = next;
            }
        outRows[2 * at + 1].w = bitsf((int)(outStart.size() - at));
    };
    /* (a list of a few dozen nodes - the Cornell room - gains 2 % from a third round of splits, lists of
     * thousands lose 7 %: profiles/r2/group_sweep.txt) */
    const int levels = std::min(
        4, std::max(0, getenv("SOLR_HIP_GROUP_LEVELS") ? atoi(getenv("SOLR_HIP_GROUP_LEVELS")) : (n <= 64 ? 3 : 2)));
    emit.siblings = [&](const std::vector<int> &sib, int from, int to) {
        if (to - from <= flatMax)
        {
            for (int k = from; k < to; ++k)
                emit.node(sib[k]);
            return;
        }
        /* a member about as large as the whole run (a wall of the room, the light cell that spans the
         * view distance) would make every group around it as large as itself and never culled: such
         * members stay where they are, ungrouped, and the runs between them are grouped on their own */
        {
            Bounds u = boundsOf(sib[from]);
            for (int k = from + 1; k < to; ++k)
                u = merge(u, boundsOf(sib[k]));
            const double limit = 0.5 * area(u);
            bool dominant = false;
            for (int k = from; k < to && !dominant; ++k)
                dominant = area(boundsOf(sib[k])) > limit;
            if (dominant)
            {
                int runStart = from;
                for (int k = from; k <= to; ++k)
                    if (k == to || area(boundsOf(sib[k])) > limit)
                    {
                        if (k > runStart)
                            emit.siblings(sib, runStart, k);
                        if (k < to)
                            emit.node(sib[k]);
                        runStart = k + 1;
                    }
                return;
            }
        }
        /* `levels` rounds of binary splits without intermediate nodes: up to 2^levels parts */
        int parts[17];
        int np = 1;
        parts[0] = from;
        parts[1] = to;
        for (int level = 0; level < levels; ++level)
        {
            int next[17];
            int nn = 0;
            for (int q = 0; q < np; ++q)
            {
                next[nn++] = parts[q];
                if (parts[q + 1] - parts[q] > 2)
                    next[nn++] = splitPoint(sib, parts[q], parts[q + 1]);
            }
            next[nn] = to;
            np = nn;
            for (int q = 0; q <= np; ++q)
                parts[q] = next[q];
        }
        for (int q = 0; q < np; ++q)
        {
            const int a = parts[q], b = parts[q + 1];
            if (b - a == 1)
            {
                emit.node(sib[a]);
                continue;
            }
            Bounds u = boundsOf(sib[a]);
            for (int k = a + 1; k < b; ++k)
                u = merge(u, boundsOf(sib[k]));
            const size_t at = outStart.size();
            outRows.push_back(make_float4(u.lo[0], u.lo[1], u.lo[2], u.hi[2]));
            outRows.push_back(make_float4(u.hi[0], u.hi[1], bitsf(0), bitsf(1)));
            outStart.push_back(0);
            outOrigin.push_back(-1);
            emit.siblings(sib, a, b);
            outRows[2 * at + 1].w = bitsf((int)(outStart.size() - at));
        }
    };
    std::vector<int> top;
    for (int j = 0; j < n; j += std::max(skipOf(j), 1))
        top.push_back(j);
    emit.siblings(top, 0, (int)top.size());
    rows.swap(outRows);
    start.swap(outStart);
    origin.swap(outOrigin);
    return (int)start.size();
}

static void h2dSceneOne(BoundingBox *boundingBoxes, int nbActiveBoxes, Primitive *primitives, int nbPrimitives, Lamp *lamps,
                        int nbLamps)
{
    if (!ready("h2d_scene"))
        return;
    quiesce();
    ARGCHECK(nbActiveBoxes >= 0 && nbPrimitives >= 0 && nbLamps >= 0, "h2d_scene: negative count");
    ARGCHECK(nbActiveBoxes == 0 || boundingBoxes, "h2d_scene: null boxes");
    ARGCHECK(nbPrimitives == 0 || primitives, "h2d_scene: null primitives");
    if (!ok())
        return;
    PhaseTimer phase;
    std::vector<float4> boxes(2 * (size_t)nbActiveBoxes);
    std::vector<int> start(nbActiveBoxes);
    for (int i = 0; i < nbActiveBoxes; ++i)
    {
        const BoundingBox &b = boundingBoxes[i];
        ARGCHECK(b.nbPrimitives >= 0 && (b.nbPrimitives == 0 || (b.startIndex >= 0 &&
                                                                  (long)b.startIndex + b.nbPrimitives <= nbPrimitives)),
                 "h2d_scene: box primitive range outside the primitive array");
        /* node record, scene_layout.h: { min.xyz, max.z } { max.xy, nbPrimitives, skip } */
        boxes[2 * i] = make_float4(b.parameters[0].x, b.parameters[0].y, b.parameters[0].z, b.parameters[1].z);
        boxes[2 * i + 1] =
            make_float4(b.parameters[1].x, b.parameters[1].y, bitsf(b.nbPrimitives), bitsf(b.indexForNextBox.x));
        start[i] = b.startIndex;
    }
    if (!ok())
        return;
    phase.mark("h2d_scene: node rows");
    g.nested = validateNesting(boundingBoxes, nbActiveBoxes);
    phase.mark("h2d_scene: nesting check");
    if (!g.nested)
    {
        /* the general walk needs at least forward progress */
        for (int i = 0; i < nbActiveBoxes; ++i)
            ARGCHECK(boundingBoxes[i].indexForNextBox.x >= 1, "h2d_scene: skip pointer < 1");
        if (!ok())
            return;
    }

    /* Collapsed walk order.  The reference's grid builder wraps most leaves in
     * a chain of inner nodes with bit-identical bounds (one per tree level,
     * GPUKernel.cpp:1008-1035).  A ray that enters the first node of such a
     * chain enters all of them - same slabs, same ray, same minDistance since
     * no primitive is tested in between - and a ray that misses it skips all
     * of them, so dropping every inner node whose only child has the same
     * bounds changes no result.  Skip pointers are recomputed in the compacted
     * numbering and stay nested. */
    /* (one pass: which nodes stay, whether every bound is ordered and finite, the compacted numbering) */
    std::vector<char> keep(nbActiveBoxes, 1);
    std::vector<int> newIndex((size_t)nbActiveBoxes + 1);
    newIndex[0] = 0;
    g.orderedExact = 1;
    g.orderedCompact = 1;
    for (int i = 0; i < nbActiveBoxes; ++i)
    {
        const BoundingBox &a = boundingBoxes[i];
        if (g.nested && a.nbPrimitives == 0)
        {
            if (i + 1 < nbActiveBoxes)
            {
                const BoundingBox &b = boundingBoxes[i + 1];
                if (a.indexForNextBox.x >= 2 && b.indexForNextBox.x == a.indexForNextBox.x - 1 &&
                    memcmp(a.parameters, b.parameters, sizeof(a.parameters)) == 0)
                    keep[i] = 0;
            }
            /* an inner node without emitted children (its cell held only lights or nothing,
             * GPUKernel.cpp:1096) leads nowhere: entering or missing it changes nothing */
            if (a.indexForNextBox.x == 1)
                keep[i] = 0;
        }
        const float *lo = &a.parameters[0].x, *hi = &a.parameters[1].x;
        bool ordered = true;
        for (int k = 0; k < 3; ++k)
            ordered = ordered && (lo[k] <= hi[k]) && (fabsf(lo[k]) < 1.0e30f) && (fabsf(hi[k]) < 1.0e30f);
        if (!ordered)
        {
            g.orderedExact = 0;
            if (keep[i])
                g.orderedCompact = 0;
        }
        newIndex[(size_t)i + 1] = newIndex[i] + (keep[i] ? 1 : 0);
    }
    const int nc = newIndex[nbActiveBoxes];
    std::vector<float4> boxesC(2 * (size_t)nc);
    std::vector<int> startC(nc), originC(nc);
    for (int i = 0; i < nbActiveBoxes; ++i)
        if (keep[i])
        {
            const int j = newIndex[i];
            originC[j] = i;
            const int end = std::min(i + boundingBoxes[i].indexForNextBox.x, nbActiveBoxes);
            boxesC[2 * j] = boxes[2 * i];
            boxesC[2 * j + 1] = boxes[2 * i + 1];
            boxesC[2 * j + 1].w = bitsf(newIndex[end] - j);
            startC[j] = start[i];
        }

    /* the order-free lists are built when the scene has stayed for a frame (maybeBuildOrderFreeLists): a host that
     * uploads the scene again for every frame - the reference's own way of animating - never pays for them */
    std::vector<float4> boxesF;
    std::vector<int> startF, originF;
    const int nbFreeNodes = 0;
    g.freeCountdown = 0;
    if (g.nested && g.orderedCompact && nc > 1 && g.grouping && !getenv("SOLR_HIP_NO_FREE_ORDER"))
        g.freeCountdown = std::max(1, getenv("SOLR_HIP_FREE_AFTER") ? atoi(getenv("SOLR_HIP_FREE_AFTER")) : 2);

    int nbWalkNodes = nc, prunedBefore = 0, prunedAfter = 0;
    if (g.nested && g.orderedCompact && nc > 0 && g.grouping)
    {
        phase.mark("h2d_scene: chain collapse");
        pruneInnerNodes(boxesC, startC, originC, &prunedBefore, getenv("SOLR_HIP_REBUILD") != nullptr); /* cells that do not cull: their children join the run above */
        phase.mark("h2d_scene: prune");
        groupSiblings(boxesC, startC, originC);
        phase.mark("h2d_scene: grouping");
        nbWalkNodes = pruneInnerNodes(boxesC, startC, originC, &prunedAfter); /* groups that do not cull either */
        phase.mark("h2d_scene: prune groups");
    }
    if (getenv("SOLR_HIP_DEBUG_TREE"))
    {
        fprintf(stderr, "solr_hip: %d nodes uploaded, %d after collapsing chains, %d in the walk list (%d + %d inner nodes that hardly cull left out)\n",
                nbActiveBoxes, nc, nbWalkNodes, prunedBefore, prunedAfter);
        if (nbWalkNodes <= 80)
            for (int i = 0; i < nbWalkNodes; ++i)
                fprintf(stderr, "  node %2d: prims %d skip %d  [%g %g %g .. %g %g %g]\n", i, bitsi(boxesC[2 * i + 1].z),
                        bitsi(boxesC[2 * i + 1].w), boxesC[2 * i].x, boxesC[2 * i].y, boxesC[2 * i].z,
                        boxesC[2 * i + 1].x, boxesC[2 * i + 1].y, boxesC[2 * i].w);
    }

    std::vector<float4> prims(8 * (size_t)nbPrimitives);
    for (int i = 0; i < nbPrimitives; ++i)
    {
        const Primitive &p = primitives[i];
        float4 *r = &prims[8 * (size_t)i];
        r[ROW_P0_TYPE] = make_float4(p.p0.x, p.p0.y, p.p0.z, bitsf(p.type & PRIM_TYPE_MASK));
        r[ROW_SIZE_MAT] = make_float4(p.size.x, p.size.y, p.size.z, bitsf(p.materialId));
        r[ROW_P1_INDEX] = make_float4(p.p1.x, p.p1.y, p.p1.z, bitsf(p.index));
        r[ROW_P2] = make_float4(p.p2.x, p.p2.y, p.p2.z, 0.f);
        r[ROW_N0] = make_float4(p.n0.x, p.n0.y, p.n0.z, p.vt0.x);
        r[ROW_N1] = make_float4(p.n1.x, p.n1.y, p.n1.z, p.vt0.y);
        r[ROW_N2] = make_float4(p.n2.x, p.n2.y, p.n2.z, p.vt1.x);
        r[ROW_UV] = make_float4(p.vt1.y, p.vt2.x, p.vt2.y, 0.f);
    }
    if (prims.empty())
        prims.assign(8, make_float4(0.f, 0.f, 0.f, 0.f)); /* inactive lanes read record 0 */
    phase.mark("h2d_scene: primitive rows");
    g.refitReady = false;
    g.exactStale = false;
    g.refitPlanPending = true;
    g.hostOriginFree.swap(originF);
    g.deviceAhead = false;
    g.nbMovable = -1;
    g.hostBoxes.swap(boxes);
    g.hostBoxesCompact.swap(boxesC);
    g.hostBoxStart.swap(start);
    g.hostBoxStartCompact.swap(startC);
    g.hostOriginCompact = originC;
    g.hostBoxesFree.swap(boxesF);
    g.hostBoxStartFree.swap(startF);
    g.freeRows = 0;
    g.freeHostValid = true;
    g.freeDirty = false;
    dropFreeStage(true);
    g.nbBoxesFree = nbFreeNodes;
    g.freeStale = false;
    g.hostPrims.swap(prims);
    retagPrimitives();
    phase.mark("h2d_scene: tags");
    HIPCHECK(hipSetDevice(g.device));
    std::vector<int> l(lamps, lamps + (lamps ? nbLamps : 0));
    upload(g.lamps, l);
    if (ok())
    {
        g.nbBoxes = nbActiveBoxes;
        g.nbBoxesCompact = nbWalkNodes;
        g.nbPrimitives = nbPrimitives;
        g.nbLamps = nbLamps;
    }
}

/* Extension: per flattened primitive, whether GPUKernel::rotatePrimitives would move it (it sits in a
 * level-0 box, is movable and is not the camera primitive).  Valid until the next h2d_scene. */
static void setMovableOne(const unsigned char *flags, int nbPrimitives)
{
    if (!ready("solr_hip_set_movable"))
        return;
    ARGCHECK(nbPrimitives >= 0 && (nbPrimitives == 0 || flags), "solr_hip_set_movable: null flags");
    if (!ok())
        return;
    g.nbMovable = -1;
    if (nbPrimitives != g.nbPrimitives)
        return;
    quiesce();
    HIPCHECK(hipSetDevice(g.device));
    std::vector<unsigned char> f(flags, flags + nbPrimitives);
    if (f.empty())
        f.push_back(0);
    upload(g.movable, f);
    if (ok())
        g.nbMovable = nbPrimitives;
}

/* Extension: GPUKernel::rotatePrimitives + compactBoxes(false) + h2d_scene on the resident scene
 * (GPUKernel.cpp:1378-1460, 1151-1281 of the reference), see k_rotatePrimitives.  Returns 1 when the
 * arena now holds the rotated scene, 0 when the request cannot be served here and the caller has to
 * take the host route (nothing was changed). */
/* can this engine rotate its resident scene?  (makes the refit plan when the lists changed; changes nothing else) */
static bool canRotateOne(const float center[3], const float cosAngles[3], const float sinAngles[3], float viewDistance)
{
    if (!ready("solr_hip_rotate_primitives") || !ok())
        return false;
    if (g.refitPlanPending)
    {
        /* which nodes to refit, in which order: made for the first rotation after the lists changed */
        g.refitPlanPending = false;
        ensureHostFreeLists();
        buildRefitPlan(g.hostBoxes, g.hostBoxesCompact, g.hostOriginCompact, g.hostBoxesFree, g.hostOriginFree);
    }
    /* the seeds of the two box updates only commute with the unions while viewDistance <= 1e6, and a
     * tree cut off at NB_MAX_BOXES has host-side children the flattened list does not show */
    if (!g.refitReady || g.nbMovable != g.nbPrimitives || g.nbPrimitives <= 0 || !(viewDistance <= 1000000.f) ||
        !(viewDistance > 0.f) || g.nbBoxes >= NB_MAX_BOXES || !center || !cosAngles || !sinAngles)
    {
        if (getenv("SOLR_HIP_DEBUG_TREE"))
            fprintf(stderr, "solr_hip_rotate_primitives refused: plan %d, flags for %d of %d primitives, viewDistance %g, %d nodes\n",
                    (int)g.refitReady, g.nbMovable, g.nbPrimitives, viewDistance, g.nbBoxes);
        return false;
    }
    return true;
}

static int rotatePrimitivesOne(const float center[3], const float cosAngles[3], const float sinAngles[3], float viewDistance)
{
    if (!canRotateOne(center, cosAngles, sinAngles, viewDistance))
        return 0;
    HIPCHECK(hipSetDevice(g.device));
    flushGeometry();
    if (!ok())
        return 0;
    quiesce();
    RotationArgs R;
    R.cx = center[0], R.cy = center[1], R.cz = center[2];
    R.cosx = cosAngles[0], R.cosy = cosAngles[1], R.cosz = cosAngles[2];
    R.sinx = sinAngles[0], R.siny = sinAngles[1], R.sinz = sinAngles[2];
    hipLaunchKernelGGL(k_rotatePrimitives, dim3((unsigned)((g.nbPrimitives + 255) / 256)), dim3(256), 0, g.stream,
                       (float4 *)g.geometry.ptr, g.offPrims, g.nbPrimitives, (const unsigned char *)g.movable.ptr, R);
    refitList(g.refitWalkLevels, g.offBoxesCompact, g.offBoxStartCompact, viewDistance);
    if (g.nbBoxesFree > 0 && !g.refitFreeLevels.empty())
        refitList(g.refitFreeLevels, g.offBoxesFree, g.offBoxStartFree, viewDistance);
    else
        g.freeStale = g.nbBoxesFree > 0; /* no plan: rotated scenes walk the reference's order until the next upload */
    buildLeafRecords(); /* the leaves' copies of their first primitive follow the primitives */
    g.exactStale = true;
    g.exactStaleViewDistance = viewDistance;
    HIPCHECK(hipGetLastError());
    /* the other flights' streams start their next frame only after this */
    HIPCHECK(hipStreamSynchronize(g.stream));
    if (!ok())
        return 0;
    g.deviceAhead = true;
    ++g.nbDeviceRotations;
    return 1;
}

int solr_hip_device_rotations(void)
{
    return g.nbDeviceRotations;
}

/* Diagnostics / tests: the resident arena's node lists and primitive records as the device holds them
 * now.  exact != 0: the reference's list, else the walk-order list.  Returns the number of float4 rows
 * written (2 per node, 8 per primitive), -1 if the capacity is too small. */
int solr_hip_read_nodes(int exact, float *rows, int capacityRows)
{
    if (!ready("solr_hip_read_nodes") || !g.geometry.ptr)
        return -1;
    flushGeometry();
    if (exact)
        refreshExactList();
    quiesce();
    int n = 2 * (exact ? g.nbBoxes : g.nbBoxesCompact);
    unsigned at = exact ? g.offBoxes : g.offBoxesCompact;
    if (exact >= 2) /* 2 ... 9: the order-free list of octant exact - 2 (0 rows when there are none) */
    {
        const bool have = exact <= 9 && g.nbBoxesFree > 0 && !g.freeStale && g.freeRows == 16 * (size_t)g.nbBoxesFree;
        n = have ? 2 * g.nbBoxesFree : 0;
        at = g.offBoxesFree + 2u * (unsigned)((exact - 2) * g.nbBoxesFree);
    }
    if (!rows)
        return n; /* size query */
    if (n > capacityRows)
        return -1;
    if (n)
        HIPCHECK(hipMemcpy(rows, (const char *)g.geometry.ptr + (size_t)at * 16, (size_t)n * 16, hipMemcpyDeviceToHost));
    return ok() ? n : -1;
}

int solr_hip_read_primitives(float *rows, int capacityRows)
{
    if (!ready("solr_hip_read_primitives") || !g.geometry.ptr)
        return -1;
    flushGeometry();
    quiesce();
    const int n = PRIM_ROWS * g.nbPrimitives;
    if (!rows)
        return n;
    if (n > capacityRows)
        return -1;
    if (n)
        HIPCHECK(hipMemcpy(rows, (const char *)g.geometry.ptr + (size_t)g.offPrims * 16, (size_t)n * 16,
                           hipMemcpyDeviceToHost));
    return ok() ? n : -1;
}

static void h2dMaterialsOne(Material *materials, int nbActiveMaterials)
{
    if (!ready("h2d_materials"))
        return;
    quiesce();
    ARGCHECK(nbActiveMaterials >= 0 && (nbActiveMaterials == 0 || materials), "h2d_materials: bad arguments");
    if (!ok())
        return;
    /* always NB_MAX_MATERIALS + 1 records on the device (zero beyond the
     * active ones) so that every id a primitive or the box-debug view can
     * produce stays inside the allocation */
    const int capacity = NB_MAX_MATERIALS + 1;
    const int active = std::min(nbActiveMaterials, capacity);
    /* (only the active records are built and copied: the 12.6 MB of the full table took 10 ms per call, and the
     * reference's hosts call this whenever one material changes) */
    std::vector<MaterialHot> hot((size_t)std::max(active, 1));
    std::vector<MaterialCold> cold((size_t)std::max(active, 1));
    memset(hot.data(), 0, hot.size() * sizeof(MaterialHot));
    memset(cold.data(), 0, cold.size() * sizeof(MaterialCold));
    g.materialTags.assign(capacity, PRIM_FAST0 | (1 << PRIM_WIDTH_SHIFT));
    g.materialAverage.assign(capacity, 0.f);
    g.textureUses.clear();
    g.textureTablesChecked = false;
    for (int i = 0; i < nbActiveMaterials && i < capacity; ++i)
    {
        Material m = materials[i];
        /* A diffuse texture id that was never loaded: GPUKernel::setMaterial then leaves the "computed texture"
         * mapping (40000 x 40000 at offset 0, GPUKernel.cpp:1893-1896) next to the id, and the mappers would
         * index gigabytes past the atlas (the reference reads whatever is there; a memory fault here).  Such a
         * material is untextured on the device. */
        if (m.textureIds.x >= 0 && m.textureMapping.x == 40000 && m.textureMapping.y == 40000 && m.textureOffset.x == 0)
            m.textureIds.x = TEXTURE_NONE;
        g.materialTags[i] = materialTag(m);
        /* the mappers fetch only for 0 <= u < mapping.x (rt_device.h): a mapping without columns - what
         * realignTexturesAndMaterials gives a material whose texture nobody loaded - never reaches the atlas */
        if (m.textureIds.x >= 0 && m.textureMapping.x > 0) /* procedural ids (Mandelbrot, Julia) are negative */
        {
            Engine::TextureUse use;
            use.material = i;
            use.texels = (long)m.textureMapping.x * (long)m.textureMapping.y * (long)m.textureMapping.w;
            if (m.textureMapping.y <= 0 || m.textureMapping.w <= 0 || m.textureOffset.x < 0)
                use.texels = 0; /* fetchTexel takes an index modulo this: refused by checkTextureTables */
            const int ids[7] = {m.textureIds.x, m.textureIds.y, m.textureIds.z, m.textureIds.w,
                                m.advancedTextureIds.x, m.advancedTextureIds.y, m.advancedTextureIds.z};
            const int offs[7] = {m.textureOffset.x, m.textureOffset.y, m.textureOffset.z, m.textureOffset.w,
                                 m.advancedTextureOffset.x, m.advancedTextureOffset.y, m.advancedTextureOffset.z};
            for (int t = 0; t < 7; ++t)
                use.offsets[t] = ids[t] != TEXTURE_NONE ? (long)offs[t] : -1L;
            g.textureUses.push_back(use);
        }
        g.materialAverage[i] = (m.color.x + m.color.y + m.color.z) / 3.f; /* same expression, same rounding */
        MaterialHot &h = hot[i];
        h.innerIllumination = make_float4(m.innerIllumination.x, m.innerIllumination.y, m.innerIllumination.z,
                                          m.innerIllumination.w);
        h.color = make_float4(m.color.x, m.color.y, m.color.z, m.color.w);
        h.specular = make_float4(m.specular.x, m.specular.y, m.specular.z, m.specular.w);
        h.reflection = m.reflection;
        h.refraction = m.refraction;
        h.transparency = m.transparency;
        h.opacity = m.opacity;
        h.attributes = make_int4(m.attributes.x, m.attributes.y, m.attributes.z, m.attributes.w);
        h.ids = make_int4(m.textureIds.x, m.advancedTextureIds.z, 0, 0);
        MaterialCold &c = cold[i];
        c.textureMapping = make_int4(m.textureMapping.x, m.textureMapping.y, m.textureMapping.z, m.textureMapping.w);
        c.textureOffset = make_int4(m.textureOffset.x, m.textureOffset.y, m.textureOffset.z, m.textureOffset.w);
        c.textureIds = make_int4(m.textureIds.x, m.textureIds.y, m.textureIds.z, m.textureIds.w);
        c.advancedTextureOffset = make_int4(m.advancedTextureOffset.x, m.advancedTextureOffset.y,
                                            m.advancedTextureOffset.z, m.advancedTextureOffset.w);
        c.advancedTextureIds = make_int4(m.advancedTextureIds.x, m.advancedTextureIds.y, m.advancedTextureIds.z,
                                         m.advancedTextureIds.w);
        c.mappingOffset = make_float2(m.mappingOffset.x, m.mappingOffset.y);
        c.pad = make_float2(0.f, 0.f);
    }
    HIPCHECK(hipSetDevice(g.device));
    const size_t tableBytes = 12 * (size_t)capacity * sizeof(float4);
    const bool fresh = !g.materials.ptr || g.materials.bytes < tableBytes;
    reserve(g.materials, tableBytes);
    if (!ok())
        return;
    char *table = (char *)g.materials.ptr;
    const size_t coldAt = 6 * (size_t)capacity * sizeof(float4);
    /* zeros beyond the active records: the whole table when it is new, else what the last call left behind */
    const int stale = fresh ? capacity : std::min(std::max(g.nbMaterials, 0), capacity);
    if (fresh)
        HIPCHECK(hipMemsetAsync(table, 0, tableBytes, g.stream));
    else if (stale > active)
    {
        HIPCHECK(hipMemsetAsync(table + (size_t)active * sizeof(MaterialHot), 0, (size_t)(stale - active) * sizeof(MaterialHot),
                                g.stream));
        HIPCHECK(hipMemsetAsync(table + coldAt + (size_t)active * sizeof(MaterialCold), 0,
                                (size_t)(stale - active) * sizeof(MaterialCold), g.stream));
    }
    if (active > 0)
    {
        HIPCHECK(hipMemcpyAsync(table, hot.data(), (size_t)active * sizeof(MaterialHot), hipMemcpyHostToDevice, g.stream));
        HIPCHECK(hipMemcpyAsync(table + coldAt, cold.data(), (size_t)active * sizeof(MaterialCold), hipMemcpyHostToDevice,
                                g.stream));
    }
    HIPCHECK(hipStreamSynchronize(g.stream)); /* pageable sources: complete for the caller when this returns */
    if (ok())
    {
        g.offMatCold = 6u * (unsigned)capacity;
        g.nbMaterials = nbActiveMaterials;
        retagPrimitives();
    }
}

namespace
{
bool shareRandoms(); /* (with the RCCL layer at the end of this file) */
}
static void noteRandomsReach(const std::vector<float> &r)
{
    /* the ambient-occlusion taps read randoms[i] and randoms[i + 100], i < 256 (CRT:1146-1153) */
    float reach = 0.f;
    for (size_t i = 0; i < r.size() && i < 356; ++i)
        reach = std::max(reach, fabsf(r[i]));
    g.randomsReach = reach;
}

static void uploadRandoms(const float *randoms, long count, const char *who)
{
    if (ready(who))
    {
        quiesce();
        std::vector<float> r(randoms, randoms + count);
        HIPCHECK(hipSetDevice(g.device));
        upload(g.randoms, r);
        if (ok())
            g.nbRandoms = count;
        noteRandomsReach(r);
    }
    /* with a communicator rank 0's buffer is everybody's: every rank ends its upload here, in whatever state */
    shareRandoms();
}

static void h2dRandomsOne(float *randoms)
{
    if (g.initialized && ok())
        ARGCHECK(randoms != nullptr, "h2d_randoms: null buffer");
    uploadRandoms(randoms, MAX_BITMAP_SIZE, "h2d_randoms");
}

/* Frames larger than the reference's 1920 x 1080 limit: its natural depth of field indexes the buffer with
 * `pixel index + timestamp % (MAX_BITMAP_SIZE - 2)` (CRT:475, the precedence as written), i.e. up to
 * W * H + 9999 + 1 - beyond MAX_BITMAP_SIZE floats as soon as the frame is larger (and by up to 9 999 floats
 * even at that size).  A host that renders such frames hands over as many values
 * as the expression can reach; reads beyond what was handed over return 0 (rt_device.h rnd()). */
static void h2dRandomsSizedOne(const float *randoms, long count)
{
    if (g.initialized && ok())
        ARGCHECK(randoms != nullptr && count >= MAX_BITMAP_SIZE && count <= (1L << 30),
                 "solr_hip_h2d_randoms_sized: needs at least MAX_BITMAP_SIZE values");
    uploadRandoms(randoms, count, "solr_hip_h2d_randoms_sized");
}

static void h2dTexturesOne(int activeTextures, TextureInfo *textureInfos)
{
    if (!ready("h2d_textures"))
        return;
    quiesce();
    ARGCHECK(activeTextures >= 0 && (activeTextures == 0 || textureInfos), "h2d_textures: bad arguments");
    for (int i = 0; ok() && i < activeTextures; ++i)
        if (textureInfos[i].buffer)
            ARGCHECK(textureInfos[i].offset >= 0 && textureInfos[i].size.x >= 0 && textureInfos[i].size.y >= 0 &&
                         textureInfos[i].size.z >= 0 &&
                         (double)textureInfos[i].size.x * textureInfos[i].size.y * textureInfos[i].size.z < 2147483648.0,
                     "h2d_textures: a texture with a negative offset or size, or larger than 2 GB");
    if (!ok())
        return;
    size_t total = 0, largest = 0;
    for (int i = 0; i < activeTextures; ++i)
        if (textureInfos[i].buffer)
        {
            size_t sz = (size_t)textureInfos[i].size.x * textureInfos[i].size.y * textureInfos[i].size.z;
            size_t end = (size_t)textureInfos[i].offset + sz;
            total = end > total ? end : total;
            largest = sz > largest ? sz : largest;
        }
    /* Slack: a texel fetch reads index .. index+2, and the secondary maps of a material (normal, bump,
     * specular ...) are read at the texel index of its DIFFUSE texture (TextureMapping.cuh:30-116): a map
     * smaller than the diffuse texture is read up to `largest` bytes past its own end.  Inside the atlas that
     * is the next texture, as in the reference; past the atlas the reference reads whatever follows its
     * buffer - here zeros, always. */
    std::vector<unsigned char> atlas(total + largest + 4, 0);
    for (int i = 0; i < activeTextures; ++i)
        if (textureInfos[i].buffer)
        {
            size_t sz = (size_t)textureInfos[i].size.x * textureInfos[i].size.y * textureInfos[i].size.z;
            memcpy(atlas.data() + textureInfos[i].offset, textureInfos[i].buffer, sz);
        }
    HIPCHECK(hipSetDevice(g.device));
    upload(g.textures, atlas);
    g.atlasBytes = ok() ? atlas.size() : 0;
    g.textureTablesChecked = false;
}

static void h2dLightInformationOne(LightInformation *lightInformation, int lightInformationSize)
{
    if (!ready("h2d_lightInformation"))
        return;
    quiesce();
    ARGCHECK(lightInformationSize >= 0 && (lightInformationSize == 0 || lightInformation),
             "h2d_lightInformation: bad arguments");
    if (!ok())
        return;
    std::vector<float4> l(3 * (size_t)lightInformationSize);
    for (int i = 0; i < lightInformationSize; ++i)
    {
        const LightInformation &s = lightInformation[i];
        l[3 * i] = make_float4(s.location.x, s.location.y, s.location.z, bitsf(s.primitiveId));
        l[3 * i + 1] = make_float4(s.color.x, s.color.y, s.color.z, s.color.w);
        l[3 * i + 2] = make_float4(bitsf(s.materialId), 0.f, 0.f, 0.f);
    }
    g.hostLights.swap(l);
    g.geometryDirty = true;
    g.nbLights = lightInformationSize;
}

/* wait == false: the copies are enqueued and d2hBitmapWait() is owed (several devices copy side by side) */
static void d2hBitmapOne(const SceneInfo &sceneInfo, BitmapBuffer *bitmap, PrimitiveXYIdBuffer *primitivesXYIds, bool wait)
{
    if (!ready("d2h_bitmap"))
        return;
    HIPCHECK(hipSetDevice(g.device));
    const int rows = stripRows();
    const int first = g.nbRows >= 0 ? g.firstRow : 0;
    const size_t pixels = (size_t)sceneInfo.size.x * rows;
    const size_t offset = (size_t)sceneInfo.size.x * first;
    /* the frame rendered last: its buffer set, on its stream */
    const hipStream_t stream = flightStream(g.current);
    const void *src = g.boundBitmap ? g.boundBitmap : flightBitmap(g.current).ptr;
    if (bitmap && src)
        HIPCHECK(hipMemcpyAsync(bitmap + offset * SOLR_COLOR_DEPTH, src, pixels * SOLR_COLOR_DEPTH,
                                hipMemcpyDeviceToHost, stream));
    if (primitivesXYIds && flightIds(g.current).ptr)
        HIPCHECK(hipMemcpyAsync(primitivesXYIds + offset, flightIds(g.current).ptr,
                                pixels * sizeof(PrimitiveXYIdBuffer), hipMemcpyDeviceToHost, stream));
    if (wait)
        HIPCHECK(hipStreamSynchronize(stream));
}
static void d2hBitmapWait()
{
    if (g.initialized && ok())
        HIPCHECK(hipStreamSynchronize(flightStream(g.current)));
}

/* Pipelined read-back of the image (SURVEY.md 8d defines the metric over cudaRender + d2h_bitmap; d2h_bitmap waits
 * for the frame and then for the copy, CudaRayTracer.cu:1647-1672, and nothing renders meanwhile).  Called after
 * cudaRender, solr_hip_d2h_image_async enqueues the copy of the RGB image of the frame rendered last - this
 * process's strip at its place in a full-size image, like d2h_bitmap - into a page-locked host image of the
 * engine's, on a copy stream of its own behind that frame's kernel, and returns a ticket at once; the render
 * streams are free for the next frames (solr_hip_set_frames_in_flight), whose kernels overlap the copy.
 * solr_hip_image_wait(ticket) waits for that one copy and returns the host image; it stays valid until
 * MAX_FLIGHTS more tickets have been handed out.  The ids stay on the device until d2h_bitmap asks for them. */
namespace
{
/* the copy stream of the current engine and its events (one per slot of the ring) */
void ensureCopyStream()
{
    if (g.copyStream)
        return;
    /* (at the render streams' priority.  Measured, profiles/r3/readback_probe.txt: with one or two render streams
     * the copies cost nothing - 0.286 ms per Cornell frame with the image against 0.285 without; with three the
     * frame takes 0.45 ms whatever the host's lag - the runtime's hardware queues are dealt out in turn and a
     * render stream ends up sharing one with this stream; a stream of the highest priority, which gets queues of
     * its own, was slower in every combination (0.33 at best).  HipKernel::setFramesInFlight therefore keeps
     * the engine at two buffer sets and puts the rest of the depth into the host's lag.) */
    HIPCHECK(hipStreamCreateWithFlags(&g.copyStream, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&g.frameRendered, hipEventDisableTiming));
    for (int i = 0; i < Engine::IMAGE_RING && ok(); ++i)
        HIPCHECK(hipEventCreateWithFlags(&g.imageDone[i], hipEventDisableTiming));
}

/* the ring of page-locked images lives in engine 0 (every in-process device copies its strip into the same image) */
bool ensureImageRing()
{
    Engine &e = gFirst;
    const size_t frameBytes = (size_t)e.width * e.height * SOLR_COLOR_DEPTH;
    if (e.pinnedBytes >= frameBytes)
        return true;
    if (e.sharedRing)
    {
        setError(-1, "the frame has grown beyond the host image the ranks share (solr_hip_image_share): share again", __FILE__, __LINE__);
        return false;
    }
    Engine *const was = gCurrent;
    gCurrent = &gFirst;
    releaseImageRing(); /* (outstanding tickets are void from here on: their serial no longer matches) */
    for (int i = 0; i < Engine::IMAGE_RING && ok(); ++i)
    {
        HIPCHECK(hipHostMalloc((void **)&g.pinnedImage[i], frameBytes, hipHostMallocPortable));
        if (ok())
            memset(g.pinnedImage[i], 0, frameBytes);
    }
    if (ok())
        g.pinnedBytes = frameBytes;
    const bool fine = ok();
    gCurrent = was;
    return fine;
}

/* the current engine's strip of the frame it rendered last -> its rows of `image`, on the engine's copy stream behind
 * that frame's kernel; `slot` names the event that says the copy has landed */
void copyStripBehindFrame(BitmapBuffer *image, int slot)
{
    if (!ok())
        return;
    HIPCHECK(hipSetDevice(g.device));
    ensureCopyStream();
    if (!ok())
        return;
    const int flight = g.current;
    const int rows = stripRows();
    const int first = g.nbRows >= 0 ? g.firstRow : 0;
    const void *src = g.boundBitmap ? g.boundBitmap : flightBitmap(flight).ptr;
    /* The copy on the frame's own stream instead of the copy stream (solr_hip_set_copy_route; SOLR_HIP_COPY_INLINE=0/1
     * overrides): it then delays that stream's next frame, not the other streams'.  Measured, profiles/r4/readback_routes.txt:
     * a whole 1080p frame is best served by two buffer sets and the copy stream (0.272 ms; three sets and their own
     * streams 0.280), a 1/8 strip - one round of waves, as slow as its slowest - by three sets and their own streams
     * (0.038 ms against 0.045). */
    static const char *forced = getenv("SOLR_HIP_COPY_INLINE");
    const bool inlineCopy = forced && forced[0] ? forced[0] == '1' : gFirst.copyOnRenderStream;
    const hipStream_t copyOn = inlineCopy ? flightStream(flight) : g.copyStream;
    if (!inlineCopy)
    {
        HIPCHECK(hipEventRecord(g.frameRendered, flightStream(flight)));
        HIPCHECK(hipStreamWaitEvent(g.copyStream, g.frameRendered, 0));
    }
    if (rows > 0 && src)
        HIPCHECK(hipMemcpyAsync(image + (size_t)g.width * first * SOLR_COLOR_DEPTH, src,
                                (size_t)g.width * rows * SOLR_COLOR_DEPTH, hipMemcpyDeviceToHost, copyOn));
    HIPCHECK(hipEventRecord(g.imageDone[slot], copyOn));
    if (!g.boundBitmap)
        g.flightCopy[flight][g.bitmapSide[flight]] = slot;
}

/* hands out the next slot of the ring; the ticket is (serial mod TICKET_PERIOD) * IMAGE_RING + slot: positive for ever
 * (`(int)(serial * IMAGE_RING + slot)` went negative after 2^31 / 6 tickets and read as an error code) */
int nextTicket(int *slot)
{
    Engine &e = gFirst;
    const long serial = ++e.imageSerial;
    if (e.sharedRing)
    {
        const long seq = ++e.shareSeq;
        *slot = (int)(seq % Engine::IMAGE_RING);
        e.slotShareSeq[*slot] = seq;
    }
    else
        *slot = (int)(serial % Engine::IMAGE_RING);
    e.slotSerial[*slot] = serial;
    return Engine::ticketOf(serial, *slot);
}

/* the slot of a ticket whose image is still the one it was handed out for (the slot's full serial says so; generations
 * are compared modulo the ticket's period) */
bool liveTicket(int ticket, int *slot)
{
    if (ticket < 0)
        return false;
    *slot = ticket % Engine::IMAGE_RING;
    const long held = gFirst.slotSerial[*slot];
    return gFirst.pinnedImage[*slot] != nullptr && held >= 0 && held % Engine::TICKET_PERIOD == (long)(ticket / Engine::IMAGE_RING);
}

/* shared ring: report, for every slot, the newest copy of this rank that has LANDED (its event has fired) - at every
 * call of the read-back API, not only when this rank's host asks for that image: a rank whose host never calls
 * solr_hip_image_wait must not keep the root waiting */
void publishLanded()
{
    Engine &e = gFirst;
    if (!e.sharedRing)
        return;
    for (int slot = 0; slot < Engine::IMAGE_RING; ++slot)
    {
        const long seq = e.slotShareSeq[slot];
        if (seq <= e.sharePublished[slot] || !e.imageDone[slot] || !e.slotOfStrips[slot])
            continue;
        if (hipEventQuery(e.imageDone[slot]) != hipSuccess)
            continue;
        e.sharedRing->done[e.shareRank][slot].store(seq, std::memory_order_release);
        e.sharePublished[slot] = seq;
    }
}
} // namespace

/* 0 (default): the pipelined read-back copies on a stream of its own behind the frame's kernel; 1: on the frame's own
 * stream (what to choose: see copyStripBehindFrame) */
void solr_hip_set_copy_route(int onTheFramesOwnStream)
{
    gFirst.copyOnRenderStream = onTheFramesOwnStream != 0;
}

int solr_hip_d2h_image_async(void)
{
    HostSpan whole("solr_hip_d2h_image_async");
    if (!ready("solr_hip_d2h_image_async"))
        return -1;
    ARGCHECK(g.width > 0 && g.height > 0, "solr_hip_d2h_image_async: no frame was rendered");
    if (!ok())
        return -1;
    HIPCHECK(hipSetDevice(g.device));
    if (!ensureImageRing())
        return -1;
    if (g.sharedRing)
    {
        publishLanded();
        /* the slot's last frame must have been handed to the root's host before this rank overwrites its rows (ranks
         * are a few frames apart at most: normally no wait at all).  The root gives an image back when it asks for the
         * NEXT one, so a host that lets IMAGE_RING - 1 tickets pile up without asking for any would wait for itself:
         * refused up front, with the limit, before a ticket is taken (the ranks' ticket sequences stay alike) */
        const long serial = g.shareSeq + 1; /* (the ring's sequence number of the ticket about to be taken) */
        ARGCHECK(serial - 1 - g.lastWaitedSeq < Engine::IMAGE_RING - 1,
                 "solr_hip_d2h_image_async: 5 tickets of the shared image ring are outstanding (IMAGE_RING - 1): ask for "
                 "the oldest one (solr_hip_image_wait) before the next frame is read back");
        if (!ok())
            return -1;
        const auto t0 = std::chrono::steady_clock::now();
        while (g.sharedRing->consumed.load(std::memory_order_acquire) < serial - Engine::IMAGE_RING)
        {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0)
            {
                setError(-1, "solr_hip_d2h_image_async: the root has not taken the frame this slot of the shared image ring "
                             "still holds (60 s)", __FILE__, __LINE__);
                return -1;
            }
            sched_yield();
        }
    }
    int slot = 0;
    const int ticket = nextTicket(&slot);
    g.slotOfStrips[slot] = g.sharedRing != nullptr;
    BitmapBuffer *const image = gFirst.pinnedImage[slot];
    onEveryDevice([&](int) { copyStripBehindFrame(image, slot); });
    return solr_hip_last_error(nullptr, 0) == 0 ? ticket : -1;
}

/* back to a ring of this process's own (after solr_hip_image_share; outstanding tickets are void) */
void solr_hip_image_unshare(void)
{
    if (!g.initialized || !g.sharedRing)
        return;
    quiesce();
    (void)hipSetDevice(g.device);
    releaseImageRing();
}

/* One host image for all ranks of a multi-process job.  The reference copies every device's strip to its place in
 * the host bitmap over that device's own link (d2h_bitmap, CudaRayTracer.cu:1647-1672); with one process per GPU the
 * strips meet in memory the processes share: the ring of page-locked images of solr_hip_d2h_image_async becomes a
 * POSIX shared-memory segment `name` (rank 0 creates it - call it there first, e.g. before a barrier - the others
 * open it), registered with the HIP runtime in every process.  From then on every rank's solr_hip_d2h_image_async
 * copies its strip to its rows of the same image, and solr_hip_image_wait on the ROOT (rank 0) returns when every
 * rank's strip of that frame has landed: the assembled frame on the host at the bandwidth of N PCIe links, not one.
 * The ranks run the same program (the same sequence of tickets).  After initialize_scene / reshape_scene (the frame
 * size is the segment's); undone by finalize_scene.  0, or -1 with the error set. */
int solr_hip_image_share(const char *name, int rank, int world)
{
    if (!ready("solr_hip_image_share"))
        return -1;
    ARGCHECK(name && name[0] == '/' && rank >= 0 && world >= 1 && world <= 64 && rank < world && gDevices == 1,
             "solr_hip_image_share: a name like /solr_frame, 0 <= rank < world <= 64, one device per process");
    ARGCHECK(g.width > 0 && g.height > 0, "solr_hip_image_share: no frame size yet (reshape_scene)");
    if (!ok())
        return -1;
    quiesce();
    HIPCHECK(hipSetDevice(g.device));
    releaseImageRing();
    const size_t frameBytes = (size_t)g.width * g.height * SOLR_COLOR_DEPTH;
    const size_t stride = (frameBytes + 4095) & ~(size_t)4095;
    const size_t head = (sizeof(SharedRing) + 4095) & ~(size_t)4095;
    const size_t bytes = head + stride * Engine::IMAGE_RING;
    int fd = -1;
    if (rank == 0)
    {
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)bytes) != 0)
        {
            close(fd);
            fd = -1;
        }
    }
    else
    {
        const auto t0 = std::chrono::steady_clock::now();
        struct stat st;
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < bytes)
        {
            if (fd >= 0)
                close(fd);
            fd = -1;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
                break;
            usleep(2000);
        }
    }
    void *base = fd >= 0 ? mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
    if (fd >= 0)
        close(fd);
    if (base == MAP_FAILED)
    {
        if (rank == 0)
            (void)shm_unlink(name); /* (created but not mapped: no name is left behind) */
        setError(-1, "solr_hip_image_share: the shared segment could not be created / opened", __FILE__, __LINE__);
        return -1;
    }
    SharedRing *ring = (SharedRing *)base;
    if (rank == 0)
    {
        memset(base, 0, head);
        ring->frameBytes = (long)frameBytes;
        ring->imageStride = (long)stride;
        ring->consumed.store(0, std::memory_order_release);
    }
    HIPCHECK(hipHostRegister(base, bytes, hipHostRegisterPortable));
    if (!ok())
    {
        (void)munmap(base, bytes);
        if (rank == 0)
            (void)shm_unlink(name);
        return -1;
    }
    g.sharedRing = ring;
    g.sharedBytes = bytes;
    g.sharedName = name;
    g.shareRank = rank;
    g.shareWorld = world;
    for (int i = 0; i < Engine::IMAGE_RING; ++i)
    {
        g.pinnedImage[i] = (BitmapBuffer *)base + head + stride * i;
        g.sharePublished[i] = g.slotShareSeq[i] = 0;
    }
    g.pinnedBytes = frameBytes;
    /* the ranks count the ring's slots alike from here (shareSeq); the tickets' generation - this process's own serial -
     * goes on counting: a ticket from before the share never names a slot of the shared ring (the serial
     * used to be reset to 0 here, and an old ticket with the same serial then returned a new frame's image) */
    g.shareSeq = 0;
    g.lastHandedOut = 0;
    g.lastWaitedSeq = 0;
    return ok() ? 0 : -1;
}

/* Once EVERY rank has opened the segment (after a barrier of the caller's) the root takes the name away: the mappings
 * stay, and a job that dies from here on leaves nothing behind in /dev/shm (150 MB per crashed 4K run).
 * Harmless on the other ranks and without a shared ring. */
void solr_hip_image_share_sealed(void)
{
    if (g.initialized && g.sharedRing && g.shareRank == 0 && !g.sharedName.empty())
    {
        (void)shm_unlink(g.sharedName.c_str());
        g.sharedName.clear();
    }
}

/* Waits for the copy (every in-process device's strip) behind `ticket` and returns the host image.  A ticket is good
 * until IMAGE_RING - 1 more have been handed out, or the frame grew and the ring with it: after that it names a
 * frame that is gone, and asking for it is an error - not, silently, a newer frame's image. */
const BitmapBuffer *solr_hip_image_wait(int ticket)
{
    HostSpan whole("solr_hip_image_wait");
    if (!ready("solr_hip_image_wait"))
        return nullptr;
    int slot = 0;
    ARGCHECK(liveTicket(ticket, &slot), "solr_hip_image_wait: no such ticket, or one so old that its image has been "
                                        "handed out again (or re-allocated for a larger frame, or shared since)");
    if (!ok())
        return nullptr;
    onEveryDevice([&](int) {
        if (g.imageDone[slot])
            HIPCHECK(hipEventSynchronize(g.imageDone[slot]));
    });
    if (g.sharedRing && ok())
    {
        /* this rank's strip of that frame has landed; the root returns when everybody's has */
        const long serial = g.slotShareSeq[slot]; /* (the shared ring's sequence number of that frame) */
        if (serial > g.lastWaitedSeq)
            g.lastWaitedSeq = serial;
        SharedRing &ring = *g.sharedRing;
        if (g.slotOfStrips[slot] && serial > g.sharePublished[slot])
        {
            ring.done[g.shareRank][slot].store(serial, std::memory_order_release);
            g.sharePublished[slot] = serial;
        }
        publishLanded();
        if (g.shareRank == 0)
        {
            /* asking for the next image gives the last one back: only now may the other ranks overwrite its rows (a
             * rank can be frames ahead of the root's host - a transport that buffers its sends lets it) */
            if (ring.consumed.load(std::memory_order_relaxed) < g.lastHandedOut)
                ring.consumed.store(g.lastHandedOut, std::memory_order_release);
            const auto t0 = std::chrono::steady_clock::now();
            /* (a ticket of solr_hip_d2h_gathered_async is the root's own copy of the assembled frame: nobody to wait for) */
            for (int r = 1; r < g.shareWorld && g.slotOfStrips[slot]; ++r)
                while (ring.done[r][slot].load(std::memory_order_acquire) < serial)
                {
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0)
                    {
                        setError(-1, "solr_hip_image_wait: a rank's strip of this frame has not landed in the shared image (60 s)",
                                 __FILE__, __LINE__);
                        return nullptr;
                    }
                    sched_yield();
                }
            g.lastHandedOut = serial;
        }
    }
    return solr_hip_last_error(nullptr, 0) == 0 ? gFirst.pinnedImage[slot] : nullptr;
}

/* the float frame buffer of the strip rendered last (strip-sized host buffer; with several in-process devices the
 * whole frame: every device's rows at their place) */
void solr_hip_d2h_postprocessing(PostProcessingBuffer *hostBuffer)
{
    onEveryDevice([&](int) {
        if (!ready("solr_hip_d2h_postprocessing"))
            return;
        ARGCHECK(hostBuffer != nullptr && flightPp(g.current).ptr != nullptr, "solr_hip_d2h_postprocessing: no buffer");
        if (!ok())
            return;
        const size_t pixels = (size_t)g.width * stripRows();
        const size_t offset = gDevices > 1 ? (size_t)g.width * (g.nbRows >= 0 ? g.firstRow : 0) : 0;
        HIPCHECK(hipMemcpyAsync(hostBuffer + offset, flightPp(g.current).ptr, pixels * sizeof(PostProcessingBuffer),
                                hipMemcpyDeviceToHost, flightStream(g.current)));
        HIPCHECK(hipStreamSynchronize(flightStream(g.current)));
    });
}

void solr_hip_h2d_postprocessing(const PostProcessingBuffer *hostBuffer, const PrimitiveXYIdBuffer *ids)
{
    if (!ready("solr_hip_h2d_postprocessing"))
        return;
    quiesce();
    allocateFrame();
    if (!ok())
        return;
    /* into the set the next refinement / accumulation pass will read: the current one */
    const size_t pixels = (size_t)g.width * stripRows();
    const hipStream_t stream = flightStream(g.current);
    if (hostBuffer)
        HIPCHECK(hipMemcpyAsync(flightPp(g.current).ptr, hostBuffer, pixels * sizeof(PostProcessingBuffer),
                                hipMemcpyHostToDevice, stream));
    if (ids)
        HIPCHECK(hipMemcpyAsync(flightIds(g.current).ptr, ids, pixels * sizeof(PrimitiveXYIdBuffer),
                                hipMemcpyHostToDevice, stream));
    HIPCHECK(hipStreamSynchronize(stream));
}


void solr_hip_render_counting(const SceneInfo *sceneInfo, const vec4i *objects,
                              const PostProcessingInfo *postProcessingInfo, const float origin[3],
                              const float direction[3], const float angles[4], unsigned long long counts[8])
{
    if (gDevices < 2)
    {
        renderImpl(*sceneInfo, *objects, *postProcessingInfo, origin, direction, angles, true, counts);
        return;
    }
    /* several in-process devices: the census of the frame is the sum over their strips */
    unsigned long long sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    onEveryDevice([&](int) {
        unsigned long long mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        renderImpl(*sceneInfo, *objects, *postProcessingInfo, origin, direction, angles, true, mine);
        for (int i = 0; i < 8; ++i)
            sum[i] += mine[i];
    });
    if (counts)
        memcpy(counts, sum, sizeof(sum));
}

/* The walk's own ceiling (SURVEY.md 8d's second yardstick; rt_device.h WalkRecord, k_walkBound).  Renders one frame
 * whose walks are recorded - a frame like any other, launched as the frames before it were - and then replays those
 * walks `repeats` times with nothing but the node loop, one launch at a time, HIP events around each.  Out:
 *   ms[0] the recorded frame's own kernel (with the stores of the record: slower than a frame), ms[1] mean, ms[2] min of
 *   the replays; stats[0] walks recorded (per wave), [1] walks left out of the replay (slots full, or not through the node
 *   loop), [2] leaf entries the replay made (per lane), [3] workgroups.
 * Engine 0, one GPU; the lean instantiations only (untextured spheres / planes / triangles / cylinders).  0, or -1. */
static unsigned long long walkLists[6];
void solr_hip_walk_bound_lists(unsigned long long out[6])
{
    if (out)
        memcpy(out, walkLists, sizeof(walkLists));
}

int solr_hip_walk_bound(const SceneInfo *sceneInfo, const vec4i *objects, const PostProcessingInfo *postProcessingInfo,
                        const float origin[3], const float direction[3], const float angles[4], int repeats, double ms[3],
                        unsigned long long stats[4])
{
    if (!ready("solr_hip_walk_bound"))
        return -1;
    ARGCHECK(gDevices == 1, "solr_hip_walk_bound: a diagnostic of one engine; this process renders on several devices");
    if (!ok())
        return -1;
    quiesce();
    HIPCHECK(hipSetDevice(g.device));
    g.recorded = false;
    g.recordNext = true;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    if (!ok())
        return -1;
    renderImpl(*sceneInfo, *objects, *postProcessingInfo, origin, direction, angles, false, nullptr);
    g.recordNext = false;
    const hipStream_t stream = flightStream(g.current);
    HIPCHECK(hipStreamSynchronize(stream));
    if (!ok() || !g.recorded)
    {
        if (ok())
            setError(-1, "solr_hip_walk_bound: the frame was not recorded", __FILE__, __LINE__);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return -1;
    }
    typedef WalkBoundFn BoundFn;
    static const int leanRows[4] = {F_SPHERE | F_PLANE, F_SPHERE | F_TRI, F_SPHERE | F_CYL, F_SPHERE | F_PLANE | F_TRI | F_CYL};
    const BoundFn fn = solrrows::walkBound(g.recordVariant, leanRows[g.recordVariant] | (g.recordDeep ? F_DEEP : 0));
    ARGCHECK(fn != nullptr, "solr_hip_walk_bound: no replay instantiation for this row");
    if (!ok())
    {
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return -1;
    }
    unsigned *visits = (unsigned *)g.walkVisits.ptr;
    unsigned *skipped = visits + (size_t)g.recordGrid * WAVE;
    double sum = 0.0, best = 1.0e30;
    repeats = repeats < 1 ? 1 : repeats;
    for (int i = 0; i < repeats + 2 && ok(); ++i)
    {
        HIPCHECK(hipMemsetAsync(skipped, 0, sizeof(unsigned), stream));
        HIPCHECK(hipEventRecord(e0, stream));
        hipLaunchKernelGGL(fn, dim3(g.recordGrid), dim3(WAVE), g.recordLds, stream, g.recordScene, (const char *)g.walkRecords.ptr,
                           visits, skipped);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipEventRecord(e1, stream));
        HIPCHECK(hipEventSynchronize(e1));
        float t = 0.f;
        HIPCHECK(hipEventElapsedTime(&t, e0, e1));
        if (i >= 2) /* (two launches to warm the instruction cache and the clocks) */
        {
            sum += t;
            best = t < best ? t : best;
        }
    }
    if (ok() && stats)
    {
        std::vector<unsigned> v((size_t)g.recordGrid * WAVE + 1);
        HIPCHECK(hipMemcpy(v.data(), visits, v.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::vector<int> heads((size_t)g.recordGrid * 4);
        HIPCHECK(hipMemcpy2D(heads.data(), 16, g.walkRecords.ptr, SOLR_WALK_SLOT_BYTES, 16, g.recordGrid, hipMemcpyDeviceToHost));
        if (const char *dump = getenv("SOLR_HIP_WALK_BOUND_DUMP"))
        {
            /* diagnostics (tools/longest_wave.py): leaf entries per lane and walks per workgroup of the replay */
            if (FILE *f = fopen(dump, "wb"))
            {
                const unsigned n = g.recordGrid;
                fwrite(&n, sizeof(n), 1, f);
                fwrite(v.data(), sizeof(unsigned), (size_t)n * WAVE, f);
                fwrite(heads.data(), sizeof(int), (size_t)n * 4, f);
                fclose(f);
            }
        }
        unsigned long long walks = 0, entries = 0;
        for (unsigned b = 0; b < g.recordGrid; ++b)
            walks += (unsigned long long)heads[4 * (size_t)b];
        {
            /* which list each recorded walk took (solr_hip_walk_bound_lists) */
            std::vector<int> kinds((size_t)g.recordGrid * 4 * (SOLR_WALK_SLOTS + 1));
            HIPCHECK(hipMemcpy2D(kinds.data(), 16 * (SOLR_WALK_SLOTS + 1), g.walkRecords.ptr, SOLR_WALK_SLOT_BYTES,
                                 16 * (SOLR_WALK_SLOTS + 1), g.recordGrid, hipMemcpyDeviceToHost));
            for (int i = 0; i < 6; ++i)
                walkLists[i] = 0;
            for (unsigned b = 0; ok() && b < g.recordGrid; ++b)
            {
                const int *slot = &kinds[(size_t)b * 4 * (SOLR_WALK_SLOTS + 1)];
                const int n = std::min(slot[0], (int)SOLR_WALK_SLOTS);
                for (int j = 0; j < n; ++j)
                {
                    const int kind = slot[4 * (1 + j)], freeList = slot[4 * (1 + j) + 1];
                    if (kind == WALK_CLOSEST || kind == WALK_SHADOW)
                        ++walkLists[2 * kind + (freeList ? 1 : 0)];
                    else
                        ++walkLists[4];
                }
                walkLists[5] += (unsigned long long)(slot[0] - n);
            }
        }
        for (size_t i = 0; i + 1 < v.size(); ++i)
            entries += v[i];
        stats[0] = walks;
        stats[1] = v.back();
        stats[2] = entries;
        stats[3] = g.recordGrid;
    }
    if (ms)
    {
        ms[0] = 0.0;
        ms[1] = sum / repeats;
        ms[2] = best;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    /* the buffers are a gigabyte for a 1080p frame: given back at once */
    release(g.walkRecords);
    release(g.walkVisits);
    g.recorded = false;
    return ok() ? 0 : -1;
}

void solr_hip_enable_timing(int enable)
{
    onEveryDevice([&](int) {
        g.timing = enable > 0 ? enable : 0;
        g.timingTick = 0;
    });
}

void solr_hip_set_frames_in_flight(int n)
{
    onEveryDevice([&](int) {
        quiesce();
        if (g.initialized)
            (void)hipSetDevice(g.device);
        g.flights = n < 1 ? 1 : (n > MAX_FLIGHTS ? MAX_FLIGHTS : n);
        g.current = 0;
        if (g.initialized && g.width > 0)
            allocateFrame();
    });
}

int solr_hip_get_frames_in_flight(void)
{
    return activeFlights();
}

void *solr_hip_flight_stream(int flight)
{
    return (flight >= 0 && flight < MAX_FLIGHTS) ? (void *)flightStream(flight) : nullptr;
}

int solr_hip_next_flight(void)
{
    return twoFlights() ? (int)(g.frameSerial % (unsigned)activeFlights()) : 0;
}

void solr_hip_set_tile_scheduling(int mode)
{
    onEveryDevice([&](int) {
        g.tileScheduling = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
        g.costFrames = 0;
        g.reorder = false;
        g.orderValid = false;
    });
}

int solr_hip_tile_scheduling_active(void)
{
    return (g.tileScheduling == 2 || (g.tileScheduling == 1 && g.reorder)) && g.orderValid ? 1 : 0;
}

void solr_hip_enable_tile_clocks(int enable)
{
    g.tileClocks = enable != 0;
}

int solr_hip_tile_clocks(unsigned long long *clocks, int capacityTiles)
{
    if (!g.initialized || !g.tileClock.ptr || !clocks || capacityTiles <= 0)
        return 0;
    const int n = g.nbTilesTimed < capacityTiles ? g.nbTilesTimed : capacityTiles;
    quiesce();
    if (hipMemcpy(clocks, g.tileClock.ptr, (size_t)n * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost) !=
        hipSuccess)
        return 0;
    return n;
}

double solr_hip_kernel_time(int *nbLaunches, int reset)
{
    if (g.initialized && g.stream)
        (void)hipStreamSynchronize(g.stream);
    collectEvents();
    double ms = g.timedMs;
    if (nbLaunches)
        *nbLaunches = g.timedLaunches;
    if (reset)
    {
        g.timedMs = 0.0;
        g.timedLaunches = 0;
        g.kernelSamples.clear();
        g.intervalSamples.clear();
    }
    return ms;
}

/* The timed launches one by one (since the last reset of solr_hip_kernel_time; call before it): kernelMs[i] the
 * duration of the renderer kernel of launch i, intervalMs[i] the time from the end of the timed launch before it to
 * its own end (-1 for the first of a batch) - with frames in flight that is what a step takes, and its spread is the
 * error bar of a short timed region.  Returns the number of samples written (at most `capacity`). */
int solr_hip_timing_samples(float *kernelMs, float *intervalMs, int capacity)
{
    if (g.initialized && g.stream)
        (void)hipStreamSynchronize(g.stream);
    collectEvents();
    const int n = std::min((int)g.kernelSamples.size(), std::max(capacity, 0));
    for (int i = 0; i < n; ++i)
    {
        if (kernelMs)
            kernelMs[i] = g.kernelSamples[i];
        if (intervalMs)
            intervalMs[i] = g.intervalSamples[i];
    }
    return n;
}

void solr_hip_set_short_ray_lists(int mode)
{
    onEveryDevice([&](int) { g.shortRayListsMode = mode < 0 ? -1 : (mode != 0 ? 1 : 0); });
}

int solr_hip_short_ray_lists(void)
{
    return shortRayListsChoice() ? 1 : 0;
}

void solr_hip_set_variant(int variant)
{
    onEveryDevice([&](int) {
        g.variant = variant;
        g.grouping = (variant != 5); /* takes effect at the next h2d_scene */
    });
}

int solr_hip_get_variant(void)
{
    return g.variant;
}

void solr_hip_memory_usage(unsigned long long bytes[4])
{
    bytes[0] = g.geometry.bytes + g.lamps.bytes + g.movable.bytes + g.refitPlan.bytes;
    bytes[1] = g.materials.bytes;
    bytes[2] = g.textures.bytes;
    bytes[3] = g.pp.bytes + g.ids.bytes + g.bitmap.bytes + g.randoms.bytes;
    for (int f = 0; f < MAX_FLIGHTS - 1; ++f)
        bytes[3] += g.ppX[f].bytes + g.idsX[f].bytes + g.bitmapX[f].bytes;
    for (int f = 0; f < MAX_FLIGHTS; ++f)
        bytes[3] += g.deepStack[f].bytes;
}

/* ---- multi-GPU from the C ABI: row strips gathered with RCCL, no torch ---------------------------------------
 * The reference splits the frame over the GPUs of one process inside cudaRender (CudaRayTracer.cu:1709-1815)
 * and assembles it with per-device copies in d2h_bitmap (:1647-1672).  Here it is one process per GPU: every
 * process sets its strip (solr_hip_set_strip with the rows of solr_hip_strip_rows), renders, and
 * solr_hip_gather_strips sends the strip to the root with RCCL - one grouped ncclSend / ncclRecv per peer over
 * xGMI - ENQUEUED ON THE STREAM THAT RENDERED THE FRAME, right behind the kernel: no event, no host wait; with
 * several frames in flight each flight has its own assembled-frame buffer on the root.  RCCL is loaded at run
 * time (dlopen; the copy a framework already mapped is reused), so the library needs it only when these entry
 * points are called.  The 128-byte id of ncclGetUniqueId travels from rank 0 to the others by whatever channel
 * the host application has (a file, a socket, MPI, torch's store: INTEGRATION.md). */
namespace
{
typedef struct ncclComm *ncclComm_t;
typedef struct
{
    char internal[128];
} ncclUniqueId;
struct Rccl
{
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *) = nullptr; /* (optional: one communicator per flight) */
    ncclComm_t comm = nullptr;
    /* One communicator per frame in flight (SOLR_HIP_COMM_PER_FLIGHT=1 / solr_hip_comm_set_per_flight).  RCCL orders
     * the operations of ONE communicator, whatever streams they are enqueued on: with frames in flight on several
     * streams, the gather of frame n + 1 (stream B) then waits for the gather of frame n (stream A) and the frames
     * partly serialise.  Communicators split off the first one (ncclCommSplit, same ranks) do not order against each
     * other; flightComm[f] carries the per-frame transfers of flight f (strip gather, depth halo), `comm` the blocking
     * collectives and flight 0.  Off by default until an N > 1 run has measured both (bench.py prints the mode). */
    ncclComm_t flightComm[MAX_FLIGHTS] = {};
    bool perFlight = false;
    int rank = 0, world = 0;
    DeviceBuffer frame[MAX_FLIGHTS]; /* root: the assembled RGB8 frame of each flight */
    int frameCopy[MAX_FLIGHTS] = {-1, -1, -1, -1}; /* the slot of the image ring whose copy still reads that frame, or -1 */
    DeviceBuffer idsFrame;           /* root: the assembled primitive ids (solr_hip_gather_ids) */
    int idsFlight = 0;               /* the flight whose stream carried that gather */
    DeviceBuffer zeros;              /* what a rank sends when it cannot send its own rows (see joinWith) */
    DeviceBuffer scratch;            /* the few floats of the blocking all-reduces */
    int lastFlight = 0;
    /* the rows of a neighbour's strip the ambient-occlusion taps reach, AGREED over the communicator (the maximum of
     * what the ranks derive from their own parameters and random buffers); -1: not agreed yet */
    int haloAgreed = -1;
    bool haloStale = true;  /* something it depends on was uploaded since (or nothing was agreed yet) */
    int haloParam2Bits = 0; /* PostProcessingInfo.param2 of the agreement */
    unsigned sharedSeed = 0; /* rank 0's draw at solr_hip_comm_init, the same on every rank (solr_hip_comm_shared_seed) */
} rccl;
const int RCCL_UINT8 = 1; /* ncclUint8, rccl.h:460 */
const int RCCL_INT32 = 2; /* ncclInt32 */
const int RCCL_FLOAT32 = 7; /* ncclFloat32 */
const int RCCL_SUM = 0, RCCL_MAX = 2; /* ncclSum, ncclMax */

bool loadRccl()
{
    if (rccl.lib)
        return true;
    /* SOLR_HIP_RCCL_LIBRARY: another build of the library (a site's own RCCL; tests/loopback_rccl.c, which lets
     * several ranks share the one GPU of a test box) */
    const char *named = getenv("SOLR_HIP_RCCL_LIBRARY");
    if (named && named[0])
    {
        if (!(rccl.lib = dlopen(named, RTLD_NOW | RTLD_GLOBAL)))
        {
            setError(-1, (std::string("SOLR_HIP_RCCL_LIBRARY=") + named + " could not be loaded: " + dlerror()).c_str(), __FILE__,
                     __LINE__);
            return false;
        }
    }
    else
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                break;
    if (!rccl.lib)
    {
        setError(-1, "RCCL (librccl.so) could not be loaded", __FILE__, __LINE__);
        return false;
    }
    rccl.GetUniqueId = (decltype(rccl.GetUniqueId))dlsym(rccl.lib, "ncclGetUniqueId");
    rccl.CommInitRank = (decltype(rccl.CommInitRank))dlsym(rccl.lib, "ncclCommInitRank");
    rccl.CommDestroy = (decltype(rccl.CommDestroy))dlsym(rccl.lib, "ncclCommDestroy");
    rccl.CommCount = (decltype(rccl.CommCount))dlsym(rccl.lib, "ncclCommCount");
    rccl.GroupStart = (decltype(rccl.GroupStart))dlsym(rccl.lib, "ncclGroupStart");
    rccl.GroupEnd = (decltype(rccl.GroupEnd))dlsym(rccl.lib, "ncclGroupEnd");
    rccl.Send = (decltype(rccl.Send))dlsym(rccl.lib, "ncclSend");
    rccl.Recv = (decltype(rccl.Recv))dlsym(rccl.lib, "ncclRecv");
    rccl.AllReduce = (decltype(rccl.AllReduce))dlsym(rccl.lib, "ncclAllReduce");
    rccl.GetErrorString = (decltype(rccl.GetErrorString))dlsym(rccl.lib, "ncclGetErrorString");
    rccl.CommSplit = (decltype(rccl.CommSplit))dlsym(rccl.lib, "ncclCommSplit");
    if (!rccl.GetUniqueId || !rccl.CommInitRank || !rccl.CommDestroy || !rccl.GroupStart || !rccl.GroupEnd ||
        !rccl.Send || !rccl.Recv || !rccl.AllReduce)
    {
        setError(-1, "librccl.so lacks an entry point the strip gather needs", __FILE__, __LINE__);
        dlclose(rccl.lib);
        rccl.lib = nullptr;
        return false;
    }
    return true;
}

bool rcclOk(int result, const char *what)
{
    if (result == 0)
        return true;
    std::string text = std::string(what) + ": " + (rccl.GetErrorString ? rccl.GetErrorString(result) : "RCCL error");
    setError(-1, text.c_str(), __FILE__, __LINE__);
    return false;
}

/* the communicator that carries the per-frame transfers of `flight` */
ncclComm_t commOf(int flight)
{
    return (rccl.perFlight && flight >= 0 && flight < MAX_FLIGHTS && rccl.flightComm[flight]) ? rccl.flightComm[flight] : rccl.comm;
}

/* ---- collectives that every rank joins ---------------------------------------------------------------------------
 * The ranks of a communicator run the same host program (the same sequence of C-ABI calls on
 * every rank).  A collective that one rank leaves out - because an argument check failed on it alone, because it is
 * in an error state, because its strip is not the one the others think it has - leaves the others waiting for
 * ever.  So nothing rank-local decides WHETHER a rank takes part, only WHAT it contributes:
 *   - the blocking all-reduces carry a failure slot: a rank in trouble contributes zeros and raises it, and all
 *     ranks fail together after the sum;
 *   - the point-to-point transfers behind a frame (strip gather, depth-halo exchange) have their sizes fixed by the
 *     strip table and the agreed halo height - facts every rank holds alike - and a rank that cannot send its own rows
 *     sends that many bytes of zeros, records its error and returns -1: the frame is wrong and says so, nobody hangs. */

/* a device allocation that does not depend on (or change) the engine's error state */
bool reserveQuietly(DeviceBuffer &b, size_t bytes, bool zero)
{
    bytes = std::max(bytes, (size_t)16);
    if (b.ptr && b.bytes >= bytes)
        return true;
    if (b.ptr)
        (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.bytes = 0;
    if (hipMalloc(&b.ptr, bytes) != hipSuccess)
    {
        b.ptr = nullptr;
        return false;
    }
    b.bytes = bytes;
    if (zero)
        (void)hipMemset(b.ptr, 0, bytes);
    return true;
}

/* `bytes` of zeros in HBM (the stand-in payload) */
const void *zeroPayload(size_t bytes)
{
    return reserveQuietly(rccl.zeros, bytes, true) ? rccl.zeros.ptr : nullptr;
}

/* blocking all-reduce (sum or max) of a few floats, in place, on the engine's first stream; every rank, same count.
 * Works in an error state too - that is the point. */
bool allReduceFloats(float *values, size_t n, int op, const char *what)
{
    if (!rccl.comm)
        return false;
    (void)hipSetDevice(g.device);
    const hipStream_t stream = flightStream(0);
    bool fine = reserveQuietly(rccl.scratch, n * sizeof(float), false);
    /* (a rank that cannot even allocate the few floats still has to show up: it reduces in the zero buffer) */
    void *buffer = fine ? rccl.scratch.ptr : (void *)zeroPayload(n * sizeof(float));
    if (!buffer)
    {
        setError(-1, (std::string(what) + ": no device memory for the all-reduce; the other ranks are left waiting").c_str(),
                 __FILE__, __LINE__);
        return false;
    }
    if (fine)
        fine = hipMemcpyAsync(buffer, values, n * sizeof(float), hipMemcpyHostToDevice, stream) == hipSuccess;
    const int result = rccl.AllReduce(buffer, buffer, n, RCCL_FLOAT32, op, rccl.comm, stream);
    if (result != 0)
    {
        (void)rcclOk(result, what);
        return false;
    }
    if (hipMemcpyAsync(values, buffer, n * sizeof(float), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        fine = false;
    if (!fine)
        setError(-1, (std::string(what) + ": a copy around the all-reduce failed").c_str(), __FILE__, __LINE__);
    return fine;
}

/* The rows of the neighbouring strips the 256 ambient-occlusion taps of a pixel can reach (CRT:1146-1153: 16 * param2
 * * |random| / 10 pixels), as ALL ranks will use it for the exchange below.  Each rank derives a figure from its own
 * post-processing parameters and random buffer; hosts seed their random buffers differently unless told otherwise
 * (GPUKernel.cpp:89, fillRandoms: srand(time(0))), and where 16 * param2 * reach / 10 straddles an integer two
 * neighbours would post sends and receives of different sizes.  So the figure is agreed - one all-reduce (max) -
 * whenever something it depends on was uploaded (communicator, random buffer) or param2 differs from the last
 * agreement's: events of the host program, the same on every rank, not values.  Called by every rank at the top of
 * every cudaRender with the ambient-occlusion post-process, whatever state the rank is in. */
int agreedHaloRows(const PostProcessingInfo &ppInfo)
{
    const float reach = 16.f * fabsf(ppInfo.param2) * g.randomsReach / 10.f;
    const int wanted = reach < 4096.f ? (int)reach + 2 : 4096;
    if (!rccl.comm || rccl.world < 2)
        return wanted;
    if (rccl.haloStale || rccl.haloAgreed < 0 || bitsi(ppInfo.param2) != rccl.haloParam2Bits)
    {
        float v[2] = {(float)wanted, ok() ? 0.f : 1.f};
        if (!allReduceFloats(v, 2, RCCL_MAX, "ncclAllReduce (rows of the depth halo)"))
            return wanted;
        rccl.haloAgreed = (int)v[0];
        rccl.haloStale = false;
        rccl.haloParam2Bits = bitsi(ppInfo.param2);
        if (v[1] > 0.f && ok())
            setError(-1, "cudaRender: another rank of the communicator is in an error state", __FILE__, __LINE__);
    }
    return rccl.haloAgreed;
}

/* Rank 0's random buffer to every rank (the buffer feeds the taps of the ambient-occlusion kernel, the depth of field
 * and the jitter of accumulation passes: strips rendered from different buffers do not assemble to the frame one GPU
 * renders).  With a communicator, rank 0's buffer is THE buffer: solr_hip_comm_init and every h2d_randoms after it
 * end with this.  Blocking; every rank. */
bool shareRandoms()
{
    if (!rccl.comm || rccl.world < 2)
        return true;
    rccl.haloStale = true;
    const long n = g.randoms.ptr ? g.nbRandoms : 0;
    /* the same count everywhere?  (two 16-bit halves: a float holds them exactly) */
    /* (two more slots: a seed of rank 0's, in 16-bit halves, for what the hosts draw per frame - see
     * solr_hip_comm_shared_seed) */
    unsigned draw = 0;
    if (rccl.rank == 0)
    {
        draw = (unsigned)std::chrono::steady_clock::now().time_since_epoch().count() * 2654435761u;
        draw = (draw ^ (draw >> 15)) | 1u;
    }
    float v[8] = {(float)(n >> 16), -(float)(n >> 16), (float)(n & 0xffff), -(float)(n & 0xffff), ok() ? 0.f : 1.f,
                  rccl.rank == 0 ? g.randomsReach : 0.f, (float)(draw >> 16), (float)(draw & 0xffffu)};
    if (!allReduceFloats(v, 8, RCCL_MAX, "ncclAllReduce (size of the random buffer)"))
        return false;
    if (rccl.sharedSeed == 0)
        rccl.sharedSeed = ((unsigned)v[6] << 16) | (unsigned)v[7];
    if (v[0] != -v[1] || v[2] != -v[3])
    {
        setError(-1, "the ranks of the communicator hold random buffers of different sizes (h2d_randoms on some only?)",
                 __FILE__, __LINE__);
        return false;
    }
    if (v[4] > 0.f)
    {
        if (ok())
            setError(-1, "another rank of the communicator is in an error state", __FILE__, __LINE__);
        return false; /* every rank leaves here */
    }
    if (n == 0)
        return true;
    (void)hipSetDevice(g.device);
    const hipStream_t stream = flightStream(0);
    bool fine = rcclOk(rccl.GroupStart(), "ncclGroupStart");
    if (fine && rccl.rank == 0)
        for (int r = 1; r < rccl.world && fine; ++r)
            fine = rcclOk(rccl.Send(g.randoms.ptr, (size_t)n, RCCL_FLOAT32, r, rccl.comm, stream), "ncclSend (random buffer)");
    else if (fine)
        fine = rcclOk(rccl.Recv(g.randoms.ptr, (size_t)n, RCCL_FLOAT32, 0, rccl.comm, stream), "ncclRecv (random buffer)");
    if (!rcclOk(rccl.GroupEnd(), "ncclGroupEnd"))
        fine = false;
    if (hipStreamSynchronize(stream) != hipSuccess)
        fine = false;
    if (fine)
        g.randomsReach = v[5];
    return fine;
}

/* The strips of all ranks when they are not the equal ones of solr_hip_strip_rows (solr_hip_set_strip_table) */
struct StripTable
{
    std::vector<int> first, count;
    int height = 0;
} stripTable;
void stripOf(int rank, int world, int height, int *first, int *count)
{
    if ((int)stripTable.first.size() == world && stripTable.height == height && rank >= 0 && rank < world)
    {
        if (first)
            *first = stripTable.first[rank];
        if (count)
            *count = stripTable.count[rank];
        return;
    }
    solr_hip_strip_rows(rank, world, height, first, count, nullptr);
}

/* Ambient occlusion on a strip: the 256 taps of a pixel reach up to `wanted` rows into the strips of the ranks above
 * and below ("exchange a 16-row halo").  Every rank packs the depths of its first and last
 * `wanted` rows and trades them with its neighbours - one grouped ncclSend / ncclRecv pair per neighbour, on the
 * stream that rendered the strip, between the renderer and the post-processing kernel - so that the assembled frame
 * is the one a single GPU renders.  The sizes follow from the strip table and from `wanted` = agreedHaloRows alone,
 * so neighbours always post matching transfers; pp == nullptr (a rank that returned early from cudaRender, or whose
 * strip is not the table's) sends zeros and sets the error. */
void exchangeDepthHalo(int flight, hipStream_t stream, const PixelRecord *pp, int W, int firstRow, int nbRows, int frameRows,
                       int wanted, DepthHalo *halo)
{
    if (!rccl.comm || rccl.world < 2 || wanted < 1 || W < 1 || frameRows < 1)
        return;
    int first = 0, count = 0;
    stripOf(rccl.rank, rccl.world, frameRows, &first, &count);
    if (count < 1)
        return; /* no row of the frame is this rank's: its neighbours know and trade nothing with it */
    int upRows = 0, downRows = 0;
    if (rccl.rank > 0)
        stripOf(rccl.rank - 1, rccl.world, frameRows, nullptr, &upRows);
    if (rccl.rank + 1 < rccl.world)
        stripOf(rccl.rank + 1, rccl.world, frameRows, nullptr, &downRows);
    const int mine = std::min(wanted, count);
    const int recvAbove = std::min(wanted, upRows), recvBelow = std::min(wanted, downRows);
    const int sendUp = upRows > 0 ? mine : 0, sendDown = downRows > 0 ? mine : 0;
    if (!(recvAbove || recvBelow || sendUp || sendDown))
        return;
    const size_t mineBytes = (size_t)mine * W * sizeof(float);
    bool own = pp != nullptr && ok() && first == firstRow && count == nbRows;
    if (pp != nullptr && ok() && !own)
        setError(-1, "cudaRender: this process's strip is not the one solr_hip_strip_rows (or the table of "
                     "solr_hip_set_strip_table) gives its rank; its neighbours received zeros for its boundary rows",
                 __FILE__, __LINE__);
    const bool room = reserveQuietly(g.haloAbove[flight], (size_t)std::max(recvAbove, 1) * W * sizeof(float), false) &&
                      reserveQuietly(g.haloBelow[flight], (size_t)std::max(recvBelow, 1) * W * sizeof(float), false);
    if (own && !(reserveQuietly(g.haloSendTop[flight], mineBytes, false) && reserveQuietly(g.haloSendBottom[flight], mineBytes, false)))
        own = false;
    const void *top = own ? g.haloSendTop[flight].ptr : zeroPayload(mineBytes);
    const void *bottom = own ? g.haloSendBottom[flight].ptr : top;
    if (!room || !top)
    {
        setError(-1, "cudaRender: no device memory for the depth-halo exchange; the neighbouring ranks are left waiting",
                 __FILE__, __LINE__);
        return;
    }
    if (own)
    {
        const dim3 grid((unsigned)((mine * W + 255) / 256)), block(256);
        if (sendUp)
            hipLaunchKernelGGL(k_packDepthRows, grid, block, 0, stream, pp, W, 0, mine, (float *)g.haloSendTop[flight].ptr);
        if (sendDown)
            hipLaunchKernelGGL(k_packDepthRows, grid, block, 0, stream, pp, W, nbRows - mine, mine,
                               (float *)g.haloSendBottom[flight].ptr);
        HIPCHECK(hipGetLastError());
    }
    bool fine = rcclOk(rccl.GroupStart(), "ncclGroupStart");
    if (fine && sendUp)
        fine = rcclOk(rccl.Send(top, (size_t)mine * W, RCCL_FLOAT32, rccl.rank - 1, commOf(flight), stream), "ncclSend (depth rows, up)");
    if (fine && sendDown)
        fine = rcclOk(rccl.Send(bottom, (size_t)mine * W, RCCL_FLOAT32, rccl.rank + 1, commOf(flight), stream),
                      "ncclSend (depth rows, down)");
    if (fine && recvAbove)
        fine = rcclOk(rccl.Recv(g.haloAbove[flight].ptr, (size_t)recvAbove * W, RCCL_FLOAT32, rccl.rank - 1, commOf(flight), stream),
                      "ncclRecv (depth rows, above)");
    if (fine && recvBelow)
        fine = rcclOk(rccl.Recv(g.haloBelow[flight].ptr, (size_t)recvBelow * W, RCCL_FLOAT32, rccl.rank + 1, commOf(flight), stream),
                      "ncclRecv (depth rows, below)");
    if (!rcclOk(rccl.GroupEnd(), "ncclGroupEnd") || !fine || !own || !halo)
        return;
    halo->above = (const float *)g.haloAbove[flight].ptr;
    halo->below = (const float *)g.haloBelow[flight].ptr;
    halo->nbAbove = recvAbove;
    halo->nbBelow = recvBelow;
}

/* (for renderImpl, which is defined before this layer) */
bool haveCommunicator()
{
    return rccl.comm != nullptr && rccl.world > 1;
}
} // namespace

/* Extension: the depths (PostProcessingBuffer.colorInfo.w) of the rows next to this process's strip that other
 * processes rendered - nbAbove rows of `width` floats just above it, nbBelow just below - for hosts that move them
 * themselves (MPI, shared memory; tests/test_gpu_parity.py does it from a full frame).  Used by the ambient-occlusion
 * kernel of the frames that follow, in place of the RCCL exchange; (NULL, 0, NULL, 0) ends it. */
/* Extension: nodes per order-free list of the resident scene if closest-hit walks of long rays use them (the
 * lists exist, every primitive lies inside its leaf's box, no rotation on the device since the upload, variant
 * not 6), else 0. */
extern "C" int solr_hip_order_free_nodes(void)
{
    return (g.initialized && orderFreeListsUsable()) ? g.nbBoxesFree : 0;
}

/* Extension: 1 if the shadow walks of the resident scene take the order-free lists as well (they are in use and
 * nothing in the scene is transparent or a textured plane), else 0. */
extern "C" int solr_hip_order_free_shadows(void)
{
    return (g.initialized && orderFreeListsUsable() && g.opaqueShadows) ? 1 : 0;
}

extern "C" void solr_hip_set_depth_halo(const float *above, int nbAbove, const float *below, int nbBelow)
{
    if (!ready("solr_hip_set_depth_halo"))
        return;
    ARGCHECK(nbAbove >= 0 && nbBelow >= 0 && (nbAbove == 0 || above) && (nbBelow == 0 || below) && nbAbove <= 4096 &&
                 nbBelow <= 4096,
             "solr_hip_set_depth_halo: bad arguments");
    if (!ok())
        return;
    quiesce();
    HIPCHECK(hipSetDevice(g.device));
    g.haloSuppliedAbove = g.haloSuppliedBelow = 0;
    if (nbAbove)
    {
        std::vector<float> rows(above, above + (size_t)nbAbove * g.width);
        upload(g.haloGivenAbove, rows);
    }
    if (nbBelow)
    {
        std::vector<float> rows(below, below + (size_t)nbBelow * g.width);
        upload(g.haloGivenBelow, rows);
    }
    if (ok())
    {
        g.haloSuppliedAbove = nbAbove;
        g.haloSuppliedBelow = nbBelow;
    }
}

/* rows [first, first + count) of a `height`-row image for rank `rank` of `world`, and the common strip height:
 * contiguous strips like the reference's (CudaRayTracer.cu:1694-1696), the last one absorbing the remainder;
 * trailing ranks get no row when there are more ranks than rows to share out (solr_hip_set_strip(first, 0)) */
void solr_hip_strip_rows(int rank, int world, int height, int *first, int *count, int *rowsPerRank)
{
    const int per = world > 0 ? (height + world - 1) / world : height;
    const int f = rank * per;
    int c = height - f;
    c = c < 0 ? 0 : (c > per ? per : c);
    if (first)
        *first = f;
    if (count)
        *count = c;
    if (rowsPerRank)
        *rowsPerRank = per;
}

/* Cost-balanced strips.  Equal strips share out rows, not work: of the 100k-triangle mesh the strip at the
 * horizon takes 0.22 ms, the one at the bottom 0.012 (profiles/r2/strip_throughput_height_field.txt), and the
 * frame is as slow as its slowest rank.  rowCost[y] is what row y costs (any unit; solr_hip_strip_row_costs,
 * summed over the ranks by the host's control plane or solr_hip_balance_strips): contiguous strips whose
 * boundaries are multiples of `align` rows (8 = the tiles' height: a tile's cost then belongs to one strip)
 * chosen where the running sum is nearest to r / world of the total.  Every rank keeps at least `align` rows
 * while there are enough; rows without a cost count as a thousandth of the mean, so a frame that has not been
 * rendered yet gives the equal split.  Pure host arithmetic, the same on every rank. */
int solr_hip_balanced_strips(const float *rowCost, int height, int world, int align, int *firstRows, int *nbRows)
{
    if (!rowCost || height < 1 || world < 1 || align < 1 || !firstRows || !nbRows)
        return -1;
    const int blocks = (height + align - 1) / align;
    std::vector<double> prefix((size_t)blocks + 1, 0.0);
    double total = 0.0;
    for (int y = 0; y < height; ++y)
        if (rowCost[y] > 0.f && rowCost[y] < 1e30f)
            total += rowCost[y];
    const double floor = total > 0.0 ? 1e-3 * total / height : 1.0;
    for (int b = 0; b < blocks; ++b)
    {
        double sum = 0.0;
        for (int y = b * align; y < std::min(height, (b + 1) * align); ++y)
            sum += floor + ((rowCost[y] > 0.f && rowCost[y] < 1e30f) ? (double)rowCost[y] : 0.0);
        prefix[(size_t)b + 1] = prefix[b] + sum;
    }
    const double all = prefix[blocks];
    std::vector<int> cut((size_t)world + 1, 0); /* in blocks */
    cut[world] = blocks;
    int at = 0;
    for (int r = 1; r < world; ++r)
    {
        const double target = all * r / world;
        while (at < blocks && prefix[(size_t)at + 1] <= target)
            ++at; /* prefix[at] <= target < prefix[at + 1] */
        int best = (at < blocks && prefix[(size_t)at + 1] - target < target - prefix[at]) ? at + 1 : at;
        /* at least one block for every rank if there are that many (else whoever the sums leave without) */
        const bool room = blocks >= world;
        best = std::max(best, cut[r - 1] + (room ? 1 : 0));
        best = std::min(best, room ? blocks - (world - r) : blocks);
        cut[r] = best;
    }
    for (int r = 0; r < world; ++r)
    {
        const int from = std::min(height, cut[r] * align), to = std::min(height, cut[r + 1] * align);
        firstRows[r] = from;
        nbRows[r] = std::max(0, to - from);
    }
    return 0;
}

/* The strips of all ranks, when they are not solr_hip_strip_rows' (balanced ones): what solr_hip_gather_strips
 * and the depth-halo exchange take the other ranks' rows from.  Contiguous, in rank order, covering the frame;
 * world = 0 forgets the table.  This process's own strip is still set with solr_hip_set_strip. */
int solr_hip_set_strip_table(const int *firstRows, const int *nbRows, int world, int height)
{
    if (world == 0 || !firstRows || !nbRows)
    {
        stripTable.first.clear();
        stripTable.count.clear();
        stripTable.height = 0;
        return 0;
    }
    int next = 0;
    bool fine = world > 0 && height > 0;
    for (int r = 0; fine && r < world; ++r)
    {
        fine = nbRows[r] >= 0 && (nbRows[r] == 0 || firstRows[r] == next);
        next += nbRows[r];
    }
    if (!fine || next != height)
    {
        setError(1, "solr_hip_set_strip_table: the strips are not contiguous, in rank order and covering the frame",
                 __FILE__, __LINE__);
        return -1;
    }
    stripTable.first.assign(firstRows, firstRows + world);
    stripTable.count.assign(nbRows, nbRows + world);
    stripTable.height = height;
    return 0;
}

/* What the rows of this process's strip cost in the frame rendered last: rowCost[y] for the rows of the strip
 * (frame coordinates; a tile's measured duration shared out over its rows), 0 elsewhere.  Needs tile scheduling
 * (solr_hip_set_tile_scheduling 1 or 2, the default) and a frame; waits for the frames in flight. */
int solr_hip_strip_row_costs(float *rowCost, int height)
{
    if (!ready("solr_hip_strip_row_costs"))
        return -1;
    ARGCHECK(rowCost != nullptr && height == g.height, "solr_hip_strip_row_costs: rowCost[height of the frame]");
    ARGCHECK(g.tileCost.ptr != nullptr && g.costFrames > 0 && g.costKey[0] > 0 && g.costKey[1] > 0,
             "solr_hip_strip_row_costs: no frame has recorded tile costs (tile scheduling off?)");
    if (!ok())
        return -1;
    quiesce();
    const int nbTiles = (int)g.costKey[0], tilesX = (int)g.costKey[1], firstRow = (int)g.costKey[2], nbRows = (int)g.costKey[3];
    std::vector<unsigned> cost((size_t)nbTiles);
    HIPCHECK(hipMemcpy(cost.data(), g.tileCost.ptr, cost.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    if (!ok())
        return -1;
    for (int y = 0; y < height; ++y)
        rowCost[y] = 0.f;
    for (int t = 0; t < nbTiles; ++t)
    {
        const int y0 = (t / tilesX) * TILE, y1 = std::min(nbRows, y0 + TILE);
        for (int y = y0; y < y1; ++y)
            if (firstRow + y < height)
                rowCost[firstRow + y] += (float)cost[t] / (float)(y1 - y0);
    }
    return 0;
}

/* Every rank, between frames, after a few frames on the current strips: the rows' costs of all ranks summed
 * (one ncclAllReduce of `height` floats), balanced strips from them, the table for the gather and this
 * process's own strip set - the next cudaRender renders it.  A host without a control plane of its own needs
 * nothing else; one that has (torch.distributed in bench.py) can do the sum there and call
 * solr_hip_balanced_strips + solr_hip_set_strip_table + solr_hip_set_strip itself.
 * Two all-reduces, and every rank that has a communicator takes part in both whatever its own state: first the
 * maximum of {rows the ambient-occlusion taps reach beyond a strip, a failure flag, the frame height and its
 * negative} - the halo exchange delivers rows of the next rank only, so no strip may be lower than the LARGEST reach
 * any rank has seen, every rank must cut with the same `align`, and ranks that disagree about the frame must not meet
 * in a sum of different lengths - then, if nobody failed, the sum of the rows' costs. */
int solr_hip_balance_strips(void)
{
    if (!g.initialized || !rccl.comm)
    {
        /* the same on every rank of a correct program: nobody is waiting */
        if (ok())
            setError(-1, !g.initialized ? "solr_hip_balance_strips: initialize_scene has not been called"
                                        : "solr_hip_balance_strips: no communicator (solr_hip_comm_init)",
                     __FILE__, __LINE__);
        return -1;
    }
    bool mine = ok();
    const int height = g.height;
    std::vector<float> cost((size_t)std::max(height, 1), 0.f);
    /* a rank that has nothing to report (an empty strip, tile scheduling off, no frame yet) contributes zeros */
    const bool recorded = mine && height > 0 && stripRows() > 0 && g.tileCost.ptr != nullptr && g.costFrames > 0 && g.costKey[0] > 0;
    if (recorded && solr_hip_strip_row_costs(cost.data(), height) != 0)
    {
        mine = false;
        std::fill(cost.begin(), cost.end(), 0.f);
    }
    quiesce();
    float head[4] = {(float)std::max(g.haloWanted, 0), mine ? 0.f : 1.f, (float)height, -(float)height};
    if (!allReduceFloats(head, 4, RCCL_MAX, "ncclAllReduce (balance: reach, failures, frame height)"))
        return -1;
    if (head[1] > 0.f || head[2] != -head[3] || height < 1)
    {
        if (ok())
            setError(-1, head[1] > 0.f ? "solr_hip_balance_strips: another rank could not report its rows' costs"
                                       : "solr_hip_balance_strips: the ranks do not render frames of the same height",
                     __FILE__, __LINE__);
        return -1; /* on every rank */
    }
    if (!allReduceFloats(cost.data(), (size_t)height, RCCL_SUM, "ncclAllReduce (balance: rows' costs)"))
        return -1;
    const int reach = (int)head[0];
    const int align = std::max(TILE, (reach + TILE - 1) / TILE * TILE);
    std::vector<int> first((size_t)rccl.world), count((size_t)rccl.world);
    if (solr_hip_balanced_strips(cost.data(), height, rccl.world, align, first.data(), count.data()) != 0 ||
        solr_hip_set_strip_table(first.data(), count.data(), rccl.world, height) != 0)
        return -1;
    solr_hip_set_strip(first[rccl.rank], count[rccl.rank]);
    return ok() ? 0 : -1;
}

int solr_hip_comm_unique_id(void *id128)
{
    if (!id128 || !loadRccl())
        return -1;
    ncclUniqueId id;
    if (!rcclOk(rccl.GetUniqueId(&id), "ncclGetUniqueId"))
        return -1;
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

/* Joins the communicator and, when it has more than one rank, makes rank 0's random buffer everybody's (see
 * shareRandoms; hosts seed theirs from the clock unless told otherwise).  Every rank, after initialize_scene and
 * after the uploads of its first frame. */
int solr_hip_comm_init(int rank, int world, const void *id128)
{
    if (gDevices > 1)
    {
        setError(-1, "solr_hip_comm_init: this process renders on several devices (occupancyParameters.x > 1); a "
                     "communicator belongs to the one-process-per-GPU model", __FILE__, __LINE__);
        return -1;
    }
    if (!ready("solr_hip_comm_init") || !loadRccl())
        return -1;
    ARGCHECK(id128 != nullptr && world >= 1 && rank >= 0 && rank < world, "solr_hip_comm_init: bad arguments");
    ARGCHECK(rccl.comm == nullptr, "solr_hip_comm_init: a communicator exists already");
    if (!ok())
        return -1;
    HIPCHECK(hipSetDevice(g.device));
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    if (!rcclOk(rccl.CommInitRank(&rccl.comm, world, id, rank), "ncclCommInitRank"))
    {
        rccl.comm = nullptr;
        return -1;
    }
    rccl.rank = rank;
    rccl.world = world;
    rccl.haloAgreed = -1;
    rccl.haloStale = true;
    for (ncclComm_t &c : rccl.flightComm)
        c = nullptr;
    for (int &slot : rccl.frameCopy)
        slot = -1;
    const char *env = getenv("SOLR_HIP_COMM_PER_FLIGHT");
    if (env && env[0])
        rccl.perFlight = env[0] != '0';
    if (rccl.perFlight)
    {
        /* every rank, in the same order: the split is a collective of the parent communicator */
        if (!rccl.CommSplit)
        {
            setError(-1, "solr_hip_comm_init: one communicator per flight was asked for and this RCCL has no ncclCommSplit",
                     __FILE__, __LINE__);
            return -1;
        }
        rccl.flightComm[0] = rccl.comm;
        for (int f = 1; f < MAX_FLIGHTS; ++f)
            if (!rcclOk(rccl.CommSplit(rccl.comm, 0, rank, &rccl.flightComm[f], nullptr), "ncclCommSplit"))
            {
                rccl.flightComm[f] = nullptr;
                return -1;
            }
    }
    quiesce();
    if (!shareRandoms())
        return -1;
    return ok() ? 0 : -1;
}

/* before solr_hip_comm_init: 1 = one communicator per frame in flight (see struct Rccl), 0 = one for everything (the
 * default; SOLR_HIP_COMM_PER_FLIGHT in the environment overrides either).  Every rank alike. */
void solr_hip_comm_set_per_flight(int on)
{
    rccl.perFlight = on != 0;
}

/* communicators this process holds: 0 without one, 1, or one per possible flight */
int solr_hip_comm_count(void)
{
    if (!rccl.comm)
        return 0;
    int n = 1;
    for (int f = 1; f < MAX_FLIGHTS; ++f)
        if (rccl.perFlight && rccl.flightComm[f])
            ++n;
    return n;
}

/* A number every rank of the communicator holds alike (rank 0 drew it at solr_hip_comm_init), 0 without a
 * communicator of more than one rank.  What a host draws per frame - GPUKernel::render_begin takes the frame's
 * timestamp from rand() (GPUKernel.cpp:2712-2727), and the timestamp indexes the random buffer in the shader, the
 * depth of field and the procedural spheres - has to be the same on every rank or the strips do not assemble to one
 * frame: hosts seed a generator of their own with this (sol-r_amd/host/HipKernel.cpp does) instead of talking to
 * each other every frame. */
unsigned solr_hip_comm_shared_seed(void)
{
    return (rccl.comm && rccl.world > 1) ? rccl.sharedSeed : 0u;
}

/* ranks of the communicator as the library itself reports them (ncclCommCount), 0 without one */
int solr_hip_comm_ranks(void)
{
    if (!rccl.comm)
        return 0;
    int n = rccl.world;
    if (rccl.CommCount && rccl.CommCount(rccl.comm, &n) != 0)
        return -1;
    return n;
}

namespace
{
/* One gather: rows [first, first + count) of this rank -> `root`, `rowBytes` per row, on `stream`; the root receives
 * every rank's rows at their place in `assembled`.  Sizes come from the strip table alone; `own` == nullptr (this
 * rank cannot send its own rows) sends zeros. */
bool gatherRows(int root, const void *own, void *assembled, size_t rowBytes, int datatype, size_t perByte, hipStream_t stream,
                ncclComm_t comm, const char *what)
{
    int first = 0, count = 0;
    stripOf(rccl.rank, rccl.world, g.height, &first, &count);
    if (rccl.world == 1) /* one process: the strip is whatever was set, the "gather" a copy into the frame */
    {
        first = g.nbRows >= 0 ? g.firstRow : 0;
        count = stripRows();
    }
    const void *payload = own;
    if (!payload && count > 0 && !(payload = zeroPayload((size_t)count * rowBytes)))
    {
        setError(-1, (std::string(what) + ": no device memory; the other ranks are left waiting").c_str(), __FILE__, __LINE__);
        return false;
    }
    bool fine = rcclOk(rccl.GroupStart(), "ncclGroupStart");
    if (fine && rccl.rank == root && assembled)
        for (int r = 0; r < rccl.world && fine; ++r)
        {
            int rf = first, rc = count;
            if (rccl.world > 1)
                stripOf(r, rccl.world, g.height, &rf, &rc);
            if (rc > 0)
                fine = rcclOk(rccl.Recv((char *)assembled + (size_t)rf * rowBytes, (size_t)rc * rowBytes / perByte, datatype, r,
                                        comm, stream),
                              "ncclRecv");
        }
    if (fine && count > 0)
        fine = rcclOk(rccl.Send(payload, (size_t)count * rowBytes / perByte, datatype, root, comm, stream), "ncclSend");
    if (!rcclOk(rccl.GroupEnd(), "ncclGroupEnd"))
        fine = false;
    return fine;
}

/* this process's strip is the one its rank has in the eyes of the others */
bool stripIsTheTables()
{
    if (rccl.world == 1)
        return true;
    int first = 0, count = 0;
    stripOf(rccl.rank, rccl.world, g.height, &first, &count);
    return stripRows() == count && (count == 0 || (g.nbRows >= 0 ? g.firstRow : 0) == first);
}

int gatherImpl(int root, bool ids, const char *who)
{
    HostSpan whole("solr_hip_gather_strips / _ids");
    if (!g.initialized || !rccl.comm || root < 0 || root >= rccl.world || g.width < 1 || g.height < 1)
    {
        /* program errors, the same on every rank: nobody is waiting */
        if (ok())
            setError(-1, (std::string(who) + (!g.initialized ? ": initialize_scene has not been called"
                                              : !rccl.comm   ? ": no communicator (solr_hip_comm_init)"
                                              : g.width < 1  ? ": no frame was rendered"
                                                             : ": no such root")).c_str(),
                     __FILE__, __LINE__);
        return -1;
    }
    (void)hipSetDevice(g.device);
    const int flight = g.current;
    const hipStream_t stream = flightStream(flight);
    const size_t rowBytes = (size_t)g.width * (ids ? sizeof(PrimitiveXYIdBuffer) : (size_t)SOLR_COLOR_DEPTH);
    /* rank-local trouble decides what is sent, not whether (see the note on collectives above) */
    bool mine = ok();
    if (mine && !stripIsTheTables())
    {
        setError(-1, (std::string(who) + ": this process's strip is not the one solr_hip_strip_rows (or the table of "
                                         "solr_hip_set_strip_table) gives its rank; the root received zeros for its rows").c_str(),
                 __FILE__, __LINE__);
        mine = false;
    }
    const void *src = ids ? flightIds(flight).ptr : (g.boundBitmap ? g.boundBitmap : flightBitmap(flight).ptr);
    if (!src)
        mine = false;
    DeviceBuffer &assembled = ids ? rccl.idsFrame : rccl.frame[flight];
    if (rccl.rank == root && !reserveQuietly(assembled, (size_t)g.height * rowBytes, true))
    {
        setError(-1, (std::string(who) + ": no device memory for the assembled frame; the other ranks are left waiting").c_str(),
                 __FILE__, __LINE__);
        return -1;
    }
    if (!ids && rccl.rank == root && rccl.frameCopy[flight] >= 0)
    {
        /* a pipelined read-back (solr_hip_d2h_gathered_async) may still be copying the frame this flight assembled
         * last: the gather that overwrites it goes behind that copy */
        if (gFirst.imageDone[rccl.frameCopy[flight]])
            (void)hipStreamWaitEvent(stream, gFirst.imageDone[rccl.frameCopy[flight]], 0);
        rccl.frameCopy[flight] = -1;
    }
    const bool fine = gatherRows(root, mine ? src : nullptr, rccl.rank == root ? assembled.ptr : nullptr, rowBytes,
                                 ids ? RCCL_INT32 : RCCL_UINT8, ids ? 4 : 1, stream, commOf(flight), who);
    if (!ids)
        rccl.lastFlight = flight;
    else
        rccl.idsFlight = flight;
    return (fine && mine && ok()) ? 0 : -1;
}
} // namespace

/* the strip of the frame rendered last -> `root`, on that frame's stream.  Every rank calls it once per frame,
 * in the same order of frames.  Returns immediately. */
int solr_hip_gather_strips(int root)
{
    return gatherImpl(root, false, "solr_hip_gather_strips");
}

/* Picking on an N-GPU frame (GPUKernel::getPrimitiveAt, GPUKernel.cpp:729-739, reads primitivesXYIds of the whole
 * frame; the reference's d2h_bitmap copies every device's strip of them after every frame, CudaRayTracer.cu:1664-1670):
 * the PrimitiveXYIdBuffer strips of the frame rendered last -> `root`, 16 bytes per pixel - five times the image, so
 * on demand, when picking asks, not per frame.  Every rank calls it; solr_hip_d2h_gathered_ids on the root waits and
 * copies the assembled height x width records to the host. */
int solr_hip_gather_ids(int root)
{
    return gatherImpl(root, true, "solr_hip_gather_ids");
}

/* root: the assembled frame of the gather issued last (device memory, height x width x 3; valid once the
 * stream has run the gather - solr_hip_d2h_gathered waits for it) */
void *solr_hip_gathered_frame(void)
{
    return rccl.frame[rccl.lastFlight].ptr;
}

int solr_hip_d2h_gathered(BitmapBuffer *hostBitmap)
{
    if (!ready("solr_hip_d2h_gathered"))
        return -1;
    ARGCHECK(hostBitmap != nullptr && rccl.frame[rccl.lastFlight].ptr != nullptr,
             "solr_hip_d2h_gathered: nothing was gathered on this rank");
    if (!ok())
        return -1;
    const hipStream_t stream = flightStream(rccl.lastFlight);
    HIPCHECK(hipMemcpyAsync(hostBitmap, rccl.frame[rccl.lastFlight].ptr, (size_t)g.height * g.width * SOLR_COLOR_DEPTH,
                            hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    return ok() ? 0 : -1;
}

/* The delivered frame of an N-GPU job, pipelined: on the root, the assembled frame of the gather issued last is copied
 * to a page-locked host image on the engine's copy stream, behind that gather, and a ticket comes back at once
 * (solr_hip_image_wait(ticket) waits for it and returns the image) - the read-back of frame n overlaps the rendering
 * and the gather of frames n + 1 ..., like solr_hip_d2h_image_async does on one GPU.  The next gather into the same
 * flight's frame waits for the copy.  On the other ranks: nothing to deliver, returns -2 (no error). */
int solr_hip_d2h_gathered_async(void)
{
    if (!ready("solr_hip_d2h_gathered_async"))
        return -1;
    if (!rccl.comm || !rccl.frame[rccl.lastFlight].ptr)
    {
        if (!g.sharedRing)
            return -2;
        /* with a ring the ranks share (solr_hip_image_share) every rank takes the ticket, so that the ranks keep
         * counting alike; only the root has something to copy */
        int slot = 0;
        const int ticket = nextTicket(&slot);
        g.slotOfStrips[slot] = false;
        return ticket;
    }
    HIPCHECK(hipSetDevice(g.device));
    if (!ensureImageRing())
        return -1;
    ensureCopyStream();
    if (!ok())
        return -1;
    const int flight = rccl.lastFlight;
    int slot = 0;
    const int ticket = nextTicket(&slot);
    HIPCHECK(hipEventRecord(g.frameRendered, flightStream(flight)));
    HIPCHECK(hipStreamWaitEvent(g.copyStream, g.frameRendered, 0));
    HIPCHECK(hipMemcpyAsync(g.pinnedImage[slot], rccl.frame[flight].ptr, (size_t)g.height * g.width * SOLR_COLOR_DEPTH,
                            hipMemcpyDeviceToHost, g.copyStream));
    HIPCHECK(hipEventRecord(g.imageDone[slot], g.copyStream));
    g.slotOfStrips[slot] = false;
    rccl.frameCopy[flight] = slot;
    return ok() ? ticket : -1;
}

int solr_hip_d2h_gathered_ids(PrimitiveXYIdBuffer *hostIds)
{
    if (!ready("solr_hip_d2h_gathered_ids"))
        return -1;
    ARGCHECK(hostIds != nullptr && rccl.idsFrame.ptr != nullptr, "solr_hip_d2h_gathered_ids: nothing was gathered on this rank");
    if (!ok())
        return -1;
    const hipStream_t stream = flightStream(rccl.idsFlight);
    HIPCHECK(hipMemcpyAsync(hostIds, rccl.idsFrame.ptr, (size_t)g.height * g.width * sizeof(PrimitiveXYIdBuffer),
                            hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    return ok() ? 0 : -1;
}

void solr_hip_comm_finalize(void)
{
    solr_hip_set_strip_table(nullptr, nullptr, 0, 0); /* the table was that communicator's */
    if (rccl.comm)
    {
        (void)hipDeviceSynchronize();
        for (int f = 1; f < MAX_FLIGHTS; ++f)
            if (rccl.flightComm[f] && rccl.flightComm[f] != rccl.comm)
                (void)rccl.CommDestroy(rccl.flightComm[f]);
        (void)rccl.CommDestroy(rccl.comm);
        rccl.comm = nullptr;
    }
    for (ncclComm_t &c : rccl.flightComm)
        c = nullptr;
    for (int &slot : rccl.frameCopy)
        slot = -1;
    for (DeviceBuffer &b : rccl.frame)
        release(b);
    release(rccl.idsFrame);
    release(rccl.zeros);
    release(rccl.scratch);
    rccl.world = 0;
    rccl.haloAgreed = -1;
    rccl.haloStale = true;
    rccl.sharedSeed = 0;
}

#ifdef SOLR_TIMING
/* development build only (tools/wave_time_split.py): shader-clock cycles summed over the waves of every frame
 * since the last reset - [0] whole kernel, [1] closest-hit walks, [2] shadow walks, [3] node loop, [4] leaves,
 * [5] calls of the node loop, [6] leaf visits, [7] waves, [8] primitiveShader (its shadow walks included), [9] launchRayTracing,
 * [10] from the end of the trace to the end of the kernel */
void solr_hip_wave_cycles(unsigned long long out[16], int reset)
{
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> slots(16 * SOLR_TIMING_SLOTS);
    (void)hipMemcpy(slots.data(), (unsigned long long *)g.counters.ptr + 16, slots.size() * sizeof(unsigned long long),
                    hipMemcpyDeviceToHost);
    if (out)
        for (int k = 0; k < 16; ++k)
        {
            out[k] = 0;
            for (size_t w = 0; w < SOLR_TIMING_SLOTS; ++w)
                out[k] += slots[16 * w + k];
        }
    if (reset)
        (void)hipMemset((unsigned long long *)g.counters.ptr + 16, 0, slots.size() * sizeof(unsigned long long));
}
/* the same counters per workgroup (16 words each; [11] the primary ray's closest-hit walk, [12] second attempts of the
 * checked unit-ray walks, [13] their count << 32 | their lanes, [14] node-loop calls of closest-hit walks: first attempt
 * << 32 | second, [15] checked walks); returns the workgroups copied */
int solr_hip_wave_cycle_slots(unsigned long long *out, int capacityWorkgroups)
{
    (void)hipDeviceSynchronize();
    const size_t n = std::min((size_t)std::max(capacityWorkgroups, 0), (size_t)SOLR_TIMING_SLOTS);
    (void)hipMemcpy(out, (unsigned long long *)g.counters.ptr + 16, 16 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    return (int)n;
}
#endif
}

/* ======================================================================= */
/* The boundary, once per in-process device                                  */
/* ======================================================================= */
/* CudaRayTracer.h:25-67 with occupancyParameters.x honoured as the reference honours it: that many devices of THIS
 * process (clamped to the devices there are, CudaRayTracer.cu:1413-1424), every upload repeated per device
 * (:1536-1625), the frame cut into equal row strips, device d rendering strip d (:1694-1696, 1709-1815), and
 * d2h_bitmap copying every device's strip to its place in the host arrays (:1647-1672).  One device - what every
 * caller of this library but a host that edits CudaKernel.cpp:90 asks for - is engine 0 alone and nothing below adds
 * to it.  occupancyParameters.y (streams per device; the reference's arithmetic for it is broken) is
 * accepted and not used: how a device's strip is scheduled is the engine's business, like blockSize.
 * Several devices in one process and the one-process-per-GPU model (solr_hip_set_strip, solr_hip_comm_*) do not
 * combine; either refuses the other.  Neighbourhood post-processing (ambient occlusion, depth of field, radiosity,
 * filter) stays inside a device's strip here, as in the reference (the multi-process path trades the boundary rows).
 * SOLR_HIP_VIRTUAL_DEVICES=n (tests on a one-GPU box): pretend n devices, engine d on device d mod the real ones. */
namespace
{
bool gSplit = false; /* the strips of the engines are the in-process split's */
int gAskNext = 1;    /* solr_hip_set_gpu_count: what solr_hip_initialize (the pointer form) asks for */

int devicesThereAre()
{
    int real = 0;
    if (hipGetDeviceCount(&real) != hipSuccess)
        real = 0;
    /* (read at every initialize_scene, not once per process: a test that sets it must be able to take it away again) */
    const char *const env = getenv("SOLR_HIP_VIRTUAL_DEVICES");
    const int pretend = env ? atoi(env) : 0;
    return (real > 0 && pretend > real) ? pretend : real;
}

/* engines 0 ... n - 1 exist; the later ones take the settings of engine 0 and the devices that follow its device */
void ensureEngines(int n)
{
    int real = 0;
    if (hipGetDeviceCount(&real) != hipSuccess || real < 1)
        real = 1;
    for (int d = 1; d < n && d < SOLR_MAX_GPU_COUNT; ++d)
    {
        if (!gEngines[d])
            gEngines[d] = new Engine;
        Engine &e = *gEngines[d];
        e.device = (gFirst.device + d) % real;
        e.variant = gFirst.variant;
        e.shortRayListsMode = gFirst.shortRayListsMode;
        e.grouping = gFirst.grouping;
        e.flights = gFirst.flights;
        e.tileScheduling = gFirst.tileScheduling;
        e.timing = gFirst.timing;
    }
}

/* equal row strips, device d the d-th (the split of solr_hip_strip_rows, what the multi-process path uses too) */
void splitRows(int height)
{
    if (gDevices < 2)
    {
        if (gSplit)
        {
            gSplit = false;
            quiesce();
            gFirst.firstRow = 0;
            gFirst.nbRows = -1;
        }
        return;
    }
    gSplit = true;
    onEveryDevice([&](int d) {
        int first = 0, count = 0;
        solr_hip_strip_rows(d, gDevices, height, &first, &count, nullptr);
        if (g.firstRow != first || g.nbRows != count)
        {
            quiesce();
            g.firstRow = first;
            g.nbRows = count;
        }
    });
}

/* The reference reads occupancyParameters.x anew in every call (CudaRayTracer.cu:1647-1672, 1694 ...): a host that
 * hands initialize_scene N and a later call another count gets that call on the devices that count names.  Here the
 * engines initialize_scene set up all hold the scene and their strips of the frame, so every call runs on all of them
 * whatever it says - a frame is never missing a strip - and a count that differs is noted once, not turned into a
 * sticky error that ends the process's frames. */
bool sameOccupancy(const vec2i &occ, const char *who)
{
    if (occ.x < 1 || occ.x == gRequested || (occ.x > SOLR_MAX_GPU_COUNT && gRequested == SOLR_MAX_GPU_COUNT))
        return true;
    static bool noted = false;
    if (!noted)
        fprintf(stderr, "solr_hip: %s was handed occupancyParameters.x = %d, initialize_scene %d: the %d device(s) set up then "
                        "serve this and every later call (noted once)\n", who, occ.x, gRequested, gDevices);
    noted = true;
    return true;
}
} // namespace

extern "C" {

void initialize_scene(vec2i occupancyParameters, SceneInfo sceneInfo, int, int, int)
{
    if (solr_hip_last_error(nullptr, 0) != 0)
        return;
    int asked = occupancyParameters.x < 1 ? 1 : occupancyParameters.x;
    if (asked > SOLR_MAX_GPU_COUNT)
        asked = SOLR_MAX_GPU_COUNT; /* CudaRayTracer.cu:1415-1416 */
    int use = asked;
    if (asked > 1)
    {
        if (rccl.comm)
        {
            setError(-1, "initialize_scene: occupancyParameters.x > 1 asks for several devices in this process, which has "
                         "joined a communicator (one process per GPU): the two do not combine", __FILE__, __LINE__);
            return;
        }
        const int have = devicesThereAre();
        if (asked > have)
        {
            /* CudaRayTracer.cu:1419-1424: "You asked for n CUDA-capable devices, but only m are available" */
            fprintf(stderr, "solr_hip: initialize_scene was asked for %d devices (occupancyParameters.x), %d are available\n",
                    asked, have);
            use = have < 1 ? 1 : have;
        }
    }
    /* engines an earlier call set up beyond what this one uses */
    for (int d = use; d < SOLR_MAX_GPU_COUNT; ++d)
        if (d > 0 && gEngines[d] && gEngines[d]->initialized)
        {
            gCurrent = gEngines[d];
            (void)hipSetDevice(g.device);
            finalizeOne();
            gCurrent = &gFirst;
        }
    gRequested = asked;
    gDevices = use;
    ensureEngines(use);
    onEveryDevice([&](int) { initializeOne(sceneInfo); });
    splitRows(sceneInfo.size.y);
}

void solr_hip_set_gpu_count(int n)
{
    gAskNext = n < 1 ? 1 : n;
}

int solr_hip_gpu_count(void)
{
    return gDevices;
}

void solr_hip_initialize(const SceneInfo *sceneInfo)
{
    vec2i occ;
    occ.x = gAskNext;
    occ.y = 1;
    initialize_scene(occ, *sceneInfo, 0, 0, 0);
}

void finalize_scene(vec2i)
{
    /* every engine that is up, whatever occupancyParameters says by now; afterwards the process is a one-device
     * process again until initialize_scene says otherwise */
    const bool several = gDevices > 1;
    for (int d = 0; d < SOLR_MAX_GPU_COUNT; ++d)
        if (gEngines[d] && gEngines[d]->initialized)
        {
            gCurrent = gEngines[d];
            if (several)
                (void)hipSetDevice(g.device);
            finalizeOne();
        }
    gCurrent = &gFirst;
    gDevices = 1;
    gRequested = 1;
    if (several)
        (void)hipSetDevice(g.device);
}

void reshape_scene(vec2i occupancyParameters, SceneInfo sceneInfo)
{
    if (!sameOccupancy(occupancyParameters, "reshape_scene"))
        return;
    splitRows(sceneInfo.size.y);
    onEveryDevice([&](int) { reshapeOne(sceneInfo); });
}

void solr_hip_reshape(const SceneInfo *sceneInfo)
{
    vec2i occ;
    occ.x = 0;
    occ.y = 1;
    reshape_scene(occ, *sceneInfo);
}

void h2d_scene(vec2i occupancyParameters, BoundingBox *boundingBoxes, int nbActiveBoxes, Primitive *primitives,
               int nbPrimitives, Lamp *lamps, int nbLamps)
{
    if (!sameOccupancy(occupancyParameters, "h2d_scene"))
        return;
    onEveryDevice([&](int) { h2dSceneOne(boundingBoxes, nbActiveBoxes, primitives, nbPrimitives, lamps, nbLamps); });
}

void h2d_materials(vec2i occupancyParameters, Material *materials, int nbActiveMaterials)
{
    if (!sameOccupancy(occupancyParameters, "h2d_materials"))
        return;
    onEveryDevice([&](int) { h2dMaterialsOne(materials, nbActiveMaterials); });
}

void h2d_randoms(vec2i occupancyParameters, float *randoms)
{
    if (!sameOccupancy(occupancyParameters, "h2d_randoms"))
        return;
    onEveryDevice([&](int) { h2dRandomsOne(randoms); });
}

void solr_hip_h2d_randoms_sized(const float *randoms, long count)
{
    onEveryDevice([&](int) { h2dRandomsSizedOne(randoms, count); });
}

void h2d_textures(vec2i occupancyParameters, int activeTextures, TextureInfo *textureInfos)
{
    if (!sameOccupancy(occupancyParameters, "h2d_textures"))
        return;
    onEveryDevice([&](int) { h2dTexturesOne(activeTextures, textureInfos); });
}

void h2d_lightInformation(vec2i occupancyParameters, LightInformation *lightInformation, int lightInformationSize)
{
    if (!sameOccupancy(occupancyParameters, "h2d_lightInformation"))
        return;
    onEveryDevice([&](int) { h2dLightInformationOne(lightInformation, lightInformationSize); });
}

void cudaRender(vec2i occupancyParameters, vec4i, SceneInfo sceneInfo, vec4i objects, PostProcessingInfo postProcessingInfo,
                vec3f origin, vec3f direction, vec4f angles)
{
    if (!sameOccupancy(occupancyParameters, "cudaRender"))
        return;
    const float o[3] = {origin.x, origin.y, origin.z};
    const float d[3] = {direction.x, direction.y, direction.z};
    const float a[4] = {angles.x, angles.y, angles.z, angles.w};
    splitRows(sceneInfo.size.y);
    /* (asynchronous: the devices render their strips side by side) */
    onEveryDevice([&](int) { renderImpl(sceneInfo, objects, postProcessingInfo, o, d, a, false, nullptr); });
}

void solr_hip_render(const SceneInfo *sceneInfo, const vec4i *objects, const PostProcessingInfo *postProcessingInfo,
                     const float origin[3], const float direction[3], const float angles[4])
{
    splitRows(sceneInfo->size.y);
    onEveryDevice([&](int) { renderImpl(*sceneInfo, *objects, *postProcessingInfo, origin, direction, angles, false, nullptr); });
}

void d2h_bitmap(vec2i occupancyParameters, SceneInfo sceneInfo, BitmapBuffer *bitmap, PrimitiveXYIdBuffer *primitivesXYIds)
{
    if (!sameOccupancy(occupancyParameters, "d2h_bitmap"))
        return;
    if (gDevices < 2)
    {
        d2hBitmapOne(sceneInfo, bitmap, primitivesXYIds, true);
        return;
    }
    /* every device's strip to its place in the host arrays (CudaRayTracer.cu:1647-1672): enqueued on all of them,
     * then waited for */
    onEveryDevice([&](int) { d2hBitmapOne(sceneInfo, bitmap, primitivesXYIds, false); });
    onEveryDevice([&](int) { d2hBitmapWait(); });
}

void solr_hip_d2h(const SceneInfo *sceneInfo, BitmapBuffer *bitmap, PrimitiveXYIdBuffer *primitivesXYIds)
{
    vec2i occ;
    occ.x = 0;
    occ.y = 1;
    d2h_bitmap(occ, *sceneInfo, bitmap, primitivesXYIds);
}

void solr_hip_synchronize(void)
{
    onEveryDevice([&](int) { synchronizeOne(); });
}

void solr_hip_set_movable(const unsigned char *flags, int nbPrimitives)
{
    onEveryDevice([&](int) { setMovableOne(flags, nbPrimitives); });
}

/* (every in-process device holds the scene and rotates its own copy; 1 only when all of them did) */
int solr_hip_rotate_primitives(const float center[3], const float cosAngles[3], const float sinAngles[3], float viewDistance)
{
    /* every resident copy must be able to follow before any of them moves: "0: nothing was changed" then holds on
     * several devices as well (one engine declining used to leave the others rotated) */
    bool all = true;
    onEveryDevice([&](int) { all = canRotateOne(center, cosAngles, sinAngles, viewDistance) && all; });
    if (!all)
        return 0;
    int status = 1; /* 1: rotated on the device */
    onEveryDevice([&](int) {
        const int mine = rotatePrimitivesOne(center, cosAngles, sinAngles, viewDistance);
        if (mine != 1 && status == 1)
            status = mine; /* (a device error past the check: the engine's error state says which) */
    });
    return status;
}
}
