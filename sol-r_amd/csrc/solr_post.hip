/*
 * solr_post.hip - the kernels that run behind the renderer: the stand-alone float -> RGB8 conversion (k_default), the five
 * post-processing effects of cudaRender's switch (CudaRayTracer.cu:1057-1358), the sort of the tiles by cost for the
 * next frames' launch order, and the packing of a strip's boundary depths for the neighbouring ranks.  The host side
 * (solr_launch.hip, solr_rccl.hip) sees plain launchers (engine.h, namespace solrpost).  gfx950 only.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "engine.h"

using namespace solrdev;

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
 * maximum and sum for the host (every 64th frame: engine.h sortPeriod) and, when the
 * host has seen a heavy tail (max > 2 x mean), sorts the tiles by cost with a counting sort in LDS
 * (64 cost classes) so that the following frames are launched most-expensive-first; the order is
 * refreshed every 64th frame.  Only the order of work changes,
 * never a result.  (Per-wave atomics for max / sum were tried first: 32 400 same-address device-scope
 * atomics per frame serialise at the memory side and tripled the frame time.) */
__device__ unsigned orderSerial = 0u;

__device__ __forceinline__ unsigned bandOfTile(const BandCuts &cuts, int tile)
{
    unsigned band = 0u;
#pragma unroll
    for (int b = 1; b < SOLR_STREAM_BANDS_MAX; ++b)
        band += (b < cuts.bands && tile >= cuts.firstTile[b]) ? 1u : 0u;
    return band;
}

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
                                                      volatile unsigned *hostStats, int sort, const BandCuts cuts)
{
    /* cuts.bands > 0 (ImageStreaming, renderer.h: the image leaves in bands of tile rows while the kernel renders): the
     * heaviest eighth of the tiles first, by cost - they are the frame's critical path wherever they lie, and a band
     * whose heavy tiles started first completes when its light ones have; then the others band after band - at most
     * eight - and by cost inside a band, so that the bands complete one after the other.  No tile is split (the host asks
     * for this order only where none would be).  Bins: band (3 bits, the first band in the highest bins), class (6), and
     * one bit of the tile index instead of four. */
    __shared__ unsigned classCount[64];
    __shared__ unsigned classFirst[64];
    __shared__ unsigned firstHeavy, nbHeavy;
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
            {
                const unsigned cls = min(63u, (unsigned)((float)c[k] * toClass));
                atomicAdd(&bins[cuts.bands > 0 ? ((7u - bandOfTile(cuts, i)) << 7) | (cls << 1) | ((unsigned)i & 1u)
                                               : (cls << 4) | ((unsigned)i & 15u)], 1u);
            }
        }
    }
    __syncthreads();
    if (cuts.bands > 0)
    {
        if (t < 64)
        {
            unsigned total = 0u;
            for (unsigned band = 0; band < 8u; ++band)
                total += bins[(band << 7) | ((unsigned)t << 1)] + bins[(band << 7) | ((unsigned)t << 1) | 1u];
            classCount[t] = total;
        }
        __syncthreads();
        if (t == 0)
        {
            /* whole classes from the top down, as many as hold an eighth of the tiles at most; their places in descending
             * order of class */
            unsigned upTo = 0u, first = 64u;
            for (int c = 63; c >= 1; --c)
            {
                if (upTo + classCount[c] > (unsigned)n / (unsigned)max(cuts.heavyShare, 1))
                    break;
                classFirst[c] = upTo;
                upTo += classCount[c];
                first = (unsigned)c;
            }
            firstHeavy = first;
            nbHeavy = upTo;
        }
        __syncthreads();
        const unsigned heavyFrom = firstHeavy, heavy = nbHeavy;
        if (t == 0)
            hostStats[5] = 0u;
        if (((unsigned)t >> 1 & 63u) >= heavyFrom) /* those tiles take their places from classFirst, not from a bin */
            bins[t] = 0u;
        __syncthreads();
        const unsigned own = bins[1023 - t];
        scan[t] = own;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1)
        {
            const unsigned add = (t >= off) ? scan[t - off] : 0u;
            __syncthreads();
            scan[t] += add;
            __syncthreads();
        }
        bins[1023 - t] = scan[t] - own;
        __syncthreads();
        for (int i = n + t; i < n + (SPLIT_PARTS - 1) * SPLIT_TILES_MAX; i += 1024)
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
                    const unsigned cls = min(63u, (unsigned)((float)c[k] * toClass));
                    if (cls >= heavyFrom)
                        order[atomicAdd(&classFirst[cls], 1u)] = (unsigned)i;
                    else
                        order[heavy + atomicAdd(&bins[((7u - bandOfTile(cuts, i)) << 7) | (cls << 1) | ((unsigned)i & 1u)], 1u)] = (unsigned)i;
                }
            }
        }
        return;
    }
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
    if (t == 0)
        hostStats[5] = split; /* (diagnostics: tiles this order renders as quadrant waves) */
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
#ifndef AO_TILES_PER_GROUP
#define AO_TILES_PER_GROUP 8
#endif
#ifndef AO_IRREGULAR_TOGETHER
#define AO_IRREGULAR_TOGETHER 40 /* irregular pixels of a tile in two binades that the workgroup takes together (256 threads a pixel's 256 taps) */
#endif
#define AO_AHEAD 4 /* window depths a thread holds for the next tile: windows of up to 256 x AO_AHEAD floats are asked for a tile ahead */
#ifndef AO_WAVES_PER_SIMD
#define AO_WAVES_PER_SIMD 8 /* 64 registers, 14 of them spilled: a tile is a chain of waits, and eight workgroups a CU hide more of
                             * them than the spills cost (6: 80 registers, none spilled, 5 % slower; the compiler's own choice, 108
                             * registers and four workgroups a CU: 25 % slower) */
#endif
__global__ __launch_bounds__(256, AO_WAVES_PER_SIMD) void k_ambientOcclusion(const SceneInfo si, const PostProcessingInfo ppi, int nbRows,
                                                          const PixelRecord *__restrict__ pp,
                                                          const float *__restrict__ randoms, long nbRandoms,
                                                          unsigned char *__restrict__ bitmap, const DepthHalo halo,
                                                          int firstRow, int windowFloats, int heavyFirst)
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
    /* the irregular pixels of a tile in two binades (below), taken together: which threads, their depths, their counts */
    __shared__ unsigned short irregularPixel[256];
    __shared__ float irregularDepth[256];
    __shared__ int irregularCount[256];
    __shared__ int nbIrregular;
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
    /* cell i < 1 024 of the window -> (i / ww, i % ww) by a multiplication: floor(i m / 2^26) with m = floor(2^26 / ww) + 1
     * is i / ww exactly while i < 2^10 and ww <= 2^13 (the excess i / 2^26 is below 1 / ww) - a division by a number only
     * known at run time is thirty instructions, four of them per thread and tile */
    const unsigned wwMagic = (1u << 26) / (unsigned)max(ww, 1) + 1u;
    float aheadDepth[AO_AHEAD];
    float4 aheadLocal = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ahead = [&](int2 t) { /* this thread's share of the window of tile (column, row), and its own pixel's record */
        const int tx0 = t.x * AO_TILE_W, ty0 = t.y * AO_TILE_H;
#pragma unroll
        for (int k = 0; k < AO_AHEAD; ++k)
        {
            const int i = (int)threadIdx.x + 256 * k;
            float d = 0.f;
            const int row = (int)(((unsigned)i * wwMagic) >> 26);
            if (i < ww * wrows)
                aoDepthAt(pp, halo, W, nbRows, tx0 - rx + (i - row * ww), ty0 - ry + row, d);
            aheadDepth[k] = d;
        }
        const int px = tx0 + (int)(threadIdx.x % AO_TILE_W), py = ty0 + (int)(threadIdx.x / AO_TILE_W);
        aheadLocal = pp[(px < W && py < nbRows) ? py * W + px : 0].colorInfo;
    };
#ifdef SOLR_AO_DEBUG
    const unsigned long long groupClock0 = __builtin_amdgcn_s_memrealtime();
#endif
    /* Which tiles, in which order?  A tile that straddles a binade costs four to twelve times a steady one
     * (tools/ao_paths.py: 12 - 40 us against 3.2) and whole tile rows and columns of a frame are such: with any fixed
     * share of the frame per workgroup the kernel was as long as its unluckiest workgroup - 0.3 ms of which the chip
     * stood half empty for 0.13.  "Steady" factorises - a tile is steady iff its tile ROW is light (window inside the
     * strip and its halo, the rows and their taps' sums in one binade) and its tile COLUMN is light - so the workgroup
     * lists the heavy and the light rows and columns once (LDS, 256 threads: a row or column each), and the frame's
     * tiles are taken in the order [heavy rows x all columns, light rows x heavy columns, light rows x light columns],
     * item blockIdx.x + k gridDim.x for k = 0, 1, ...: every workgroup gets its share of the heavy tiles, to within
     * one, and gets them FIRST - the tail of the kernel is made of steady tiles.  The grid is the workgroups the chip
     * holds at once.  (heavyFirst == 0: tile `run` of this workgroup is a stride of the grid further, AO_TILES_PER_GROUP
     * of them - solr_hip_set_variant(9), the probes of the paths.) */
    __shared__ unsigned short rowList[1024], colList[512]; /* heavy ones first */
    __shared__ int listCount[4];                           /* [0] heavy rows, [2] heavy columns */
    const int tilesY = (nbRows + AO_TILE_H - 1) / AO_TILE_H;
    const bool ordered = heavyFirst != 0 && tiled && tilesY <= 1024 && tilesX <= 512;
    if (ordered)
    {
        auto rowLight = [&](int ty) {
            const int y0_ = ty * AO_TILE_H, wy = y0_ - ry;
            const int ylo = y0_ + firstRow - ry, yhi = y0_ + firstRow + AO_TILE_H - 1 + ry;
            return wy >= -halo.nbAbove && wy + wrows <= nbRows + halo.nbBelow && ylo >= 1 && __clz(ylo) == __clz(yhi);
        };
        auto colLight = [&](int tx) {
            const int x0_ = tx * AO_TILE_W, wx = x0_ - rx;
            const int xlo = x0_ - rx, xhi = x0_ + AO_TILE_W - 1 + rx;
            return wx >= 0 && wx + ww <= W && xlo >= 1 && __clz(xlo) == __clz(xhi);
        };
        /* the heavy ones take the front of each list, the light ones the rest, both in ascending order - EVERY workgroup
         * must make the same lists (they share out one order of the frame's tiles): one wave compacts them with
         * ballots, 64 rows or columns a step */
        if (threadIdx.x < 64)
        {
            const int lane = (int)threadIdx.x;
            const unsigned long long below = (1ull << lane) - 1ull;
            int count = 0;
            for (int sweep = 0; sweep < 2; ++sweep)
            {
                for (int base = 0; base < tilesY; base += 64)
                {
                    const int ty = base + lane;
                    const bool take = ty < tilesY && rowLight(ty) == (sweep == 1);
                    const unsigned long long taken = __ballot(take);
                    if (take)
                        rowList[count + __popcll(taken & below)] = (unsigned short)ty;
                    count += __popcll(taken);
                }
                if (sweep == 0 && lane == 0)
                    listCount[0] = count;
            }
            count = 0;
            for (int sweep = 0; sweep < 2; ++sweep)
            {
                for (int base = 0; base < tilesX; base += 64)
                {
                    const int tx = base + lane;
                    const bool take = tx < tilesX && colLight(tx) == (sweep == 1);
                    const unsigned long long taken = __ballot(take);
                    if (take)
                        colList[count + __popcll(taken & below)] = (unsigned short)tx;
                    count += __popcll(taken);
                }
                if (sweep == 0 && lane == 0)
                    listCount[2] = count;
            }
        }
        __syncthreads();
    }
    const int heavyRows = ordered ? listCount[0] : 0, heavyCols = ordered ? listCount[2] : 0;
    const int lightCols = tilesX - heavyCols;
    const int firstBlock = heavyRows * tilesX, secondBlock = (tilesY - heavyRows) * heavyCols;
    /* item of that order (or, unordered, the tile's row-major index) -> the tile's (column, row).  The scalar unit is what
     * this kernel runs at (profiles/r5/ao_heavy_tiles_first.txt: 485 scalar instructions per wave and tile against 426
     * vector ones, one scalar unit a CU), and a division by a number only known at run time is some thirty-five of
     * them - four per tile as this was first written.  k / d for k < 2^20 and d <= 2^10 is floor(k m / 2^40) with
     * m = floor(2^40 / d) + 1 exactly (the excess k / 2^40 is below 1 / d): one division per divisor and kernel, a 64-bit
     * multiplication per tile, and the (column, row) pair is kept instead of being divided out of the index again. */
    auto magicOf = [](int d) { return (1ull << 40) / (unsigned long long)max(d, 1) + 1ull; };
    const unsigned long long magicTilesX = magicOf(tilesX), magicHeavyCols = magicOf(heavyCols), magicLightCols = magicOf(lightCols);
    auto over = [](int k, unsigned long long magic) { return (int)(((unsigned long long)(unsigned)k * magic) >> 40); };
    const bool smallFrame = nbTiles < (1 << 20) && tilesX <= 1024;
    auto tileOf = [&](int item) {
        if (!ordered)
        {
            const int row = smallFrame ? over(item, magicTilesX) : item / tilesX;
            return make_int2(item - row * tilesX, row);
        }
        if (item < firstBlock)
        {
            const int q = over(item, magicTilesX);
            return make_int2(item - q * tilesX, (int)rowList[q]);
        }
        if (item < firstBlock + secondBlock)
        {
            const int k = item - firstBlock, q = over(k, magicHeavyCols);
            return make_int2((int)colList[k - q * heavyCols], (int)rowList[heavyRows + q]);
        }
        const int k = item - firstBlock - secondBlock, q = over(k, magicLightCols);
        return make_int2((int)colList[heavyCols + k - q * lightCols], (int)rowList[heavyRows + q]);
    };
    const bool persistent = heavyFirst != 0; /* the launch's grid: the workgroups the chip holds, each until the frame is done */
    int2 tileNext = make_int2(0, 0);
    for (int run = 0; persistent || run < AO_TILES_PER_GROUP; ++run)
    {
        const int item = (int)blockIdx.x + run * (int)gridDim.x;
        if (item >= nbTiles)
            break;
        const int2 tileAt = run == 0 ? tileOf(item) : tileNext; /* (column, row) */
        /* the tile after this one (the window of its depths is asked for while this one is compared) */
        const int itemAfter = (persistent || run + 1 < AO_TILES_PER_GROUP) ? item + (int)gridDim.x : nbTiles;
        const bool another = itemAfter < nbTiles;
        if (another)
            tileNext = tileOf(itemAfter);
#ifdef SOLR_AO_DEBUG
        const unsigned long long tileClock0 = __builtin_amdgcn_s_memrealtime(); /* (tools/ao_paths.py) */
#endif
        const int x0 = tileAt.x * AO_TILE_W;
        const int y0 = tileAt.y * AO_TILE_H;
        const int wx0 = x0 - rx, wy0 = y0 - ry;
        const int x = x0 + (int)(threadIdx.x % AO_TILE_W);
        const int y = y0 + (int)(threadIdx.x / AO_TILE_W);
        const bool mine = x < W && y < nbRows;
        const int index = mine ? y * W + x : 0;
        /* A window of up to 1 024 depths (taps that reach 12 pixels) is asked for ONE TILE AHEAD, into registers, behind
         * the barrier below: the loads of tile n + 1 are in flight while tile n is compared and stored, and a tile is
         * no longer two memory latencies long. */
        if (pipelined && run == 0)
            ahead(tileAt);
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
        bool regularX = true, regularY = true; /* (every column and row of a steady tile is) */
        if (!steady)
        {
            regularX = x >= 1 && (int)(__float_as_uint((float)x + tapLowX) >> 23) == e0x &&
                       (int)(__float_as_uint((float)x + tapHighX) >> 23) == e0x;
            regularY = y + firstRow >= 1 && (int)(__float_as_uint((float)(y + firstRow) + tapLowY) >> 23) == e0y &&
                       (int)(__float_as_uint((float)(y + firstRow) + tapHighY) >> 23) == e0y;
        }
        const bool windowInside = tiled && wx0 >= 0 && wy0 >= -halo.nbAbove && wx0 + ww <= W && wy0 + wrows <= nbRows + halo.nbBelow;
        bool classed = tiled && !steady && binsX * binsY <= 256;
        if (classed)
        {
            const int i = threadIdx.x;
            tableKey = 0; /* (the histograms take the place of the steady tiles' table) */
            if (i < 8)
                cls[i] = (i == 0 || i == 2 || i == 4 || i == 6) ? 0x7fffffff : -1;
            if (i == 0)
                nbIrregular = 0;
            irregularCount[i] = 0;
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
        /* The pixels of the irregular columns and rows of such a tile - cfg4: the one column AT the power of two, eight
         * pixels - used to take the 256-tap loop inside their waves, and a wave is as long as its slowest lane: every
         * wave of every tile that straddles a power of two in x ran the whole loop for one lane in thirty-two (17 000
         * waves of a 4K frame: 0.11 ms of the kernel's 0.22).  They are taken TOGETHER instead: each registers (thread,
         * depth), and for one such pixel after the other the workgroup's 256 threads evaluate one tap each - the
         * reference's expression for that pixel and that tap - and count by ballot: eight trips of a dozen instructions
         * for the column, not 256 trips in every wave.  (Up to AO_IRREGULAR_TOGETHER pixels; a tile with more - whole
         * rows of them - keeps the loop in the waves that hold them.) */
        int irregularSlot = -1;
        if (classed && mine && !(regularX && regularY))
        {
            irregularSlot = atomicAdd(&nbIrregular, 1);
            irregularPixel[irregularSlot] = (unsigned short)threadIdx.x;
            irregularDepth[irregularSlot] = local.w;
        }
        __syncthreads(); /* the window is in LDS, and so are the offsets */
        if (pipelined && another)
            ahead(tileNext);
        const int together = classed ? nbIrregular : 0;
        const bool takenTogether = together > 0 && together <= AO_IRREGULAR_TOGETHER;
        if (takenTogether)
        {
            const int origin = -((wy0 + firstRow) * ww + wx0);
            const float tx = tapX[threadIdx.x], ty = tapY[threadIdx.x];
            for (int j = 0; j < together; ++j)
            {
                const int p = irregularPixel[j];
                const float fx = (float)(x0 + p % AO_TILE_W), fy = (float)(y0 + p / AO_TILE_W + firstRow);
                const int xx = (int)(fx + tx);
                const int yy = (int)(fy + ty);
                const bool in = xx >= 0 && xx < W && yy - firstRow >= -halo.nbAbove && yy - firstRow < nbRows + halo.nbBelow;
                const float tap = window[in ? __mul24(yy, ww) + xx + origin : 0];
                const unsigned long long hits = __ballot(!in || tap >= irregularDepth[j]);
                if ((threadIdx.x & 63) == 0)
                    atomicAdd(&irregularCount[j], (int)__popcll(hits));
            }
            __syncthreads();
        }
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
                else if (takenTogether && irregularSlot >= 0)
                    count = irregularCount[irregularSlot];
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
            occ *= 0.00390625f; /* occ / c (CRT:1166) with c == 256 on either path: a division by a power of two is this multiplication, bit for bit */
            (void)c;
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
#ifdef SOLR_AO_DEBUG
            /* development build (tools/ao_paths.py): the image holds, per pixel, the path its tile took, the tile's time
             * in units of 0.64 us (100 MHz ticks / 64) and the tile's place in its workgroup's run */
            const unsigned long long ticks = __builtin_amdgcn_s_memrealtime() - tileClock0;
            bitmap[3 * index + 0] = (unsigned char)(!tiled ? 5 : steady ? 1 : (classed && regularX && regularY) ? 2 : classed ? 3 : windowInside ? 4 : 6);
            bitmap[3 * index + 1] = (unsigned char)min(255ull, ticks / 64ull);
            bitmap[3 * index + 2] = (unsigned char)run;
#endif
        }
        __syncthreads(); /* the next tile's window goes where this one's is still being read */
    }
#ifdef SOLR_AO_DEBUG
    /* ... and the first four pixels of the workgroup's first tile (the fourth: a mark the tool finds them by): when the workgroup began its tiles and when it ended
     * them (24 bits of the 100 MHz clock each) and where it ran (HW_ID: wave, SIMD, CU, SH, SE; XCC_ID) */
    if (threadIdx.x == 0 && (int)blockIdx.x < nbTiles)
    {
        const int2 mark = tileOf((int)blockIdx.x); /* this workgroup's first tile */
        const int first = mark.y * AO_TILE_H * W + mark.x * AO_TILE_W;
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        unsigned hw = 0u, xcc = 0u;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned where = ((xcc & 15u) << 16) | (((hw >> 13) & 7u) << 8) | ((hw >> 8) & 15u); /* XCC, SE, CU */
        const unsigned words[4] = {(unsigned)groupClock0 & 0xffffffu, (unsigned)t1 & 0xffffffu, where, 0xefcdabu};
        for (int k = 0; k < 4; ++k)
        {
            bitmap[3 * (first + k) + 0] = (unsigned char)(words[k] & 255u);
            bitmap[3 * (first + k) + 1] = (unsigned char)((words[k] >> 8) & 255u);
            bitmap[3 * (first + k) + 2] = (unsigned char)((words[k] >> 16) & 255u);
        }
    }
#endif
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


/* ---- plain launchers (engine.h): what the host side of the engine sees of this file ---------------------------------- */
namespace solrpost
{
void defaultConversion(hipStream_t stream, const SceneInfo &si, int nbPixels, const PixelRecord *pp, unsigned char *bitmap)
{
    hipLaunchKernelGGL(k_default, dim3((unsigned)((nbPixels + 255) / 256)), dim3(256), 0, stream, si, nbPixels, pp, bitmap);
}

void ambientOcclusion(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
                      const float *randoms, long nbRandoms, unsigned char *bitmap, const DepthHalo &halo, int firstRow,
                      float randomsReach, bool heavyFirst)
{
    /* the window the taps of this random buffer and this param2 can need (the kernel takes its own, exact reach and
     * gathers from memory if this should ever be too small): |tap| <= 16 |param2| max|random| / 10 */
    const float aoReach = 16.f * fabsf(ppi.param2) * randomsReach / 10.f;
    const int aoR = aoReach < 4096.f ? (int)aoReach + 3 : 4096;
    const long aoCells = (long)(AO_TILE_W + 2 * aoR) * (AO_TILE_H + 2 * aoR);
    const int aoWindow = (int)std::min<long>(std::max<long>(aoCells, 64), AO_WINDOW_FLOATS);
    const int tiles = ((si.size.x + AO_TILE_W - 1) / AO_TILE_W) * ((nbRows + AO_TILE_H - 1) / AO_TILE_H);
    /* heavy tiles first: the workgroups the chip holds at once (AO_WAVES_PER_SIMD of them per CU, a wave on each SIMD),
     * or fewer when the frame has fewer tiles */
    unsigned groups = (unsigned)((tiles + AO_TILES_PER_GROUP - 1) / AO_TILES_PER_GROUP);
    if (heavyFirst)
    {
        static int cusOf[SOLR_MAX_GPU_COUNT + 1] = {}; /* (per device of this process: asked once) */
        int device = 0;
        (void)hipGetDevice(&device);
        int &cus = cusOf[(device >= 0 && device < SOLR_MAX_GPU_COUNT) ? device : SOLR_MAX_GPU_COUNT];
        if (cus == 0)
        {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0)
                n = 256;
            cus = n;
        }
        groups = std::max(1u, std::min((unsigned)tiles, (unsigned)(cus * AO_WAVES_PER_SIMD)));
    }
    hipLaunchKernelGGL(k_ambientOcclusion, dim3(groups), dim3(256), (size_t)aoWindow * sizeof(float), stream, si, ppi, nbRows, pp,
                       randoms, nbRandoms, bitmap, halo, firstRow, aoWindow, heavyFirst ? 1 : 0);
}

static dim3 pixelsGrid(const SceneInfo &si, int nbRows) { return dim3((unsigned)((si.size.x * nbRows + 255) / 256)); }

void depthOfField(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
                  const float *randoms, long nbRandoms, unsigned char *bitmap)
{
    hipLaunchKernelGGL(k_depthOfField, pixelsGrid(si, nbRows), dim3(256), 0, stream, si, ppi, nbRows, pp, randoms, nbRandoms, bitmap);
}

void radiosity(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
               const int4 *ids, const float *randoms, long nbRandoms, unsigned char *bitmap)
{
    hipLaunchKernelGGL(k_radiosity, pixelsGrid(si, nbRows), dim3(256), 0, stream, si, ppi, nbRows, pp, ids, randoms, nbRandoms, bitmap);
}

void filter(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
            unsigned char *bitmap)
{
    hipLaunchKernelGGL(k_filter, pixelsGrid(si, nbRows), dim3(256), 0, stream, si, ppi, nbRows, pp, bitmap);
}

void cartoon(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
             unsigned char *bitmap)
{
    hipLaunchKernelGGL(k_cartoon, pixelsGrid(si, nbRows), dim3(256), 0, stream, si, ppi, nbRows, pp, bitmap);
}

void orderTiles(hipStream_t stream, const unsigned *cost, unsigned *snapshot, unsigned *order, int nbTiles,
                volatile unsigned *hostStats, int flights, const BandCuts &cuts)
{
    hipLaunchKernelGGL(k_orderTiles, dim3(1), dim3(1024), 0, stream, cost, snapshot, order, nbTiles, hostStats, flights, cuts);
}

void packDepthRows(hipStream_t stream, const PixelRecord *pp, int W, int row0, int n, float *out)
{
    hipLaunchKernelGGL(k_packDepthRows, dim3((unsigned)((W * n + 255) / 256)), dim3(256), 0, stream, pp, W, row0, n, out);
}
} // namespace solrpost
