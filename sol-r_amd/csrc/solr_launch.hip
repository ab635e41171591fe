/*
 * solr_launch.hip - a frame of the MI355X rendering engine: the per-pixel buffers, the launch table (which instantiation of
 * k_standardRenderer a scene and a frame get: csrc/rows), the cost-ordered launch, cudaRender's post-processing switch
 * (CudaRayTracer.cu:1694-1890) and the read-back of d2h_bitmap (:1647-1672).
 * Part of the engine's host side (engine.h); the boundary that calls into it is solr_hip.hip.  gfx950 only.
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

#include "renderer.h"
#include "engine.h"

using namespace solreng;

/* the renderer's instantiations live in the files under csrc/rows (one object per row of renderImpl's table) */
namespace solrrows
{
RendererFn renderer(int count, int features, bool volume)
{
    if (volume || (features & ~(F_DEEP | F_STACK | F_STREAM)) == F_ALL)
        return (features & (F_STACK | F_STREAM)) ? nullptr : everything(count, features, volume);
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

namespace solreng
{
void allocateFrame()
{
    const int rows = stripRows();
    const size_t pixels = (size_t)std::max(g.width, 1) * (size_t)std::max(rows, 1);
    const bool grow = pixels * sizeof(PostProcessingBuffer) > g.pp.bytes;
    reserve(g.pp, pixels * sizeof(PostProcessingBuffer));
    reserve(g.ids, pixels * sizeof(PrimitiveXYIdBuffer));
    reserve(g.bitmap, pixels * SOLR_COLOR_DEPTH);
    /* (a buffer set's second RGB image - renderImpl makes it when a read-back still holds the first - grows with the
     * frame too: the set may be on that side when the frame is re-shaped) */
    for (int f = 0; f < MAX_FLIGHTS; ++f)
        if (g.bitmapAlt[f].ptr)
            reserve(g.bitmapAlt[f], pixels * SOLR_COLOR_DEPTH);
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
        if (ok())
            solrpost::ambientOcclusion(stream, sceneInfo, ppInfo, nbRows, (const PixelRecord *)flightPp(flight).ptr,
                                       (const float *)g.randoms.ptr, g.randoms.ptr ? g.nbRandoms : 0L, bitmap, halo, firstRow,
                                       g.randomsReach, g.variant != 9);
    }
    else if (ppInfo.type == ppe_depthOfField)
        solrpost::depthOfField(stream, sceneInfo, ppInfo, nbRows, (const PixelRecord *)flightPp(flight).ptr,
                               (const float *)g.randoms.ptr, g.randoms.ptr ? g.nbRandoms : 0L, bitmap);
    else if (ppInfo.type == ppe_radiosity)
        solrpost::radiosity(stream, sceneInfo, ppInfo, nbRows, (const PixelRecord *)flightPp(flight).ptr,
                            (const int4 *)flightIds(flight).ptr, (const float *)g.randoms.ptr, g.randoms.ptr ? g.nbRandoms : 0L,
                            bitmap);
    else if (ppInfo.type == ppe_filter)
        solrpost::filter(stream, sceneInfo, ppInfo, nbRows, (const PixelRecord *)flightPp(flight).ptr, bitmap);
    else
        solrpost::cartoon(stream, sceneInfo, ppInfo, nbRows, (const PixelRecord *)flightPp(flight).ptr, bitmap);
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
    F.tilesX = (sceneInfo.size.x + TILE_W - 1) / TILE_W;
    const int tilesY = (F.nbRows + TILE_H - 1) / TILE_H;
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
    /* ImageStreaming (renderer.h): asked for (solr_hip_stream_next_image), and this is a frame whose image the kernel
     * itself writes, whole, on one device, one frame at a time */
    g.streamedValid = false;
    BandCuts streamCuts = {};
    {
        int rows[SOLR_STREAM_BANDS_MAX + 1];
        /* (is there an instantiation that counts tiles for the kernel this scene takes?  Asked of the frame before: the
         * launch order, made further up than the choice of the kernel, has to know) */
        if (g.streamNext && !counting && !g.recordNext && F.fuseDefault && sceneInfo.frameBufferType != ftBGR && !twoFlights() &&
            g.nbRows < 0 && gDevices == 1 && !g.boundBitmap && !g.sharedRing && g.lastMask >= 0 &&
            solrrows::renderer(0, g.lastMask | F_STREAM, false) != nullptr && imageStreamingCuts(tilesY, rows, &streamCuts.bands, g.streamNext == 2))
        {
            for (int b = 0; b <= streamCuts.bands; ++b)
                streamCuts.firstTile[b] = rows[b] * F.tilesX;
            static const int share = getenv("SOLR_HIP_STREAM_HEAVY") ? std::max(1, atoi(getenv("SOLR_HIP_STREAM_HEAVY"))) : 8;
            streamCuts.heavyShare = share;
        }
    }
    const bool streamIds = g.streamNext == 2;
    g.streamNext = 0;
    bool streamCandidate = streamCuts.bands > 0;
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
            /* a frame whose longest tile would be rendered by four quadrant waves (k_orderTiles' criterion, for one frame
             * in flight) keeps the order that puts those first: it is not streamed */
            const float mean = (float)sum / (float)grid.x;
            const float critical = fmaxf(2.f * mean, (float)sum / 5120.f);
            if (g.reorder && (unsigned)(critical * (64.f / ((float)mx + 1.f))) < 63u)
                streamCandidate = false;
        }
        F.tileCost = (unsigned *)g.tileCost.ptr;
        /* statistics (and, in cost order, a fresh order) every sortPeriod()-th frame, and at once when the
         * decision has just changed; in between the last order is reused */
        const bool ordered = g.costFrames > 0 && (g.tileScheduling == 2 || g.reorder);
        /* a streamed frame takes its tiles band after band (k_orderTiles), any other by cost alone: the order is re-made
         * at once when the frame at hand is of the other kind */
        const BandCuts cuts = streamCandidate ? streamCuts : BandCuts();
        if (ordered && g.orderValid && memcmp(&g.orderCuts, &cuts, sizeof(cuts)) != 0)
            g.orderValid = false;
        const bool refresh = g.costFrames > 0 && (g.costFrames % sortPeriod() == 1 || (ordered && !g.orderValid));
        const bool sort = ordered && refresh;
        if (!ordered)
            g.orderValid = false;
        if (refresh)
        {
            /* a new order goes to the buffer no frame in flight is reading; the other stream waits for
             * the sort before its next frame picks that buffer up */
            const int target = sort ? (g.orderBuffer ^ 1) : g.orderBuffer;
            DeviceBuffer &orderOut = target ? g.tileOrder2 : g.tileOrder;
            solrpost::orderTiles(stream, (const unsigned *)g.tileCost.ptr, (unsigned *)g.tileCostSnapshot.ptr,
                                 (unsigned *)orderOut.ptr, (int)grid.x, (volatile unsigned *)g.hostStatsDev, sort ? activeFlights() : 0, cuts);
            HIPCHECK(hipGetLastError());
            if (sort)
            {
                g.orderValid = true;
                g.orderCuts = cuts;
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
    int chosenMask = -1; /* features of the row chosen: what an instantiation with another epilogue (F_STREAM) is asked for with */
    if (!counting)
    {
        fn = solrrows::renderer(0, F_ALL | F_DEEP, volumeCamera);
        int row = 0, chosen = -1;
        chosenMask = -1;
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
                chosenMask = mask;
                break;
            }
            ++row;
        }
        g.recordVariant = chosen;
    }
    ARGCHECK(fn != nullptr, "cudaRender: no instantiation of the renderer for this scene (csrc/rows)");
    if (!ok())
        return;
    /* (the census kernel adds into them; a frame's own kernel does not touch them: zeroing them on the stream of EVERY frame
     * was a fill kernel of 3.6 us - and a launch - in front of every renderer, 1.4 % of the GPU's time in the profile) */
    if (counting)
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
        /* the deep slots are never zeroed: every slot a lane reads was written by the trip that made it (rt_device.h
         * launchRayTracing).  Variant 10 proves it: NaNs in every slot before the launch, the same frame after */
        if (g.variant == 10)
            HIPCHECK(hipMemsetAsync(F.deepStack, 0xff, (size_t)deepSlots * (size_t)F.deepStride * sizeof(float4), stream));
    }
    /* (tiles in launch order, or band after band: an order by cost alone completes every band at the end; the epilogue that
     * counts tiles is in instantiations of its own - the lean rows have them, rt_device.h F_STREAM) */
    bool streamed = false;
    const int streamMask = chosenMask | (deepSlots > 0 ? F_STACK : 0) | F_STREAM;
    if (streamCandidate && cntPtr == (unsigned long long *)g.counters.ptr && chosenMask >= 0 &&
        (!F.tileOrder || memcmp(&g.orderCuts, &streamCuts, sizeof(streamCuts)) == 0))
        if (KernelFn streaming = solrrows::renderer(0, streamMask, false))
            if (armImageStreaming(F, tilesY, stream, streamIds))
            {
                fn = streaming;
                streamed = true;
            }
    g.lastMask = chosenMask;
    if (streamed)
        F.fuseDefault |= streamIds ? 6 : 2;
    g.streamedIds = streamed && streamIds;
    if (streamed && g.variant == 13) /* (tests: the waves write no band's word - the host goes by the end of the kernel) */
        F.streamSerial = 0x7fffff00u;
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
    if (streamed)
    {
        markStreamedFrame(stream);
        g.streamedValid = ok();
        g.streamedBitmap = bitmap;
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
/* wait == false: the copies are enqueued and d2hBitmapWait() is owed (several devices copy side by side) */
void d2hBitmapOne(const SceneInfo &sceneInfo, BitmapBuffer *bitmap, PrimitiveXYIdBuffer *primitivesXYIds, bool wait)
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
void d2hBitmapWait()
{
    if (g.initialized && ok())
        HIPCHECK(hipStreamSynchronize(flightStream(g.current)));
}

} // namespace solreng

extern "C" {
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

} // extern "C"
