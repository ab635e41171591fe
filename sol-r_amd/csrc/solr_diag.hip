/*
 * solr_diag.hip - knobs and diagnostics of the MI355X rendering engine (include/solr_hip.h, part 2): frames in flight,
 * tile scheduling, variants, kernel timing and tile clocks, the walk's own ceiling (a frame's walks recorded and replayed
 * with nothing but the node loop), memory usage - and what the test-only probes (solr_probes.hip) ask of the engine.
 * Nothing here is on a frame's path.  gfx950 only.
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
        solrpost::defaultConversion(stream, sceneInfo, (int)pixels, (const PixelRecord *)flightPp(flight).ptr, bitmap);
        HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipMemcpyAsync(bitmapOut, bitmap, pixels * SOLR_COLOR_DEPTH, hipMemcpyDeviceToHost, stream));
    HIPCHECK(hipStreamSynchronize(stream));
    return ok() ? 0 : -1;
}
} // namespace solrprobe

extern "C" {
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

/* a frame whose walks are recorded: the records stay in g.walkRecords (a gigabyte for a 1080p frame) */
static int recordFrame(const SceneInfo *sceneInfo, const vec4i *objects, const PostProcessingInfo *postProcessingInfo,
                       const float origin[3], const float direction[3], const float angles[4], const char *who)
{
    quiesce();
    HIPCHECK(hipSetDevice(g.device));
    g.recorded = false;
    g.recordNext = true;
    if (!ok())
        return -1;
    renderImpl(*sceneInfo, *objects, *postProcessingInfo, origin, direction, angles, false, nullptr);
    g.recordNext = false;
    HIPCHECK(hipStreamSynchronize(flightStream(g.current)));
    if (!ok() || !g.recorded)
    {
        if (ok())
            setError(-1, "solr_hip_walk_bound: the frame was not recorded", __FILE__, __LINE__);
        return -1;
    }
    (void)who;
    return 0;
}

/* the first `grid` workgroup slots of g.walkRecords replayed `repeats` times with `ldsBytes` of dynamic LDS a wave */
static int replayRecords(unsigned grid, size_t ldsBytes, int repeats, double ms[3], unsigned long long stats[4], bool lists)
{
    typedef WalkBoundFn BoundFn;
    static const int leanRows[4] = {F_SPHERE | F_PLANE, F_SPHERE | F_TRI, F_SPHERE | F_CYL, F_SPHERE | F_PLANE | F_TRI | F_CYL};
    ARGCHECK(g.recordVariant >= 0 && g.recordVariant < 4 && g.walkRecords.ptr && grid > 0 &&
                 (size_t)grid * SOLR_WALK_SLOT_BYTES <= g.walkRecords.bytes,
             "solr_hip_walk_replay: no recorded frame, or more workgroups than its buffer holds");
    if (!ok())
        return -1;
    const BoundFn fn = solrrows::walkBound(g.recordVariant, leanRows[g.recordVariant] | (g.recordDeep ? F_DEEP : 0));
    ARGCHECK(fn != nullptr, "solr_hip_walk_bound: no replay instantiation for this row");
    if (!ok())
        return -1;
    reserve(g.walkVisits, (size_t)grid * WAVE * sizeof(unsigned) + 64);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    if (!ok())
        return -1;
    const hipStream_t stream = flightStream(g.current);
    unsigned *visits = (unsigned *)g.walkVisits.ptr;
    unsigned *skipped = visits + (size_t)grid * WAVE;
    double sum = 0.0, best = 1.0e30;
    repeats = repeats < 1 ? 1 : repeats;
    for (int i = 0; i < repeats + 2 && ok(); ++i)
    {
        HIPCHECK(hipMemsetAsync(skipped, 0, sizeof(unsigned), stream));
        HIPCHECK(hipEventRecord(e0, stream));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(WAVE), ldsBytes, stream, g.recordScene, (const char *)g.walkRecords.ptr,
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
        std::vector<unsigned> v((size_t)grid * WAVE + 1);
        HIPCHECK(hipMemcpy(v.data(), visits, v.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::vector<int> heads((size_t)grid * 4);
        HIPCHECK(hipMemcpy2D(heads.data(), 16, g.walkRecords.ptr, SOLR_WALK_SLOT_BYTES, 16, grid, hipMemcpyDeviceToHost));
        if (const char *dump = getenv("SOLR_HIP_WALK_BOUND_DUMP"))
        {
            /* diagnostics (tools/longest_wave.py): leaf entries per lane and walks per workgroup of the replay */
            if (FILE *f = fopen(dump, "wb"))
            {
                const unsigned n = grid;
                fwrite(&n, sizeof(n), 1, f);
                fwrite(v.data(), sizeof(unsigned), (size_t)n * WAVE, f);
                fwrite(heads.data(), sizeof(int), (size_t)n * 4, f);
                fclose(f);
            }
        }
        unsigned long long walks = 0, entries = 0;
        for (unsigned b = 0; b < grid; ++b)
            walks += (unsigned long long)heads[4 * (size_t)b];
        if (lists)
        {
            /* which list each recorded walk took (solr_hip_walk_bound_lists) */
            std::vector<int> kinds((size_t)grid * 4 * (SOLR_WALK_SLOTS + 1));
            HIPCHECK(hipMemcpy2D(kinds.data(), 16 * (SOLR_WALK_SLOTS + 1), g.walkRecords.ptr, SOLR_WALK_SLOT_BYTES,
                                 16 * (SOLR_WALK_SLOTS + 1), grid, hipMemcpyDeviceToHost));
            for (int i = 0; i < 6; ++i)
                walkLists[i] = 0;
            for (unsigned b = 0; ok() && b < grid; ++b)
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
        stats[3] = grid;
    }
    if (ms)
    {
        ms[0] = 0.0;
        ms[1] = sum / repeats;
        ms[2] = best;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ok() ? 0 : -1;
}

static bool keepWalkRecords = false;

int solr_hip_walk_bound(const SceneInfo *sceneInfo, const vec4i *objects, const PostProcessingInfo *postProcessingInfo,
                        const float origin[3], const float direction[3], const float angles[4], int repeats, double ms[3],
                        unsigned long long stats[4])
{
    if (!ready("solr_hip_walk_bound"))
        return -1;
    ARGCHECK(gDevices == 1, "solr_hip_walk_bound: a diagnostic of one engine; this process renders on several devices");
    if (!ok())
        return -1;
    if (recordFrame(sceneInfo, objects, postProcessingInfo, origin, direction, angles, "solr_hip_walk_bound") != 0)
        return -1;
    const int rc = replayRecords(g.recordGrid, g.recordLds, repeats, ms, stats, true);
    if (!keepWalkRecords)
    {
        /* the buffers are a gigabyte for a 1080p frame: given back at once */
        release(g.walkRecords);
        release(g.walkVisits);
        g.recorded = false;
    }
    return rc;
}

/* The record behind solr_hip_walk_bound in the caller's hands (tools/ray_regroup.py: what would sorting a frame's rays
 * by where they go buy the node loop?).  keep(1): the next solr_hip_walk_bound leaves its records on the device.
 * info: {workgroups recorded, bytes per workgroup slot, dynamic LDS of the recorded launch, walk slots per workgroup}.
 * copy: the first `grid` slots to (toDevice == 0) or from the host.  replay: those slots with nothing but the node loop,
 * `ldsBytes` of dynamic LDS a wave (< 0: the recorded launch's; 0: as many waves as the replay kernel's 64 registers
 * allow).  release: the buffers given back. */
void solr_hip_walk_records_keep(int keep)
{
    keepWalkRecords = keep != 0;
}

int solr_hip_walk_records_info(unsigned long long info[4])
{
    if (!ready("solr_hip_walk_records_info") || !info)
        return -1;
    info[0] = g.recorded ? g.recordGrid : 0;
    info[1] = SOLR_WALK_SLOT_BYTES;
    info[2] = g.recordLds;
    info[3] = SOLR_WALK_SLOTS;
    return 0;
}

int solr_hip_walk_records_copy(void *host, unsigned grid, int toDevice)
{
    if (!ready("solr_hip_walk_records_copy"))
        return -1;
    ARGCHECK(host && grid > 0 && g.recorded, "solr_hip_walk_records_copy: no recorded frame");
    if (!ok())
        return -1;
    const size_t bytes = (size_t)grid * SOLR_WALK_SLOT_BYTES;
    HIPCHECK(hipSetDevice(g.device));
    if (toDevice)
    {
        reserve(g.walkRecords, bytes);
        if (!ok())
            return -1;
        HIPCHECK(hipMemcpy(g.walkRecords.ptr, host, bytes, hipMemcpyHostToDevice));
    }
    else
    {
        ARGCHECK(bytes <= g.walkRecords.bytes, "solr_hip_walk_records_copy: more workgroups than were recorded");
        if (!ok())
            return -1;
        HIPCHECK(hipMemcpy(host, g.walkRecords.ptr, bytes, hipMemcpyDeviceToHost));
    }
    return ok() ? 0 : -1;
}

int solr_hip_walk_replay(unsigned grid, long ldsBytes, int repeats, double ms[3], unsigned long long stats[4])
{
    if (!ready("solr_hip_walk_replay"))
        return -1;
    ARGCHECK(g.recorded, "solr_hip_walk_replay: no recorded frame (solr_hip_walk_records_keep(1), then solr_hip_walk_bound)");
    if (!ok())
        return -1;
    HIPCHECK(hipSetDevice(g.device));
    return replayRecords(grid, ldsBytes < 0 ? g.recordLds : (size_t)ldsBytes, repeats, ms, stats, false);
}

void solr_hip_walk_records_release(void)
{
    release(g.walkRecords);
    release(g.walkVisits);
    g.recorded = false;
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
        const int before = g.flights;
        g.flights = n < 1 ? 1 : (n > MAX_FLIGHTS ? MAX_FLIGHTS : n);
        g.current = 0;
        /* (which tiles are worth four quadrant waves depends on how many frames overlap: the launch order is made anew
         * with the next frame, not at the next regular sort) */
        if (g.flights != before)
            g.orderValid = false;
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

/* tiles the current launch order renders as four quadrant waves each (0: none, or no order) */
int solr_hip_split_tiles(void)
{
    return (g.hostStats && g.orderValid) ? (int)g.hostStats[5] : 0;
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
} // extern "C"
