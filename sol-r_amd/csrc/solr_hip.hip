/*
 * solr_hip.hip - the C ABI of the MI355X rendering engine (include/solr_hip.h): the reference's ten entry points
 * (CudaRayTracer.h:25-67) once per in-process device, device / stream / strip selection, the life of an engine
 * (initialize_scene ... finalize_scene).  The rest of the engine's host side: engine.h (state), solr_scene.hip (the
 * resident scene and its lists), solr_launch.hip (a frame), solr_post.hip, solr_image_ring.hip, solr_rccl.hip,
 * solr_diag.hip; the kernels: rt_device.h, renderer_kernel.h, rows/.  gfx950 only.
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

namespace solreng
{
/* (the engine's state - struct Engine, one per device of this process - and the helpers every part shares: engine.h) */
Engine gFirst;
Engine *gEngines[SOLR_MAX_GPU_COUNT] = {&gFirst};
int gDevices = 1;
int gRequested = 1;
Engine *gCurrent = &gFirst;
HostProfile gHostProfile;
} // namespace solreng

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
    releaseImageStreaming();
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


} // extern "C"

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
        if (communicatorUp())
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

