/*
 * engine.h - the state of the MI355X rendering engine and the helpers every part of its host side uses: what
 * solr_hip.hip (the boundary), solr_scene.hip (scene upload, list builders), solr_launch.hip (the renderer's launch), solr_diag.hip (knobs and diagnostics), solr_image_ring.hip (the pipelined
 * read-back), solr_rccl.hip (strips, communicator, gather, halo) and solr_post.hip (the post-processing kernels) share.
 * One Engine per device this process renders on; `g` is the engine a function works on.  gfx950 only.
 */
#ifndef SOLR_ENGINE_H
#define SOLR_ENGINE_H

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/solr_hip.h"
#include "renderer.h"
#include "lists_device.h"

namespace solreng
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
extern HostProfile gHostProfile; /* (solr_hip.hip) */
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
/* Frames between two sorts of the tiles by cost (k_orderTiles: one workgroup, 46 us for the 32 400 tiles of a 1080p frame, on the
 * frame's own stream: 2.9 us of every Cornell frame at sixteen, which it was until round 6; 64: delivered frames 0.2442 ->
 * 0.2419 ms, one at a time 0.2734 -> 0.2712.  A decision that changes - cost order on / off, a streamed frame's bands - is
 * still acted on at once.  On a stream of its own the sort cost the delivered frames a third: one stream more, and the
 * runtime's four hardware queues are dealt out differently; on the copy stream it delays the images).
 * SOLR_HIP_SORT_PERIOD (environment): experiments. */
inline int sortPeriod()
{
    static const int period = getenv("SOLR_HIP_SORT_PERIOD") ? std::max(2, atoi(getenv("SOLR_HIP_SORT_PERIOD"))) : 64;
    return period;
}

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
    bool sortedFree = false;                      /* the copy of the order-free lists with sorted bounds is up to date */
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
    BandCuts orderCuts = {};          /* ... band after band (ImageStreaming); bands = 0: by cost alone */
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

    /* ImageStreaming (renderer.h): the next frame counts its tiles if it can (solr_hip_stream_next_image), and
     * solr_hip_d2h_streamed_image then sends its image off band by band as the bands' words come */
    int streamNext = 0;                  /* 0 no, 1 the image, 2 the image and the primitive ids */
    bool streamedIds = false;            /* the frame rendered last stored its ids for the bands too */
    bool streamedValid = false;          /* the frame rendered last counted its tiles: serial, image and bands below */
    const void *streamedBitmap = nullptr;
    unsigned streamSerial = 0;           /* streamed frames since the counters were zeroed */
    long streamKey[3] = {0, 0, 0};       /* tilesX, tile rows (+ 100000 x the number of bands), image width the counters belong to */
    DeviceBuffer streamCounters;         /* rowDone | bandDone | the StreamPlan */
    StreamPlan streamPlan = {};          /* host image of the plan */
    int streamBands = 0;
    unsigned *streamHostWords = nullptr; /* StreamPlan::hostWord, the host's address */
    hipEvent_t streamRendered = nullptr; /* behind the kernel of the streamed frame rendered last */
    long streamedDelivered = 0;          /* images that left in bands */
    int lastMask = -1;                   /* features of the lean row the frame before took (-1: another kernel, or none yet) */
    int streamSupport = -1;              /* 1 / 0; -1: not asked yet (SOLR_HIP_NO_IMAGE_STREAMING) */

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
extern Engine gFirst;
extern Engine *gEngines[SOLR_MAX_GPU_COUNT];
extern int gDevices;   /* engines in use since initialize_scene: min(occupancyParameters.x, devices visible) */
extern int gRequested; /* occupancyParameters.x as initialize_scene was given it */
extern Engine *gCurrent;
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
inline int activeFlights()
{
    if (g.flights < 2 || !(g.ownStream || g.callerStreams))
        return 1;
    int n = 1;
    while (n < g.flights && n < MAX_FLIGHTS && g.extraStream[n - 1])
        ++n;
    return n;
}
inline bool twoFlights() { return activeFlights() > 1; }
inline hipStream_t flightStream(int f) { return f ? g.extraStream[f - 1] : g.stream; }
inline DeviceBuffer &flightPp(int f) { return f ? g.ppX[f - 1] : g.pp; }
inline DeviceBuffer &flightIds(int f) { return f ? g.idsX[f - 1] : g.ids; }
inline DeviceBuffer &flightBitmap(int f) { return g.bitmapSide[f] ? g.bitmapAlt[f] : (f ? g.bitmapX[f - 1] : g.bitmap); }
/* nothing may touch scene or frame buffers while a frame is still in flight on the other stream */
inline void quiesce()
{
    for (hipStream_t extra : g.extraStream)
        if (extra)
            (void)hipStreamSynchronize(extra);
    if (g.stream)
        (void)hipStreamSynchronize(g.stream);
    if (g.copyStream)
        (void)hipStreamSynchronize(g.copyStream);
}

inline void setError(int code, const char *what, const char *file, int line)
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

inline bool ok()
{
    return g.errorCode == 0;
}

inline bool ready(const char *who)
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

inline void release(DeviceBuffer &b)
{
    if (b.ptr)
        (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.bytes = 0;
}

/* grow-only device allocation */
inline void reserve(DeviceBuffer &b, size_t bytes)
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

inline int stripRows()
{
    return g.nbRows >= 0 ? g.nbRows : g.height;
}


/* ---- what the parts ask of each other (defined in the file named) ---------------------------------------------------- */
/* solr_scene.hip: the resident scene - uploads (one engine's share of the boundary's h2d_* calls), its lists, rotation */
void h2dSceneOne(BoundingBox *boundingBoxes, int nbActiveBoxes, Primitive *primitives, int nbPrimitives, Lamp *lamps, int nbLamps);
void h2dMaterialsOne(Material *materials, int nbActiveMaterials);
void h2dRandomsOne(float *randoms);
void h2dRandomsSizedOne(const float *randoms, long count);
void h2dTexturesOne(int activeTextures, TextureInfo *textureInfos);
void h2dLightInformationOne(LightInformation *lightInformation, int lightInformationSize);
void setMovableOne(const unsigned char *flags, int nbPrimitives);
bool canRotateOne(const float center[3], const float cosAngles[3], const float sinAngles[3], float viewDistance);
int rotatePrimitivesOne(const float center[3], const float cosAngles[3], const float sinAngles[3], float viewDistance);
void checkTextureTables();
void maybeBuildOrderFreeLists();
void flushGeometry();
void refreshExactList();
void dropFreeStage(bool originToo);
bool orderFreeListsUsable();
bool shortRayListsChoice();
SceneArgs makeScene(bool exactNodes);
int tightListsFor(const SceneArgs &S, const SceneInfo &sceneInfo, bool exactNodes);
/* solr_launch.hip: a frame - buffers, the launch, post-processing, read-back */
void allocateFrame();
int neededFeatures(const SceneInfo &sceneInfo, bool full);
void renderImpl(const SceneInfo &sceneInfo, const vec4i &objects, const PostProcessingInfo &ppInfo, const float origin[3],
                const float direction[3], const float angles[4], bool counting, unsigned long long counts[8]);
void collectEvents();
void d2hBitmapOne(const SceneInfo &sceneInfo, BitmapBuffer *bitmap, PrimitiveXYIdBuffer *primitivesXYIds, bool wait);
void d2hBitmapWait();
/* solr_image_ring.hip: the ring of page-locked host images behind solr_hip_d2h_image_async */
void releaseImageRing();
void releaseImageStreaming();
bool imageStreamingCuts(int tileRows, int firstRow[SOLR_STREAM_BANDS_MAX + 1], int *bands, bool withIds);
bool armImageStreaming(FrameArgs &F, int tileRows, hipStream_t stream, bool withIds);
void markStreamedFrame(hipStream_t stream);
void ensureCopyStream();
bool ensureImageRing();
int nextTicket(int *slot);
/* solr_rccl.hip */
bool haveCommunicator();
bool communicatorUp(); /* a communicator of any size exists (initialize_scene refuses several in-process devices then) */
int agreedHaloRows(const PostProcessingInfo &ppInfo);
void exchangeDepthHalo(int flight, hipStream_t stream, const PixelRecord *pp, int W, int firstRow, int nbRows, int frameRows,
                       int wanted, DepthHalo *halo);
bool shareRandoms(); /* every rank takes rank 0's seed for the random sequence (solr_hip_comm_shared_seed) */

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
/* solr_launch.hip: the neighbourhood post-processing of a frame (also what the test-only solr_hip_probe_postprocess runs) */
void launchPostProcess(const SceneInfo &sceneInfo, const PostProcessingInfo &ppInfo, int flight, hipStream_t stream, int firstRow,
                       int nbRows, unsigned char *bitmap, HaloDebt &debt);
} // namespace solreng

/* solr_post.hip: the post-processing kernels of cudaRender (CRT:1057-1358) and the tile sort, behind plain launchers */
namespace solrpost
{
void defaultConversion(hipStream_t stream, const SceneInfo &si, int nbPixels, const PixelRecord *pp, unsigned char *bitmap);
void ambientOcclusion(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
                      const float *randoms, long nbRandoms, unsigned char *bitmap, const DepthHalo &halo, int firstRow,
                      float randomsReach, bool heavyFirst);
void depthOfField(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
                  const float *randoms, long nbRandoms, unsigned char *bitmap);
void radiosity(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
               const int4 *ids, const float *randoms, long nbRandoms, unsigned char *bitmap);
void filter(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
            unsigned char *bitmap);
void cartoon(hipStream_t stream, const SceneInfo &si, const PostProcessingInfo &ppi, int nbRows, const PixelRecord *pp,
             unsigned char *bitmap);
void orderTiles(hipStream_t stream, const unsigned *cost, unsigned *snapshot, unsigned *order, int nbTiles,
                volatile unsigned *hostStats, int flights, const BandCuts &cuts);
void packDepthRows(hipStream_t stream, const PixelRecord *pp, int W, int row0, int n, float *out);
} // namespace solrpost

#endif
