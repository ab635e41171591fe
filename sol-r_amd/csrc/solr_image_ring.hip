/*
 * solr_image_ring.hip - the pipelined read-back of the engine (include/solr_hip.h: solr_hip_d2h_image_async,
 * solr_hip_image_wait, solr_hip_image_share ...): a ring of page-locked host images, a copy stream, read-back tickets,
 * and the ring as a POSIX shared-memory segment that the ranks of a multi-process job fill together; and the read-back of
 * a frame taken one at a time, whose image leaves in bands of tile rows while its kernel still renders
 * (solr_hip_stream_next_image, solr_hip_d2h_streamed_image; renderer.h, ImageStreaming).  gfx950 only.
 */
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>

#include "engine.h"

using namespace solrdev;
using namespace solreng;

namespace solreng
{
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

/* (at finalize only: a ring that is re-made for a larger frame must leave the counters alone - the frame that is being
 * read back may still be counting into them) */
void releaseImageStreaming()
{
    if (g.streamRendered)
        (void)hipEventDestroy(g.streamRendered);
    g.streamRendered = nullptr;
    if (g.streamHostWords)
        (void)hipHostFree(g.streamHostWords);
    g.streamHostWords = nullptr;
    release(g.streamCounters);
    g.streamPlan = StreamPlan();
    g.streamBands = 0;
    g.streamSerial = 0;
    g.streamKey[0] = g.streamKey[1] = g.streamKey[2] = 0;
    g.streamedValid = false;
    g.streamSupport = -1;
}

/* ImageStreaming, the host's part before a launch (renderImpl): counters, plan and band words for this frame geometry
 * - made, or zeroed, when the geometry changes or the row counts near 2^32 - and the frame's serial.  False: this frame is
 * not streamed (SOLR_HIP_NO_IMAGE_STREAMING=1, too few tile rows, an allocation failed: the read-back then takes the plain route). */
/* the bands of a frame of that many tile rows; false: such a frame is not streamed (SOLR_HIP_NO_IMAGE_STREAMING=1, a frame
 * of fewer than sixteen tile rows) */
bool imageStreamingCuts(int tileRows, int firstRow[SOLR_STREAM_BANDS_MAX + 1], int *bands, bool withIds)
{
    if (g.streamSupport < 0)
    {
        static const char *off = getenv("SOLR_HIP_NO_IMAGE_STREAMING");
        g.streamSupport = (off && off[0] == '1') ? 0 : 1;
    }
    if (!g.streamSupport || tileRows < 2 * SOLR_STREAM_BANDS_MAX)
        return false;
    /* Five bands of equal height.  The copy engine moves rows twice as fast as the Cornell kernel renders them: every
     * band but the last has landed before the next is complete, and the last is what the frame waits for behind its
     * kernel.  More bands make that one shorter - and cost the host more calls and the kernel its order: the launch has
     * to respect the bands (k_orderTiles: the heaviest eighth of the tiles first wherever they lie, the others band after
     * band).  Per frame, Cornell / molecule (profiles/r6/stream_frame.txt): 3 bands 0.313 / 0.398 ms, 4: 0.304 / 0.388,
     * 5: 0.301 / 0.385, 6: 0.387 / 0.432 (the host's calls no longer keep up); read back behind the kernel 0.394 / 0.483.
     * SOLR_HIP_STREAM_BANDS (experiments): 1 ... SOLR_STREAM_BANDS_MAX; SOLR_HIP_STREAM_EQUAL=0: bands of n : n - 1 : ... : 1,
     * the last copy the shortest; SOLR_HIP_STREAM_HEAVY=h: the heaviest 1 / h of the tiles first (8; with none first the
     * molecule's frame is 0.412 ms, with half of them 0.42 and the Cornell box's 0.54) */
    /* SOLR_HIP_STREAM_CUTS="0.33,0.67,0.89" (experiments): the bands end at these fractions of the frame's tile rows */
    if (const char *given = getenv("SOLR_HIP_STREAM_CUTS"))
    {
        int n = 0;
        firstRow[0] = 0;
        for (const char *at = given; *at && n + 1 < SOLR_STREAM_BANDS_MAX;)
        {
            char *end = nullptr;
            const double f = strtod(at, &end);
            if (end == at)
                break;
            const int row = std::max(firstRow[n] + 1, std::min(tileRows - 1, (int)(f * tileRows + 0.5)));
            if (row > firstRow[n] && row < tileRows)
                firstRow[++n] = row;
            at = (*end == ',') ? end + 1 : end;
        }
        firstRow[n + 1] = tileRows;
        *bands = n + 1;
        return true;
    }
    /* (with the primitive ids - 39 MB a frame, PCIe busy from the first band to the last - three: every band is two copies,
     * and five cost 0.943 ms a Cornell frame where three cost 0.873; behind the kernel 1.0) */
    static const int asked = getenv("SOLR_HIP_STREAM_BANDS") ? std::max(1, std::min(SOLR_STREAM_BANDS_MAX, atoi(getenv("SOLR_HIP_STREAM_BANDS")))) : 0;
    const int wanted = asked ? asked : (withIds ? 3 : 5);
    static const bool equal = !(getenv("SOLR_HIP_STREAM_EQUAL") && getenv("SOLR_HIP_STREAM_EQUAL")[0] == '0');
    const int total = equal ? wanted : wanted * (wanted + 1) / 2;
    int row = 0, weight = 0;
    for (int b = 0; b < wanted; ++b)
    {
        firstRow[b] = row;
        weight += equal ? 1 : wanted - b;
        row = std::max(row + 1, (int)((long)tileRows * weight / total));
    }
    firstRow[wanted] = tileRows;
    *bands = wanted;
    return true;
}

bool armImageStreaming(FrameArgs &F, int tileRows, hipStream_t stream, bool withIds)
{
    int cuts[SOLR_STREAM_BANDS_MAX + 1], bands = 0;
    if (!imageStreamingCuts(tileRows, cuts, &bands, withIds))
        return false;
    const long key[3] = {F.tilesX, tileRows + 100000l * bands, g.width};
    const unsigned perRow = (unsigned)(F.tilesX * SPLIT_PARTS);
    const bool fresh = memcmp(key, g.streamKey, sizeof(key)) != 0 || !g.streamCounters.ptr;
    /* (variant 14, tests: as if the row counts neared 2^32 every third frame) */
    if (fresh || (unsigned long long)(g.streamSerial + 2u) * perRow >= 0xffffffffull || (g.variant == 14 && g.streamSerial >= 3u))
    {
        /* nothing of an earlier streamed frame may be under way: its waves count into these words */
        quiesce();
        const size_t words = (size_t)64 * tileRows + (size_t)64 * SOLR_STREAM_BANDS_MAX;
        reserve(g.streamCounters, words * sizeof(unsigned) + sizeof(StreamPlan));
        if (!ok())
            return false;
        HIPCHECK(hipMemset(g.streamCounters.ptr, 0, words * sizeof(unsigned)));
        if (!g.streamHostWords)
        {
            HIPCHECK(hipHostMalloc((void **)&g.streamHostWords, SOLR_STREAM_BANDS_MAX * sizeof(unsigned), hipHostMallocMapped));
            if (ok())
                HIPCHECK(hipHostGetDevicePointer((void **)&g.streamPlan.hostWord, g.streamHostWords, 0));
        }
        if (!ok())
        {
            /* no page-locked words: the frames are rendered and read back as they always were */
            (void)solr_hip_clear_error();
            g.streamSupport = 0;
            return false;
        }
        for (int b = 0; b < SOLR_STREAM_BANDS_MAX; ++b)
            g.streamHostWords[b] = 0u;
        g.streamPlan.bandDone = (unsigned *)g.streamCounters.ptr + (size_t)64 * tileRows;
        g.streamPlan.bands = bands;
        for (int b = 0; b <= bands; ++b)
            g.streamPlan.firstRow[b] = cuts[b];
        HIPCHECK(hipMemcpy((char *)g.streamCounters.ptr + words * sizeof(unsigned), &g.streamPlan, sizeof(StreamPlan), hipMemcpyHostToDevice));
        HIPCHECK(hipStreamSynchronize(stream));
        if (!ok())
            return false;
        memcpy(g.streamKey, key, sizeof(key));
        g.streamBands = bands;
        g.streamSerial = 0;
    }
    F.rowDone = (unsigned *)g.streamCounters.ptr;
    F.streamPlan = (const StreamPlan *)((const char *)g.streamCounters.ptr + ((size_t)64 * tileRows + (size_t)64 * SOLR_STREAM_BANDS_MAX) * sizeof(unsigned));
    F.streamSerial = ++g.streamSerial;
    return true;
}

/* ... and behind the launch, on the frame's own stream: the event that says the kernel has ended */
void markStreamedFrame(hipStream_t stream)
{
    if (!g.streamRendered)
        HIPCHECK(hipEventCreateWithFlags(&g.streamRendered, hipEventDisableTiming));
    if (ok())
        HIPCHECK(hipEventRecord(g.streamRendered, stream));
}
} // namespace solreng

/* Pipelined read-back of the image (SURVEY.md 8d defines the metric over cudaRender + d2h_bitmap; d2h_bitmap waits
 * for the frame and then for the copy, CudaRayTracer.cu:1647-1672, and nothing renders meanwhile).  Called after
 * cudaRender, solr_hip_d2h_image_async enqueues the copy of the RGB image of the frame rendered last - this
 * process's strip at its place in a full-size image, like d2h_bitmap - into a page-locked host image of the
 * engine's, on a copy stream of its own behind that frame's kernel, and returns a ticket at once; the render
 * streams are free for the next frames (solr_hip_set_frames_in_flight), whose kernels overlap the copy.
 * solr_hip_image_wait(ticket) waits for that one copy and returns the host image; it stays valid until
 * MAX_FLIGHTS more tickets have been handed out.  The ids stay on the device until d2h_bitmap asks for them. */
namespace solreng
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
} // namespace solreng

extern "C" {

/* 0 (default): the pipelined read-back copies on a stream of its own behind the frame's kernel; 1: on the frame's own
 * stream (what to choose: see copyStripBehindFrame) */
void solr_hip_set_copy_route(int onTheFramesOwnStream)
{
    gFirst.copyOnRenderStream = onTheFramesOwnStream != 0;
}

/* ImageStreaming: the frame the next cudaRender launches counts its tiles, if it is a whole frame of one device with
 * the RGB conversion fused in, one frame in flight; solr_hip_d2h_streamed_image, called behind that cudaRender, sends
 * its image off band by band while the kernel renders (renderer.h).  Returns 1 when frames can be streamed, 0 when they
 * will be read back the plain way; on = -1 asks, behind a cudaRender, whether that frame was such a frame; -2 how many
 * images have left in bands so far. */
int solr_hip_stream_next_image(int on)
{
    if (!ready("solr_hip_stream_next_image"))
        return 0;
    if (on == -2) /* (tests, tools: images that have left in bands since initialize_scene) */
        return (int)(g.streamedDelivered & 0x7fffffff);
    if (on < 0) /* (asked behind a cudaRender: did that frame count its tiles?) */
        return g.streamedValid ? 1 : 0;
    g.streamNext = on >= 2 ? 2 : (on != 0 ? 1 : 0);
    return g.streamSupport != 0 ? 1 : 0;
}

/* Behind a cudaRender that counted its tiles: every band of the image is copied to its rows of `image` - host memory of
 * any kind - as soon as the band's word has come, and the call returns when the last has landed.  1: done; 0: that frame
 * was not such a frame and nothing was copied (d2h_bitmap is the way then); -1: error. */
static int streamedReadBack(BitmapBuffer *image, PrimitiveXYIdBuffer *primitiveIds, const char *who)
{
    HostSpan whole(who);
    if (!ready(who))
        return -1;
    ARGCHECK(image != nullptr, "solr_hip_d2h_streamed_image: no image");
    if (!ok())
        return -1;
    const int flight = g.current;
    const void *src = flightBitmap(flight).ptr;
    if (!g.streamedValid || src != g.streamedBitmap || g.nbRows >= 0 || g.streamBands <= 0 || (primitiveIds && !g.streamedIds))
        return 0;
    HIPCHECK(hipSetDevice(g.device));
    ensureCopyStream();
    /* the host watches the bands' words (page-locked memory the waves write to); a band whose word has not come is copied
     * when the kernel has ended - everything it wrote is in memory then, whatever became of the word.  (`image` is
     * pageable as a rule, and a copy into pageable memory returns when it is done: the loop is the pipeline.) */
    const size_t rowBytes = (size_t)g.width * SOLR_COLOR_DEPTH, rowIds = (size_t)g.width;
    const PrimitiveXYIdBuffer *srcIds = (const PrimitiveXYIdBuffer *)flightIds(flight).ptr;
    bool ended = false;
    for (int b = 0; b < g.streamBands && ok(); ++b)
    {
        const int y0 = g.streamPlan.firstRow[b] * TILE_H, y1 = std::min(g.height, g.streamPlan.firstRow[b + 1] * TILE_H);
        volatile unsigned *word = g.streamHostWords + b;
        while (!ended && (int)(*word - g.streamSerial) < 0)
            ended = !g.streamRendered || hipEventQuery(g.streamRendered) != hipErrorNotReady;
        std::atomic_thread_fence(std::memory_order_acquire);
        if (y1 > y0)
            HIPCHECK(hipMemcpyAsync(image + rowBytes * y0, (const char *)src + rowBytes * y0, rowBytes * (y1 - y0), hipMemcpyDeviceToHost,
                                    g.copyStream));
        if (y1 > y0 && primitiveIds && ok())
            HIPCHECK(hipMemcpyAsync(primitiveIds + rowIds * y0, srcIds + rowIds * y0, rowIds * (y1 - y0) * sizeof(PrimitiveXYIdBuffer),
                                    hipMemcpyDeviceToHost, g.copyStream));
    }
    HIPCHECK(hipStreamSynchronize(g.copyStream));
    g.streamedValid = false;
    ++g.streamedDelivered;
    return ok() ? 1 : -1;
}

int solr_hip_d2h_streamed_image(BitmapBuffer *image)
{
    return streamedReadBack(image, nullptr, "solr_hip_d2h_streamed_image");
}

/* ... with the primitive ids of every pixel (16 bytes each: five times the image), as d2h_bitmap hands both over
 * (CudaRayTracer.cu:1647-1672) - for a frame asked for with solr_hip_stream_next_image(2), whose waves store their ids
 * with device scope as well */
int solr_hip_d2h_streamed(BitmapBuffer *image, PrimitiveXYIdBuffer *primitivesXYIds)
{
    ARGCHECK(primitivesXYIds != nullptr, "solr_hip_d2h_streamed: no array for the ids");
    if (!ok())
        return -1;
    return streamedReadBack(image, primitivesXYIds, "solr_hip_d2h_streamed");
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

} // extern "C"

/* test-only hooks of include/solr_hip_probes.h (csrc/solr_probes.hip) */
namespace solrprobe
{
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

} // namespace solrprobe
