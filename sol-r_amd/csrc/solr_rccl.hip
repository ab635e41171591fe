/*
 * solr_rccl.hip - the multi-process side of the engine (include/solr_hip.h: solr_hip_comm_*, solr_hip_strip_rows,
 * solr_hip_balance_strips, solr_hip_gather_strips, solr_hip_d2h_gathered ...): one process per GPU, row strips of the
 * frame, RCCL loaded at run time for the gather to the root, the depth rows the ambient-occlusion kernel needs from
 * the neighbouring strips, and the shared random seed.  gfx950 only.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "engine.h"

using namespace solrdev;
using namespace solreng;

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
namespace /* this file's own: the library handle, the communicators, the strip table */
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
} // namespace
namespace solreng /* (engine.h) */
{
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
} // namespace solreng
namespace
{

/* Rank 0's random buffer to every rank (the buffer feeds the taps of the ambient-occlusion kernel, the depth of field
 * and the jitter of accumulation passes: strips rendered from different buffers do not assemble to the frame one GPU
 * renders).  With a communicator, rank 0's buffer is THE buffer: solr_hip_comm_init and every h2d_randoms after it
 * end with this.  Blocking; every rank. */
} // namespace
namespace solreng /* (engine.h) */
{
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
} // namespace solreng
namespace
{

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
} // namespace
namespace solreng /* (engine.h) */
{
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
        if (sendUp)
            solrpost::packDepthRows(stream, pp, W, 0, mine, (float *)g.haloSendTop[flight].ptr);
        if (sendDown)
            solrpost::packDepthRows(stream, pp, W, nbRows - mine, mine, (float *)g.haloSendBottom[flight].ptr);
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
} // namespace solreng
namespace
{

/* (for renderImpl, which is defined before this layer) */
} // namespace
namespace solreng /* (engine.h) */
{
bool haveCommunicator()
{
    return rccl.comm != nullptr && rccl.world > 1;
}
} // namespace solreng
namespace
{
} // namespace
namespace solreng /* (engine.h) */
{
bool communicatorUp()
{
    return rccl.comm != nullptr;
}
} // namespace solreng
namespace
{
} // namespace

extern "C" {

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
        const int y0 = (t / tilesX) * TILE_H, y1 = std::min(nbRows, y0 + TILE_H);
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
} // extern "C"
