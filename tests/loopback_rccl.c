/*
 * loopback_rccl.c - TEST INFRASTRUCTURE: a stand-in for librccl.so that lets several processes ON ONE GPU (or, in
 * host mode, on no GPU at all) run the engine's multi-rank code - solr_hip_comm_init, solr_hip_gather_strips,
 * solr_hip_gather_ids, solr_hip_balance_strips, the depth-halo exchange inside cudaRender - with world sizes the
 * one-GPU test box cannot give RCCL (which refuses two ranks on one device).  It implements exactly the entry
 * points sol-r_amd/csrc/solr_hip.hip resolves with dlsym and nothing else; messages travel as files in a directory
 * (default /dev/shm), staged through the host.
 *
 * Loaded only when SOLR_HIP_RCCL_LIBRARY names it (tests/multi_rank_worker.py, and bench.py's rehearsal mode on a
 * one-GPU box); never part of the product, never on the path of a measured number.
 *
 * Semantics kept from the real library, because they are what the engine's code relies on:
 *   - point-to-point operations between ncclGroupStart / ncclGroupEnd are issued together: all sends are posted
 *     before any receive blocks, so a rank may send to and receive from the same peers in one group;
 *   - messages between a pair of ranks arrive in the order they were sent (a sequence number per ordered pair);
 *   - an operation is ordered after the work already enqueued on its stream (the stream is synchronised first) and
 *     before anything enqueued later (the copy has completed when the call returns).
 * Stricter than the real library on purpose: a receive whose byte count differs from the matching send's fails
 * with ncclInvalidUsage, and a peer that does not show up within SOLR_LOOPBACK_TIMEOUT seconds (default 60) fails
 * with ncclSystemError - where RCCL would hang or truncate, a test gets an error code.
 *
 * Build: gcc -O2 -shared -fPIC -o libloopback_rccl.so loopback_rccl.c -ldl   (no HIP headers: the three runtime
 * calls it needs are looked up in the process, i.e. in the copy of the HIP runtime the engine itself uses).
 * SOLR_LOOPBACK_HOST=1: buffers are host memory (memcpy) - the transport's own CPU test.
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

enum
{
    ncclSuccess = 0,
    ncclUnhandledCudaError = 1,
    ncclSystemError = 2,
    ncclInternalError = 3,
    ncclInvalidArgument = 4,
    ncclInvalidUsage = 5
};
enum
{
    LB_SUM = 0,
    LB_MAX = 2
}; /* ncclSum, ncclMax (rccl.h) */
enum
{
    LB_UINT8 = 1,
    LB_INT32 = 2,
    LB_FLOAT32 = 7
}; /* ncclUint8, ncclInt32, ncclFloat32 */

typedef struct
{
    char internal[128];
} ncclUniqueId;

#define LB_MAX_RANKS 64
#define LB_MAX_OPS 256
typedef struct
{
    int send; /* 1 send, 0 receive */
    void *buffer;
    size_t bytes;
    int peer;
    void *stream;
} Op;
typedef struct ncclComm
{
    char token[64];
    char dir[160];
    int rank, world;
    unsigned long sent[LB_MAX_RANKS], received[LB_MAX_RANKS], reductions, splits;
} Comm;
typedef Comm *ncclComm_t;

static __thread int groupDepth = 0;
static __thread int nbOps = 0;
static __thread Op ops[LB_MAX_OPS];
static __thread Comm *opComm[LB_MAX_OPS];

/* the HIP runtime of the process (hipMemcpyDeviceToHost = 2, hipMemcpyHostToDevice = 1) */
static int (*p_hipMemcpy)(void *, const void *, size_t, int) = NULL;
static int (*p_hipStreamSynchronize)(void *) = NULL;
static int hostMode = -1;

static int setup(void)
{
    if (hostMode >= 0)
        return 0;
    const char *h = getenv("SOLR_LOOPBACK_HOST");
    hostMode = (h && h[0] == '1') ? 1 : 0;
    if (!hostMode)
    {
        p_hipMemcpy = (int (*)(void *, const void *, size_t, int))dlsym(RTLD_DEFAULT, "hipMemcpy");
        p_hipStreamSynchronize = (int (*)(void *))dlsym(RTLD_DEFAULT, "hipStreamSynchronize");
        if (!p_hipMemcpy || !p_hipStreamSynchronize)
        {
            fprintf(stderr, "loopback_rccl: no HIP runtime in this process (set SOLR_LOOPBACK_HOST=1 for host buffers)\n");
            hostMode = -1;
            return -1;
        }
    }
    return 0;
}

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

static double timeoutSeconds(void)
{
    const char *t = getenv("SOLR_LOOPBACK_TIMEOUT");
    double s = t ? atof(t) : 60.0;
    return s > 0.0 ? s : 60.0;
}

static void nap(void)
{
    struct timespec t = {0, 50000}; /* 50 us */
    nanosleep(&t, NULL);
}

/* write a message under a temporary name, then rename: a reader never sees half a file */
static int putFile(const char *path, const void *data, size_t bytes)
{
    char tmp[256];
    snprintf(tmp, sizeof(tmp), "%s.part", path);
    int fd = open(tmp, O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0)
        return -1;
    const char *p = (const char *)data;
    size_t left = bytes;
    while (left)
    {
        ssize_t n = write(fd, p, left);
        if (n < 0)
        {
            if (errno == EINTR)
                continue;
            close(fd);
            unlink(tmp);
            return -1;
        }
        p += n;
        left -= (size_t)n;
    }
    close(fd);
    return rename(tmp, path);
}

/* wait for `path`, read exactly `bytes` from it; 0 fine, 1 timed out, 2 wrong size, 3 i/o */
static int getFile(const char *path, void *data, size_t bytes, int removeIt)
{
    const double deadline = now() + timeoutSeconds();
    int fd;
    while ((fd = open(path, O_RDONLY)) < 0)
    {
        if (now() > deadline)
            return 1;
        nap();
    }
    struct stat st;
    if (fstat(fd, &st) != 0)
    {
        close(fd);
        return 3;
    }
    if ((size_t)st.st_size != bytes)
    {
        close(fd);
        return 2;
    }
    char *p = (char *)data;
    size_t left = bytes;
    while (left)
    {
        ssize_t n = read(fd, p, left);
        if (n <= 0)
        {
            if (n < 0 && errno == EINTR)
                continue;
            close(fd);
            return 3;
        }
        p += n;
        left -= (size_t)n;
    }
    close(fd);
    if (removeIt)
        unlink(path);
    return 0;
}

static size_t elementSize(int datatype)
{
    switch (datatype)
    {
    case LB_UINT8:
        return 1;
    case LB_INT32:
    case LB_FLOAT32:
        return 4;
    default:
        return 0;
    }
}

static int fromDevice(void *host, const void *device, size_t bytes, void *stream)
{
    if (hostMode)
    {
        memcpy(host, device, bytes);
        return 0;
    }
    if (p_hipStreamSynchronize(stream) != 0)
        return -1;
    return p_hipMemcpy(host, device, bytes, 2) == 0 ? 0 : -1;
}

static int toDevice(void *device, const void *host, size_t bytes, void *stream)
{
    if (hostMode)
    {
        memcpy(device, host, bytes);
        return 0;
    }
    if (p_hipStreamSynchronize(stream) != 0)
        return -1;
    return p_hipMemcpy(device, host, bytes, 1) == 0 ? 0 : -1;
}

static int doSend(Comm *c, const Op *op)
{
    void *host = malloc(op->bytes ? op->bytes : 1);
    if (!host)
        return ncclSystemError;
    int rc = ncclSuccess;
    if (fromDevice(host, op->buffer, op->bytes, op->stream) != 0)
        rc = ncclUnhandledCudaError;
    else
    {
        char path[256];
        snprintf(path, sizeof(path), "%s/%s.p2p.%d-%d.%lu", c->dir, c->token, c->rank, op->peer, c->sent[op->peer]);
        if (putFile(path, host, op->bytes) != 0)
            rc = ncclSystemError;
        else
            c->sent[op->peer]++;
    }
    free(host);
    return rc;
}

static int doRecv(Comm *c, const Op *op)
{
    void *host = malloc(op->bytes ? op->bytes : 1);
    if (!host)
        return ncclSystemError;
    char path[256];
    snprintf(path, sizeof(path), "%s/%s.p2p.%d-%d.%lu", c->dir, c->token, op->peer, c->rank, c->received[op->peer]);
    int rc = ncclSuccess;
    const int got = getFile(path, host, op->bytes, 1);
    if (got == 1)
    {
        fprintf(stderr, "loopback_rccl: rank %d waited %.0f s for message %lu of rank %d (%zu bytes): no matching send\n",
                c->rank, timeoutSeconds(), c->received[op->peer], op->peer, op->bytes);
        rc = ncclSystemError;
    }
    else if (got == 2)
    {
        fprintf(stderr, "loopback_rccl: rank %d expects %zu bytes from rank %d (message %lu), the send has another size\n",
                c->rank, op->bytes, op->peer, c->received[op->peer]);
        unlink(path);
        c->received[op->peer]++;
        rc = ncclInvalidUsage;
    }
    else if (got != 0)
        rc = ncclSystemError;
    else
    {
        c->received[op->peer]++;
        if (toDevice(op->buffer, host, op->bytes, op->stream) != 0)
            rc = ncclUnhandledCudaError;
    }
    free(host);
    return rc;
}

static int flushOps(void)
{
    int rc = ncclSuccess;
    for (int i = 0; i < nbOps; ++i)
        if (ops[i].send)
        {
            const int r = doSend(opComm[i], &ops[i]);
            rc = rc ? rc : r;
        }
    for (int i = 0; i < nbOps; ++i)
        if (!ops[i].send)
        {
            const int r = doRecv(opComm[i], &ops[i]);
            rc = rc ? rc : r;
        }
    nbOps = 0;
    return rc;
}

static int enqueue(Comm *c, int send, void *buffer, size_t count, int datatype, int peer, void *stream)
{
    if (!c || peer < 0 || peer >= c->world || elementSize(datatype) == 0 || (count && !buffer))
        return ncclInvalidArgument;
    if (nbOps >= LB_MAX_OPS)
        return ncclInternalError;
    ops[nbOps].send = send;
    ops[nbOps].buffer = buffer;
    ops[nbOps].bytes = count * elementSize(datatype);
    ops[nbOps].peer = peer;
    ops[nbOps].stream = stream;
    opComm[nbOps] = c;
    ++nbOps;
    return groupDepth > 0 ? ncclSuccess : flushOps();
}

/* ---- the entry points solr_hip.hip resolves ------------------------------------------------------------------ */

int ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id)
        return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    unsigned long r = (unsigned long)getpid() * 2654435761ul ^ (unsigned long)(now() * 1e6);
    snprintf(id->internal, sizeof(id->internal), "lb%08lx%04x", r & 0xfffffffful, (unsigned)(rand() & 0xffff));
    return ncclSuccess;
}

int ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > LB_MAX_RANKS || rank < 0 || rank >= nranks)
        return ncclInvalidArgument;
    if (setup() != 0)
        return ncclSystemError;
    Comm *c = (Comm *)calloc(1, sizeof(Comm));
    if (!c)
        return ncclSystemError;
    memcpy(c->token, id.internal, sizeof(c->token) - 1);
    c->token[sizeof(c->token) - 1] = 0;
    const char *dir = getenv("SOLR_LOOPBACK_DIR");
    snprintf(c->dir, sizeof(c->dir), "%s", (dir && dir[0]) ? dir : "/dev/shm");
    c->rank = rank;
    c->world = nranks;
    /* rendezvous: every rank announces itself and waits for the others */
    char path[256];
    snprintf(path, sizeof(path), "%s/%s.here.%d", c->dir, c->token, rank);
    if (putFile(path, &rank, sizeof(rank)) != 0)
    {
        free(c);
        return ncclSystemError;
    }
    for (int r = 0; r < nranks; ++r)
    {
        int who = -1;
        snprintf(path, sizeof(path), "%s/%s.here.%d", c->dir, c->token, r);
        if (getFile(path, &who, sizeof(who), 0) != 0 || who != r)
        {
            fprintf(stderr, "loopback_rccl: rank %d did not see rank %d join\n", rank, r);
            free(c);
            return ncclSystemError;
        }
    }
    *comm = c;
    return ncclSuccess;
}

/* ncclCommSplit with one colour: a second communicator of the same ranks, whose messages and sequence numbers are its
 * own (the engine asks for one per frame in flight: operations of different communicators do not order against each
 * other).  Collective on the parent, called by every rank in the same order: the child's token is the parent's plus
 * the number of the split. */
int ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t *newcomm, void *config)
{
    (void)config;
    if (!comm || !newcomm || color != 0 || key != comm->rank)
        return ncclInvalidArgument;
    ncclUniqueId id;
    memset(&id, 0, sizeof(id));
    snprintf(id.internal, sizeof(id.internal), "%.40ss%lu", comm->token, ++comm->splits);
    return ncclCommInitRank(newcomm, comm->world, id, comm->rank);
}

int ncclCommCount(const ncclComm_t comm, int *count)
{
    if (!comm || !count)
        return ncclInvalidArgument;
    *count = comm->world;
    return ncclSuccess;
}

int ncclCommDestroy(ncclComm_t comm)
{
    if (!comm)
        return ncclInvalidArgument;
    /* every rank says goodbye; rank 0 waits for all of them (a peer may still be reading a reduction file) and
     * then sweeps everything that carries the communicator's token */
    char path[512];
    snprintf(path, sizeof(path), "%s/%s.bye.%d", comm->dir, comm->token, comm->rank);
    (void)putFile(path, &comm->rank, sizeof(comm->rank));
    if (comm->rank == 0)
    {
        for (int r = 0; r < comm->world; ++r)
        {
            int who;
            snprintf(path, sizeof(path), "%s/%s.bye.%d", comm->dir, comm->token, r);
            (void)getFile(path, &who, sizeof(who), 0);
        }
        DIR *d = opendir(comm->dir);
        if (d)
        {
            struct dirent *e;
            const size_t n = strlen(comm->token);
            while ((e = readdir(d)))
                if (strncmp(e->d_name, comm->token, n) == 0 && e->d_name[n] == '.')
                {
                    snprintf(path, sizeof(path), "%s/%s", comm->dir, e->d_name);
                    unlink(path);
                }
            closedir(d);
        }
    }
    free(comm);
    return ncclSuccess;
}

int ncclGroupStart(void)
{
    ++groupDepth;
    return ncclSuccess;
}

int ncclGroupEnd(void)
{
    if (groupDepth <= 0)
        return ncclInvalidUsage;
    if (--groupDepth > 0)
        return ncclSuccess;
    return flushOps();
}

int ncclSend(const void *sendbuff, size_t count, int datatype, int peer, ncclComm_t comm, void *stream)
{
    return enqueue(comm, 1, (void *)sendbuff, count, datatype, peer, stream);
}

int ncclRecv(void *recvbuff, size_t count, int datatype, int peer, ncclComm_t comm, void *stream)
{
    return enqueue(comm, 0, recvbuff, count, datatype, peer, stream);
}

/* float32 sum / max, reduced in rank order on every rank (all ranks get the same bits) */
int ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op, ncclComm_t comm, void *stream)
{
    if (!comm || datatype != LB_FLOAT32 || (op != LB_SUM && op != LB_MAX) || (count && (!sendbuff || !recvbuff)))
        return ncclInvalidArgument;
    if (groupDepth > 0)
        return ncclInvalidUsage;
    const size_t bytes = count * sizeof(float);
    float *mine = (float *)malloc(bytes ? bytes : 4), *other = (float *)malloc(bytes ? bytes : 4);
    if (!mine || !other)
    {
        free(mine);
        free(other);
        return ncclSystemError;
    }
    int rc = ncclSuccess;
    char path[256];
    if (fromDevice(mine, sendbuff, bytes, stream) != 0)
        rc = ncclUnhandledCudaError;
    const unsigned long n = comm->reductions++;
    snprintf(path, sizeof(path), "%s/%s.red.%lu.%d", comm->dir, comm->token, n, comm->rank);
    if (!rc && putFile(path, mine, bytes) != 0)
        rc = ncclSystemError;
    float *acc = (float *)malloc(bytes ? bytes : 4);
    if (!acc)
        rc = ncclSystemError;
    for (int r = 0; !rc && r < comm->world; ++r)
    {
        snprintf(path, sizeof(path), "%s/%s.red.%lu.%d", comm->dir, comm->token, n, r);
        const int got = getFile(path, other, bytes, 0);
        if (got == 1)
        {
            fprintf(stderr, "loopback_rccl: rank %d waited %.0f s for rank %d in all-reduce %lu: it never joined\n",
                    comm->rank, timeoutSeconds(), r, n);
            rc = ncclSystemError;
        }
        else if (got == 2)
        {
            fprintf(stderr, "loopback_rccl: all-reduce %lu: rank %d's count differs from rank %d's\n", n, r, comm->rank);
            rc = ncclInvalidUsage;
        }
        else if (got != 0)
            rc = ncclSystemError;
        else
            for (size_t i = 0; i < count; ++i)
                acc[i] = r == 0 ? other[i] : (op == LB_SUM ? acc[i] + other[i] : (other[i] > acc[i] ? other[i] : acc[i]));
    }
    if (!rc && toDevice(recvbuff, acc, bytes, stream) != 0)
        rc = ncclUnhandledCudaError;
    free(mine);
    free(other);
    free(acc);
    return rc;
}

const char *ncclGetErrorString(int result)
{
    switch (result)
    {
    case ncclSuccess:
        return "no error";
    case ncclUnhandledCudaError:
        return "loopback: a HIP call failed";
    case ncclSystemError:
        return "loopback: a peer did not show up in time, or the message directory failed";
    case ncclInternalError:
        return "loopback: too many operations in one group";
    case ncclInvalidArgument:
        return "loopback: invalid argument";
    case ncclInvalidUsage:
        return "loopback: counts of a send and its receive differ";
    default:
        return "loopback: unknown error";
    }
}
