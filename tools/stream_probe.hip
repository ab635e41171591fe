// stream_probe.hip - can a frame's image leave for the host WHILE its kernel still renders, without a launch boundary?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/stream_probe tools/stream_probe.hip && /tmp/stream_probe
// A producer shaped like the renderer (one wave per 8x8 tile of a 1920x1080 frame, tiles in launch order, ~30 us of work,
// RGB bytes stored at the end) and a few resident copier waves on a second stream that watch per-tile flags and move
// each band of 32 rows to mapped host memory as soon as all its tiles have stored.  Questions:
//   1. what a handful of waves gets over PCIe with 16-byte stores (the copier must keep pace: 6.2 MB in 0.25 ms),
//   2. whether device-scope stores + a wait + a device-scope flag are enough for a wave on ANOTHER XCD to read the bytes
//      (every frame's host image is compared with what the producer must have written),
//   3. what the frame costs end to end against kernel + hipMemcpyAsync.
// ... and, further down, the ways that do without a second kernel: the last wave of a tile row copies it; every wave stores
// its tile to the host; the copy engine per band behind hipStreamWaitValue32.  What the engine does in the end - the bands'
// words in page-locked memory, the host watching them - came out of these (profiles/r6/stream_probe.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x)                                                                                         \
    do                                                                                                   \
    {                                                                                                    \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess)                                                                            \
        {                                                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));            \
            exit(1);                                                                                     \
        }                                                                                                \
    } while (0)

static const int W = 1920, H = 1080, TX = W / 8, TY = H / 8, NT = TX * TY;
static const int BAND_TILE_ROWS = 4, NB = (TY + BAND_TILE_ROWS - 1) / BAND_TILE_ROWS;

__host__ __device__ inline unsigned char pixelByte(int x, int y, int c, unsigned serial)
{
    return (unsigned char)((x * 7 + y * 13 + c * 101 + serial * 29) & 255);
}

// mode 0: plain stores, no flag (the renderer as it is); 1: device-scope byte stores, wait, device-scope flag
__global__ __launch_bounds__(64) void k_producer(unsigned char *image, unsigned *flags, unsigned serial, int work, int mode, float *sink,
                                                 long long *stamps)
{
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0)
        stamps[2 * NB + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    if (stamps && blockIdx.x == NT - 1 && threadIdx.x == 0)
        stamps[2 * NB + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    extern __shared__ float lds[];
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int ty = tile / TX, tx = tile - ty * TX;
    const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    float a = (float)lane * 1e-3f + (float)tile * 1e-7f;
    for (int i = 0; i < work; ++i) // dependent chain: ~4 cycles each
        a = __builtin_fmaf(a, 0.999f, 0.001f);
    lds[lane] = a;
    if (a == 123.456f)
        sink[0] = a;
    unsigned char *p = image + ((size_t)y * W + x) * 3;
    if (mode == 0)
    {
        p[0] = pixelByte(x, y, 0, serial);
        p[1] = pixelByte(x, y, 1, serial);
        p[2] = pixelByte(x, y, 2, serial);
    }
    else
    {
        __hip_atomic_store(p + 0, pixelByte(x, y, 0, serial), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, pixelByte(x, y, 1, serial), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 2, pixelByte(x, y, 2, serial), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_store(flags + tile, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One kernel: every wave counts itself into its tile row when its bytes are out; the wave that completes the row's count
// moves the row's 8 x 1920 pixels (45 KB) to the host itself.  No second stream, nobody polls.
__global__ __launch_bounds__(64) void k_producerCopies(unsigned char *image, unsigned *rowCount, unsigned serial, int work, float *sink,
                                                       unsigned char *host, int copy, unsigned nth, long long *rowStamps,
                                                       unsigned **signals, int rowsPerBand, unsigned nth2)
{
    if (rowStamps && blockIdx.x == 0 && threadIdx.x == 0)
        rowStamps[2 * TY] = (long long)__builtin_amdgcn_s_memrealtime();
    extern __shared__ float lds[];
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int ty = tile / TX, tx = tile - ty * TX;
    const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    float a = (float)lane * 1e-3f + (float)tile * 1e-7f;
    for (int i = 0; i < work; ++i)
        a = __builtin_fmaf(a, 0.999f, 0.001f);
    lds[lane] = a;
    if (a == 123.456f)
        sink[0] = a;
    unsigned char *p = image + ((size_t)y * W + x) * 3;
    if (copy == -1) // the renderer as it is: plain stores, nobody counts
    {
        p[0] = pixelByte(x, y, 0, serial);
        p[1] = pixelByte(x, y, 1, serial);
        p[2] = pixelByte(x, y, 2, serial);
        return;
    }
    __hip_atomic_store(p + 0, pixelByte(x, y, 0, serial), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, pixelByte(x, y, 1, serial), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, pixelByte(x, y, 2, serial), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (copy == 2) // every wave sends its own tile: 8 runs of 24 bytes, one dword per lane (48 lanes)
    {
        if (lane < 48)
        {
            const int r = lane / 6, d = lane - 6 * r;
            unsigned word = 0u;
            for (int k = 0; k < 4; ++k)
            {
                const int byte = 4 * d + k;
                word |= (unsigned)pixelByte(tx * 8 + byte / 3, ty * 8 + r, byte % 3, serial) << (8 * k);
            }
            *(unsigned *)(host + ((size_t)(ty * 8 + r) * W + tx * 8) * 3 + 4 * d) = word;
        }
        return;
    }
    if (copy == 3) // ... or byte by byte, as the image leaves the renderer now
    {
        unsigned char *q = host + ((size_t)y * W + x) * 3;
        q[0] = pixelByte(x, y, 0, serial);
        q[1] = pixelByte(x, y, 1, serial);
        q[2] = pixelByte(x, y, 2, serial);
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (copy == 4) // groups of 8 tiles side by side: 8 runs of 192 bytes; the last wave of a group sends them
    {
        unsigned old = 0u;
        if (lane == 0)
            old = __hip_atomic_fetch_add(rowCount + 64 * TY + (ty * (TX / 8) + tx / 8), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old + 1u != nth * 8u)
            return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // 8 rows x 12 uint4 = 96 uint4: lanes 0...63 take the first 64, lanes 0...31 the rest
        for (int i = lane; i < 96; i += 64)
        {
            const int r = i / 12, c = i - 12 * r;
            const size_t at = ((size_t)(ty * 8 + r) * W + (tx / 8) * 64) * 3 + 16 * c;
            *(uint4 *)(host + at) = *(const uint4 *)(image + at);
        }
        return;
    }
    unsigned before = 0u;
    if (lane == 0)
        before = __hip_atomic_fetch_add(rowCount + 64 * ty, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    before = __builtin_amdgcn_readfirstlane(before);
    if (before + 1u != nth * (unsigned)TX || !copy)
        return;
    if (copy == 5) // the copy engine moves the band when its rows are all there: tell the command processor
    {
        // (an atomic add on the word itself would be a PCIe atomic; a second counter on the device instead, and the wave
        // that completes the band's count stores the value the command processor waits for)
        const int band = ty / rowsPerBand;
        const int rows = min(TY, (band + 1) * rowsPerBand) - band * rowsPerBand;
        unsigned rowsBefore = 0u;
        if (lane == 0)
            rowsBefore = __hip_atomic_fetch_add(rowCount + 64 * TY + TY * TX / 8 + 64 * band, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rowsBefore = __builtin_amdgcn_readfirstlane(rowsBefore);
        if (rowsBefore + 1u == nth2 * (unsigned)rows && lane == 0)
            __hip_atomic_store(signals[band], nth2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (rowStamps && lane == 0)
        rowStamps[2 * ty] = (long long)__builtin_amdgcn_s_memrealtime();
    const size_t first = (size_t)ty * 8 * W * 3;
    const uint4 *src = (const uint4 *)(image + first);
    uint4 *dst = (uint4 *)(host + first);
    const int units = 8 * W * 3 / 1024; // 45
    for (int u = 0; u < units; u += 9)
    {
        uint4 v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            v[k] = src[(size_t)(u + k) * 64 + lane];
#pragma unroll
        for (int k = 0; k < 9; ++k)
            dst[(size_t)(u + k) * 64 + lane] = v[k];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (rowStamps && lane == 0)
        rowStamps[2 * ty + 1] = (long long)__builtin_amdgcn_s_memrealtime();
}

// NW waves, resident before the producer starts.  Every wave watches every band in turn and copies its share of it.
__global__ __launch_bounds__(64) void k_copier(const unsigned char *image, const unsigned *flags, unsigned serial,
                                               unsigned char *host, unsigned *gaveUp, long long timeoutTicks,
                                               long long *stamps, int option)
{
    const int lane = threadIdx.x, w = blockIdx.x, NW = gridDim.x;
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int b = 0; b < NB; ++b)
    {
        const int firstTile = b * BAND_TILE_ROWS * TX;
        const int lastTile = min(NT, firstTile + BAND_TILE_ROWS * TX);
        for (;;)
        {
            bool all = true;
            for (int t = firstTile + lane; t < lastTile; t += 64)
                all = all && (__hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == serial);
            if (__all(all))
                break;
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > timeoutTicks)
            {
                if (lane == 0)
                    atomicAdd(gaveUp, 1u);
                return;
            }
            if (option & 1)
                for (int k = 0; k < 16; ++k)
                    __builtin_amdgcn_s_sleep(127);
            else
                __builtin_amdgcn_s_sleep(8);
        }
        if (!(option & 2))
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (stamps && w == 0 && lane == 0)
            stamps[2 * b] = (long long)__builtin_amdgcn_s_memrealtime();
        const size_t first = (size_t)b * BAND_TILE_ROWS * 8 * W * 3;
        const size_t bytes = (size_t)(min(H, (b + 1) * BAND_TILE_ROWS * 8) - b * BAND_TILE_ROWS * 8) * W * 3;
        const uint4 *src = (const uint4 *)(image + first);
        uint4 *dst = (uint4 *)(host + first);
        const int units = (int)(bytes / 1024); // (1920 * 3 * 8 rows is a multiple of 1024)
        for (int u = w; u < units; u += 4 * NW)
        {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (u + k * NW < units)
                    v[k] = src[(size_t)(u + k * NW) * 64 + lane];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (u + k * NW < units)
                    dst[(size_t)(u + k * NW) * 64 + lane] = v[k];
        }
        if (stamps && w == 0 && lane == 0)
            stamps[2 * b + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if (stamps && w == 0 && lane == 0)
        stamps[2 * NB] = t0;
}

__global__ __launch_bounds__(64) void k_copyAll(const unsigned char *image, unsigned char *host, size_t bytes)
{
    const uint4 *src = (const uint4 *)image;
    uint4 *dst = (uint4 *)host;
    const size_t n = bytes / 16;
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n; i += (size_t)gridDim.x * 64 * 4)
    {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + (size_t)k * gridDim.x * 64 < n)
                v[k] = src[i + (size_t)k * gridDim.x * 64];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + (size_t)k * gridDim.x * 64 < n)
                dst[i + (size_t)k * gridDim.x * 64] = v[k];
    }
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

static double now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int work = argc > 1 ? atoi(argv[1]) : 6000;
    const int only = argc > 2 ? atoi(argv[2]) : 0; // 4: only the copy-engine part
    const size_t bytes = (size_t)W * H * 3;
    unsigned char *image, *host;
    unsigned *flags, *gaveUp;
    float *sink;
    long long *stamps;
    CHECK(hipHostMalloc((void **)&stamps, (2 * NB + 4) * sizeof(long long), hipHostMallocDefault));
    CHECK(hipMalloc((void **)&image, bytes));
    CHECK(hipMalloc((void **)&flags, NT * sizeof(unsigned)));
    CHECK(hipMalloc((void **)&sink, 64));
    CHECK(hipHostMalloc((void **)&host, bytes, hipHostMallocDefault));
    CHECK(hipHostMalloc((void **)&gaveUp, 64, hipHostMallocDefault));
    CHECK(hipMemset(flags, 0, NT * sizeof(unsigned)));
    *gaveUp = 0;
    hipStream_t A, B;
    CHECK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    const size_t lds = 10 * 1024; // 16 producer waves per CU, as the renderer has
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    // 1. a few waves over PCIe
    for (int nw : {8, 16, 32, 64, 128, 256})
    {
        std::vector<double> t;
        for (int r = 0; r < 12; ++r)
        {
            CHECK(hipEventRecord(e0, A));
            hipLaunchKernelGGL(k_copyAll, dim3(nw), dim3(64), 0, A, image, host, bytes);
            CHECK(hipEventRecord(e1, A));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2)
                t.push_back(ms);
        }
        printf("copy kernel, %3d waves of 16-byte stores to mapped host memory: %.3f ms for %.1f MB = %.1f GB/s\n", nw, median(t),
               bytes / 1e6, bytes / median(t) / 1e6);
    }
    {
        std::vector<double> t;
        for (int r = 0; r < 12; ++r)
        {
            const double a = now();
            CHECK(hipMemcpyAsync(host, image, bytes, hipMemcpyDeviceToHost, A));
            CHECK(hipStreamSynchronize(A));
            if (r >= 2)
                t.push_back(now() - a);
        }
        printf("hipMemcpyAsync + wait: %.3f ms (host clock) = %.1f GB/s\n", median(t), bytes / median(t) / 1e6);
    }

    // 2. + 3.
    unsigned serial = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (int r = 0; r < 1; ++r)
        {
            std::vector<double> t;
            for (int f = 0; f < 60; ++f)
            {
                ++serial;
                CHECK(hipEventRecord(e0, A));
                hipLaunchKernelGGL(k_producer, dim3(NT), dim3(64), lds, A, image, flags, serial, work, mode, sink, nullptr);
                CHECK(hipEventRecord(e1, A));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (f >= 10)
                    t.push_back(ms);
            }
            printf("producer alone, %s: %.4f ms (HIP events)\n", mode ? "device-scope stores + flag" : "plain stores", median(t));
        }
    {
        std::vector<double> t;
        for (int f = 0; f < 210; ++f)
        {
            ++serial;
            const double a = now();
            hipLaunchKernelGGL(k_producer, dim3(NT), dim3(64), lds, A, image, flags, serial, work, 0, sink, nullptr);
            CHECK(hipMemcpyAsync(host, image, bytes, hipMemcpyDeviceToHost, A));
            CHECK(hipStreamSynchronize(A));
            if (f >= 10)
                t.push_back(now() - a);
        }
        printf("frame = producer + hipMemcpyAsync + wait: %.4f ms (host clock, median of 200)\n", median(t));
    }
    if (only != 4)
    for (int option : {0, 2})
    for (int nw : {32})
    {
        printf("option %d (1: polls 50 us apart; 2: no buffer_inv after a band is seen)\n", option);
        std::vector<double> t, tk, tl;
        long wrong = 0, checked = 0;
        for (int f = 0; f < 260; ++f)
        {
            ++serial;
            const bool verify = f < 60;
            if (verify)
                memset(host, 0xee, bytes);
            const double a = now();
            hipLaunchKernelGGL(k_copier, dim3(nw), dim3(64), 0, B, image, flags, serial, host, gaveUp, 50000000ll, stamps, option);
            CHECK(hipEventRecord(e0, A));
            hipLaunchKernelGGL(k_producer, dim3(NT), dim3(64), lds, A, image, flags, serial, work, 1, sink, stamps);
            CHECK(hipEventRecord(e1, A));
            const double l = now();
            CHECK(hipStreamSynchronize(B));
            const double b = now();
            CHECK(hipStreamSynchronize(A));
            float kms;
            CHECK(hipEventElapsedTime(&kms, e0, e1));
            if (f >= 60)
            {
                tk.push_back(kms);
                tl.push_back(l - a);
            }
            if (f >= 60)
                t.push_back(b - a);
            if (verify)
            {
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x)
                        for (int c = 0; c < 3; ++c)
                            wrong += host[((size_t)y * W + x) * 3 + c] != pixelByte(x, y, c, serial);
                checked += (long)bytes;
            }
        }
        printf("frame = copier (%2d waves, resident first) + producer, wait for the copier: %.4f ms (host clock, median of 200); "
               "%ld wrong bytes of %ld checked over 60 frames; copier gave up %u times\n", nw, median(t), wrong, checked, *gaveUp);
        printf("    the producer in such a frame: %.4f ms (HIP events); the three launches took the host %.4f ms\n", median(tk), median(tl));
        // the last frame's clock: 100 MHz ticks from the copier's start
        const long long c0 = stamps[2 * NB];
        printf("    last frame, us after the copier's start: producer's first wave %.1f, last tile's wave starts %.1f; band seen complete / copied:",
               (stamps[2 * NB + 1] - c0) * 0.01, (stamps[2 * NB + 2] - c0) * 0.01);
        for (int b = 0; b < NB; b += 1)
            printf("  [%d] %.1f / %.1f", b, (stamps[2 * b] - c0) * 0.01, (stamps[2 * b + 1] - c0) * 0.01);
        printf("\n");
    }
    unsigned *rowCount;
    long long *rowStamps;
    CHECK(hipHostMalloc((void **)&rowStamps, (2 * TY + 2) * sizeof(long long), hipHostMallocDefault));
    CHECK(hipMalloc((void **)&rowCount, (TY * 64 + TY * TX / 8 + 64 * 64) * sizeof(unsigned)));
    CHECK(hipMemset(rowCount, 0, (TY * 64 + TY * TX / 8 + 64 * 64) * sizeof(unsigned)));
    unsigned rowSerial = 0, rowFrames = 0;
    for (int copy = -1; copy < 5; ++copy)
    {
        std::vector<double> t, tk;
        long wrong = 0, checked = 0;
        for (int f = 0; f < 260; ++f)
        {
            ++rowSerial;
            const bool verify = copy && f < 60;
            if (verify)
                memset(host, 0xee, bytes);
            const double a = now();
            CHECK(hipEventRecord(e0, A));
            hipLaunchKernelGGL(k_producerCopies, dim3(NT), dim3(64), lds, A, image, rowCount, rowSerial, work, sink, host, copy, copy == 4 ? (unsigned)(f + 1) : (copy == 0 || copy == 1) ? ++rowFrames : 0u, rowStamps, nullptr, 1, 0u);
            CHECK(hipEventRecord(e1, A));
            CHECK(hipStreamSynchronize(A));
            const double b = now();
            float kms;
            CHECK(hipEventElapsedTime(&kms, e0, e1));
            if (f >= 60)
            {
                t.push_back(b - a);
                tk.push_back(kms);
            }
            if (verify)
            {
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < W; ++x)
                        for (int c = 0; c < 3; ++c)
                            wrong += host[((size_t)y * W + x) * 3 + c] != pixelByte(x, y, c, rowSerial);
                checked += (long)bytes;
            }
        }
        printf("frame = ONE kernel, every wave counts itself into its tile row%s: %.4f ms (host clock, median of 200), the kernel %.4f ms "
               "(HIP events); %ld wrong bytes of %ld checked\n", copy == -1 ? " - no: plain stores, nobody counts, nothing copied" : copy == 0 ? " (nothing copied)" : copy == 1 ? ", the last wave of a row copies it to the host" : copy == 2 ? " - no: every wave stores its tile to the host, a dword per lane" : copy == 3 ? " - no: every wave stores its tile to the host, byte by byte" : " - no, into its group of 8 tiles, whose last wave copies 8 x 192 bytes",
               median(t), median(tk), wrong, checked);
        if (copy == 1)
        {
            printf("    last frame, us after the first wave's start: tile row complete / copied:");
            for (int r = 0; r < TY; r += (r < TY - 8 ? 8 : 1))
                printf("  [%d] %.1f / %.1f", r, (rowStamps[2 * r] - rowStamps[2 * TY]) * 0.01, (rowStamps[2 * r + 1] - rowStamps[2 * TY]) * 0.01);
            printf("\n");
        }
    }
    // 4. the copy engine, band by band, each copy behind a wait of the command processor on the band's word
    hipStream_t BS[3];
    BS[0] = B;
    CHECK(hipStreamCreateWithFlags(&BS[1], hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&BS[2], hipStreamNonBlocking));
    int can = 0;
    CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue: %d\n", can);
    if (can)
        for (int copyStreams : {1, 2, 3})
        for (int bands : {3, 4, 6, 8})
        {
            const int rowsPerBand = (TY + bands - 1) / bands;
            const int nb = (TY + rowsPerBand - 1) / rowsPerBand;
            unsigned **signals, **signalsDev;
            CHECK(hipHostMalloc((void **)&signals, nb * sizeof(unsigned *), hipHostMallocDefault));
            for (int b = 0; b < nb; ++b)
            {
                CHECK(hipExtMallocWithFlags((void **)&signals[b], 8, hipMallocSignalMemory));
                CHECK(hipStreamWriteValue32(A, signals[b], 0u, 0));
                CHECK(hipStreamSynchronize(A));
            }
            signalsDev = signals;
            CHECK(hipMemset(rowCount, 0, (TY * 64 + TY * TX / 8 + 64 * 64) * sizeof(unsigned)));
            std::vector<double> t, tk, tq;
            long wrong = 0, checked = 0;
            std::vector<unsigned> done(nb, 0u);
            for (int f = 0; f < 260; ++f)
            {
                ++rowSerial;
                const bool verify = f < 60;
                if (verify)
                    memset(host, 0xee, bytes);
                const double a = now();
                CHECK(hipEventRecord(e0, A));
                hipLaunchKernelGGL(k_producerCopies, dim3(NT), dim3(64), lds, A, image, rowCount, rowSerial, work, sink, host, 5, (unsigned)(f + 1),
                                   nullptr, signalsDev, rowsPerBand, (unsigned)(f + 1));
                CHECK(hipEventRecord(e1, A));
                for (int b = 0; b < nb; ++b)
                {
                    const int rows = std::min(TY, (b + 1) * rowsPerBand) - b * rowsPerBand;
                    hipStream_t Bs = BS[b % copyStreams];
                    CHECK(hipStreamWaitValue32(Bs, signals[b], (unsigned)(f + 1), hipStreamWaitValueGte, 0xffffffffu));
                    const size_t off = (size_t)b * rowsPerBand * 8 * W * 3;
                    CHECK(hipMemcpyAsync(host + off, image + off, (size_t)rows * 8 * W * 3, hipMemcpyDeviceToHost, Bs));
                }
                const double q = now();
                for (int k = 0; k < copyStreams; ++k)
                    CHECK(hipStreamSynchronize(BS[k]));
                const double b2 = now();
                CHECK(hipStreamSynchronize(A));
                float kms;
                CHECK(hipEventElapsedTime(&kms, e0, e1));
                if (f >= 60)
                {
                    t.push_back(b2 - a);
                    tk.push_back(kms);
                    tq.push_back(q - a);
                }
                if (verify)
                {
                    for (int y = 0; y < H; ++y)
                        for (int x = 0; x < W; ++x)
                            for (int c = 0; c < 3; ++c)
                                wrong += host[((size_t)y * W + x) * 3 + c] != pixelByte(x, y, c, rowSerial);
                    checked += (long)bytes;
                }
            }
            printf("frame = ONE kernel + the copy engine in %d bands on %d stream(s), each behind hipStreamWaitValue32: %.4f ms (host clock, median of 200), "
                   "the kernel %.4f ms (HIP events), the host had everything enqueued after %.4f ms; %ld wrong bytes of %ld checked\n",
                   nb, copyStreams, median(t), median(tk), median(tq), wrong, checked);
        }
    return 0;
}
