// Development aid (not part of the product): what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for the renderer's own
// access patterns?  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE is half the bytes of a wide (16 B per lane) streaming read;
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Each kernel
// below moves a KNOWN number of bytes the way k_standardRenderer / k_ambientOcclusion do - one wave per 8 x 8 pixel tile
// of a 3840 x 2160 frame - and tools/fetch_calibration.sh divides that by what the counters say.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calibration tools/fetch_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                                       \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess)                                                                                          \
        {                                                                                                              \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                    \
            exit(1);                                                                                                   \
        }                                                                                                              \
    } while (0)

struct Record
{
    float4 colorInfo, sceneInfo;
};
constexpr int W = 3840, H = 2160, TILES_X = W / 8;

__device__ __forceinline__ int pixelOfLane()
{
    const int tile = blockIdx.x, lane = threadIdx.x;
    const int ty = tile / TILES_X, tx = tile - ty * TILES_X;
    return (ty * 8 + (lane >> 3)) * W + tx * 8 + (lane & 7);
}

// the ids read of a refinement / accumulation pass (renderer_kernel.h: id = ids[index0]): 16 B per lane, 128 B per tile row
__global__ __launch_bounds__(64) void read_ids_int4_per_pixel(const int4 *__restrict__ ids, int *__restrict__ sink)
{
    const int4 v = ids[pixelOfLane()];
    if (v.x + v.y + v.z + v.w == 0x7fffffff)
        sink[0] = 1;
}
// the frame-buffer read of those passes (pp[index].colorInfo, pp[index].sceneInfo): two 16-B loads into a 32-B record
__global__ __launch_bounds__(64) void read_pp_two_float4_per_pixel(const Record *__restrict__ pp, int *__restrict__ sink)
{
    const int p = pixelOfLane();
    const float4 a = pp[p].colorInfo, b = pp[p].sceneInfo;
    if (a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w == 12345.f)
        sink[0] = 1;
}
// the depth k_ambientOcclusion's window is made of (pp[...].colorInfo.w): 4 B of every 32-B record
__global__ __launch_bounds__(64) void read_depth_4_of_32_bytes(const Record *__restrict__ pp, int *__restrict__ sink)
{
    const float d = pp[pixelOfLane()].colorInfo.w;
    if (d == 12345.f)
        sink[0] = 1;
}
// the guide's own case, for reference: 16 B per lane, a wave's 1 KB contiguous
__global__ __launch_bounds__(64) void read_linear_16_bytes_per_lane(const int4 *__restrict__ ids, int *__restrict__ sink)
{
    const int4 v = ids[blockIdx.x * 64 + threadIdx.x];
    if (v.x + v.y + v.z + v.w == 0x7fffffff)
        sink[0] = 1;
}
// the walks' node records (rt_device.h: one s_load_dwordx8 of a 32-B record per node, the index wave-uniform): 64
// dependent scalar loads a wave, at pseudo-random places of a 265 MB table (every load a line nobody else has fetched)
__global__ __launch_bounds__(64) void read_scalar_32_byte_records(const float4 *__restrict__ table, int *__restrict__ sink)
{
    const unsigned records = (unsigned)((size_t)W * H);   // 32-B records in the table
    unsigned at = __builtin_amdgcn_readfirstlane(blockIdx.x * 2654435761u);
    float acc = 0.f;
    for (int i = 0; i < 64; ++i)
    {
        at = (at * 1664525u + 1013904223u);
        const unsigned r = __builtin_amdgcn_readfirstlane(at % records);
        const float4 a = table[2 * (size_t)r], b = table[2 * (size_t)r + 1];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
        at += (unsigned)__builtin_amdgcn_readfirstlane((int)(acc == 12345.f)); // (the next index depends on the load)
    }
    if (acc == 54321.f)
        sink[0] = 1;
}
// the stores of a pass: ids (16 B), the record (2 x 16 B), the RGB image (3 single bytes per lane)
__global__ __launch_bounds__(64) void write_ids_pp_rgb_per_pixel(int4 *__restrict__ ids, Record *__restrict__ pp,
                                                                unsigned char *__restrict__ rgb, float seed)
{
    const int p = pixelOfLane();
    ids[p] = make_int4(p, 1, 2, 3);
    pp[p].colorInfo = make_float4(seed, seed, seed, seed);
    pp[p].sceneInfo = make_float4(seed, seed, seed, seed);
    rgb[3 * p] = (unsigned char)p;
    rgb[3 * p + 1] = (unsigned char)(p >> 8);
    rgb[3 * p + 2] = (unsigned char)(p >> 16);
}

int main()
{
    const size_t pixels = (size_t)W * H;
    int4 *ids;
    Record *pp;
    unsigned char *rgb;
    int *sink;
    char *flush;
    const size_t flushBytes = 768ull << 20; /* three times the Infinity Cache: what a kernel reads comes from HBM */
    CHECK(hipMalloc(&ids, pixels * sizeof(int4)));
    CHECK(hipMalloc(&pp, pixels * sizeof(Record)));
    CHECK(hipMalloc(&rgb, pixels * 3));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&flush, flushBytes));
    CHECK(hipMemset(ids, 1, pixels * sizeof(int4)));
    CHECK(hipMemset(pp, 1, pixels * sizeof(Record)));
    const dim3 grid((unsigned)(pixels / 64)), block(64);
    for (int round = 0; round < 4; ++round)
    {
        CHECK(hipMemset(flush, round, flushBytes));
        hipLaunchKernelGGL(read_ids_int4_per_pixel, grid, block, 0, 0, ids, sink);
        CHECK(hipMemset(flush, round + 1, flushBytes));
        hipLaunchKernelGGL(read_pp_two_float4_per_pixel, grid, block, 0, 0, pp, sink);
        CHECK(hipMemset(flush, round + 2, flushBytes));
        hipLaunchKernelGGL(read_depth_4_of_32_bytes, grid, block, 0, 0, pp, sink);
        CHECK(hipMemset(flush, round + 3, flushBytes));
        hipLaunchKernelGGL(read_linear_16_bytes_per_lane, grid, block, 0, 0, ids, sink);
        CHECK(hipMemset(flush, round + 5, flushBytes));
        hipLaunchKernelGGL(read_scalar_32_byte_records, dim3(32768), block, 0, 0, (const float4 *)pp, sink);
        CHECK(hipMemset(flush, round + 4, flushBytes));
        hipLaunchKernelGGL(write_ids_pp_rgb_per_pixel, grid, block, 0, 0, ids, pp, rgb, (float)round);
        CHECK(hipDeviceSynchronize());
    }
    printf("known_bytes read_ids_int4_per_pixel %zu\n", pixels * 16);
    printf("known_bytes read_pp_two_float4_per_pixel %zu\n", pixels * 32);
    printf("known_bytes read_depth_4_of_32_bytes %zu (useful; the records it touches: %zu)\n", pixels * 4, pixels * 32);
    printf("known_bytes read_linear_16_bytes_per_lane %zu\n", pixels * 16);
    printf("known_bytes write_ids_pp_rgb_per_pixel %zu\n", pixels * 51);
    return 0;
}
