// Development aid (not part of the product): what does one SIMD of gfx950 sustain in vector, scalar and
// mixed instruction issue at 1, 2, 4 and 8 resident waves?  The figures price `roofline.valu_issue` in
// bench.py (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32; one wave
// alone sustains one per 4).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue_bench tools/valu_issue_bench.hip && /tmp/valu_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

// 16 independent accumulators, 64 vector instructions per loop trip
#define V16(OP)                                                                                                        \
    OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7) OP(b0) OP(b1) OP(b2) OP(b3) OP(b4) OP(b5) OP(b6) OP(b7)

template <int MODE>
__global__ void __launch_bounds__(64) k_issue(float *out, int trips, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6,
          a7 = seed + 7, b0 = seed + 8, b1 = seed + 9, b2 = seed + 10, b3 = seed + 11, b4 = seed + 12, b5 = seed + 13,
          b6 = seed + 14, b7 = seed + 15;
    const float m = 1.0001f + threadIdx.x * 1e-7f, c = 0.5f;
    int s0 = trips, s1 = 1, s2 = 2, s3 = 3;
    for (int t = 0; t < trips; ++t)
    {
        if (MODE == 0) // vector only: 64 v_fma_f32
        {
#define FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(c));
            V16(FMA) V16(FMA) V16(FMA) V16(FMA)
        }
        else if (MODE == 1) // scalar only: 64 s_add_i32 on four chains
        {
#define SADD4 asm volatile("s_add_i32 %0, %0, %4\n s_add_i32 %1, %1, %4\n s_add_i32 %2, %2, %4\n s_add_i32 %3, %3, %4" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(t));
            SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4 SADD4
        }
        else if (MODE == 2) // one scalar per two vector, interleaved: 64 v_fma_f32 + 32 s_add_i32
        {
#define FMA2S(r, q) asm volatile("v_fma_f32 %0, %0, %3, %4\n s_add_i32 %2, %2, %5\n v_fma_f32 %1, %1, %3, %4" : "+v"(r), "+v"(q), "+s"(s0) : "v"(m), "v"(c), "s"(t));
#define ROUND FMA2S(a0, a1) FMA2S(a2, a3) FMA2S(a4, a5) FMA2S(a6, a7) FMA2S(b0, b1) FMA2S(b2, b3) FMA2S(b4, b5) FMA2S(b6, b7)
            ROUND ROUND ROUND ROUND
        }
        else if (MODE == 3) // packed: 64 v_pk_mul_f32 (two floats each)
        {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {b0, b1}, p5 = {b2, b3}, p6 = {b4, b5},
               p7 = {b6, b7}, mm = {m, m};
#define PK(r) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r) : "v"(mm));
#define PK8 PK(p0) PK(p1) PK(p2) PK(p3) PK(p4) PK(p5) PK(p6) PK(p7)
            PK8 PK8 PK8 PK8 PK8 PK8 PK8 PK8
            a0 = p0.x, a1 = p0.y, a2 = p1.x, a3 = p1.y, a4 = p2.x, a5 = p2.y, a6 = p3.x, a7 = p3.y;
            b0 = p4.x, b1 = p4.y, b2 = p5.x, b3 = p5.y, b4 = p6.x, b5 = p6.y, b6 = p7.x, b7 = p7.y;
        }
        else if (MODE == 4) // one scalar per vector: 64 + 64
        {
#define FMA1S(r) asm volatile("v_fma_f32 %0, %0, %2, %3\n s_add_i32 %1, %1, %4" : "+v"(r), "+s"(s0) : "v"(m), "v"(c), "s"(t));
            V16(FMA1S) V16(FMA1S) V16(FMA1S) V16(FMA1S)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + (float)(s0 + s1 + s2 + s3);
    if (r == 12345.678f)
        out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int MODE>
static void run(const char *name, int vectorPerTrip, int scalarPerTrip, float *out, int cus)
{
    const int trips = 20000;
    for (int wavesPerSimd : {1, 2, 4, 8})
    {
        // one-wave workgroups; the dispatcher spreads them over CUs and SIMDs round-robin
        const int grid = cus * 4 * wavesPerSimd;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_issue<MODE>, dim3(grid), dim3(64), 0, 0, out, trips, 1.f);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep)
        {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_issue<MODE>, dim3(grid), dim3(64), 0, 0, out, trips, 1.f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best)
                best = ms;
        }
        const double sec = best * 1e-3;
        const double v = (double)grid * trips * vectorPerTrip / sec, s = (double)grid * trips * scalarPerTrip / sec;
        const double simds = cus * 4.0;
        printf("%-28s waves/SIMD %d  %8.3f ms  vector %7.1f G wave-inst/s (%.2f cycles per inst per SIMD at 2.4 GHz)"
               "  scalar %7.1f G/s (%.2f cycles per inst per CU)\n",
               name, wavesPerSimd, best, v * 1e-9, v > 0 ? 2.4e9 * simds / v : 0.0, s * 1e-9,
               s > 0 ? 2.4e9 * cus / s : 0.0);
    }
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
    float *out;
    CHECK(hipMalloc(&out, 1 << 24));
    run<0>("v_fma_f32 only", 64, 0, out, cus);
    run<3>("v_pk_mul_f32 only", 64, 0, out, cus);
    run<1>("s_add_i32 only", 0, 64, out, cus);
    run<2>("2 v_fma_f32 : 1 s_add_i32", 64, 32, out, cus);
    run<4>("1 v_fma_f32 : 1 s_add_i32", 64, 64, out, cus);
    return 0;
}
