// what does launching a wave cost?  32400 one-wave workgroups of W iterations each against 4096 workgroups (as many as are
// resident: 256 CUs x 16) of W * 32400 / 4096 iterations each - the same work, one launch per wave slot instead of eight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k_work(float *out, int work)
{
    extern __shared__ float lds[];
    float a = (float)threadIdx.x * 1e-3f + (float)blockIdx.x * 1e-7f, b = a + 1.f, c = a + 2.f, d = a + 3.f;
    for (int i = 0; i < work; ++i)
    {
        a = __builtin_fmaf(a, 0.999f, 0.001f);
        b = __builtin_fmaf(b, 0.998f, 0.002f);
        c = __builtin_fmaf(c, 0.997f, 0.003f);
        d = __builtin_fmaf(d, 0.996f, 0.004f);
    }
    lds[threadIdx.x] = a + b + c + d;
    if (lds[threadIdx.x] == 123.456f)
        out[0] = a;
    out[1 + blockIdx.x * 64 + threadIdx.x] = lds[threadIdx.x];
}
int main()
{
    float *out;
    CHECK(hipMalloc((void **)&out, (1 + 32400 * 64) * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int work : {500, 1000, 2000})
        for (int grid : {32400, 4096, 32400, 4096})
        {
            const int w = grid == 32400 ? work : (int)((long)work * 32400 / 4096);
            std::vector<float> t;
            for (int r = 0; r < 30; ++r)
            {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_work, dim3(grid), dim3(64), 10 * 1024, 0, out, w);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 5)
                    t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf("%5d workgroups x %6d iterations: %.4f ms (median of 25)\n", grid, w, t[t.size() / 2]);
        }
    return 0;
}
