// Development aid (not part of the product): what does fetching a 32-byte node record cost a wave when the next
// address depends on the record - through the scalar cache (s_load_dwordx8, what the walk does), from LDS
// (2 x ds_read_b128 of a staged copy, what the north star proposed for scenes that fit), and through the
// scalar cache from a list too large for it (L2)?  Dependent chains: cycles per hop = the latency a walk
// cannot hide inside one wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/node_fetch_bench tools/node_fetch_bench.hip && /tmp/node_fetch_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

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

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t; /* volatile: stays where it is written, between the other volatile statements */
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

struct Node
{
    float lo[3], hiz, hix, hiy;
    int count, next; /* next: index of the node to visit after this one */
};

__global__ void __launch_bounds__(64) k_scalar(const Node *nodes, int hops, unsigned long long *cycles, int *sink)
{
    int cur = 0;
    const unsigned long long t0 = now();
    for (int h = 0; h < hops; ++h)
    {
        int next;
        asm volatile("s_lshl_b32 s20, %1, 5\n"
                     "s_load_dwordx8 s[8:15], %2, s20\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     "s_mov_b32 %0, s15\n"
                     : "=s"(next)
                     : "s"(cur), "s"(nodes)
                     : "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s20", "scc");
        cur = __builtin_amdgcn_readfirstlane(next);
    }
    const unsigned long long t1 = now();
    if (threadIdx.x == 0)
    {
        atomicAdd(cycles, t1 - t0);
        *sink = cur;
    }
}

__global__ void __launch_bounds__(64) k_lds(const Node *nodes, int n, int hops, unsigned long long *cycles, int *sink)
{
    extern __shared__ float4 staged[]; /* 2 float4 per node */
    for (int i = threadIdx.x; i < 2 * n; i += 64)
        staged[i] = ((const float4 *)nodes)[i];
    __syncthreads();
    int cur = 0;
    const unsigned long long t0 = now();
    for (int h = 0; h < hops; ++h)
    {
        /* every lane reads the same two rows (a broadcast read), as a staged walk would */
        const float4 a = staged[2 * cur], b = staged[2 * cur + 1];
        asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z));
        cur = __builtin_amdgcn_readfirstlane(__float_as_int(b.w));
    }
    const unsigned long long t1 = now();
    if (threadIdx.x == 0)
    {
        atomicAdd(cycles, t1 - t0);
        *sink = cur;
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs; shader-clock cycles per dependent fetch of a 32-byte node record\n", prop.gcnArchName, prop.multiProcessorCount);
    unsigned long long *cycles;
    int *sink;
    CHECK(hipMalloc(&cycles, 8));
    CHECK(hipMalloc(&sink, 4));
    const int hops = 20000;
    for (int n : {35, 2000, 500000})
    {
        std::vector<Node> nodes(n);
        std::vector<int> order(n);
        std::iota(order.begin(), order.end(), 0);
        std::mt19937 rng(7);
        std::shuffle(order.begin() + 1, order.end(), rng); /* one cycle through all nodes, in random order */
        for (int i = 0; i < n; ++i)
        {
            Node &nd = nodes[order[i]];
            nd.lo[0] = nd.lo[1] = nd.lo[2] = nd.hiz = nd.hix = nd.hiy = 1.f;
            nd.count = 1;
            nd.next = order[(i + 1) % n];
        }
        Node *d;
        CHECK(hipMalloc(&d, n * sizeof(Node)));
        CHECK(hipMemcpy(d, nodes.data(), n * sizeof(Node), hipMemcpyHostToDevice));
        for (int wavesPerCu : {1, 16})
        {
            const int blocks = wavesPerCu == 1 ? 1 : prop.multiProcessorCount * wavesPerCu;
            for (int rep = 0; rep < 2; ++rep)
            {
                CHECK(hipMemset(cycles, 0, 8));
                hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(64), 0, 0, d, hops, cycles, sink);
                CHECK(hipDeviceSynchronize());
            }
            unsigned long long c;
            CHECK(hipMemcpy(&c, cycles, 8, hipMemcpyDeviceToHost));
            printf("s_load_dwordx8   %7d nodes (%8.1f KB)  %2d wave(s) per CU: %7.1f cycles per hop\n", n, n * 32 / 1024.0,
                   wavesPerCu, (double)c / blocks / hops);
            if ((size_t)n * 32 <= 8192)
            {
                for (int rep = 0; rep < 2; ++rep)
                {
                    CHECK(hipMemset(cycles, 0, 8));
                    hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(64), n * 32, 0, d, n, hops, cycles, sink);
                    CHECK(hipDeviceSynchronize());
                }
                CHECK(hipMemcpy(&c, cycles, 8, hipMemcpyDeviceToHost));
                printf("2 x ds_read_b128 %7d nodes (%8.1f KB)  %2d wave(s) per CU: %7.1f cycles per hop\n", n, n * 32 / 1024.0,
                       wavesPerCu, (double)c / blocks / hops);
            }
        }
        CHECK(hipFree(d));
    }
    return 0;
}
