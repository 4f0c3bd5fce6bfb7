// Micro-benchmark: how fast can gfx950 launch single-wave workgroups?  (tools only, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS, int SPIN>
__global__ __launch_bounds__(64) void k_empty(unsigned *out)
{
    __shared__ unsigned s[LDS / 4 > 0 ? LDS / 4 : 1];
    if (LDS) s[threadIdx.x] = blockIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (SPIN) while (__builtin_readcyclecounter() - t0 < SPIN) { }
    if (out && blockIdx.x == 0xffffffffu) out[0] = LDS ? s[63 - threadIdx.x] : 0;
}
template <int LDS, int SPIN>
static void run(const char *name, int n)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_empty<LDS, SPIN>), dim3(n), dim3(64), 0, 0, nullptr);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_empty<LDS, SPIN>), dim3(n), dim3(64), 0, 0, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s n=%d  %.2f us/launch  %.1f WG/us\n", name, n, ms * 1000 / 20, n / (ms * 1000 / 20));
}
int main()
{
    run<0, 0>("empty", 39000);
    run<0, 14000>("spin14000 lds0", 39000);
    run<4096, 14000>("spin14000 lds4096", 39000);
    run<5120, 14000>("spin14000 lds5120", 39000);
    run<5632, 14000>("spin14000 lds5632", 39000);
    run<6144, 14000>("spin14000 lds6144", 39000);
    run<6400, 14000>("spin14000 lds6400", 39000);
    run<6608, 14000>("spin14000 lds6608", 39000);
    run<7168, 14000>("spin14000 lds7168", 39000);
    run<8192, 14000>("spin14000 lds8192", 39000);
    run<10240, 14000>("spin14000 lds10240", 39000);
    return 0;
}
