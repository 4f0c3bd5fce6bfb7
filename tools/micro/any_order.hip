// Micro-benchmark: does hipExtAnyOrderLaunch let independent kernels of one stream overlap on gfx950?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_spin(unsigned *out, int spin)
{
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { }
    if (out && blockIdx.x == 0xffffffffu) out[0] = 1;
}
static float run(int flags, int nk, int wgs, int spin)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipStream_t st; hipStreamCreate(&st);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a, st);
        for (int i = 0; i < 10; ++i) {
            for (int k = 0; k < nk; ++k)
                hipExtLaunchKernelGGL(k_spin, dim3(wgs), dim3(64), 0, st, nullptr, nullptr, k ? flags : 0, (unsigned *)nullptr, spin);
        }
        hipEventRecord(b, st); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000 / 10;
}
int main()
{
    printf("3 kernels x 1024 WGs x 20000 clk, ordered   : %.2f us per group\n", run(0, 3, 1024, 20000));
    printf("3 kernels x 1024 WGs x 20000 clk, any-order : %.2f us per group\n", run(hipExtAnyOrderLaunch, 3, 1024, 20000));
    printf("3 kernels x 1 WG x 0 clk, ordered           : %.2f us per group\n", run(0, 3, 1, 0));
    printf("3 kernels x 1 WG x 0 clk, any-order         : %.2f us per group\n", run(hipExtAnyOrderLaunch, 3, 1, 0));
    return 0;
}
