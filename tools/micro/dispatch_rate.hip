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
// same, with a register budget like k_itx<4,64> (80 VGPRs -> 6 waves / SIMD)
template <int LDS, int SPIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(80))) void k_fat(unsigned *out)
{
    __shared__ unsigned s[LDS / 4 > 0 ? LDS / 4 : 1];
    if (LDS) s[threadIdx.x] = blockIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (SPIN) while (__builtin_readcyclecounter() - t0 < SPIN) { }
    if (out && blockIdx.x == 0xffffffffu) out[0] = LDS ? s[63 - threadIdx.x] : 0;
}
template <int LDS, int SPIN>
static void run_fat(const char *name, int n)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_fat<LDS, SPIN>), dim3(n), dim3(64), 0, 0, nullptr);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_fat<LDS, SPIN>), dim3(n), dim3(64), 0, 0, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s n=%d  %.2f us/launch  %.1f WG/us\n", name, n, ms * 1000 / 20, n / (ms * 1000 / 20));
}
int main()
{
    run<0, 0>("empty", 39000);
    run<1536, 7700>("spin7700 lds1536", 14447);
    run_fat<1536, 7700>("fat spin7700 lds1536", 14447);
    run_fat<1536, 3000>("fat spin3000 lds1536", 14447);
    run_fat<1536, 0>("fat spin0 lds1536", 14447);
    run<0, 0>("empty 14447", 14447);
    run<0, 0>("empty 1886", 1886);
    run<0, 0>("empty 1", 1);
    return 0;
}
