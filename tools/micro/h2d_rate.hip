// Aggregate host-to-device rate of T threads, each copying `bytes` per "picture" as `parts` asynchronous copies on its own stream
// out of page-locked memory, then waiting for them (what a frame thread's flush does with its recorder arrays):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_rate tools/micro/h2d_rate.hip -lpthread && /tmp/h2d_rate 16 8738592 10 200
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static size_t g_bytes; static int g_parts, g_iters, g_kernel, g_wgs = 64;
struct Parts { const uint4 *src[16]; uint4 *dst[16]; unsigned n16[16]; int n; };
// one launch copies every part: the GPU reads the page-locked host memory itself (no DMA engine, no per-copy set-up)
__global__ __launch_bounds__(256) void k_copy(Parts p)
{
    for (int k = 0; k < p.n; ++k)
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < p.n16[k]; i += gridDim.x * 256) p.dst[k][i] = p.src[k][i];
}
static pthread_barrier_t g_bar;
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void *worker(void *)
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    char *h, *d; hipHostMalloc((void **)&h, g_bytes, hipHostMallocDefault); hipMalloc((void **)&d, g_bytes);
    for (size_t i = 0; i < g_bytes; i += 4096) h[i] = (char)i;
    const size_t part = g_bytes / g_parts;
    pthread_barrier_wait(&g_bar);
    Parts P; P.n = g_parts < 16 ? g_parts : 16;
    for (int p = 0; p < P.n; ++p) { P.src[p] = (const uint4 *)(h + p * (g_bytes / P.n)); P.dst[p] = (uint4 *)(d + p * (g_bytes / P.n)); P.n16[p] = (unsigned)(g_bytes / P.n / 16); }
    for (int it = 0; it < g_iters; ++it) {
        if (g_kernel) hipLaunchKernelGGL(k_copy, dim3(g_wgs), dim3(256), 0, s, P);
        else for (int p = 0; p < g_parts; ++p) hipMemcpyAsync(d + p * part, h + p * part, part, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
    }
    pthread_barrier_wait(&g_bar);
    return NULL;
}
int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 16;
    g_bytes = argc > 2 ? (size_t)atoll(argv[2]) : 8738592; g_parts = argc > 3 ? atoi(argv[3]) : 10; g_iters = argc > 4 ? atoi(argv[4]) : 200;
    g_kernel = argc > 5 ? atoi(argv[5]) : 0; if (argc > 6) g_wgs = atoi(argv[6]);
    pthread_barrier_init(&g_bar, NULL, T + 1);
    pthread_t th[256];
    for (int i = 0; i < T; ++i) pthread_create(&th[i], NULL, worker, NULL);
    pthread_barrier_wait(&g_bar);
    const double t0 = now();
    pthread_barrier_wait(&g_bar);
    const double dt = now() - t0;
    printf("%s threads %d, %zu bytes per picture in %d copies: %.1f GB/s, %.0f pictures/s\n", g_kernel ? "copy kernel," : "DMA,", T, g_bytes, g_parts, T * (double)g_iters * g_bytes / dt / 1e9, T * (double)g_iters / dt);
    for (int i = 0; i < T; ++i) pthread_join(th[i], NULL);
    return 0;
}
