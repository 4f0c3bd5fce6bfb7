// How long does a one-sample hand-over from workgroup to workgroup take, by cache scope and XCD placement?  (The ordered pass of
// kernels_intra.hip is a chain of such hand-overs: ~2100 per 4K I picture.)
// A chain of single-wave workgroups: link i polls a word written by link i - 1 until it carries the tag, adds one, stores it
// write-through.  Variants: links on consecutive workgroups (= round-robin over the 8 XCDs) or all on one XCD (every 8th
// workgroup), polled with agent-scope (sc1: memory side) or workgroup-scope (sc0: the XCD's own L2) loads.
// hipcc --offload-arch=gfx950 -O3 xcd_handover.hip -o /tmp/xcd_handover && /tmp/xcd_handover
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define TAG 0x80000000u

template <int SCOPE_LOCAL, int WORK>
__global__ __launch_bounds__(64) void k_chain(unsigned *words, unsigned *xcc, int stride, int n_links, int spread)
{
    const int wg = blockIdx.x;
    if (wg % stride) return;                       // only every stride-th workgroup is a link
    const int link = wg / stride;
    if (link >= n_links) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[link] = id & 0xf;
    unsigned v = 0;
    if (link > 0) {
        unsigned *src = words + (size_t)(link - 1) * spread;
        unsigned spins = 0;
        for (;;) {
            v = SCOPE_LOCAL ? __hip_atomic_load(src + (threadIdx.x & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                            : __hip_atomic_load(src + (threadIdx.x & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v & TAG)) break;
            if (++spins > (1u << 20)) { v = TAG | 0x7fffff; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    // stand-in for the prediction: WORK dependent multiply-adds
    unsigned acc = v & ~TAG;
#pragma unroll 1
    for (int i = 0; i < WORK; ++i) acc = acc * 3u + 1u - 2u * acc - 1u;      // == acc, but the compiler keeps the chain
    acc += 1;
    unsigned *dst = words + (size_t)link * spread + (threadIdx.x & 15);
    asm volatile("global_store_dword %0, %1, off sc1" :: "v"(dst), "v"(acc | TAG) : "memory");
}

template <int L, int W> float run(unsigned *d_words, unsigned *d_xcc, int stride, int n_links, int spread, int reps, unsigned *h_last)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        hipMemset(d_words, 0, (size_t)n_links * spread * 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k_chain<L, W>), dim3(n_links * stride), dim3(64), 0, 0, d_words, d_xcc, stride, n_links, spread);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipMemcpy(h_last, d_words + (size_t)(n_links - 1) * spread, 4, hipMemcpyDeviceToHost);
    return best;
}

int main()
{
    const int N = 2048, SPREAD = 64;            // links, words between two links' cells (a cache line each)
    unsigned *d_words, *d_xcc, h_xcc[2048], last;
    hipMalloc(&d_words, (size_t)N * SPREAD * 4); hipMalloc(&d_xcc, N * 4);
#define RUN(L, W, stride, what) { float ms = run<L, W>(d_words, d_xcc, stride, N, SPREAD, 5, &last); \
        hipMemcpy(h_xcc, d_xcc, N * 4, hipMemcpyDeviceToHost); int same = 1; for (int i = 1; i < N; ++i) same &= h_xcc[i] == h_xcc[0]; \
        printf("%-66s %7.3f us per link  (chain %s, last = %u%s; XCC of links 0..9: %u %u %u %u %u %u %u %u %u %u%s)\n", what, ms * 1e3 / N, \
               (last & ~TAG) == (unsigned)N ? "complete" : "BROKEN", last & ~TAG, (last & ~TAG) == 0x7fffff ? " (gave up)" : "", \
               h_xcc[0], h_xcc[1], h_xcc[2], h_xcc[3], h_xcc[4], h_xcc[5], h_xcc[6], h_xcc[7], h_xcc[8], h_xcc[9], same ? ", all links on one XCD" : ""); }
    RUN(0, 0, 1, "consecutive workgroups (round robin over XCDs), agent-scope polls");
    RUN(0, 0, 8, "every 8th workgroup (one XCD), agent-scope polls");
    RUN(1, 0, 8, "every 8th workgroup (one XCD), workgroup-scope polls (L2)");
    RUN(1, 0, 1, "consecutive workgroups, workgroup-scope polls (expected: stale, gives up)");
    RUN(0, 200, 1, "consecutive, agent-scope, + 200 dependent instructions per link");
    RUN(1, 200, 8, "one XCD, workgroup-scope, + 200 dependent instructions per link");
    return 0;
}
