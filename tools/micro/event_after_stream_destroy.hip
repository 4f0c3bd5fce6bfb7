// Does an event outlive the stream it was last recorded on?  (ROCm 7.2 / MI355X: hipEventSynchronize on such an event returned
// "operation not permitted when stream is capturing" in bench.py once a stream driver had been destroyed -- the reason the
// library keeps its streams in a pool instead of destroying them, ovvc_engine.hip.)
// hipcc --offload-arch=gfx950 event_after_stream_destroy.hip -o /tmp/event_after_stream_destroy && /tmp/event_after_stream_destroy
#include <hip/hip_runtime.h>
#include <stdio.h>
int main()
{
    hipStream_t a, b[8];
    hipEvent_t e;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    void *d; hipMalloc(&d, 1 << 20);
    hipMemsetAsync(d, 0, 1 << 20, a);
    hipEventRecord(e, a);
    printf("sync while the stream lives: %s\n", hipGetErrorString(hipEventSynchronize(e)));
    hipStreamDestroy(a);
    printf("sync after hipStreamDestroy: %s\n", hipGetErrorString(hipEventSynchronize(e)));
    for (int i = 0; i < 8; ++i) hipStreamCreateWithFlags(&b[i], hipStreamNonBlocking);
    printf("sync after 8 new streams:    %s\n", hipGetErrorString(hipEventSynchronize(e)));
    hipMemsetAsync(d, 0, 1 << 20, b[3]);
    printf("re-record on a live stream:  %s\n", hipGetErrorString(hipEventRecord(e, b[3])));
    printf("sync after re-record:        %s\n", hipGetErrorString(hipEventSynchronize(e)));
    return 0;
}
