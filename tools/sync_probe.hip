// Runs ON THE GPU BOX (hipcc --offload-arch=gfx950 -O2 tools/sync_probe.hip -o /tmp/syncp && /tmp/syncp): what one host wait costs -- a tiny kernel and a
// 64-byte copy to page-locked memory, then (a) hipStreamSynchronize, (b) a spin on hipEventQuery, (c) a spin on a flag the copy itself writes.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(unsigned long long *d, unsigned long long v) { if (threadIdx.x == 0) d[0] = v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long *d; hipMalloc(&d, 64);
    volatile unsigned long long *h; hipHostMalloc((void **)&h, 64, hipHostMallocDefault);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int N = 2000;
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            const double t0 = now();
            for (int i = 1; i <= N; i++) {
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (unsigned long long)(i + rep * N + mode * 10 * N));
                hipMemcpyAsync((void *)h, d, 64, hipMemcpyDeviceToHost, s);
                if (mode == 0) hipStreamSynchronize(s);
                else if (mode == 1) { hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) { } }
                else { const unsigned long long want = (unsigned long long)(i + rep * N + mode * 10 * N); while (h[0] != want) { } }
            }
            const double dt = now() - t0;
            if (rep) printf("%s: %.2f us per launch + copy + wait\n", mode == 0 ? "hipStreamSynchronize" : (mode == 1 ? "spin on hipEventQuery" : "spin on the copied word"), dt / N * 1e6);
        }
    }
    return 0;
}
