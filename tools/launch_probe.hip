// GPU-box probe: what does the HOST pay per kernel launch on this runtime?  N back-to-back launches of an empty kernel on one stream -- with an 8-byte argument, with a
// 400-byte struct argument (the row stage passes its table of pointers by value), with a hipMemsetAsync between launches -- host time to enqueue them all, and the time
// until the stream is idle.  A stage whose kernels run 5-30 us each is bound by whichever is larger.   build: hipcc --offload-arch=gfx950 -O2 tools/launch_probe.hip -o /tmp/launch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { const void *p[50]; };
__global__ void k_small(int *x) { if (x && threadIdx.x == 9999) *x = 1; }
__global__ void k_big(Big b) { if (b.p[0] && threadIdx.x == 9999) *(int *)b.p[0] = 1; }
__global__ void k_work(int *x, int n) { int s = 0; for (int i = 0; i < n; i++) s += i * threadIdx.x; if (s == 123456789) *x = s; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *d; hipMalloc(&d, 1 << 20);
    Big b; for (auto &q : b.p) q = nullptr;
    const int N = 2000;
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipStreamSynchronize(s);
            const double t0 = now();
            for (int i = 0; i < N; i++) {
                if (mode == 0) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, (int *)nullptr);
                else if (mode == 1) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b);
                else if (mode == 2) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, (int *)nullptr); hipMemsetAsync(d, 0, 64, s); }
                else if (mode == 3) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, d, 2000);          // ~10 us of GPU work each
                else hipLaunchKernelGGL(k_big, dim3(1024), dim3(256), 0, s, b);
            }
            const double t1 = now();
            hipStreamSynchronize(s);
            const double t2 = now();
            if (rep == 2) printf("%-58s host enqueue %6.2f us per launch, stream idle after %6.2f us per launch\n",
                                 mode == 0 ? "empty kernel, 8-byte argument" : mode == 1 ? "empty kernel, 400-byte struct argument" : mode == 2 ? "empty kernel + hipMemsetAsync(64 B) (per pair)" :
                                 mode == 3 ? "256 x 256 threads, ~10 us of work, 12-byte arguments" : "1024 x 256 threads, empty, 400-byte struct argument", (t1 - t0) / N, (t2 - t0) / N);
        }
    }
    return 0;
}
