// Runs ON THE GPU BOX (hipcc --offload-arch=gfx950 -O2 tools/occ_probe.hip -o /tmp/occ && /tmp/occ): how many workgroups of 256 threads a CU really holds
// at a given LDS size -- a kernel of blocks that spin for a fixed time, resident blocks = grid x spin / kernel time.  MI355X, ROCm 7.2: 1 KB 6.7, 12 KB 7.5,
// 20 KB 6.9, 32 KB 3.4-3.8 per CU (the occupancy API says 5 for 32 KB).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB> __global__ void spin(int *o, long long cycles) {
    __shared__ int s[KB * 256];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (s[(threadIdx.x * 7) % (KB * 256)] == -1) o[0] = 1;
}
template <int KB> void run(int grid, long long ticks, const char *what) {
    int *d; hipMalloc(&d, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(spin<KB>, dim3(grid), dim3(256), 0, 0, d, ticks); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(spin<KB>, dim3(grid), dim3(256), 0, 0, d, ticks); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s: LDS %d KB, grid %d, per-block spin %.1f us -> kernel %.1f us => concurrent blocks %.0f (%.2f per CU)\n", what, KB, grid, ticks / 100.0, ms * 1e3, grid * (ticks / 100.0) / (ms * 1e3), grid * (ticks / 100.0) / (ms * 1e3) / 256);
    hipFree(d);
}
int main() {
    // wall_clock64 ticks at 100 MHz
    run<32>(9660, 800, "k_pairs-like");
    run<1>(9660, 800, "no LDS");
    run<32>(9660, 8000, "long blocks 32KB");
    run<20>(9350, 1700, "k_line-like 20KB");
    run<12>(87200, 500, "cfg-like 12KB");
    run<1>(100000, 100, "tiny blocks");
    return 0;
}
