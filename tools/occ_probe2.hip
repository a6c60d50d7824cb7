// Runs ON THE GPU BOX (hipcc --offload-arch=gfx950 -O2 tools/occ_probe2.hip -o /tmp/occ2 && /tmp/occ2): workgroups of 128 threads a CU really holds as a
// function of their static LDS size (k_map's shape: 312,500 short workgroups of two waves), resident = grid x spin / kernel time.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES, int BLOCK> __global__ __launch_bounds__(BLOCK) void spin(int *o, long long cycles) {
    __shared__ int s[BYTES / 4];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (s[(threadIdx.x * 7) % (BYTES / 4)] == -1) o[0] = 1;
}
template <int BYTES, int BLOCK> void run(int grid, long long ticks) {
    int *d; hipMalloc(&d, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((spin<BYTES, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, d, ticks); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL((spin<BYTES, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, d, ticks); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double conc = grid * (ticks / 100.0) / (ms * 1e3);
    printf("block %d, LDS %6d B, grid %d, spin %.1f us -> kernel %.1f us => resident workgroups %.0f (%.2f per CU = %.2f waves per SIMD; LDS in use %.0f KB per CU)\n", BLOCK, BYTES, grid,
           ticks / 100.0, ms * 1e3, conc, conc / 256, conc / 256 * (BLOCK / 64) / 4, conc / 256 * BYTES / 1024.0);
    hipFree(d);
}
int main() {
    run<1024, 128>(312500, 1000); run<4096, 128>(312500, 1000); run<8192, 128>(312500, 1000); run<9216, 128>(312500, 1000); run<10240, 128>(312500, 1000);
    run<11264, 128>(312500, 1000); run<12288, 128>(312500, 1000); run<12608, 128>(312500, 1000); run<13608, 128>(312500, 1000); run<14808, 128>(312500, 1000);
    run<16384, 128>(312500, 1000); run<20480, 128>(312500, 1000);
    run<1024, 128>(312500, 300); run<10240, 128>(312500, 300); run<12608, 128>(312500, 300);
    run<1024, 64>(625000, 1000); run<6304, 64>(625000, 1000); run<1024, 256>(156250, 1000); run<25216, 256>(156250, 1000);
    return 0;
}
