// Issue-rate microbenchmarks for gfx950 (MI355X): how many wave64 vector-ALU, scalar-ALU and LDS instructions the chip retires per
// second.  K_map is integer / branch work whose HBM traffic is far below the memory roofline; these rates are the ceiling its
// instruction stream is measured against (bench.py: roofline.issue).  Each kernel runs a long unrolled block of independent
// instructions of ONE kind (inline assembly, so the compiler cannot fold them), `waves_per_simd` waves on every SIMD of the chip,
// timed with HIP events; the result is wave-instructions per second over the whole chip.
#include "phz_internal.h"

namespace {

constexpr int UB_UNROLL = 64;

// 64 independent-enough VALU instructions per iteration over eight accumulators (v_add_u32: one issue slot each)
__global__ __launch_bounds__(256) void k_ub_valu(int iters, uint32_t *out) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t y = blockIdx.x | 1u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UB_UNROLL / 8; u++) {
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0xDEADBEEFu) out[0] = 1;
}
// the same with scalar adds (s_add_u32: the scalar unit the walk's control flow runs on)
__global__ __launch_bounds__(256) void k_ub_salu(int iters, uint32_t *out) {
    uint32_t a0 = blockIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t y = blockIdx.x | 1u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UB_UNROLL / 8; u++) {
            asm volatile("s_add_u32 %0, %0, %8\n s_add_u32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_add_u32 %3, %3, %8\n"
                         "s_add_u32 %4, %4, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_add_u32 %7, %7, %8\n"
                         : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4), "+s"(a5), "+s"(a6), "+s"(a7) : "s"(y) : "scc");
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0xDEADBEEFu) out[0] = 1;
}
// LDS: ds_read_b32 of consecutive dwords (no bank conflicts), eight in flight
__global__ __launch_bounds__(256) void k_ub_lds(int iters, uint32_t *out) {
    __shared__ uint32_t s[2048];
    for (int j = threadIdx.x; j < 2048; j += 256) s[j] = (uint32_t)j;
    __syncthreads();
    uint32_t acc = 0;
    const uint32_t addr = (uint32_t)(size_t)(&s[threadIdx.x]);         // low word of the generic address of an LDS object = its LDS byte offset
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UB_UNROLL / 8; u++) {
            uint32_t b0, b1, b2, b3, b4, b5, b6, b7;
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:1024\n ds_read_b32 %2, %8 offset:2048\n ds_read_b32 %3, %8 offset:3072\n"
                         "ds_read_b32 %4, %8 offset:4096\n ds_read_b32 %5, %8 offset:5120\n ds_read_b32 %6, %8 offset:6144\n ds_read_b32 %7, %8 offset:7168\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3), "=v"(b4), "=v"(b5), "=v"(b6), "=v"(b7) : "v"(addr) : "memory");
            acc += b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
        }
    }
    if (acc == 0xDEADBEEFu) out[0] = 1;
}

}  // namespace

// kind 0 VALU (v_add_u32), 1 SALU (s_add_u32), 2 LDS (ds_read_b32).  One launch fills every SIMD of the device with waves_per_simd waves
// (workgroups of 256 threads = one wave per SIMD of a CU); *wave_insts_per_s = measured wave-instructions per second over the chip,
// *n_cu = compute units of the device, *clock_mhz = its reported engine clock.
extern "C" int phz_microbench(phz_ctx *ctx, int kind, int waves_per_simd, int iters, double *wave_insts_per_s, int *n_cu, int *clock_mhz) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || kind < 0 || kind > 2 || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || !wave_insts_per_s) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    int cus = 0, mhz = 0;
    PHZ_HIP(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    PHZ_HIP(ctx, hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, ctx->device));
    if (n_cu) *n_cu = cus;
    if (clock_mhz) *clock_mhz = mhz / 1000;
    if (int s = phz_reserve(ctx, ctx->scalars, 64)) return s;
    uint32_t *out = (uint32_t *)ctx->scalars.p;
    const unsigned grid = (unsigned)(cus * waves_per_simd);
    double best = 0;
    for (int rep = 0; rep < 4; rep++) {          // the first repetition warms the clocks up; the best of the rest is reported
        PHZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        if (kind == 0) hipLaunchKernelGGL(k_ub_valu, dim3(grid), dim3(256), 0, ctx->stream, iters, out);
        else if (kind == 1) hipLaunchKernelGGL(k_ub_salu, dim3(grid), dim3(256), 0, ctx->stream, iters, out);
        else hipLaunchKernelGGL(k_ub_lds, dim3(grid), dim3(256), 0, ctx->stream, iters, out);
        PHZ_HIP(ctx, hipGetLastError());
        PHZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        PHZ_HIP(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0;
        PHZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        const double insts = (double)grid * 4.0 * (double)iters * UB_UNROLL;       // 4 waves per workgroup
        if (rep > 0 && ms > 0) best = std::max(best, insts / (ms * 1e-3));
    }
    *wave_insts_per_s = best;
    return PHZ_OK;
}
