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

// ---------------------------------------------------------------------------------------------------------------------------------
// Memory-side calibration kernels (round 5): access patterns with a KNOWN byte count, so that rocprofv3's FETCH_SIZE / WRITE_SIZE can
// be calibrated on gfx950 for the patterns K_map really uses (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly half of the bytes of a
// wide coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own
// access pattern").  tools/prof_calib.sh runs phz_membench under --pmc passes; bench.py applies the measured factors.
//   stream<W>   : every lane reads W consecutive bytes, a wave 64*W consecutive bytes: bytes read = array size
//   gather<S,P> : one 1-byte load per S-byte unit of the array, always inside the unit's FIRST 32 bytes; P = 1 visits the units in a
//                 pseudo-random permutation (consecutive lanes land megabytes apart: K_map's base / quality bytes under a het SNP seen
//                 from one wave), P = 0 in address order (lane i -> unit i: what the memory system sees over a whole tile).
//                 Requests = array size / S; bytes the DRAM side must deliver = requests * its fetch granule -- the counter per request
//                 IS the calibration (32-byte granule: S = 32 reads the whole array, S = 64 half of it, ...).
//   write<W>    : every lane writes W consecutive bytes (K_map's staging records are 8-byte coalesced stores)
namespace {

template <int W> struct MbWord;
template <> struct MbWord<4> { typedef uint32_t T; };
template <> struct MbWord<8> { typedef uint2 T; };
template <> struct MbWord<16> { typedef uint4 T; };
__device__ inline uint32_t mb_fold(uint32_t x) { return x; }
__device__ inline uint32_t mb_fold(uint2 x) { return x.x ^ x.y; }
__device__ inline uint32_t mb_fold(uint4 x) { return x.x ^ x.y ^ x.z ^ x.w; }

template <int W> __global__ __launch_bounds__(256) void k_mb_stream(const uint8_t *src, size_t n_words, uint32_t *out) {
    typedef typename MbWord<W>::T T;
    const T *p = (const T *)src;
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) acc ^= mb_fold(p[i]);
    if (acc == 0xDEADBEEFu) out[0] = acc;
}

template <int S, int PERM> __global__ __launch_bounds__(256) void k_mb_gather(const uint8_t *src, uint32_t log2_units, uint32_t *out) {
    const uint32_t n_units = 1u << log2_units, mask = n_units - 1;
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_units; i += gridDim.x * 256) {
        uint32_t u = i;
        if (PERM) {                                   // bijection of [0, 2^k): odd multiplier, xor-shift, odd multiplier
            u = (u * 0x9E3779B1u) & mask;
            u ^= u >> (log2_units / 2);
            u = (u * 0x85EBCA6Bu) & mask;
        }
        const uint32_t within = (i * 2654435761u >> 27) & 31u;          // some byte of the unit's first 32
        acc += src[(size_t)u * S + within];
    }
    if (acc == 0xDEADBEEFu) out[0] = acc;
}

template <int W> __global__ __launch_bounds__(256) void k_mb_write(uint8_t *dst, size_t n_words, uint32_t seed) {
    typedef typename MbWord<W>::T T;
    T *p = (T *)dst;
    T v; memset(&v, 0, sizeof v); *(uint32_t *)&v = seed;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) p[i] = v;
}

}  // namespace

// kind: 0 stream 4 B/lane, 1 stream 16 B/lane, 2 gather permuted, 3 gather in address order, 4 write 4 B/lane, 5 write 8 B/lane,
// 6 write 16 B/lane.  log2_bytes: size of the array (28..33); unit: bytes per gathered unit for kinds 2 / 3 (32, 64, 128, 256 or 512),
// ignored otherwise.  Outputs: *known_bytes = bytes the pattern reads (streams: the array; gathers: requests, i.e. one per unit) or
// writes; *requests = loads / stores issued per lane-element; *seconds = HIP-event time of the best of `reps` launches.
extern "C" int phz_membench(phz_ctx *ctx, int kind, int log2_bytes, int unit, int reps, double *seconds, int64_t *known_bytes, int64_t *requests) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || kind < 0 || kind > 6 || log2_bytes < 20 || log2_bytes > 34 || reps < 1) return PHZ_E_ARG;
    if ((kind == 2 || kind == 3) && unit != 32 && unit != 64 && unit != 128 && unit != 256 && unit != 512) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)1 << log2_bytes;
    uint8_t *buf = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess) { (void)hipGetLastError(); return phz_fail(ctx, PHZ_E_NOMEM, "phz_membench: hipMalloc"); }
    struct Free { uint8_t *p; ~Free() { (void)hipFree(p); } } guard{buf};
    if (int s = phz_reserve(ctx, ctx->scalars, 64)) return s;
    uint32_t *out = (uint32_t *)ctx->scalars.p;
    PHZ_HIP(ctx, hipMemsetAsync(buf, 0x5A, bytes, ctx->stream));
    int cus = 0;
    PHZ_HIP(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    const unsigned grid = (unsigned)cus * 32;              // 8 waves per SIMD, grid-stride loops
    double best = 0;
    for (int rep = 0; rep < reps; rep++) {
        PHZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        int lg = 0;
        switch (kind) {
        case 0: hipLaunchKernelGGL(k_mb_stream<4>, dim3(grid), dim3(256), 0, ctx->stream, buf, bytes / 4, out); break;
        case 1: hipLaunchKernelGGL(k_mb_stream<16>, dim3(grid), dim3(256), 0, ctx->stream, buf, bytes / 16, out); break;
        case 2: case 3:
            lg = log2_bytes - (unit == 32 ? 5 : unit == 64 ? 6 : unit == 128 ? 7 : unit == 256 ? 8 : 9);
#define PHZ_MB_G(S) if (kind == 2) hipLaunchKernelGGL((k_mb_gather<S, 1>), dim3(grid), dim3(256), 0, ctx->stream, buf, (uint32_t)lg, out); \
                    else hipLaunchKernelGGL((k_mb_gather<S, 0>), dim3(grid), dim3(256), 0, ctx->stream, buf, (uint32_t)lg, out)
            if (unit == 32) { PHZ_MB_G(32); } else if (unit == 64) { PHZ_MB_G(64); } else if (unit == 128) { PHZ_MB_G(128); }
            else if (unit == 256) { PHZ_MB_G(256); } else { PHZ_MB_G(512); }
#undef PHZ_MB_G
            break;
        case 4: hipLaunchKernelGGL(k_mb_write<4>, dim3(grid), dim3(256), 0, ctx->stream, buf, bytes / 4, (uint32_t)rep); break;
        case 5: hipLaunchKernelGGL(k_mb_write<8>, dim3(grid), dim3(256), 0, ctx->stream, buf, bytes / 8, (uint32_t)rep); break;
        default: hipLaunchKernelGGL(k_mb_write<16>, dim3(grid), dim3(256), 0, ctx->stream, buf, bytes / 16, (uint32_t)rep); break;
        }
        PHZ_HIP(ctx, hipGetLastError());
        PHZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        PHZ_HIP(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0;
        PHZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        if (ms > 0 && (best == 0 || ms * 1e-3 < best)) best = ms * 1e-3;
    }
    const int64_t n_req = (kind == 2 || kind == 3) ? (int64_t)(bytes / (size_t)unit)
                                                   : (int64_t)(bytes / (size_t)(kind == 0 || kind == 4 ? 4 : kind == 5 ? 8 : 16));
    if (seconds) *seconds = best;
    if (known_bytes) *known_bytes = (kind == 2 || kind == 3) ? n_req : (int64_t)bytes;
    if (requests) *requests = n_req;
    return PHZ_OK;
}
