// Device-wide exclusive scan of uint32 counts (hand-written; three passes: chunk sums, one-block scan of the sums, chunk-local scan
// + base).  Header-only so that every translation unit that needs it gets its own copy of the kernels.
#pragma once
#include "phz_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------ exclusive scan (uint32)
// out[i] = sum(in[0..i)), out[n] = total.  Three passes: chunk sums, one-block scan of the sums, chunk-local scan + base.
constexpr int SCAN_ITEMS = 16, SCAN_CHUNK = SCAN_ITEMS * 256;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
    return x;
}

__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t *in, int64_t n, uint32_t *partial) {
    __shared__ uint32_t s[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { const int64_t i = base + k * 256 + threadIdx.x; if (i < n) x += in[i]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(1024) void k_scan_partials(uint32_t *partial, int64_t nb, uint32_t *total) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
        const int64_t i = b0 + tid;
        const uint32_t v = i < nb ? partial[i] : 0;
        const uint32_t x = wave_incl_scan(v, lane);
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t before = s_carry;
        for (int w2 = 0; w2 < wave; w2++) before += s_w[w2];
        if (i < nb) partial[i] = before + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + x;
        __syncthreads();
    }
    if (tid == 0) *total = s_carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *in, uint32_t *out, int64_t n, const uint32_t *partial) {
    __shared__ uint32_t s_v[SCAN_CHUNK + SCAN_CHUNK / 16];      // padded: item i at i + i/16 (thread t's 16 items: no bank conflicts)
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int j = k * 256 + tid;
        const int64_t i = base + j;
        s_v[j + (j >> 4)] = i < n ? in[i] : 0;
    }
    __syncthreads();
    uint32_t loc[SCAN_ITEMS];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { loc[k] = s_v[tid * 17 + k]; sum += loc[k]; }
    const uint32_t incl = wave_incl_scan(sum, lane);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t run = partial[blockIdx.x] + incl - sum;
    for (int w2 = 0; w2 < wave; w2++) run += s_w[w2];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { s_v[tid * 17 + k] = run; run += loc[k]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int j = k * 256 + tid;
        const int64_t i = base + j;
        if (i < n) out[i] = s_v[j + (j >> 4)];
    }
}

int scan_excl(phz_ctx *ctx, const uint32_t *in, uint32_t *out /* [n + 1] */, int64_t n, DevBuf &tmp) {
    hipStream_t sm = ctx->stream;
    if (n <= 0) { PHZ_HIP(ctx, hipMemsetAsync(out, 0, 4, sm)); return PHZ_OK; }
    const int64_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (int s = phz_reserve(ctx, tmp, (size_t)nb * 4 + 16)) return s;
    uint32_t *partial = (uint32_t *)tmp.p;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(256), 0, sm, in, n, partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, sm, partial, nb, out + n);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, sm, in, out, n, (const uint32_t *)partial);
    PHZ_HIP(ctx, hipGetLastError());
    return PHZ_OK;
}


}  // namespace
