// Device-wide primitives written for this library (no rocPRIM / hipCUB): a generic exclusive scan and a stable LSD radix sort of
// (key, value) pairs.  Header-only: every translation unit that needs them gets its own copy of the kernels.
//
// Radix sort: 8 bits per pass over the significant bits only.  One wave per tile of 1024 keys (16 rows of 64): the stable rank of a
// key inside its row comes from eight wave ballots (lanes holding the same digit), the running per-digit offsets of the tile live
// in LDS, so a pass is three launches: per-tile digit histograms, one scan over the [digit][tile] table, scatter.
#pragma once
#include "phz_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------ exclusive scan TI -> TO
// out[i] = sum(in[0..i)), out[n] = total.  Three passes: chunk sums, one-block scan of the sums, chunk-local scan + base.
constexpr int GS_ITEMS = 16, GS_CHUNK = GS_ITEMS * 256;

// optional transform of the input on load (a scan over f(in[i]) without materialising f(in)): default = the values themselves
struct GsIdentity { template <class T> __device__ static __forceinline__ T f(T x) { return x; } };

template <class T> __device__ __forceinline__ T gs_wave_incl(T x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const T y = __shfl_up(x, d); if (lane >= d) x += y; }
    return x;
}

template <class TI, class TO, class F = GsIdentity> __global__ __launch_bounds__(256) void k_gs_reduce(const TI *in, int64_t n, TO *partial) {
    __shared__ TO s[4];
    const int64_t base = (int64_t)blockIdx.x * GS_CHUNK;
    TO x = 0;
#pragma unroll
    for (int k = 0; k < GS_ITEMS; k++) { const int64_t i = base + k * 256 + threadIdx.x; if (i < n) x += (TO)F::f(in[i]); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <class TO> __global__ __launch_bounds__(1024) void k_gs_partials(TO *partial, int64_t nb, TO *total) {
    __shared__ TO s_w[16];
    __shared__ TO s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
        const int64_t i = b0 + tid;
        const TO v = i < nb ? partial[i] : (TO)0;
        const TO x = gs_wave_incl(v, lane);
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        TO before = s_carry;
        for (int w2 = 0; w2 < wave; w2++) before += s_w[w2];
        if (i < nb) partial[i] = before + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + x;
        __syncthreads();
    }
    if (tid == 0) *total = s_carry;
}

// A workgroup's chunk of GS_CHUNK elements as GS_ROWS rows of 256 x 4: thread t holds elements (row * 256 + t) * 4 .. + 3 of every row, so a wave
// reads / writes 1 KB of consecutive memory per instruction (16 consecutive elements per thread made every access instruction touch 64 cache lines:
// the 18.8 M-element scan of a genome's read-label widths ran at 0.9 TB/s).  Exclusive prefix of the chunk with ONE barrier: wave scans of the four
// row sums, the waves' totals of every row in LDS.
constexpr int GS_ROWS = GS_ITEMS / 4;
template <class TI, class TO, class F = GsIdentity> __device__ __forceinline__ void gs_load_rows(const TI *in, int64_t n, int64_t chunk0, int tid, TO v[GS_ROWS][4]) {
#pragma unroll
    for (int r = 0; r < GS_ROWS; r++) {
        const int64_t i = chunk0 + ((int64_t)r * 256 + tid) * 4;
        if (sizeof(TI) == 4 && i + 3 < n && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0)) {
            const uint4 x = *reinterpret_cast<const uint4 *>(in + i);
            v[r][0] = (TO)F::f((TI)x.x); v[r][1] = (TO)F::f((TI)x.y); v[r][2] = (TO)F::f((TI)x.z); v[r][3] = (TO)F::f((TI)x.w);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) v[r][j] = i + j < n ? (TO)F::f(in[i + j]) : (TO)0;
        }
    }
}
// -> exclusive prefix (inside the chunk) of the first element of every row of this thread; *chunk_sum = sum of the chunk.  s_w: [GS_ROWS][4] in LDS
template <class TO> __device__ __forceinline__ void gs_chunk_scan(const TO v[GS_ROWS][4], int tid, TO (*s_w)[4], TO base[GS_ROWS], TO *chunk_sum) {
    const int lane = tid & 63, wave = tid >> 6;
    TO rs[GS_ROWS], incl[GS_ROWS];
#pragma unroll
    for (int r = 0; r < GS_ROWS; r++) {
        rs[r] = v[r][0] + v[r][1] + v[r][2] + v[r][3];
        incl[r] = gs_wave_incl(rs[r], lane);
        if (lane == 63) s_w[r][wave] = incl[r];
    }
    __syncthreads();
    TO carry = 0;
#pragma unroll
    for (int r = 0; r < GS_ROWS; r++) {
        TO before = carry;
        for (int w2 = 0; w2 < wave; w2++) before += s_w[r][w2];
        base[r] = before + incl[r] - rs[r];
        carry += s_w[r][0] + s_w[r][1] + s_w[r][2] + s_w[r][3];
    }
    *chunk_sum = carry;
}
template <class TO> __device__ __forceinline__ void gs_store_rows(TO *out, int64_t n, int64_t chunk0, int tid, const TO v[GS_ROWS][4], const TO base[GS_ROWS], TO offset) {
#pragma unroll
    for (int r = 0; r < GS_ROWS; r++) {
        const int64_t i = chunk0 + ((int64_t)r * 256 + tid) * 4;
        TO run = offset + base[r];
        TO o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { o[j] = run; run += v[r][j]; }
        if (sizeof(TO) == 4 && i + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
            *reinterpret_cast<uint4 *>(out + i) = make_uint4((uint32_t)o[0], (uint32_t)o[1], (uint32_t)o[2], (uint32_t)o[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) if (i + j < n) out[i + j] = o[j];
        }
        if (i <= n - 1 && n - 1 < i + 4) out[n] = run;      // the thread holding the last element also holds the total (elements beyond n read as 0)
    }
}

template <class TI, class TO, class F = GsIdentity> __global__ __launch_bounds__(256) void k_gs_apply(const TI *in, TO *out, int64_t n, const TO *partial) {
    __shared__ TO s_w[GS_ROWS][4];
    const int tid = threadIdx.x;
    const int64_t chunk0 = (int64_t)blockIdx.x * GS_CHUNK;
    TO v[GS_ROWS][4], base[GS_ROWS], sum;
    gs_load_rows<TI, TO, F>(in, n, chunk0, tid, v);
    gs_chunk_scan<TO>(v, tid, s_w, base, &sum);
    gs_store_rows<TO>(out, n, chunk0, tid, v, base, partial[blockIdx.x]);
}

// ---- the same scan in ONE launch for 32-bit sums (decoupled look-back): tiles take tickets in start order, publish their sum, and the
// first wave of a tile looks back over its predecessors' status words (64 at a time) until it meets one that already knows its prefix.
// A status word = epoch:30 | state:2 | value:32, written and read as one 64-bit access; words of older scans carry an older epoch and
// read as "not there yet", so nothing is cleared between scans.
constexpr unsigned GS_AGG = 1u, GS_PREFIX = 2u;
__device__ __forceinline__ unsigned long long gs_word(uint32_t epoch, unsigned state, uint32_t value) {
    return ((unsigned long long)epoch << 34) | ((unsigned long long)state << 32) | value;
}
template <class TI, class F = GsIdentity> __global__ __launch_bounds__(256) void k_gs_lookback(const TI *in, uint32_t *out, int64_t n, unsigned long long *status, uint32_t *ticket,
                                                                          uint32_t ticket_base, uint32_t epoch) {
    __shared__ uint32_t s_w[GS_ROWS][4];
    __shared__ uint32_t s_tile, s_prefix;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    const uint32_t tile = s_tile;
    const int64_t chunk0 = (int64_t)tile * GS_CHUNK;
    uint32_t v[GS_ROWS][4], base[GS_ROWS], block_sum;
    gs_load_rows<TI, uint32_t, F>(in, n, chunk0, tid, v);
    gs_chunk_scan<uint32_t>(v, tid, s_w, base, &block_sum);
    if (wave == 0) {
        if (lane == 0) __hip_atomic_store(&status[tile], gs_word(epoch, tile == 0 ? GS_PREFIX : GS_AGG, block_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (tile > 0) {
            int64_t j = (int64_t)tile - 1;
            for (;;) {
                const int64_t idx = j - lane;
                const unsigned long long w = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : gs_word(epoch, GS_PREFIX, 0u);
                const unsigned state = (uint32_t)(w >> 34) == epoch ? (unsigned)(w >> 32) & 3u : 0u;
                // usable: every word from the nearest predecessor up to the first one that holds a prefix
                const unsigned long long have = __ballot(state != 0u), pref = __ballot(state == GS_PREFIX);
                const unsigned long long upto = pref ? ((pref & (~pref + 1ull)) << 1) - 1ull : ~0ull;        // lanes 0 .. first prefix lane
                if ((have & upto) != upto) { __builtin_amdgcn_s_sleep(1); continue; }                  // a predecessor in that stretch has not published yet
                uint32_t x = ((upto >> lane) & 1ull) ? (uint32_t)w : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
                excl += x;
                if (pref) break;
                j -= 64;
            }
            if (lane == 0) __hip_atomic_store(&status[tile], gs_word(epoch, GS_PREFIX, excl + block_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_prefix = excl;
    }
    __syncthreads();
    gs_store_rows<uint32_t>(out, n, chunk0, tid, v, base, s_prefix);
}

template <class TI, class TO, class F = GsIdentity> int gscan_excl(phz_ctx *ctx, const TI *in, TO *out /* [n + 1] */, int64_t n, DevBuf &tmp) {
    hipStream_t sm = ctx->stream;
    if (n <= 0) { PHZ_HIP(ctx, hipMemsetAsync(out, 0, sizeof(TO), sm)); return PHZ_OK; }
    const int64_t nb = (n + GS_CHUNK - 1) / GS_CHUNK;
    // (beyond a few million elements the prefix front of the look-back -- 64 tiles per round trip -- is slower than two streaming passes:
    //  18.8 M elements took 161 us in one launch)
    if (sizeof(TO) == 4 && n < (int64_t)(4 << 20)) {
      if constexpr (sizeof(TO) == 4) {
        const size_t before = ctx->scan_state.cap;
        if (int s = phz_reserve(ctx, ctx->scan_state, 64 + (size_t)nb * 8)) return s;
        if (ctx->scan_state.cap != before || ctx->scan_epoch >= (1u << 30) - 2u) {
            PHZ_HIP(ctx, hipMemsetAsync(ctx->scan_state.p, 0, ctx->scan_state.cap, sm));
            ctx->scan_epoch = 0; ctx->scan_ticket_base = 0;
        }
        const uint32_t epoch = ++ctx->scan_epoch;
        hipLaunchKernelGGL((k_gs_lookback<TI, F>), dim3((unsigned)nb), dim3(256), 0, sm, in, (uint32_t *)out, n, (unsigned long long *)((char *)ctx->scan_state.p + 64),
                           (uint32_t *)ctx->scan_state.p, ctx->scan_ticket_base, epoch);
        ctx->scan_ticket_base += (uint32_t)nb;
        PHZ_HIP(ctx, hipGetLastError());
        (void)tmp;
        return PHZ_OK;
      }
    }
    {
        if (int s = phz_reserve(ctx, tmp, (size_t)nb * sizeof(TO) + 16)) return s;
        TO *partial = (TO *)tmp.p;
        hipLaunchKernelGGL((k_gs_reduce<TI, TO, F>), dim3((unsigned)nb), dim3(256), 0, sm, in, n, partial);
        hipLaunchKernelGGL((k_gs_partials<TO>), dim3(1), dim3(1024), 0, sm, partial, nb, out + n);
        hipLaunchKernelGGL((k_gs_apply<TI, TO, F>), dim3((unsigned)nb), dim3(256), 0, sm, in, out, n, (const TO *)partial);
        PHZ_HIP(ctx, hipGetLastError());
        return PHZ_OK;
    }
}

// ------------------------------------------------------------------------------------------------ LSD radix sort of pairs
constexpr int RS_ROWS = 16, RS_TILE = 64 * RS_ROWS;

template <class K> __global__ __launch_bounds__(64) void k_rs_hist(const K *keys, int64_t n, int shift, uint32_t *cnt, uint32_t ntile) {
    __shared__ uint32_t s_h[256];
    const int lane = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; j++) s_h[lane + 64 * j] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const int64_t i = base + r * 64 + lane;
        if (i < n) atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) cnt[(size_t)(lane + 64 * j) * ntile + blockIdx.x] = s_h[lane + 64 * j];
}

template <class K, class V> __global__ __launch_bounds__(64) void k_rs_scatter(const K *kin, const V *vin, K *kout, V *vout, int64_t n, int shift,
                                                                               const uint32_t *off, uint32_t ntile) {
    __shared__ uint32_t s_off[256];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    K rk[RS_ROWS]; V rv[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {                   // the tile's 16 rows AND its 256 digit offsets requested together: one memory round trip
        const int64_t i = base + r * 64 + lane;
        rk[r] = i < n ? kin[i] : (K)0; rv[r] = i < n ? vin[i] : (V)0;
    }
    uint32_t o4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) o4[j] = off[(size_t)(lane + 64 * j) * ntile + blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++) s_off[lane + 64 * j] = o4[j];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const int64_t i = base + r * 64 + lane;
        const bool valid = i < n;
        const K k = rk[r];
        const uint32_t d = (uint32_t)(k >> shift) & 255u;
        unsigned long long peers = __ballot(valid ? 1 : 0);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((int)((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const int rank = __popcll(peers & below);
        const uint32_t pos = valid ? s_off[d] + (uint32_t)rank : 0u;
        __syncthreads();
        if (valid && rank == 0) s_off[d] += (uint32_t)__popcll(peers);      // the lowest lane of every digit group moves the digit's cursor
        __syncthreads();
        if (valid) { kout[pos] = k; vout[pos] = rv[r]; }
    }
}

// Sorts n (key, value) pairs by bits [bit_lo, bit_hi) of the key, stable.  Buffers ping-pong between (k0, v0) and (k1, v1); *where says
// which pair holds the result (0 or 1).  cnt / tmp: scratch.
template <class K, class V>
int radix_sort_pairs(phz_ctx *ctx, K *k0, K *k1, V *v0, V *v1, int64_t n, int bit_lo, int bit_hi, DevBuf &cnt, DevBuf &tmp, int *where, V *last_v = nullptr, K *last_k = nullptr) {
    // last_v (and last_k): the LAST pass scatters into these arrays instead of the ping-pong buffer (*where = 2 then): no copy of the result afterwards
    *where = 0;
    if (n <= 1 || bit_hi <= bit_lo) return PHZ_OK;
    if (n >= (1ll << 32)) return phz_fail(ctx, PHZ_E_ARG, "radix sort of more than 2^32 items");
    const uint32_t ntile = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const size_t ncnt = (size_t)256 * ntile;
    if (int s = phz_reserve(ctx, cnt, (ncnt + 1) * 4)) return s;
    uint32_t *c = (uint32_t *)cnt.p;
    K *ka = k0, *kb = k1; V *va = v0, *vb = v1;
    for (int shift = bit_lo; shift < bit_hi; shift += 8) {
        hipLaunchKernelGGL((k_rs_hist<K>), dim3(ntile), dim3(64), 0, ctx->stream, (const K *)ka, n, shift, c, ntile);
        if (int s = gscan_excl<uint32_t, uint32_t>(ctx, c, c, (int64_t)ncnt, tmp)) return s;
        const bool last = last_v != nullptr && shift + 8 >= bit_hi;
        // (the keys of the last pass go to the ping-pong buffer when the caller does not want them)
        hipLaunchKernelGGL((k_rs_scatter<K, V>), dim3(ntile), dim3(64), 0, ctx->stream, (const K *)ka, (const V *)va, last && last_k ? last_k : kb, last ? last_v : vb, n, shift,
                           (const uint32_t *)c, ntile);
        if (last) { *where = 2; break; }
        std::swap(ka, kb); std::swap(va, vb);
        *where ^= 1;
    }
    PHZ_HIP(ctx, hipGetLastError());
    return PHZ_OK;
}

// ------------------------------------------------------------------------------------------------ LSD radix sort, ONE launch per pass
// (round 5; the row stage's six ordering sorts were 23 passes x 3 launches + two copies each).  The digit histograms of ALL passes of a sort
// come from one read of the keys (k_os_hist -> k_os_bases: exclusive digit bases per pass); a pass is then ONE kernel, k_os_pass: workgroups of
// four waves take tiles of 4,096 keys in ticket order (wave w owns the tile's keys [w * 1024, (w + 1) * 1024): 16 rows of 64, the stable rank
// inside a row from eight ballots as in k_rs_scatter), add up their per-wave digit counts, publish the tile's 256 counts and look back over
// the status words of their predecessors -- thread d follows digit d's chain, a wave reads 64 consecutive words per step -- until a tile that
// already knows its inclusive prefix (decoupled look-back; tickets make every predecessor a tile that has started).  A status word =
// state:2 | count:30 (n < 2^30), zeroed once per sort for all its passes.  The last pass writes straight into the caller's arrays.
constexpr int OS_WAVES = 4, OS_TILE = OS_WAVES * RS_TILE, OS_MAXPASS = 8, OS_LOOK = 8;
constexpr uint32_t OS_AGG = 1u << 30, OS_PREFIX = 2u << 30, OS_VALUE = (1u << 30) - 1u;
struct OsShifts { int n; int shift[OS_MAXPASS]; };

// lanes of the wave that hold the same 8-bit digit as this lane (among the `valid` ones): eight ballots
__device__ __forceinline__ unsigned long long os_peers(uint32_t d, bool valid) {
    unsigned long long peers = __ballot(valid ? 1 : 0);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const unsigned long long m = __ballot((int)((d >> b) & 1u));
        peers &= ((d >> b) & 1u) ? m : ~m;
    }
    return peers;
}
// (a plain `atomicAdd(&s_h[digit], 1)` per key was 40 us per sort: the high digits of these keys are nearly constant, so all 256 threads of a workgroup
//  queued on one LDS word; now the lowest lane of every group of equal digits adds the group's size)
template <class K> __global__ __launch_bounds__(256) void k_os_hist(const K *keys, int64_t n, OsShifts sh, uint32_t *ghist) {
    __shared__ uint32_t s_h[OS_MAXPASS * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int j = tid; j < sh.n * 256; j += 256) s_h[j] = 0;
    __syncthreads();
    const int64_t rounds = (n + (int64_t)gridDim.x * 256 - 1) / ((int64_t)gridDim.x * 256);          // the same trip count for every lane: the ballots need whole waves
    for (int64_t q = 0; q < rounds; q++) {
        const int64_t i = (q * gridDim.x + blockIdx.x) * 256 + tid;
        const bool valid = i < n;
        const K k = valid ? keys[i] : (K)0;
        for (int p = 0; p < sh.n; p++) {
            const uint32_t d = (uint32_t)(k >> sh.shift[p]) & 255u;
            const unsigned long long peers = os_peers(d, valid);
            if (valid && (peers & below) == 0ull) atomicAdd(&s_h[p * 256 + d], (uint32_t)__popcll(peers));
        }
    }
    __syncthreads();
    for (int j = tid; j < sh.n * 256; j += 256) if (s_h[j]) atomicAdd(&ghist[j], s_h[j]);
}
// one workgroup per pass: ghist[p][d] -> number of keys whose digit of pass p is smaller than d
__global__ __launch_bounds__(256) void k_os_bases(uint32_t *ghist) {
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *h = ghist + (size_t)blockIdx.x * 256;
    const uint32_t v = h[tid];
    const uint32_t incl = gs_wave_incl(v, lane);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; w++) before += s_w[w];
    h[tid] = before + incl - v;
}
// wave-level ordering point between LDS accesses of different lanes of ONE wave: the hardware executes a wave's LDS instructions in order, so the compiler
// barrier is all that is needed
#define PHZ_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
template <class K, class V> __global__ __launch_bounds__(256) void k_os_pass(const K *kin, const V *vin, K *kout, V *vout, int64_t n, int shift, const uint32_t *base,
                                                                            uint32_t *status, uint32_t *ticket) {
    __shared__ uint32_t s_cnt[OS_WAVES][256];          // per-wave digit counts (built row by row: a key's place among its wave's keys of that digit), then the
                                                       // wave's offset inside the tile's stretch of the digit
    __shared__ uint32_t s_base[256];                   // output position of the tile's first key of a digit
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < OS_WAVES; w++) s_cnt[w][tid] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const int64_t wbase = (int64_t)tile * OS_TILE + (int64_t)wave * RS_TILE;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    K rk[RS_ROWS]; V rv[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const int64_t i = wbase + r * 64 + lane;
        rk[r] = i < n ? kin[i] : (K)0; rv[r] = i < n ? vin[i] : (V)0;
    }
    // place of every key among the keys of its wave with the same digit (stable: rows in order, lanes in order inside a row), without atomics: the lanes of a
    // row that share a digit read the wave's running count, the lowest of them adds the group's size
    uint32_t off[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const bool valid = wbase + r * 64 + lane < n;
        const uint32_t d = (uint32_t)(rk[r] >> shift) & 255u;
        const unsigned long long peers = os_peers(d, valid);
        const uint32_t before = s_cnt[wave][d];
        off[r] = before + (uint32_t)__popcll(peers & below);
        PHZ_WAVE_SYNC();
        if (valid && (peers & below) == 0ull) s_cnt[wave][d] = before + (uint32_t)__popcll(peers);
        PHZ_WAVE_SYNC();
    }
    __syncthreads();
    {
        // thread d: digit d of this tile -- counts of the waves -> exclusive offsets, publish, look back
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < OS_WAVES; w++) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = total; total += c; }
        uint32_t *mine = status + (size_t)tile * 256 + tid;
        __hip_atomic_store(mine, (tile == 0 ? OS_PREFIX : OS_AGG) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (tile > 0) {
            // OS_LOOK predecessors per round trip (independent loads), consumed nearest first up to the first one that has not published yet or that knows
            // its prefix.  All tiles of a small sort start together, so most predecessors hold an aggregate only: tile t meets a resolved tile about t / 2
            // tiles back, i.e. after t / (2 OS_LOOK) round trips (one word at a time that was hundreds of dependent loads).  Measured on a genome's six ordering
            // sorts: 8 words per step 25 passes = 598 us, 32 words 690 us -- the look-back's own L2 traffic (tiles x 256 digits x words) is what a pass pays for
            int64_t t = (int64_t)tile - 1;
            bool done = false;
            while (!done) {
                uint32_t w[OS_LOOK];
#pragma unroll
                for (int j = 0; j < OS_LOOK; j++) w[j] = t - j >= 0 ? __hip_atomic_load(status + (size_t)(t - j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : OS_PREFIX;
                int used = 0;
#pragma unroll
                for (int j = 0; j < OS_LOOK; j++) {
                    if (done || used != j) continue;                 // stopped at an earlier word of this batch
                    if ((w[j] >> 30) == 0u) continue;                // that tile has its ticket and will publish without waiting for anybody: ask again
                    excl += w[j] & OS_VALUE; used = j + 1;
                    if (w[j] & OS_PREFIX) done = true;
                }
                t -= used;
                if (!used) __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(mine, OS_PREFIX | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_base[tid] = base[tid] + excl;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        if (wbase + r * 64 + lane < n) {
            const K k = rk[r];
            const uint32_t d = (uint32_t)(k >> shift) & 255u;
            const uint32_t pos = s_base[d] + s_cnt[wave][d] + off[r];
            kout[pos] = k; vout[pos] = rv[r];
        }
    }
}

// Sorts n (key, value) pairs by the bit ranges [lo, hi) of `ranges`, least significant range first, stable.  Buffers ping-pong between (k0, v0) and
// (k1, v1); with dst_val (and dst_key) given the LAST pass writes there and *where = 2, else *where = 0 / 1 says which pair holds the result.
template <class K, class V>
int radix_sort_ranges(phz_ctx *ctx, K *k0, K *k1, V *v0, V *v1, int64_t n, const int (*ranges)[2], int nranges, DevBuf &cnt, DevBuf &tmp, int *where, V *dst_val = nullptr, K *dst_key = nullptr,
                      bool force_one_launch = false) {
    *where = 0;
    OsShifts sh; sh.n = 0;
    bool fits = n < (1ll << 30);
    for (int r = 0; r < nranges; r++)
        for (int s = ranges[r][0]; s < ranges[r][1]; s += 8) { if (sh.n < OS_MAXPASS) sh.shift[sh.n] = s; sh.n++; }
    if (sh.n > OS_MAXPASS) fits = false;
    // Which sort: the one-launch passes are correct and take 60 launches fewer per phasing pass, but on the six ordering sorts of a genome (1.5 M keys each)
    // they LOSE to the three-launch passes -- 7.89 / 7.93 against 7.77 / 7.79 ms per pass, same box, alternating runs (profiles/r05/ab_sort.txt): with 366
    // tiles all in flight at once a tile's look-back walks ~t / 16 round trips, and the launches it saves cost only ~1.5 us each here.  So the three-launch
    // passes stay the default; PHZ_SORT_ONE_LAUNCH=1 selects the other one (tests run both).
    static const bool one_launch = getenv("PHZ_SORT_ONE_LAUNCH") != nullptr;
    if (n <= 1 || sh.n == 0 || !fits || !(one_launch || force_one_launch)) {
        K *ka = k0, *kb = k1; V *va = v0, *vb = v1;
        int last_range = -1;
        for (int r = 0; r < nranges; r++) if (ranges[r][1] > ranges[r][0]) last_range = r;
        for (int r = 0; r < nranges; r++) {
            int w = 0;
            const bool fin = dst_val != nullptr && r == last_range && n > 1;
            if (int s = radix_sort_pairs<K, V>(ctx, ka, kb, va, vb, n, ranges[r][0], ranges[r][1], cnt, tmp, &w, fin ? dst_val : (V *)nullptr, fin ? dst_key : (K *)nullptr)) return s;
            if (w == 2) { *where = 2; return PHZ_OK; }
            if (w) { std::swap(ka, kb); std::swap(va, vb); *where ^= 1; }
        }
        if (dst_val && n > 0) {
            PHZ_HIP(ctx, hipMemcpyAsync(dst_val, va, (size_t)n * sizeof(V), hipMemcpyDeviceToDevice, ctx->stream));
            if (dst_key) PHZ_HIP(ctx, hipMemcpyAsync(dst_key, ka, (size_t)n * sizeof(K), hipMemcpyDeviceToDevice, ctx->stream));
            *where = 2;
        }
        return PHZ_OK;
    }
    hipStream_t sm = ctx->stream;
    const uint32_t ntile = (uint32_t)((n + OS_TILE - 1) / OS_TILE);
    const size_t hist_words = (size_t)OS_MAXPASS * 256 + 64, status_words = (size_t)sh.n * ntile * 256;       // [hist P x 256][tickets P .. pad][status P x ntile x 256]
    if (int s = phz_reserve(ctx, cnt, (hist_words + status_words) * 4)) return s;
    uint32_t *ghist = (uint32_t *)cnt.p, *tickets = ghist + OS_MAXPASS * 256, *status = ghist + hist_words;
    PHZ_HIP(ctx, hipMemsetAsync(cnt.p, 0, (hist_words + status_words) * 4, sm));
    int cus = 256;
    // (256 workgroups at most: every workgroup ends with one global atomic per (pass, digit) -- 2,048 workgroups queueing on the same 768 words was most of this
    //  kernel's 37 us)
    hipLaunchKernelGGL((k_os_hist<K>), dim3((unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)cus)), dim3(256), 0, sm, (const K *)k0, n, sh, ghist);
    hipLaunchKernelGGL(k_os_bases, dim3((unsigned)sh.n), dim3(256), 0, sm, ghist);
    K *ka = k0, *kb = k1; V *va = v0, *vb = v1;
    for (int p = 0; p < sh.n; p++) {
        const bool last = p + 1 == sh.n && dst_val != nullptr;
        K *ko = last && dst_key ? dst_key : kb; V *vo = last ? dst_val : vb;
        hipLaunchKernelGGL((k_os_pass<K, V>), dim3(ntile), dim3(256), 0, sm, (const K *)ka, (const V *)va, ko, vo, n, sh.shift[p], (const uint32_t *)(ghist + (size_t)p * 256),
                           status + (size_t)p * ntile * 256, tickets + p);
        if (last) { *where = 2; break; }
        std::swap(ka, kb); std::swap(va, vb); *where ^= 1;
    }
    PHZ_HIP(ctx, hipGetLastError());
    (void)tmp;
    return PHZ_OK;
}

inline int bits_for(uint64_t max_value) { int b = 1; while (b < 64 && (max_value >> b)) b++; return b; }

}  // namespace
