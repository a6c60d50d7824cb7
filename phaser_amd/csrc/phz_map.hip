// K_map: read -> het-variant allele mapper for gfx950 (MI355X).
//
// Computes what phaser/read_variant_map.py:3-258 computes (do_read_variant_map -> split_read ->
// identify_allele), restated as a stateless rule per (record, N-split segment, variant) -- SURVEY.md 3.3:
//   emit iff  seg_start <= var.pos - POS  and  var.pos - POS + ref_len <= seg_start + len(pseudo_read)
//   and the extracted text is neither "" nor "N".
// Integer / branch work on an HBM-resident structure of arrays; no MFMA.
//
// Launch shape: one 256-thread workgroup per tile of 1024 consecutive (coordinate-sorted) reads, four
// reads per lane strided by 256 so pos/cigar_off/seq_off loads coalesce.  Per tile:
//   1. the het-SNP window starting at lower_bound(vpos, POS of the tile's first read) is staged in LDS
//      (2048 positions = 8 KiB); lanes binary-search it per aligned run, falling through to global
//      memory only for introns that reach past the window,
//   2. pass A walks each read's packed CIGAR and counts its calls (seq/qual bytes are touched only
//      under a variant), a wave shuffle scan + LDS combine gives per-read output offsets,
//   3. a decoupled look-back over 8-byte {status,value} tile descriptors (single agent-scope atomics,
//      wave-parallel with ballot) yields the tile's global output base, so the call list comes out in
//      exact mapper order without a second launch,
//   4. pass B re-walks only the reads that produced calls and writes them.
#include "phz_internal.h"

namespace {

constexpr int MAP_BLOCK = 256;
constexpr int MAP_RPT = 4;
constexpr int MAP_TILE = MAP_BLOCK * MAP_RPT;
constexpr int MAP_WIN = 2048;

constexpr uint32_t OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_EQ = 7, OP_X = 8, OP_G = 9;

constexpr uint64_t ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62, ST_MASK = (1ull << 62) - 1;

struct MapArgs {
    const int32_t *pos;
    const uint32_t *cigar_off, *cigar, *seq_off;
    const uint8_t *seq2, *qual;
    int64_t n;
    const int32_t *vpos;
    int nv;
    int baseq;
    int32_t *o_read, *o_var;
    uint8_t *o_code;
    uint32_t *o_aux0, *o_aux1;
    int64_t cap;
    const int32_t *tile_w0;
    uint64_t *desc;
    uint32_t *ticket;
    unsigned long long *total;
    int64_t ntiles;
};

struct VarWin {
    const int32_t *g;
    const int32_t *lds;
    int w0, nv;
    __device__ __forceinline__ int at(int i) const {
        unsigned d = (unsigned)(i - w0);
        return d < (unsigned)MAP_WIN ? lds[d] : g[i];
    }
    // first index in [lo, nv) whose position is >= key
    __device__ __forceinline__ int lower_bound(int lo, int key) const {
        int hi = nv;
        if (lo >= hi) return hi;
        // gallop first: the answer is almost always within a few entries of lo
        int step = 1;
        while (lo + step < hi && at(lo + step) < key) { lo += step; step <<= 1; }
        if (lo < hi && at(lo) >= key) return lo;
        int h = lo + step < hi ? lo + step : hi;
        int l = lo + 1;
        while (l < h) {
            int m = (l + h) >> 1;
            if (at(m) < key) l = m + 1; else h = m;
        }
        return l;
    }
};

// symbol of read base x after baseq masking: 0..3 = ACGT, 4 = 'N', 5 = other IUPAC character
__device__ __forceinline__ int masked_base(const MapArgs &a, uint32_t soff, int x) {
    uint32_t q = a.qual[(size_t)soff * 4 + x];
    uint32_t s = (a.seq2[(size_t)soff + (x >> 2)] >> (2 * (x & 3))) & 3;
    if ((int)(q & 0x7f) < a.baseq) return 4;
    if (q & 0x80) return s == 0 ? 4 : 5;
    return (int)s;
}

template <bool EMIT>
__device__ int walk_read(const MapArgs &a, const VarWin &vw, int64_t r, int64_t out_base) {
    const int pos = a.pos[r];
    const uint32_t c0 = a.cigar_off[r], c1 = a.cigar_off[r + 1];
    int i = vw.lower_bound(vw.w0, pos);
    if (i >= vw.nv) return 0;
    const uint32_t soff = a.seq_off[r];
    int cnt = 0;
    int gpos = 0, rpos = 0, seg_start = 0, plen = 0, seg_rpos = 0;
    uint32_t seg_op = c0;
    for (uint32_t k = c0; k < c1; k++) {
        uint32_t w = a.cigar[k];
        const int len = (int)(w >> 4);
        const uint32_t op = w & 15;
        if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) {
            const int lo = pos + seg_start + plen, hi = lo + len;
            if (i < vw.nv && vw.at(i) < lo) i = vw.lower_bound(i, lo);
            while (i < vw.nv) {
                const int vp = vw.at(i);
                if (vp >= hi) break;
                const int p = vp - pos - seg_start;            // index into the segment's pseudo read
                const int rb = rpos + (p - plen);              // read offset of that base (M runs only)
                // insertion recorded under key == p in THIS segment (key is read-relative: quirk kept)
                int ioff = 0, ilen = 0;
                if (c1 - c0 > 1) {
                    int g2 = seg_start, r2 = seg_rpos;
                    for (uint32_t k2 = seg_op; k2 < c1; k2++) {
                        uint32_t w2 = a.cigar[k2];
                        const int l2 = (int)(w2 >> 4);
                        const uint32_t o2 = w2 & 15;
                        if (o2 == OP_N) break;
                        if (o2 == OP_M || o2 == OP_EQ || o2 == OP_X) { g2 += l2; r2 += l2; }
                        else if (o2 == OP_D || o2 == OP_G) g2 += l2;
                        else if (o2 == OP_I) { if (g2 - 1 == p) { ioff = r2; ilen = l2; } r2 += l2; }
                        else if (o2 == OP_S) r2 += l2;
                    }
                }
                int nchars = (op != OP_D) + ilen;
                int code = -1;
                if (nchars == 1) {
                    const int s = masked_base(a, soff, op != OP_D ? rb : ioff);
                    code = s < 4 ? s : (s == 4 ? -1 : 4);
                } else if (nchars > 1) {
                    code = 4;
                }
                if (code >= 0) {
                    if (EMIT) {
                        const int64_t o = out_base + cnt;
                        if (o < a.cap) {
                            a.o_read[o] = (int32_t)r;
                            a.o_var[o] = i;
                            a.o_code[o] = (uint8_t)code;
                            a.o_aux0[o] = op != OP_D ? (uint32_t)rb : 0xFFFFFFFFu;
                            a.o_aux1[o] = ilen > 0 ? (((uint32_t)ioff << 12) | (uint32_t)(ilen > 4095 ? 4095 : ilen)) : 0u;
                        }
                    }
                    cnt++;
                }
                i++;
            }
            plen += len; gpos += len;
            if (op != OP_D) rpos += len;
        } else if (op == OP_I || op == OP_S) {
            rpos += len;
        } else if (op == OP_N) {
            gpos += len; seg_start = gpos; plen = 0; seg_op = k + 1; seg_rpos = rpos;
        } else if (op == OP_G) {
            gpos += len;
        }
    }
    return cnt;
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src); hi = __shfl(hi, src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
    return ((uint64_t)hi << 32) | lo;
}

// per-tile window start: tile_w0[t] = lower_bound(vpos, pos[t * MAP_TILE])
__global__ void k_tile_window(const int32_t *pos, int64_t n, const int32_t *vpos, int nv, int32_t *tile_w0, int64_t ntiles) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const int key = pos[t * MAP_TILE];
    int lo = 0, hi = nv;
    while (lo < hi) {
        int m = (lo + hi) >> 1;
        if (vpos[m] < key) lo = m + 1; else hi = m;
    }
    tile_w0[t] = lo;
}

__global__ __launch_bounds__(MAP_BLOCK) void k_map(MapArgs a) {
    __shared__ int32_t s_vpos[MAP_WIN];
    __shared__ int s_wsum[MAP_RPT][MAP_BLOCK / 64];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_tile;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const int64_t tile = s_tile;
    const int64_t r0 = tile * MAP_TILE;

    VarWin vw;
    vw.g = a.vpos; vw.lds = s_vpos; vw.nv = a.nv; vw.w0 = a.tile_w0[tile];
    for (int j = tid; j < MAP_WIN; j += MAP_BLOCK) {
        const int idx = vw.w0 + j;
        s_vpos[j] = idx < a.nv ? a.vpos[idx] : 0x7fffffff;
    }
    __syncthreads();

    // ---- pass A: count
    int cnt[MAP_RPT], incl[MAP_RPT];
#pragma unroll
    for (int k = 0; k < MAP_RPT; k++) {
        const int64_t r = r0 + k * MAP_BLOCK + tid;
        cnt[k] = r < a.n ? walk_read<false>(a, vw, r, 0) : 0;
        int x = cnt[k];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        incl[k] = x;
        if (lane == 63) s_wsum[k][wave] = x;
    }
    __syncthreads();
    int off[MAP_RPT];
    int running = 0;
#pragma unroll
    for (int k = 0; k < MAP_RPT; k++) {
        int before = running;
#pragma unroll
        for (int w2 = 0; w2 < MAP_BLOCK / 64; w2++) {
            const int s = s_wsum[k][w2];
            if (w2 < wave) before += s;
            running += s;
        }
        off[k] = before + incl[k] - cnt[k];
    }
    const uint64_t T = (uint64_t)running;   // calls of this tile

    // ---- decoupled look-back for the tile's global base (wave 0)
    if (wave == 0) {
        if (lane == 0 && tile > 0)
            __hip_atomic_store(&a.desc[tile], ST_AGG | T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t sum = 0;
        int64_t j = tile - 1;
        for (;;) {
            const int64_t idx = j - lane;
            uint64_t d = idx >= 0 ? __hip_atomic_load(&a.desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_PREFIX;
            const uint32_t st = (uint32_t)(d >> 62);
            const uint64_t m_prefix = __ballot(st == 2), m_empty = __ballot(st == 0);
            const int fp = m_prefix ? __ffsll((unsigned long long)m_prefix) - 1 : 64;
            const uint64_t upto = fp >= 63 ? ~0ull : ((2ull << fp) - 1);
            if (m_empty & upto) { __builtin_amdgcn_s_sleep(2); continue; }
            uint64_t v = lane <= fp ? (d & ST_MASK) : 0;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
            sum += v;
            if (fp < 64) break;
            j -= 64;
        }
        if (lane == 0) {
            __hip_atomic_store(&a.desc[tile], ST_PREFIX | (sum + T), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = sum;
            if (tile == a.ntiles - 1) *a.total = sum + T;
        }
    }
    __syncthreads();
    const int64_t base = (int64_t)s_base;

    // ---- pass B: emit
#pragma unroll
    for (int k = 0; k < MAP_RPT; k++) {
        if (cnt[k] > 0) {
            const int64_t r = r0 + k * MAP_BLOCK + tid;
            walk_read<true>(a, vw, r, base + off[k]);
        }
    }
}

}  // namespace

int phz_launch_map(phz_ctx *ctx, const phz_reads &r, const phz_variants &v, int baseq, const phz_calls &out,
                   int64_t *n_calls) {
    *n_calls = 0;
    if (r.n_reads == 0 || v.n == 0) return PHZ_OK;
    if (v.n > 0x7fffffff) return phz_fail(ctx, PHZ_E_ARG, "too many variants in one shard");
    const int64_t ntiles = (r.n_reads + MAP_TILE - 1) / MAP_TILE;
    if (int s = phz_reserve(ctx, ctx->desc, (size_t)ntiles * 8)) return s;
    if (int s = phz_reserve(ctx, ctx->tile_w0, (size_t)ntiles * 4)) return s;
    if (int s = phz_reserve(ctx, ctx->scalars, 64)) return s;
    PHZ_HIP(ctx, hipMemsetAsync(ctx->desc.p, 0, (size_t)ntiles * 8, ctx->stream));
    PHZ_HIP(ctx, hipMemsetAsync(ctx->scalars.p, 0, 64, ctx->stream));
    MapArgs a;
    a.pos = r.pos; a.cigar_off = r.cigar_off; a.cigar = r.cigar; a.seq_off = r.seq_off; a.seq2 = r.seq2; a.qual = r.qual;
    a.n = r.n_reads; a.vpos = v.pos; a.nv = (int)v.n; a.baseq = baseq;
    a.o_read = out.read_idx; a.o_var = out.var_idx; a.o_code = out.code; a.o_aux0 = out.aux0; a.o_aux1 = out.aux1;
    a.cap = out.cap;
    a.tile_w0 = (const int32_t *)ctx->tile_w0.p;
    a.desc = (uint64_t *)ctx->desc.p;
    a.ticket = (uint32_t *)ctx->scalars.p;
    a.total = (unsigned long long *)((char *)ctx->scalars.p + 8);
    a.ntiles = ntiles;
    hipLaunchKernelGGL(k_tile_window, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, ctx->stream,
                       r.pos, r.n_reads, v.pos, (int)v.n, (int32_t *)ctx->tile_w0.p, ntiles);
    PHZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_map, dim3((unsigned)ntiles), dim3(MAP_BLOCK), 0, ctx->stream, a);
    PHZ_HIP(ctx, hipGetLastError());
    PHZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    unsigned long long total = 0;
    PHZ_HIP(ctx, hipMemcpyAsync(&total, a.total, 8, hipMemcpyDeviceToHost, ctx->stream));
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    PHZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms[PHZ_T_MAP] = ms; ctx->total_ms[PHZ_T_MAP] += ms; ctx->launches[PHZ_T_MAP]++;
    *n_calls = (int64_t)total;
    return (int64_t)total > out.cap ? PHZ_E_CAPACITY : PHZ_OK;
}
