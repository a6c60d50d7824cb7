// K_map_general: the mapper for variant sets that are not pure SNPs (phASER's --include_indels 1): REF longer than one
// base and/or multi-base alleles.  Same rule as K_map -- phaser/read_variant_map.py:236-258 with xvar.ref_length > 1:
//   emit iff 0 <= rs and rs + ref_len <= len(pseudo_read of the segment); text = pseudo[rs:rs+ref_len] with the segment's
//   insertions spliced after their (read-relative, quirk kept) keys and 'D' placeholders removed; "" and "N" suppressed
// but the text can be any length, so the call is classified on the device against the individual's two allele strings:
//   code 5 = text equals allele 0, 6 = equals allele 1 (first match wins, as `list.index` does at phaser.py:1317),
//   0..3 = some other single base, 4 = any other text (its read offsets go to an optional pool so the host can print it).
// This mode is off by default in phASER ("will likely result in poor quality phasing", phaser.py:48); it is built for
// completeness: one lane per record, count pass + exclusive scan + emit pass.  Records are coordinate-sorted, so a workgroup's 256
// records share one window start (one uniform search per workgroup, every per-segment search gallops from there), and the emit pass
// returns at once for the records the count pass found empty (most of them).
#include <cstring>
#include "phz_internal.h"
#include "phz_scan.h"

namespace {

constexpr uint32_t OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_EQ = 7, OP_X = 8, OP_G = 9;

struct GenArgs {
    const int32_t *pos;
    const uint32_t *cigar_off, *cigar, *seq_off;
    const uint8_t *seq2, *qual;
    int64_t n;
    const int32_t *vpos;
    const uint8_t *ref_len;
    const uint32_t *aoff;
    const char *abytes;
    int nv, baseq;
    uint32_t *n_calls, *n_text;                  // per record (count pass)
    const uint32_t *call_base, *text_base;       // exclusive scans (emit pass)
    int32_t *o_read, *o_var; uint8_t *o_code; uint32_t *o_aux0, *o_aux1;
    uint32_t *o_text_off; uint32_t *o_text;      // optional
    int64_t cap, text_cap;
};

__device__ __forceinline__ int sym_at(const GenArgs &a, uint32_t soff, int x) {
    const uint32_t q = a.qual[(size_t)soff * 4 + x];
    const uint32_t s = (a.seq2[(size_t)soff + (x >> 2)] >> (2 * (x & 3))) & 3;
    if ((int)(q & 0x7f) < a.baseq) return 4;
    if (q & 0x80) return s == 0 ? 4 : 5;
    return (int)s;
}

struct Compose {
    int n, first;
    bool m0, m1;
    uint32_t p0, l0, p1, l1;
    const char *ab;
    uint32_t *text;           // where read offsets of this call's characters go (EMIT + pool present), else nullptr
    int64_t text_room;
    __device__ __forceinline__ void add(int sym, uint32_t roff) {
        const char ch = "ACGTN?"[sym];
        if (n == 0) first = sym;
        m0 = m0 && ((uint32_t)n < l0) && ab[p0 + n] == ch;
        m1 = m1 && ((uint32_t)n < l1) && ab[p1 + n] == ch;
        if (text && n < text_room) text[n] = roff;
        n++;
    }
};

// first index in [lo, nv) with vpos >= key, galloping from lo (the answer is almost always a few entries away)
__device__ __forceinline__ int gallop_lb(const int32_t *vpos, int nv, int lo, long long key) {
    if (lo >= nv || (long long)vpos[lo] >= key) return lo;
    int step = 1;
    while (lo + step < nv && (long long)vpos[lo + step] < key) { lo += step; step <<= 1; }
    int l = lo + 1, h = lo + step < nv ? lo + step : nv;
    while (l < h) { const int m = (l + h) >> 1; if ((long long)vpos[m] < key) l = m + 1; else h = m; }
    return l;
}

template <bool EMIT>
__device__ void gen_read(const GenArgs &a, int64_t r, int w0) {
    const int pos = a.pos[r];
    const uint32_t c0 = a.cigar_off[r], c1 = a.cigar_off[r + 1];
    const uint32_t soff = a.seq_off[r];
    uint32_t ncalls = 0, ntext = 0;
    const uint64_t cbase = EMIT ? a.call_base[r] : 0, tbase = EMIT ? a.text_base[r] : 0;
    if (EMIT && a.call_base[r + 1] == a.call_base[r]) return;          // nothing under this record (the count pass knows)
    int gpos = 0, rpos = 0;
    uint32_t k = c0;
    for (;;) {
        const int seg_start = gpos, seg_rpos = rpos;
        int plen = 0;
        uint32_t k2 = k;
        for (; k2 < c1; k2++) {
            const uint32_t w = a.cigar[k2], op = w & 15;
            if (op == OP_N) break;
            if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) plen += (int)(w >> 4);
        }
        const long long lo = (long long)pos + seg_start;
        int i = gallop_lb(a.vpos, a.nv, w0, lo);
        for (; i < a.nv && (long long)a.vpos[i] < lo + plen; i++) {
            const int rs = (int)((long long)a.vpos[i] - lo), rl = a.ref_len[i];
            if (rs + rl > plen) continue;
            Compose c;
            // pass 0 classifies; a code-4 call in the emit pass is composed once more to record its read offsets (never
            // speculatively: a neighbouring record owns the pool space right after this record's share)
            for (int pass = 0; pass < 2; pass++) {
                c.n = 0; c.first = -1; c.m0 = true; c.m1 = true; c.ab = a.abytes;
                c.p0 = a.aoff[2 * i]; c.l0 = a.aoff[2 * i + 1] - c.p0; c.p1 = a.aoff[2 * i + 1]; c.l1 = a.aoff[2 * i + 2] - c.p1;
                c.text = nullptr; c.text_room = 0;
                if (pass == 1) {
                    c.text = a.o_text + tbase + ntext;
                    c.text_room = a.text_cap - (int64_t)(tbase + ntext);
                    if (c.text_room < 0) c.text_room = 0;
                }
                int pi = 0, ro = seg_rpos;
                for (uint32_t kk = k; kk < k2; kk++) {
                    const uint32_t w = a.cigar[kk], op = w & 15;
                    const int len = (int)(w >> 4);
                    if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) {
                        const int from = pi > rs ? pi : rs, to = (pi + len) < (rs + rl) ? (pi + len) : (rs + rl);
                        for (int p = from; p < to; p++) {
                            if (op != OP_D) c.add(sym_at(a, soff, ro + (p - pi)), (uint32_t)(ro + (p - pi)));
                            // insertion stored under key == p in this segment (later one wins; key is read-relative)
                            int ioff = 0, ilen = 0, g2 = seg_start, r2 = seg_rpos;
                            for (uint32_t q = k; q < k2; q++) {
                                const uint32_t w2 = a.cigar[q], o2 = w2 & 15;
                                const int l2 = (int)(w2 >> 4);
                                if (o2 == OP_M || o2 == OP_EQ || o2 == OP_X) { g2 += l2; r2 += l2; }
                                else if (o2 == OP_D || o2 == OP_G) g2 += l2;
                                else if (o2 == OP_I) { if (g2 - 1 == p) { ioff = r2; ilen = l2; } r2 += l2; }
                                else if (o2 == OP_S) r2 += l2;
                            }
                            for (int t = 0; t < ilen; t++) c.add(sym_at(a, soff, ioff + t), (uint32_t)(ioff + t));
                        }
                        pi += len;
                        if (op != OP_D) ro += len;
                        if (pi >= rs + rl) break;
                    } else if (op == OP_I || op == OP_S) {
                        ro += len;
                    }
                }
                const bool is_call = !(c.n == 0 || (c.n == 1 && c.first == 4));
                const bool other_text = is_call && !(c.m0 && (uint32_t)c.n == c.l0) && !(c.m1 && (uint32_t)c.n == c.l1) && !(c.n == 1 && c.first < 4);
                if (!(EMIT && a.o_text && other_text && pass == 0)) break;
            }
            if (c.n == 0 || (c.n == 1 && c.first == 4)) continue;
            int code;
            if (c.m0 && (uint32_t)c.n == c.l0) code = 5;
            else if (c.m1 && (uint32_t)c.n == c.l1) code = 6;
            else if (c.n == 1 && c.first < 4) code = c.first;
            else code = 4;
            if (EMIT) {
                const int64_t o = (int64_t)(cbase + ncalls);
                if (o < a.cap) {
                    a.o_read[o] = (int32_t)r; a.o_var[o] = i; a.o_code[o] = (uint8_t)code;
                    a.o_aux0[o] = 0xFFFFFFFFu; a.o_aux1[o] = 0;
                    if (a.o_text_off) a.o_text_off[o] = (uint32_t)(tbase + ntext);
                }
            }
            ncalls++;
            if (code == 4) ntext += (uint32_t)c.n;
        }
        for (uint32_t kk = k; kk < k2; kk++) {
            const uint32_t w = a.cigar[kk], op = w & 15;
            const int len = (int)(w >> 4);
            if (op == OP_M || op == OP_EQ || op == OP_X) { gpos += len; rpos += len; }
            else if (op == OP_D || op == OP_G) gpos += len;
            else if (op == OP_I || op == OP_S) rpos += len;
        }
        if (k2 >= c1) break;
        gpos += (int)(a.cigar[k2] >> 4);
        k = k2 + 1;
    }
    if (!EMIT) { a.n_calls[r] = ncalls; a.n_text[r] = ntext; }
}

template <bool EMIT>
__global__ __launch_bounds__(256) void k_map_general(GenArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // window start of the workgroup: first variant at or after the first record's POS (uniform search: every lane reads the
    // same addresses); later records only move forward from it
    const int key = a.pos[(int64_t)blockIdx.x * 256];
    int lo = 0, hi = a.nv;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (a.vpos[m] < key) lo = m + 1; else hi = m; }
    if (r < a.n) gen_read<EMIT>(a, r, lo);
}

}  // namespace

extern "C" int phz_map_reads_general(phz_ctx *ctx, const phz_reads *reads, const phz_variants_general *vars, int baseq,
                                     phz_calls *out, int64_t *n_calls, uint32_t *call_text_off, uint32_t *text_roff,
                                     int64_t text_cap, int64_t *n_text, int space) {
    if (!ctx || !reads || !vars || !out || !n_calls) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    *n_calls = 0;
    if (n_text) *n_text = 0;
    const int64_t n = reads->n_reads, nv = vars->n;
    if (n == 0 || nv == 0) return PHZ_OK;
    if (nv > 0x7fffffff) return phz_fail(ctx, PHZ_E_ARG, "too many variants in one shard");
    Staging st(ctx);
    GenArgs a;
    memset(&a, 0, sizeof a);
    if (int s = st.in(reads->pos, (size_t)n, space, &a.pos)) return s;
    if (int s = st.in(reads->cigar_off, (size_t)n + 1, space, &a.cigar_off)) return s;
    if (int s = st.in(reads->cigar, (size_t)reads->n_ops, space, &a.cigar)) return s;
    if (int s = st.in(reads->seq_off, (size_t)n + 1, space, &a.seq_off)) return s;
    if (int s = st.in(reads->seq2, (size_t)reads->n_seq_bytes, space, &a.seq2)) return s;
    if (int s = st.in(reads->qual, (size_t)reads->n_seq_bytes * 4, space, &a.qual)) return s;
    if (int s = st.in(vars->pos, (size_t)nv, space, &a.vpos)) return s;
    if (int s = st.in(vars->ref_len, (size_t)nv, space, &a.ref_len)) return s;
    if (int s = st.in(vars->allele_off, (size_t)nv * 2 + 1, space, &a.aoff)) return s;
    if (int s = st.in(vars->allele_bytes, (size_t)vars->n_allele_bytes, space, &a.abytes)) return s;
    a.n = n; a.nv = (int)nv; a.baseq = baseq;
    DevBuf *S = ctx->scratch;
    if (int s = phz_reserve(ctx, S[0], (size_t)n * 4)) return s;
    if (int s = phz_reserve(ctx, S[1], (size_t)n * 4)) return s;
    if (int s = phz_reserve(ctx, S[2], (size_t)(n + 1) * 4)) return s;
    if (int s = phz_reserve(ctx, S[3], (size_t)(n + 1) * 4)) return s;
    a.n_calls = (uint32_t *)S[0].p; a.n_text = (uint32_t *)S[1].p;
    uint32_t *cb = (uint32_t *)S[2].p, *tb = (uint32_t *)S[3].p;
    a.call_base = cb; a.text_base = tb;
    hipStream_t sm = ctx->stream;
    const unsigned grid = (unsigned)((n + 255) / 256);
    PHZ_HIP(ctx, hipEventRecord(ctx->ev0, sm));
    hipLaunchKernelGGL(k_map_general<false>, dim3(grid), dim3(256), 0, sm, a);
    if (int s = scan_excl(ctx, a.n_calls, cb, n, S[6])) return s;
    if (int s = scan_excl(ctx, a.n_text, tb, n, S[6])) return s;
    uint32_t last[2];
    PHZ_HIP(ctx, hipMemcpyAsync(&last[0], cb + n, 4, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipMemcpyAsync(&last[1], tb + n, 4, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    const int64_t total = (int64_t)last[0], ttotal = (int64_t)last[1];
    *n_calls = total;
    if (n_text) *n_text = ttotal;
    if (total > out->cap || (text_roff && ttotal > text_cap)) return PHZ_E_CAPACITY;
    uint32_t *d_toff = nullptr, *d_text = nullptr;
    if (int s = st.out(out->read_idx, (size_t)out->cap, space, &a.o_read)) return s;
    if (int s = st.out(out->var_idx, (size_t)out->cap, space, &a.o_var)) return s;
    if (int s = st.out(out->code, (size_t)out->cap, space, &a.o_code)) return s;
    if (int s = st.out(out->aux0, (size_t)out->cap, space, &a.o_aux0)) return s;
    if (int s = st.out(out->aux1, (size_t)out->cap, space, &a.o_aux1)) return s;
    if (call_text_off && text_roff) {
        if (int s = st.out(call_text_off, (size_t)out->cap + 1, space, &d_toff)) return s;
        if (int s = st.out(text_roff, (size_t)(text_cap ? text_cap : 1), space, &d_text)) return s;
    }
    a.o_text_off = d_toff; a.o_text = d_text; a.cap = out->cap; a.text_cap = text_cap;
    hipLaunchKernelGGL(k_map_general<true>, dim3(grid), dim3(256), 0, sm, a);
    PHZ_HIP(ctx, hipGetLastError());
    PHZ_HIP(ctx, hipEventRecord(ctx->ev1, sm));
    if (d_toff) { const uint32_t tt = (uint32_t)ttotal; PHZ_HIP(ctx, hipMemcpyAsync(d_toff + total, &tt, 4, hipMemcpyHostToDevice, sm)); }
    if (space == PHZ_HOST && total) {
        PHZ_HIP(ctx, hipMemcpyAsync(out->read_idx, a.o_read, (size_t)total * 4, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(out->var_idx, a.o_var, (size_t)total * 4, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(out->code, a.o_code, (size_t)total, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(out->aux0, a.o_aux0, (size_t)total * 4, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(out->aux1, a.o_aux1, (size_t)total * 4, hipMemcpyDeviceToHost, sm));
        if (d_toff) {
            PHZ_HIP(ctx, hipMemcpyAsync(call_text_off, d_toff, ((size_t)total + 1) * 4, hipMemcpyDeviceToHost, sm));
            if (ttotal) PHZ_HIP(ctx, hipMemcpyAsync(text_roff, d_text, (size_t)ttotal * 4, hipMemcpyDeviceToHost, sm));
        }
    }
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    float ms = 0;
    PHZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms[PHZ_T_MAP] = ms; ctx->total_ms[PHZ_T_MAP] += ms; ctx->launches[PHZ_T_MAP]++;
    return PHZ_OK;
}
