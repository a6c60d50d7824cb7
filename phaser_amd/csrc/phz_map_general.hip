// K_map_general: the mapper for variant sets that are not pure SNPs (phASER's --include_indels 1): REF longer than one
// base and/or multi-base alleles.  Same rule as K_map -- phaser/read_variant_map.py:236-258 with xvar.ref_length > 1:
//   emit iff 0 <= rs and rs + ref_len <= len(pseudo_read of the segment); text = pseudo[rs:rs+ref_len] with the segment's
//   insertions spliced after their (read-relative, quirk kept) keys and 'D' placeholders removed; "" and "N" suppressed
// but the text can be any length, so the call is classified on the device against the individual's two allele strings:
//   code 5 = text equals allele 0, 6 = equals allele 1 (first match wins, as `list.index` does at phaser.py:1317),
//   0..3 = some other single base, 4 = any other text (its read offsets go to an optional pool so the host can print it).
// This mode is off by default in phASER ("will likely result in poor quality phasing", phaser.py:48).  Pipeline of one call:
//   k_gen_window / k_gen_desc  per tile of 1024 records the window of variants it can touch; per variant a packed descriptor
//                              (REF length, the two alleles when they are single characters)
//   k_map_general              fast pass, four records per lane: window + the tile's CIGAR words staged in LDS, an LDS-only walk that
//                              collects up to two candidates per record, then all their (qual, seq) bytes requested together,
//                              classified from the descriptor; leaves per-record counts, the calls packed in side[r], tile totals.
//                              A record with anything else under it (REF > 1 base, an I / D / G op in the segment, a third
//                              candidate, a step outside the staged data) is handed to a work list instead of being walked in
//                              place: one such lane used to hold its wave for the whole general composition.
//   k_map_general_list<false>  the complete rule (gen_read) for the listed records, dense waves; records whose result is at most
//                              two calls without text are packed like the fast ones
//   scan over ~n/1024 tile totals; host reads the totals and sizes the output
//   k_gen_emit                 streams side[r] to each record's place (tile base + an in-workgroup scan of the counts)
//   k_map_general_list<true>   writes the calls of the few listed records that did not fit the packed form
// Measured on the configs[1] shard (50M records, 40k het SNPs as allele strings, 10.2M calls): 3.6 ms before this layout, 1.6-1.7 ms
// now, K_map on the same shard 0.90 ms; the fast pass is bound by instruction issue (2,076 vector + 1,730 scalar instructions per wave of 256 records).
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
    const int32_t *win;                          // [2 * grid] window start, length per workgroup (k_gen_window)
    const uint32_t *desc;                        // [nv] ref_len | allele0 char << 8 | allele1 char << 16 (0 = not one character)
    uint4 *side;                                 // [n] packed calls of the fast pass (k_map_general -> k_gen_emit)
    uint32_t *wl_n, *wl;                         // records left to the general composition (count pass appends, both list passes read)
    uint32_t *n_calls;                           // per record (count pass)
    uint32_t *tile_calls, *tile_text;            // per workgroup tile of the fast pass: calls / text characters under its records
    const uint32_t *tile_cbase, *tile_tbase;     // their exclusive scans
    uint32_t *call_base, *text_base;             // per record, written by k_gen_emit for the handed-over records only
    int32_t *o_read, *o_var; uint8_t *o_code; uint32_t *o_aux0, *o_aux1;
    uint32_t *o_text_off; uint32_t *o_text;      // optional
    int64_t cap, text_cap;
};

// "ACGTN?"[sym] without the trip to constant memory
__device__ __forceinline__ uint32_t sym_char(int sym) { return (uint32_t)((0x3F4E54474341ull >> (8 * sym)) & 0xFFu); }
__device__ __forceinline__ int sym_of(int baseq, uint32_t q, uint32_t sbyte, int x) {
    const uint32_t s = (sbyte >> (2 * (x & 3))) & 3;
    if ((int)(q & 0x7f) < baseq) return 4;
    if (q & 0x80) return s == 0 ? 4 : 5;
    return (int)s;
}
__device__ __forceinline__ int sym_at(const GenArgs &a, uint32_t soff, int x) {
    return sym_of(a.baseq, a.qual[(size_t)soff * 4 + x], a.seq2[(size_t)soff + (x >> 2)], x);
}

struct Compose {
    int n, first;
    bool m0, m1;
    uint32_t p0, l0, p1, l1;
    const char *ab;
    uint32_t *text;           // where read offsets of this call's characters go (EMIT + pool present), else nullptr
    int64_t text_room;
    __device__ __forceinline__ void add(int sym, uint32_t roff) {
        const char ch = (char)sym_char(sym);
        if (n == 0) first = sym;
        m0 = m0 && ((uint32_t)n < l0) && ab[p0 + n] == ch;
        m1 = m1 && ((uint32_t)n < l1) && ab[p1 + n] == ch;
        if (text && n < text_room) text[n] = roff;
        n++;
    }
};

// first index in [lo, nv) with vpos >= key, galloping from lo (the answer is almost always a few entries away)
template <class VP>
__device__ __forceinline__ int gallop_lb(const VP &vpos, int nv, int lo, long long key) {
    if (lo >= nv || (long long)vpos(lo) >= key) return lo;
    int step = 1;
    while (lo + step < nv && (long long)vpos(lo + step) < key) { lo += step; step <<= 1; }
    int l = lo + 1, h = lo + step < nv ? lo + step : nv;
    while (l < h) { const int m = (l + h) >> 1; if ((long long)vpos(m) < key) l = m + 1; else h = m; }
    return l;
}

constexpr int GEN_RPL = 4;         // records per lane: their loads are requested together, stage by stage
constexpr int GEN_TILE = 256 * GEN_RPL;
constexpr int GEN_CIG = 4 * GEN_TILE;   // CIGAR words staged per workgroup (max); the rest is read from global memory
constexpr int GEN_WIN = 1024;      // variants staged per workgroup (max); beyond it the same arrays are read from global memory
constexpr int GEN_COVER = 65536;   // the staged window reaches POS(last record of the workgroup) + GEN_COVER

// The complete rule for one record (any REF length, any CIGAR): run by k_map_general_list on the records the fast pass hands over.
// w0: first variant at or after the POS of the first record of r's tile (k_gen_window): every search starts there or later.
template <bool EMIT>
__device__ void gen_read(const GenArgs &a, int64_t r, int w0) {
    uint32_t ncalls = 0, ntext = 0, pkv[2] = {0u, 0u}, pkw[2] = {0u, 0u};
    // emit pass: nothing under this record, or its calls went out with the packed ones (k_gen_emit)
    if (EMIT && (a.n_calls[r] == 0 || !(a.side[r].y & 0x40000000u))) return;
    const uint64_t cbase = EMIT ? a.call_base[r] : 0, tbase = EMIT ? a.text_base[r] : 0;
    const int pos = a.pos[r];
    const uint32_t c0 = a.cigar_off[r], c1 = a.cigar_off[r + 1], soff = a.seq_off[r];
    auto VP = [&](int i) -> int { return a.vpos[i]; };
    auto CIG = [&](uint32_t kx) -> uint32_t { return a.cigar[kx]; };
    auto DESC = [&](int i) -> uint32_t { return a.desc[i]; };
    int gpos = 0, rpos = 0, istart = w0;        // segments move forward: each search starts where the previous one ended
    uint32_t k = c0;
    for (;;) {
        const int seg_start = gpos, seg_rpos = rpos;
        int plen = 0;
        uint32_t k2 = k;
        bool plain = true;            // no I / D / G op in the segment: pseudo-read index <-> read offset is a walk over M runs
        for (; k2 < c1; k2++) {
            const uint32_t w = CIG(k2), op = w & 15;
            if (op == OP_N) break;
            if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) plen += (int)(w >> 4);
            plain = plain && op != OP_I && op != OP_D && op != OP_G;
        }
        const long long lo = (long long)pos + seg_start;
        int i = gallop_lb(VP, a.nv, istart, lo);
        for (; i < a.nv; i++) {
            const long long vp = VP(i);
            if (vp >= lo + plen) break;
            const uint32_t d = DESC(i);
            const int rs = (int)(vp - lo), rl = (int)(d & 0xFFu);
            if (rs + rl > plen) continue;
            if (plain && rl == 1) {
                int pi = 0, ro = seg_rpos, x = 0;
                for (uint32_t kk = k; kk < k2; kk++) {
                    const uint32_t w = CIG(kk), op = w & 15;
                    const int len = (int)(w >> 4);
                    if (op == OP_M || op == OP_EQ || op == OP_X) {
                        if (rs < pi + len) { x = ro + (rs - pi); break; }
                        pi += len; ro += len;
                    } else if (op == OP_S) ro += len;
                }
                const int sym = sym_at(a, soff, x);
                if (sym == 4) continue;
                const uint32_t ch = sym_char(sym);
                const int code = ch == ((d >> 8) & 0xFFu) ? 5 : ch == ((d >> 16) & 0xFFu) ? 6 : sym < 4 ? sym : 4;
                if (EMIT) {
                    const int64_t o = (int64_t)(cbase + ncalls);
                    if (o < a.cap) {
                        a.o_read[o] = (int32_t)r; a.o_var[o] = i; a.o_code[o] = (uint8_t)code;
                        a.o_aux0[o] = 0xFFFFFFFFu; a.o_aux1[o] = 0;
                        if (a.o_text_off) a.o_text_off[o] = (uint32_t)(tbase + ntext);
                    }
                    if (code == 4 && a.o_text && (int64_t)(tbase + ntext) < a.text_cap) a.o_text[tbase + ntext] = (uint32_t)x;
                }
                if (!EMIT && ncalls < 2) { pkv[ncalls] = (uint32_t)i; pkw[ncalls] = ((uint32_t)code << 24) | 0x80000000u; }
                ncalls++;
                if (code == 4) ntext++;
                continue;
            }
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
                    const uint32_t w = CIG(kk), op = w & 15;
                    const int len = (int)(w >> 4);
                    if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) {
                        const int from = pi > rs ? pi : rs, to = (pi + len) < (rs + rl) ? (pi + len) : (rs + rl);
                        for (int p = from; p < to; p++) {
                            if (op != OP_D) c.add(sym_at(a, soff, ro + (p - pi)), (uint32_t)(ro + (p - pi)));
                            // insertion stored under key == p in this segment (later one wins; key is read-relative)
                            int ioff = 0, ilen = 0, g2 = seg_start, r2 = seg_rpos;
                            for (uint32_t q = k; q < k2; q++) {
                                const uint32_t w2 = CIG(q), o2 = w2 & 15;
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
            if (!EMIT && ncalls < 2) { pkv[ncalls] = (uint32_t)i; pkw[ncalls] = ((uint32_t)code << 24) | 0x80000000u; }
            ncalls++;
            if (code == 4) ntext += (uint32_t)c.n;
        }
        for (uint32_t kk = k; kk < k2; kk++) {
            const uint32_t w = CIG(kk), op = w & 15;
            const int len = (int)(w >> 4);
            if (op == OP_M || op == OP_EQ || op == OP_X) { gpos += len; rpos += len; }
            else if (op == OP_D || op == OP_G) gpos += len;
            else if (op == OP_I || op == OP_S) rpos += len;
        }
        if (k2 >= c1) break;
        gpos += (int)(CIG(k2) >> 4);
        k = k2 + 1;
    }
    if (!EMIT) {
        // at most two calls and no text: the record's calls fit the fast pass's packed form and k_gen_emit writes them; otherwise
        // the record stays marked for k_map_general_list<true>
        a.n_calls[r] = ncalls;
        a.side[r] = ncalls <= 2 && ntext == 0 ? make_uint4(pkv[0], pkw[0], pkv[1], pkw[1]) : make_uint4(ncalls, 0x40000000u, ntext, 0u);
        if (ncalls) atomicAdd(&a.tile_calls[r / GEN_TILE], ncalls);
        if (ntext) atomicAdd(&a.tile_text[r / GEN_TILE], ntext);
    }
}

// per workgroup of `tile` records: first variant at or after the first record's POS, and how many variants lie below
// POS(last record) + GEN_COVER, plus the one after them (at most GEN_WIN); later records only move forward from the start
__global__ void k_gen_window(const int32_t *pos, int64_t n, int tile, const int32_t *vpos, int nv, int64_t grid, int32_t *win) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= grid) return;
    const int64_t last = (b + 1) * tile - 1 < n ? (b + 1) * tile - 1 : n - 1;
    const long long k0 = pos[b * tile], k1 = (long long)pos[last] + GEN_COVER;
    int l0 = 0, h0 = nv, l1 = 0, h1 = nv;
    while (l0 < h0 || l1 < h1) {
        if (l0 < h0) { const int m = (l0 + h0) >> 1; if ((long long)vpos[m] < k0) l0 = m + 1; else h0 = m; }
        if (l1 < h1) { const int m = (l1 + h1) >> 1; if ((long long)vpos[m] < k1) l1 = m + 1; else h1 = m; }
    }
    int len = l1 - l0 + 1;                      // one past the cover: the entry that ends most searches
    if (len > nv - l0) len = nv - l0;
    win[2 * b] = l0; win[2 * b + 1] = len < 0 ? 0 : (len > GEN_WIN ? GEN_WIN : len);
}

__global__ void k_gen_desc(const uint8_t *ref_len, const uint32_t *aoff, const char *ab, int nv, uint32_t *desc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const uint32_t p0 = aoff[2 * i], p1 = aoff[2 * i + 1], p2 = aoff[2 * i + 2];
    const uint32_t c0 = p1 - p0 == 1 ? (uint8_t)ab[p0] : 0u, c1 = p2 - p1 == 1 ? (uint8_t)ab[p1] : 0u;
    desc[i] = (uint32_t)ref_len[i] | (c0 << 8) | (c1 << 16);
}

constexpr int GEN_NC = 2;          // candidates a record keeps in registers in the fast pass; a record with more goes to the work list
static_assert(GEN_NC == 2, "side[r] packs two calls per record");

// Fast pass, step 1: walk one record against the staged window without touching its bases.  Returns the number of candidates
// (variant index, descriptor, read offset of the base under it), or -1 when the record has to take the complete rule: a candidate
// whose REF is longer than a base or whose segment holds an I / D / G op, or more than GEN_NC candidates.  The general composition is
// a long chain of dependent loads, and one such lane used to hold the other 63 of its wave for its whole length; the handed-over
// records are redone from scratch by dense waves (k_map_general_list).
__device__ __forceinline__ int walk_fast(const GenArgs &a, int w0, int wlen, const int32_t *s_vpos, const uint32_t *s_desc, int pos, uint32_t c0,
                                         uint32_t c1, uint32_t cb, uint32_t ncig, const uint32_t *s_cig, int *ci, uint32_t *cd, int *cx) {
    // Everything the fast walk reads is in LDS, through explicit LDS pointers and 32-bit arithmetic: a step outside the staged window
    // or the staged CIGAR words (or a coordinate near 2^30) hands the record over instead of reaching for global memory, which keeps
    // the per-step instruction count down -- the pass is bound by VALU issue, not by latency.
    typedef const __attribute__((address_space(3))) int32_t *lds_i32;
    typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
    const lds_i32 l_vpos = (lds_i32)s_vpos;
    const lds_u32 l_desc = (lds_u32)s_desc, l_cig = (lds_u32)s_cig;
    if ((unsigned)pos >= (1u << 30) || c1 - cb > ncig) return -1;
    const int wend = w0 + wlen;                 // staged variants: [w0, wend); wend == nv or vpos[wend - 1] is past the cover
    int gpos = 0, rpos = 0, i = w0, cnt = 0;
    bool bail = false;                          // sticky: the record goes to the work list (no early returns: every divergent exit
    uint32_t k = c0 - cb, kend = c1 - cb;       // costs a handful of exec-mask instructions, and the pass is bound by those)
    bool more = true;
    while (more) {
        const int seg_start = gpos, seg_rpos = rpos;
        int plen = 0, runs = 0, run_ro = 0;      // runs of bases in the segment, read offset of the first one
        uint32_t k2 = k;
        bool plain = true;
        for (; k2 < kend; k2++) {                 // op classes as bit sets, selects instead of branches
            const uint32_t w = l_cig[k2], op = w & 15, bit = 1u << op;
            const int len = (int)(w >> 4);
            if (op == OP_N) break;
            const bool bases = (bit & 0x181u) != 0;                          // M = X
            run_ro = (bases && runs == 0) ? rpos : run_ro;
            runs += bases ? 1 : 0;
            plen += (bit & 0x185u) ? len : 0;                                // M = X D
            gpos += (bit & 0x385u) ? len : 0;                                // M = X D G
            rpos += (bit & 0x193u) ? len : 0;                                // M = X I S
            plain = plain && (bit & 0x206u) == 0;                            // no I D G
        }
        bail = bail || (unsigned)gpos >= (1u << 30);
        const int lo = pos + seg_start, hi = lo + plen;
        // first staged variant at or after lo, galloping from where the previous segment ended
        if (i < wend && l_vpos[i - w0] < lo) {
            int step = 1;
            while (i + step < wend && l_vpos[i + step - w0] < lo) { i += step; step <<= 1; }
            int l = i + 1, h = i + step < wend ? i + step : wend;
            while (l < h) { const int m = (l + h) >> 1; if (l_vpos[m - w0] < lo) l = m + 1; else h = m; }
            i = l;
        }
        bool go = !bail;
        while (go) {
            const bool inw = i < wend;
            const int t = inw ? i - w0 : 0;
            const int vp = l_vpos[t];
            const uint32_t d = l_desc[t];
            bail = bail || (!inw && wend < a.nv);                            // ran off the staged window with variants left
            const bool inside = inw && vp < hi;
            const int rs = vp - lo, rl = (int)(d & 0xFFu);
            const bool cand = inside && rs + rl <= plen;
            bail = bail || (cand && (!(plain && rl == 1) || cnt >= GEN_NC));
            const bool take = cand && !bail;
            int x = run_ro + rs;                 // one run of bases in a plain segment: the usual case
            if (take && runs > 1) {
                int pi = 0, ro = seg_rpos;
                for (uint32_t kk = k; kk < k2; kk++) {
                    const uint32_t w = l_cig[kk], op = w & 15;
                    const int len = (int)(w >> 4);
                    if (op == OP_M || op == OP_EQ || op == OP_X) {
                        if (rs < pi + len) { x = ro + (rs - pi); break; }
                        pi += len; ro += len;
                    } else if (op == OP_S) ro += len;
                }
            }
#pragma unroll
            for (int c = 0; c < GEN_NC; c++) {
                const bool put = take && c == cnt;
                ci[c] = put ? i : ci[c]; cd[c] = put ? d : cd[c]; cx[c] = put ? x : cx[c];
            }
            cnt += take ? 1 : 0;
            go = inside && !bail;
            i += go ? 1 : 0;
        }
        more = !bail && k2 < kend;
        gpos += more ? (int)(l_cig[k2 < kend ? k2 : 0u] >> 4) : 0;
        k = k2 + 1;
    }
    return bail ? -1 : cnt;
}

// Fast pass (count): GEN_RPL records per lane.  Besides the per-record counts it leaves each record's calls packed in side[r]
// ({variant, offset | code << 24 | 1 << 31} x GEN_NC; .y == 1 << 30 marks a handed-over record), so that the emit pass is a plain
// streaming kernel and nothing is walked twice.
__global__ __launch_bounds__(256) void k_map_general(GenArgs a) {
    __shared__ int32_t s_vpos[GEN_WIN];
    __shared__ uint32_t s_desc[GEN_WIN];
    __shared__ uint32_t s_cig[GEN_CIG];
    __shared__ uint32_t s_wl[GEN_TILE], s_wln, s_wlbase, s_sum[8];
    uint32_t cb, ncig;
    if (threadIdx.x == 0) s_wln = 0;
    const int64_t r0 = (int64_t)blockIdx.x * GEN_TILE;
    const int w0 = a.win[2 * blockIdx.x], wlen = a.win[2 * blockIdx.x + 1];
    // stage 1: the record words of this lane's GEN_RPL records
    int pos[GEN_RPL]; uint32_t c0[GEN_RPL], c1[GEN_RPL], soff[GEN_RPL]; bool live[GEN_RPL];
#pragma unroll
    for (int j = 0; j < GEN_RPL; j++) {
        const int64_t r = r0 + j * 256 + threadIdx.x;
        live[j] = r < a.n;
        pos[j] = 0; c0[j] = c1[j] = soff[j] = 0;
        if (live[j]) { pos[j] = a.pos[r]; c0[j] = a.cigar_off[r]; c1[j] = a.cigar_off[r + 1]; soff[j] = a.seq_off[r]; }
    }
    for (int t = threadIdx.x; t < wlen; t += 256) { s_vpos[t] = a.vpos[w0 + t]; s_desc[t] = a.desc[w0 + t]; }
    // stage 2: the tile's CIGAR words (one contiguous range, records being stored in order) into LDS
    {
        const int64_t re = r0 + GEN_TILE < a.n ? r0 + GEN_TILE : a.n;
        cb = a.cigar_off[r0];
        const uint32_t ce = a.cigar_off[re];
        ncig = ce - cb < (uint32_t)GEN_CIG ? ce - cb : (uint32_t)GEN_CIG;
        for (uint32_t t = threadIdx.x; t < ncig; t += 256) s_cig[t] = a.cigar[cb + t];
    }
    __syncthreads();
    // stage 3: walk (LDS only), then request the bytes under every candidate of the lane's records together
    int cnt[GEN_RPL], ci[GEN_RPL][GEN_NC], cx[GEN_RPL][GEN_NC]; uint32_t cd[GEN_RPL][GEN_NC], q[GEN_RPL][GEN_NC], sb[GEN_RPL][GEN_NC];
#pragma unroll
    for (int j = 0; j < GEN_RPL; j++) {
#pragma unroll
        for (int c = 0; c < GEN_NC; c++) { ci[j][c] = 0; cx[j][c] = 0; cd[j][c] = 0; }
        cnt[j] = live[j] ? walk_fast(a, w0, wlen, s_vpos, s_desc, pos[j], c0[j], c1[j], cb, ncig, s_cig, ci[j], cd[j], cx[j]) : 0;
    }
#pragma unroll
    for (int j = 0; j < GEN_RPL; j++) {
#pragma unroll
        for (int c = 0; c < GEN_NC; c++) {
            q[j][c] = sb[j][c] = 0;
            if (c < cnt[j]) { q[j][c] = a.qual[(size_t)soff[j] * 4 + cx[j][c]]; sb[j][c] = a.seq2[(size_t)soff[j] + (cx[j][c] >> 2)]; }
        }
    }
    uint32_t sum_c = 0, sum_t = 0;
#pragma unroll
    for (int j = 0; j < GEN_RPL; j++) {
        const int64_t r = r0 + j * 256 + threadIdx.x;
        if (!live[j]) continue;
        if (cnt[j] < 0) {                                         // hand-over, gathered per workgroup; the list kernel writes its counts
            s_wl[atomicAdd(&s_wln, 1u)] = (uint32_t)r;
            a.side[r] = make_uint4(0u, 0x40000000u, 0u, 0u);
            continue;
        }
        uint32_t ncalls = 0, ntext = 0, pk[2 * GEN_NC];
#pragma unroll
        for (int c = 0; c < 2 * GEN_NC; c++) pk[c] = 0;
#pragma unroll
        for (int c = 0; c < GEN_NC; c++) {
            if (c >= cnt[j]) break;
            const int x = cx[j][c];
            const int sym = sym_of(a.baseq, q[j][c], sb[j][c], x);
            if (sym == 4) continue;
            const uint32_t d = cd[j][c], ch = sym_char(sym);
            const uint32_t code = ch == ((d >> 8) & 0xFFu) ? 5u : ch == ((d >> 16) & 0xFFu) ? 6u : sym < 4 ? (uint32_t)sym : 4u;
            const uint32_t w = ((uint32_t)x & 0xFFFFFFu) | (code << 24) | 0x80000000u;
#pragma unroll
            for (int e = 0; e < GEN_NC; e++) if ((uint32_t)e == ncalls) { pk[2 * e] = (uint32_t)ci[j][c]; pk[2 * e + 1] = w; }
            ncalls++;
            if (code == 4) ntext++;
        }
        a.n_calls[r] = ncalls;
        if (ncalls) a.side[r] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        sum_c += ncalls; sum_t += ntext;
    }
    // the tile's totals over its fast records (the list pass adds the handed-over ones)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { sum_c += __shfl_xor(sum_c, d); sum_t += __shfl_xor(sum_t, d); }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum_c; s_sum[4 + (threadIdx.x >> 6)] = sum_t; }
    // one global atomic per workgroup: a shared counter hit once per wave cost more than the rest of the pass
    __syncthreads();
    if (threadIdx.x == 0) {
        a.tile_calls[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        a.tile_text[blockIdx.x] = s_sum[4] + s_sum[5] + s_sum[6] + s_sum[7];
    }
    if (s_wln) {
        if (threadIdx.x == 0) s_wlbase = atomicAdd(a.wl_n, s_wln);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < s_wln; t += 256) a.wl[s_wlbase + t] = s_wl[t];
    }
}

// Emit, fast records: one workgroup per tile of the count pass, four consecutive records per thread.  The offsets of a record's
// calls are the tile's base (a scan over ~n/1024 tile totals) plus a scan inside the workgroup, so no per-record scan of the whole
// shard exists; text characters are counted from the packed calls themselves.  Handed-over records only get their two offsets
// written down for k_map_general_list<true>.
static_assert(GEN_TILE == 1024, "k_gen_emit: 256 threads x 4 records");
__global__ __launch_bounds__(256) void k_gen_emit(GenArgs a) {
    __shared__ uint32_t s_c[4], s_t[4];
    if (a.tile_cbase[blockIdx.x + 1] == a.tile_cbase[blockIdx.x]) return;
    const int64_t rb = (int64_t)blockIdx.x * GEN_TILE + 4 * threadIdx.x;
    uint32_t nc[4] = {0u, 0u, 0u, 0u}, nt[4] = {0u, 0u, 0u, 0u};
    uint4 sd[4];
    if (rb + 3 < a.n) { const uint4 v = *(const uint4 *)(a.n_calls + rb); nc[0] = v.x; nc[1] = v.y; nc[2] = v.z; nc[3] = v.w; }
    else { for (int j = 0; j < 4; j++) if (rb + j < a.n) nc[j] = a.n_calls[rb + j]; }
    uint32_t tc = 0, tt = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        sd[j] = make_uint4(0u, 0u, 0u, 0u);
        if (nc[j]) {
            sd[j] = a.side[rb + j];
            if (sd[j].y & 0x40000000u) nt[j] = sd[j].z;
            else nt[j] = (uint32_t)(((sd[j].y >> 24) & 0x87u) == 0x84u) + (uint32_t)(((sd[j].w >> 24) & 0x87u) == 0x84u);
        }
        tc += nc[j]; tt += nt[j];
    }
    uint32_t ic = tc, it = tt;                                    // inclusive scans over the wave, then over the four waves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t uc = __shfl_up(ic, d), ut = __shfl_up(it, d);
        if (lane >= d) { ic += uc; it += ut; }
    }
    if (lane == 63) { s_c[wv] = ic; s_t[wv] = it; }
    __syncthreads();
    uint32_t ec = a.tile_cbase[blockIdx.x] + (ic - tc), et = a.tile_tbase[blockIdx.x] + (it - tt);
    for (int w = 0; w < wv; w++) { ec += s_c[w]; et += s_t[w]; }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!nc[j]) continue;
        const int64_t r = rb + j;
        if (sd[j].y & 0x40000000u) { a.call_base[r] = ec; a.text_base[r] = et; }
        else {
            uint32_t ntext = 0;
#pragma unroll
            for (int c = 0; c < GEN_NC; c++) {
                const uint32_t var = c ? sd[j].z : sd[j].x, w = c ? sd[j].w : sd[j].y;
                if (!(w & 0x80000000u)) break;
                const uint32_t code = (w >> 24) & 7u;
                const int64_t o = (int64_t)ec + c;
                if (o < a.cap) {
                    a.o_read[o] = (int32_t)r; a.o_var[o] = (int32_t)var; a.o_code[o] = (uint8_t)code;
                    a.o_aux0[o] = 0xFFFFFFFFu; a.o_aux1[o] = 0;
                    if (a.o_text_off) a.o_text_off[o] = et + ntext;
                }
                if (code == 4) {
                    if (a.o_text && (int64_t)et + ntext < a.text_cap) a.o_text[(size_t)et + ntext] = w & 0xFFFFFFu;
                    ntext++;
                }
            }
        }
        ec += nc[j]; et += nt[j];
    }
}

// terminator of the per-call text offsets: call_text_off[total] = text total
__global__ void k_gen_tail(const uint32_t *total, const uint32_t *ttotal, uint32_t *toff, int64_t cap) {
    if ((int64_t)*total <= cap) toff[*total] = *ttotal;
}

// the records the fast pass left: one lane each, dense
template <bool EMIT>
__global__ __launch_bounds__(256) void k_map_general_list(GenArgs a) {
    const uint32_t m = *a.wl_n;
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < m; t += gridDim.x * 256) {
        const int64_t r = a.wl[t];
        const int w0 = a.win[2 * (r / GEN_TILE)];
        gen_read<EMIT>(a, r, w0);
    }
}

}  // namespace

extern "C" int phz_map_reads_general(phz_ctx *ctx, const phz_reads *reads, const phz_variants_general *vars, int baseq,
                                     phz_calls *out, int64_t *n_calls, uint32_t *call_text_off, uint32_t *text_roff,
                                     int64_t text_cap, int64_t *n_text, int space) {
    PhzEnter phz_guard_(ctx);
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
    const unsigned grid = (unsigned)((n + GEN_TILE - 1) / GEN_TILE);
    if (int s = phz_reserve(ctx, S[1], (size_t)(grid + 1) * 16)) return s;
    if (int s = phz_reserve(ctx, S[2], (size_t)n * 4)) return s;
    if (int s = phz_reserve(ctx, S[3], (size_t)n * 4)) return s;
    a.n_calls = (uint32_t *)S[0].p;
    a.tile_calls = (uint32_t *)S[1].p; a.tile_text = a.tile_calls + grid;
    uint32_t *cb = a.tile_text + grid, *tb = cb + grid + 1;
    a.tile_cbase = cb; a.tile_tbase = tb;
    a.call_base = (uint32_t *)S[2].p; a.text_base = (uint32_t *)S[3].p;
    hipStream_t sm = ctx->stream;
    if (int s = phz_reserve(ctx, S[4], (size_t)grid * 8)) return s;
    if (int s = phz_reserve(ctx, S[5], (size_t)nv * 4)) return s;
    if (n >= (1ll << 32)) return phz_fail(ctx, PHZ_E_ARG, "too many records in one shard");
    if (int s = phz_reserve(ctx, S[7], (size_t)n * 4)) return s;
    if (int s = phz_reserve(ctx, S[8], 64)) return s;
    if (int s = phz_reserve(ctx, S[9], (size_t)n * 16)) return s;
    a.side = (uint4 *)S[9].p;
    a.win = (const int32_t *)S[4].p; a.desc = (const uint32_t *)S[5].p;
    a.wl = (uint32_t *)S[7].p; a.wl_n = (uint32_t *)S[8].p;
    PHZ_HIP(ctx, hipMemsetAsync(a.wl_n, 0, 4, sm));
    PHZ_HIP(ctx, hipEventRecord(ctx->ev0, sm));
    hipLaunchKernelGGL(k_gen_window, dim3((grid + 255) / 256), dim3(256), 0, sm, a.pos, n, GEN_TILE, a.vpos, (int)nv, (int64_t)grid, (int32_t *)S[4].p);
    hipLaunchKernelGGL(k_gen_desc, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, sm, a.ref_len, a.aoff, a.abytes, (int)nv, (uint32_t *)S[5].p);
    hipLaunchKernelGGL(k_map_general, dim3(grid), dim3(256), 0, sm, a);
    hipLaunchKernelGGL(k_map_general_list<false>, dim3(2048), dim3(256), 0, sm, a);     // grid-stride over a list whose length only the device knows
    if (int s = scan_excl(ctx, a.tile_calls, cb, (int64_t)grid, S[6])) return s;
    if (int s = scan_excl(ctx, a.tile_text, tb, (int64_t)grid, S[6])) return s;
    uint32_t last[2], n_listed = 0;
    if (space == PHZ_DEVICE) {
        // device-resident outputs already exist at their capacity: nothing on the host has to know the totals before the emit
        // kernels run (every store is bounded by cap / text_cap), so the whole call is queued at once and waited for once
        a.o_read = out->read_idx; a.o_var = out->var_idx; a.o_code = out->code; a.o_aux0 = out->aux0; a.o_aux1 = out->aux1;
        a.o_text_off = (call_text_off && text_roff) ? call_text_off : nullptr; a.o_text = (call_text_off && text_roff) ? text_roff : nullptr;
        a.cap = out->cap; a.text_cap = text_cap;
        hipLaunchKernelGGL(k_gen_emit, dim3(grid), dim3(256), 0, sm, a);
        hipLaunchKernelGGL(k_map_general_list<true>, dim3(2048), dim3(256), 0, sm, a);
        if (a.o_text_off) hipLaunchKernelGGL(k_gen_tail, dim3(1), dim3(1), 0, sm, (const uint32_t *)(cb + grid), (const uint32_t *)(tb + grid), a.o_text_off, a.cap);
        PHZ_HIP(ctx, hipGetLastError());
        PHZ_HIP(ctx, hipEventRecord(ctx->ev1, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(&n_listed, a.wl_n, 4, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(&last[0], cb + grid, 4, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(&last[1], tb + grid, 4, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipStreamSynchronize(sm));
        if (getenv("PHZ_GEN_DBG")) fprintf(stderr, "K_map_general: %lld records, %u handed to the list, %u calls, %u text\n", (long long)n, n_listed, last[0], last[1]);
        *n_calls = (int64_t)last[0];
        if (n_text) *n_text = (int64_t)last[1];
        float ms = 0;
        PHZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        ctx->last_ms[PHZ_T_MAP] = ms; ctx->total_ms[PHZ_T_MAP] += ms; ctx->launches[PHZ_T_MAP]++;
        if ((int64_t)last[0] > out->cap || (text_roff && (int64_t)last[1] > text_cap)) return PHZ_E_CAPACITY;
        return PHZ_OK;
    }
    PHZ_HIP(ctx, hipMemcpyAsync(&n_listed, a.wl_n, 4, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipMemcpyAsync(&last[0], cb + grid, 4, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipMemcpyAsync(&last[1], tb + grid, 4, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    const int64_t total = (int64_t)last[0], ttotal = (int64_t)last[1];
    if (getenv("PHZ_GEN_DBG")) fprintf(stderr, "K_map_general: %lld records, %u handed to the list, %lld calls, %lld text\n", (long long)n, n_listed, (long long)total, (long long)ttotal);
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
    hipLaunchKernelGGL(k_gen_emit, dim3(grid), dim3(256), 0, sm, a);
    if (n_listed) hipLaunchKernelGGL(k_map_general_list<true>, dim3((n_listed + 255) / 256), dim3(256), 0, sm, a);
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
