// K_map: read -> het-variant allele mapper for gfx950 (MI355X).
//
// Computes what phaser/read_variant_map.py:3-258 computes (do_read_variant_map -> split_read ->
// identify_allele), restated as a stateless rule per (record, N-split segment, variant) -- SURVEY.md 3.3:
//   emit iff  seg_start <= var.pos - POS  and  var.pos - POS + ref_len <= seg_start + len(pseudo_read)
//   and the extracted text is neither "" nor "N".
// Integer / branch work on an HBM-resident structure of arrays; no MFMA.
//
// Launch shape: one 128-thread workgroup per tile of 256 consecutive (coordinate-sorted) reads (template
// parameters; 64..256 threads x 2..4 reads per lane were swept on MI355X, 128 x 2 is fastest), two
// reads per lane strided by the workgroup size so every streaming load (pos, cigar_off, seq_off, cigar words) is coalesced.
// Per tile (k_map):
//   1. LDS staging with coalesced block-wide loads: the tile's cigar_off slice, its contiguous run of packed
//      CIGAR words, and the het-SNP window that starts at lower_bound(vpos, POS of the tile's first read),
//      sized by a pre-pass (k_tile_window) to what the tile's reads reach without introns; lanes binary-search
//      the window per aligned run and fall through to global memory only for introns reaching past it,
//   2. records made of one aligned run (the common case) stay in registers: a branch-free LDS binary search
//      with a tile-uniform trip count yields (first het SNP, count); their seq/qual bytes are gathered in a
//      per-lane `for o < count` loop, so iteration o is one gather for every lane that has an o-th SNP,
//   3. spliced / gapped / clipped records are re-packed densely and walk their CIGAR out of LDS; their
//      candidates go to an LDS buffer tagged (read, ordinal) and are resolved by all lanes at once
//      (seq/qual bytes are touched only under a variant: two bytes per call),
//   4. a wave shuffle scan + LDS combine turns per-read call counts into offsets and the calls are flushed in
//      exact mapper order into the tile's slot of a staging area (tile x slot_cap calls); a tile whose
//      candidates overflow the LDS buffer resolves its complex records in-lane instead.
// k_chunk_scan / k_chunk_base prefix-sum the per-tile totals; k_compact copies every slot to its final
// offset, so the call list is in mapper order with no inter-workgroup dependency inside k_map (a
// decoupled look-back was measured 0.8 ms slower here: tiles finish faster than descriptors travel).
#include "phz_internal.h"
#include <stdlib.h>
#include <string.h>

namespace {

#ifndef PHZ_MAP_WIN
#define PHZ_MAP_WIN 512
#endif
#ifndef PHZ_CIG_X2
#define PHZ_CIG_X2 5
#endif
#ifndef PHZ_CAND_X4
#define PHZ_CAND_X4 3
#endif
constexpr int MAP_WIN = PHZ_MAP_WIN;       // het-SNP positions staged per tile (max)

constexpr uint32_t OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_EQ = 7, OP_X = 8, OP_G = 9;

struct MapArgs {
    const int32_t *pos;
    const uint32_t *cigar_off, *cigar, *seq_off;
    const uint8_t *seq2, *qual;
    const uint8_t *bq;            // one byte per base (phz_reads.bq), or NULL
    int64_t n;
    const int32_t *vpos;
    int nv;
    int baseq;
    // staging slots: global tile t owns stage[t*slot_cap, (t+1)*slot_cap); one packed 8-byte record per call (stage_put):
    // one store per call here, one load in k_compact
    uint2 *stage;
    uint32_t *side;               // [2*slots] aux words that do not fit the packed record (insertions, offsets >= 2^16)
    int64_t slots;
    const int32_t *tile_w0;       // [4*ntiles]: window start, window length | complete flag, first CIGAR word, CIGAR word count
    int32_t *tile_total;          // [ntiles] calls per tile
    int slot_cap;
    int64_t ntiles;
    int dbg;          // ablation switches for profiling (PHZ_MAP_DBG env); 0 in production
    bool one;         // base and quality of a call from the one-byte plane `bq` (a compile-time constant of the instantiation, like dbg == 0)
    const struct ShardDev *shp;   // ONE instantiation: the shard record, read again by the rare escape (seq2 / qual stay out of the registers)
};

// one shard of a batch; the array lives in device memory for the duration of the launches
struct ShardDev {
    const int32_t *pos;
    const uint32_t *cigar_off, *cigar, *seq_off;
    const uint8_t *seq2, *qual;
    const uint8_t *bq;
    int64_t n;
    const int32_t *vpos;
    int nv, pad;
    int32_t *o_read, *o_var;          // final (dense, mapper-order) outputs of the shard
    uint8_t *o_code;
    uint32_t *o_aux0, *o_aux1;
    int64_t cap;
};

// all shards of one submission share ONE grid: global tile T belongs to the shard s with tile0[s] <= T < tile0[s+1]
struct MapBatch {
    const ShardDev *shards;
    const int64_t *tile0;         // [n_shards + 1]
    int n_shards, baseq;
    uint2 *stage;                 // staging slots: global tile T owns [T*slot_cap, (T+1)*slot_cap)
    uint32_t *side; int64_t slots;
    int32_t *tile_w0;             // [4*ntiles]
    int32_t *tile_total;          // [ntiles]
    int slot_cap, dbg;
    int64_t ntiles;
    unsigned long long *prof;     // PHZ_MAP_DBG bit 2048 (profiling build of the kernel only): 8 clock stamps per tile
    // overflow area of the staging slots: a tile with more calls than slot_cap takes a stretch of [ovf->base, ovf->base + ovf->cap) (slot indices of the same
    // stage / side arrays) with one cursor step and leaves its first slot in ovf->tile_first[tile] (0xFFFFFFFF: the area was too small; the host redoes the
    // batch).  Behind ONE pointer, read by the rare tile that needs it: four more kernel arguments cost the kernel 50 scalar registers spilled to vector lanes
    const struct OvfArea *ovf;
};
struct OvfArea { uint32_t *tile_first; unsigned long long *cursor; long long base, cap; };

__device__ __forceinline__ int shard_of(const int64_t *tile0, int n_shards, int64_t T) {
    int lo = 0, hi = n_shards - 1;          // largest s with tile0[s] <= T
    while (lo < hi) {
        const int m = (lo + hi + 1) >> 1;
        if (tile0[m] <= T) lo = m; else hi = m - 1;
    }
    return lo;
}

// the escape of the one-byte plane (a non-ACGT base or a phred above 62: one base in two thousand): the two planes, found through the shard record
__device__ __noinline__ int masked_base_escape(const MapArgs &a, uint32_t soff, int x) {
    const uint8_t *qual = a.shp->qual, *seq2 = a.shp->seq2;
    const uint32_t q = qual[(size_t)soff * 4 + x], s = (seq2[(size_t)soff + (x >> 2)] >> (2 * (x & 3))) & 3;
    if ((int)(q & 0x7f) < a.baseq) return 4;
    if (q & 0x80) return s == 0 ? 4 : 5;
    return (int)s;
}

struct VarWin {
    const int32_t *g;
    const int32_t *lds;
    int w0, nv, wlen;
    __device__ __forceinline__ int at(int i) const {
        unsigned d = (unsigned)(i - w0);
        return d < (unsigned)wlen ? lds[d] : g[i];
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

// Staging record of one call: {variant index, rec[0:10) | code[10:14) | wide1[14] | wide0[15] | aux0[16:32)}.  aux0 (offset of the
// base in the read) below 2^16 and aux1 == 0 (no inserted text) is all but a handful of calls; the others keep the full words in the
// side planes at the same slot, so that the common call is 8 bytes out of k_map and 8 bytes into k_compact.
__device__ __forceinline__ void stage_put(uint2 *stage, uint32_t *side, int64_t slots, int64_t g, uint32_t var, uint32_t a0, uint32_t a1,
                                          uint32_t rec, uint32_t code) {
    const bool w0 = a0 >= 0x10000u, w1 = a1 != 0u;
    stage[g] = make_uint2(var, rec | (code << 10) | ((uint32_t)w1 << 14) | ((uint32_t)w0 << 15) | (w0 ? 0u : a0 << 16));
    if (w0) side[g] = a0;
    if (w1) side[slots + g] = a1;
}

// the tile's packed CIGAR words: LDS for the staged prefix, global beyond it
struct CigWin {
    const uint32_t *g;
    const uint32_t *lds;
    uint32_t c_begin, cap;
    __device__ __forceinline__ uint32_t at(uint32_t k) const {
        const uint32_t d = k - c_begin;
        return d < cap ? lds[d] : g[k];
    }
};

// LDS candidate buffer (one per workgroup): every (read, variant) pair that passes the position rule.  Eight bytes per entry plus a
// short side list (the buffer's size decides how many tiles a CU holds, and the kernel's time follows the residency): the variant as a
// 16-bit offset from the tile's window start, the base's read offset in 16 bits, inserted text -- one candidate in hundreds -- in the side
// list.  What does not fit those widths (a variant 65,535 entries beyond the window start, a base beyond offset 65,534, more than
// CAND_INS insertion candidates in a tile) flags the buffer as overflowed: the tile then takes the in-lane fallback, which needs no buffer.
#ifndef PHZ_CAND_INS
#define PHZ_CAND_INS 16
#endif
constexpr int CAND_INS = PHZ_CAND_INS;
constexpr uint32_t KEY_HAS_INS = 0x4000u;
struct CandBuf {
    uint32_t *key;    // local read index << 16 | lean-origin << 15 | has inserted text << 14 | ordinal within the read << 8 | code (filled by the resolve phase)
    uint16_t *var;    // variant index - w0
    uint16_t *x0;     // read offset of the base, 0xFFFF = none (deleted base)
    unsigned long long *ins;      // slot << 32 | (offset << 12 | length) of the inserted text
    int *n;           // candidates appended; > cap (or an ordinal >= 32) sends the tile down the in-lane fallback
    int *nins;
    int cap, w0;
};
__device__ __forceinline__ void cand_put(const CandBuf &cb, int slot, uint32_t key, int var, uint32_t x0, uint32_t x1) {
    const uint32_t dv = (uint32_t)(var - cb.w0);
    bool wide = dv >= 0xFFFFu || (x0 != 0xFFFFFFFFu && x0 >= 0xFFFFu);
    if (x1 != 0u) {
        const int t = atomicAdd(cb.nins, 1);
        if (t < CAND_INS) cb.ins[t] = ((unsigned long long)(uint32_t)slot << 32) | x1; else wide = true;
        key |= KEY_HAS_INS;
    }
    if (wide) atomicOr(cb.n, 1 << 30);
    cb.key[slot] = key;
    cb.var[slot] = (uint16_t)dv;
    cb.x0[slot] = (uint16_t)(x0 == 0xFFFFFFFFu ? 0xFFFFu : x0);
}
__device__ __forceinline__ uint32_t cand_x0(const CandBuf &cb, int e) { const uint32_t v = cb.x0[e]; return v == 0xFFFFu ? 0xFFFFFFFFu : v; }
__device__ __forceinline__ uint32_t cand_x1(const CandBuf &cb, int e, uint32_t key) {
    if (!(key & KEY_HAS_INS)) return 0u;
    const int m = *cb.nins < CAND_INS ? *cb.nins : CAND_INS;
    for (int t = 0; t < m; t++) { const unsigned long long w = cb.ins[t]; if ((int)(w >> 32) == e) return (uint32_t)w; }
    return 0u;
}

// experiment (PHZ_MAP_DBG bit 1024, timing only -- wrong bases): the byte that holds a base's two bits is read from the quality plane, inside the
// 4-byte group of its quality byte, i.e. what a layout with both in one memory sector would fetch
#define PHZ_SEQ_BYTE(a, soff, x) (((a).dbg & 1024) ? (a).qual[(size_t)(soff) * 4 + ((x) | 3)] : (a).seq2[(size_t)(soff) + ((x) >> 2)])
// Base and quality of a call from ONE byte (phz_reads.bq: base << 6 | min(phred, 62), 63 = escape to the two planes): one 128-byte line per call instead of two.
// The ONE instantiations of the kernel read it (every shard of the submission carries the plane); A/B on the profiling instantiation: PHZ_MAP_DBG bit 4096
// (tools/ab_kmap_oneplane.py: -3.7 % on the profiling instantiation, -0.1 % on the production one, call lists identical)
#define PHZ_ONEPLANE(a) ((a).one)
__device__ __forceinline__ int masked_base_two(const MapArgs &a, uint32_t soff, int x) {
    const uint32_t q = a.qual[(size_t)soff * 4 + x], s = (a.seq2[(size_t)soff + (x >> 2)] >> (2 * (x & 3))) & 3;
    if ((int)(q & 0x7f) < a.baseq) return 4;
    if (q & 0x80) return s == 0 ? 4 : 5;
    return (int)s;
}
__device__ int masked_base_escape(const MapArgs &a, uint32_t soff, int x);
__device__ __forceinline__ int sym_from_bq(const MapArgs &a, uint32_t soff, int x, uint32_t b) {
    const uint32_t q6 = b & 63u;
    if (q6 == 63u) return a.qual ? masked_base_two(a, soff, x) : masked_base_escape(a, soff, x);
    return (int)q6 < a.baseq ? 4 : (int)(b >> 6);
}
// symbol of read base x after baseq masking: 0..3 = ACGT, 4 = 'N', 5 = other IUPAC character
__device__ __forceinline__ int masked_base(const MapArgs &a, uint32_t soff, int x) {
    if (PHZ_ONEPLANE(a)) return sym_from_bq(a, soff, x, a.bq[(size_t)soff * 4 + x]);
    uint32_t q = a.qual[(size_t)soff * 4 + x];
    uint32_t s = (PHZ_SEQ_BYTE(a, soff, x) >> (2 * (x & 3))) & 3;
    if ((int)(q & 0x7f) < a.baseq) return 4;
    if (q & 0x80) return s == 0 ? 4 : 5;
    return (int)s;
}

// MODE 0: position rule only, candidates appended to the LDS buffer (no global loads in the divergent walk)
// MODE 1: count calls, resolving bases in-lane (fallback)      MODE 2: emit calls in-lane (fallback)
template <int MODE>
__device__ int walk_read(const MapArgs &a, const VarWin &vw, const CigWin &cw, const CandBuf &cb, int j, int64_t r, int pos,
                         uint32_t c0, uint32_t c1, uint32_t soff, int64_t out_base, int64_t out_limit) {
    int i = vw.lower_bound(vw.w0, pos);
    if (i >= vw.nv) return 0;
    int cnt = 0;
    int gpos = 0, rpos = 0, seg_start = 0, plen = 0, seg_rpos = 0;
    uint32_t seg_op = c0;
    for (uint32_t k = c0; k < c1; k++) {
        uint32_t w = cw.at(k);
        const int len = (int)(w >> 4);
        const uint32_t op = w & 15;
        if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) {
            const int lo = pos + seg_start + plen, hi = lo + len;
            if (i < vw.nv && vw.at(i) < lo) i = vw.lower_bound(i, lo);
            while (i < vw.nv && !(a.dbg & 16)) {
                const int vp = vw.at(i);
                if (vp >= hi) break;
                const int p = vp - pos - seg_start;            // index into the segment's pseudo read
                const int rb = rpos + (p - plen);              // read offset of that base (M runs only)
                // insertion recorded under key == p in THIS segment (key is read-relative: quirk kept)
                int ioff = 0, ilen = 0;
                if (c1 - c0 > 1) {
                    int g2 = seg_start, r2 = seg_rpos;
                    for (uint32_t k2 = seg_op; k2 < c1; k2++) {
                        uint32_t w2 = cw.at(k2);
                        const int l2 = (int)(w2 >> 4);
                        const uint32_t o2 = w2 & 15;
                        if (o2 == OP_N) break;
                        if (o2 == OP_M || o2 == OP_EQ || o2 == OP_X) { g2 += l2; r2 += l2; }
                        else if (o2 == OP_D || o2 == OP_G) g2 += l2;
                        else if (o2 == OP_I) { if (g2 - 1 == p) { ioff = r2; ilen = l2; } r2 += l2; }
                        else if (o2 == OP_S) r2 += l2;
                    }
                }
                const int nchars = (op != OP_D) + ilen;
                if (nchars > 0) {
                    const uint32_t x0 = op != OP_D ? (uint32_t)rb : 0xFFFFFFFFu;
                    const uint32_t x1 = ilen > 0 ? (((uint32_t)ioff << 12) | (uint32_t)(ilen > 4095 ? 4095 : ilen)) : 0u;
                    if (MODE == 0) {
                        // code 7 = one character, to be resolved from seq/qual; 4 = composite text (always a call)
                        const int slot = cnt < 32 ? atomicAdd(cb.n, 1) : (atomicOr(cb.n, 1 << 30), 1 << 30);      // OR, not ADD: thousands of these in one tile must not wrap the counter
                        if (slot < cb.cap) cand_put(cb, slot, ((uint32_t)j << 16) | ((uint32_t)cnt << 8) | (nchars == 1 ? 7u : 4u), i, x0, x1);
                        cnt++;
                    } else {
                        int code = 4;
                        if (nchars == 1) {
                            const int s = masked_base(a, soff, op != OP_D ? rb : ioff);
                            code = s < 4 ? s : (s == 4 ? -1 : 4);
                        }
                        if (code >= 0) {
                            if (MODE == 2) {
                                const int64_t o = out_base + cnt;
                                if (o < out_limit) {
                                    stage_put(a.stage, a.side, a.slots, o, (uint32_t)i, x0, x1, (uint32_t)j, (uint32_t)code);
                                }
                            }
                            cnt++;
                        }
                    }
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

// Lean walker for multi-op records: any mix of M / = / X / D / N / S / H / P and up to two insertions, every run inside the staged
// window and every CIGAR word in LDS.  Same rule as walk_read<0> (candidates appended to the LDS buffer, position rule only), organised
// for a short, branch-light instruction stream -- this phase is VALU-issue bound, and a wave pays the stream once whatever its lanes do:
//   * ONE pass over the ops: the window search runs unconditionally (empty interval for ops that are not runs), wave-uniform ballots
//     skip the insertion bookkeeping and the search when no lane needs them;
//   * what the pass cannot express ('G' ops, a third insertion, an insertion not directly after a non-empty reference-consuming op,
//     a run beyond the window) is found on the way: the record is POISONED (its lean-origin candidates, key bit 15, are dropped by
//     the resolve phase) and left to the general walker.
// (Tried and dropped, round 3: a second, bookkeeping-free walker for the pure spliced shapes M N M / M N M N M, listed apart.  It runs
// in half the instructions of this one, but a 256-record tile has ~80 multi-op records: split by shape they fill three partial waves
// instead of two, the instruction total does not move and the extra barrier lengthens the tile: 1.57 ms against 1.28 ms.)
// Insertions follow the reference's keying (key = genome offset - 1 at the I op, looked up SEGMENT-relative): while seg_start is 0
// that is the last base of the op before the I (found by peeking at the next op); after an N the key lands seg_start bases further
// on, i.e. in a later run of the same segment (carried forward in two register slots, the later insertion wins).
// Lower bound over the staged window, branch-free per step: the window is padded with INT_MAX up to MAP_WIN entries, so a probe needs
// no bounds test, and the number of steps (wave-uniform, computed once per tile from the window length) selects the entry point of a
// fully unrolled halving chain -- three vector instructions and one LDS read per step, no loop bookkeeping on the scalar unit.
__device__ __forceinline__ int window_depth(int wlen) {          // steps 2^(d-1) .. 1, then one closing probe: reaches every count <= wlen
    const int d = wlen > 0 ? 32 - __builtin_clz((unsigned)wlen) : 0;
    constexpr int dmax = 31 - __builtin_clz((unsigned)MAP_WIN);
    return d < dmax ? d : dmax;
}
__device__ __forceinline__ int lds_lower_bound(const int32_t *w, int depth, int key) {
    static_assert(MAP_WIN >= 2 && MAP_WIN <= 1024 && (MAP_WIN & (MAP_WIN - 1)) == 0, "MAP_WIN: a power of two up to 1024");
    int base = 0;
#define PHZ_LB_STEP(S) base = (w[base + (S) - 1] < key) ? base + (S) : base
    switch (depth) {
    default: PHZ_LB_STEP(512); [[fallthrough]];
    case 9: PHZ_LB_STEP(256); [[fallthrough]];
    case 8: PHZ_LB_STEP(128); [[fallthrough]];
    case 7: PHZ_LB_STEP(64); [[fallthrough]];
    case 6: PHZ_LB_STEP(32); [[fallthrough]];
    case 5: PHZ_LB_STEP(16); [[fallthrough]];
    case 4: PHZ_LB_STEP(8); [[fallthrough]];
    case 3: PHZ_LB_STEP(4); [[fallthrough]];
    case 2: PHZ_LB_STEP(2); [[fallthrough]];
    case 1: PHZ_LB_STEP(1); [[fallthrough]];
    case 0: break;
    }
#undef PHZ_LB_STEP
    base += (w[base] < key) ? 1 : 0;
    return base;
}

__device__ bool walk_lean(const int32_t *s_vpos, int wlen, int wdepth, int w0, const uint32_t *s_cig, uint32_t c_begin, uint32_t cig_cap,
                          const CandBuf &cb, uint32_t *s_poison, int j, int pos, uint32_t c0, uint32_t c1, long long cover, int dbg) {
    if (c1 - c_begin > cig_cap) return false;
    int g = 0, seg_start = 0, cnt = 0, nins = 0; uint32_t r = 0;
    bool bad = false;
    int t0 = -1, t1 = -1; uint32_t o0 = 0, o1 = 0, l0 = 0, l1 = 0;      // insertions carried forward inside a later segment
    uint32_t prev = 0x10u, prev_len = 1;                                // "previous op" of the first op: S-like
    for (uint32_t k = c0; k < c1; k++) {
        const uint32_t w = s_cig[k - c_begin];
        const int len = (int)(w >> 4); const uint32_t op = w & 15, bit = 1u << op;
        const bool mlike = (bit & 0x181u) != 0;
        if (__builtin_amdgcn_ballot_w64(op == OP_I) != 0) {
            if (op == OP_I) {
                bad |= (prev & 0x272u) != 0 || prev_len == 0 || nins == 2;      // after I / S / H / P / G / an empty op; a third insertion
                nins++;
                t1 = t0; o1 = o0; l1 = l0; t0 = g - 1 + seg_start; o0 = r; l0 = (uint32_t)(len > 4095 ? 4095 : len);
            }
        }
        bad |= op == OP_G;
        const bool search = (mlike || op == OP_D) && !(dbg & 64);
        if (__builtin_amdgcn_ballot_w64(search) != 0) {
            const int lo = pos + g;
            bad |= search && (long long)lo + len > cover;                       // the run must lie inside the staged window
            const int hi = (search && !bad) ? lo + len : lo;
            int i = lds_lower_bound(s_vpos, wdepth, lo);
            while (i < wlen && !(dbg & 128)) {
                const int vp = s_vpos[i];
                if (vp >= hi) break;
                uint32_t ioff = 0, ilen = 0;
                if (seg_start == 0) {           // key = offset of the base before the I: the last base of this op
                    if (vp == hi - 1 && k + 1 < c1) {
                        const uint32_t wn = s_cig[k + 1 - c_begin];
                        if ((wn & 15) == OP_I) { ilen = wn >> 4; ilen = ilen > 4095u ? 4095u : ilen; ioff = r + (mlike ? (uint32_t)len : 0u); }
                    }
                } else {
                    const int off = vp - pos;
                    if (t0 == off) { ioff = o0; ilen = l0; } else if (t1 == off) { ioff = o1; ilen = l1; }
                }
                const int nchars = (mlike ? 1 : 0) + (int)ilen;
                if (nchars > 0) {
                    const int slot = cnt < 32 ? atomicAdd(cb.n, 1) : (atomicOr(cb.n, 1 << 30), 1 << 30);      // OR, not ADD: thousands of these in one tile must not wrap the counter
                    if (slot < cb.cap)
                        cand_put(cb, slot, ((uint32_t)j << 16) | ((uint32_t)(cnt | 0x80) << 8) | (nchars == 1 ? 7u : 4u), w0 + i,
                                 mlike ? (uint32_t)(r + (uint32_t)(vp - lo)) : 0xFFFFFFFFu, ilen > 0 ? ((ioff << 12) | ilen) : 0u);
                    cnt++;
                }
                i++;
            }
        }
        g += (bit & 0x38Du) ? len : 0;
        r += (bit & 0x193u) ? (uint32_t)len : 0u;
        if (op == OP_N) { seg_start = g; t0 = -1; t1 = -1; }
        prev = bit; prev_len = (uint32_t)len;
    }
    if (bad) atomicOr(&s_poison[j >> 5], 1u << (j & 31));
    return !bad;
}

constexpr int MAP_COVER = 65536;   // the staged window holds every het SNP below POS(last read of the tile) + MAP_COVER ...
constexpr int MAP_SLACK = 8;       // ... plus this many further entries (probe overshoot / loop sentinels)

// per-tile het-SNP window: start = lower_bound(vpos, POS of the tile's first read); tile_w[4t+1] = staged length,
// bit 30 set when the window is complete (not truncated by MAP_WIN), which enables the LDS-only fast path
__global__ void k_tile_window(MapBatch bt, int tile_reads) {
    const int64_t T = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (T == 0) *bt.ovf->cursor = 0ull;                     // the overflow area of this submission's staging starts empty
    if (T >= bt.ntiles) return;
    const int si = shard_of(bt.tile0, bt.n_shards, T);
    const ShardDev &sh = bt.shards[si];
    const int64_t t = T - bt.tile0[si];
    const int32_t *pos = sh.pos; const uint32_t *cigar_off = sh.cigar_off; const int32_t *vpos = sh.vpos;
    const int nv = sh.nv; const int64_t n = sh.n;
    int32_t *tile_w = bt.tile_w0;
    const int key = pos[t * tile_reads];
    const int64_t last = (t + 1) * tile_reads - 1 < n ? (t + 1) * tile_reads - 1 : n - 1;
    const long long key2 = (long long)pos[last] + MAP_COVER;
    // both lower bounds in ONE loop over the whole table: their probes are independent, so the two chains of dependent loads
    // overlap instead of following each other
    int lo = 0, hi = nv, lo2 = 0, hi2 = nv;
    while (lo < hi || lo2 < hi2) {
        const int m = (lo + hi) >> 1, m2 = (lo2 + hi2) >> 1;
        const bool go = lo < hi, go2 = lo2 < hi2;
        const int v = go ? vpos[m] : 0, v2 = go2 ? vpos[m2] : 0;
        if (go) { if (v < key) lo = m + 1; else hi = m; }
        if (go2) { if ((long long)v2 < key2) lo2 = m2 + 1; else hi2 = m2; }
    }
    int len = lo2 - lo + MAP_SLACK;
    int complete = 1 << 30;
    if (len > MAP_WIN) { len = MAP_WIN; complete = 0; }
    const uint32_t n_ops = cigar_off[last + 1] - cigar_off[t * tile_reads];
    tile_w[4 * T] = lo;
    tile_w[4 * T + 1] = len | complete | (si << 16);          // bits 0-15 window length, 16-28 shard of the tile, 30 window complete
    tile_w[4 * T + 2] = (int32_t)cigar_off[t * tile_reads];            // first CIGAR word of the tile
    tile_w[4 * T + 3] = (int32_t)n_ops;
}

// inclusive prefix sum over the 64 lanes of a wave in seven DPP adds (row shifts inside the rows of 16 lanes, then the row
// broadcasts of gfx9): the shuffle version is six ds_bpermute round trips with their address arithmetic and selects
__device__ __forceinline__ int wave_incl_scan(int v0) {
    int v1 = v0 + __builtin_amdgcn_update_dpp(0, v0, 0x111, 0xf, 0xf, true);      // row_shr:1
    v1 += __builtin_amdgcn_update_dpp(0, v0, 0x112, 0xf, 0xf, true);              // row_shr:2
    v1 += __builtin_amdgcn_update_dpp(0, v0, 0x113, 0xf, 0xf, true);              // row_shr:3  -> sums of 4
    v1 += __builtin_amdgcn_update_dpp(0, v1, 0x114, 0xf, 0xe, true);              // row_shr:4, banks 1-3 -> sums of 8
    v1 += __builtin_amdgcn_update_dpp(0, v1, 0x118, 0xf, 0xc, true);              // row_shr:8, banks 2-3 -> sums inside the row
    v1 += __builtin_amdgcn_update_dpp(0, v1, 0x142, 0xa, 0xf, true);              // row_bcast:15 into rows 1 and 3
    v1 += __builtin_amdgcn_update_dpp(0, v1, 0x143, 0xc, 0xf, true);              // row_bcast:31 into rows 2 and 3
    return v1;
}

// maximum over the 64 lanes of a wave (same DPP ladder, the result of lane 63 read into a scalar)
__device__ __forceinline__ int wave_max(int v0) {
    const int lo = (int)0x80000000;
    auto mx = [](int a, int b) { return a > b ? a : b; };
    int v1 = mx(v0, __builtin_amdgcn_update_dpp(lo, v0, 0x111, 0xf, 0xf, false));
    v1 = mx(v1, __builtin_amdgcn_update_dpp(lo, v0, 0x112, 0xf, 0xf, false));
    v1 = mx(v1, __builtin_amdgcn_update_dpp(lo, v0, 0x113, 0xf, 0xf, false));
    v1 = mx(v1, __builtin_amdgcn_update_dpp(lo, v1, 0x114, 0xf, 0xe, false));
    v1 = mx(v1, __builtin_amdgcn_update_dpp(lo, v1, 0x118, 0xf, 0xc, false));
    v1 = mx(v1, __builtin_amdgcn_update_dpp(lo, v1, 0x142, 0xa, 0xf, false));
    v1 = mx(v1, __builtin_amdgcn_update_dpp(lo, v1, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v1, 63);
}

// block-wide exclusive scan of RPT per-thread values laid out at [k*MAP_BLOCK + tid]; returns the block total
template <int MAP_BLOCK, int RPT>
__device__ __forceinline__ int block_scan(const int (&cnt)[RPT], int (&excl)[RPT], int (*s_wsum)[MAP_BLOCK / 64], int lane, int wave) {
    int incl[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int x = wave_incl_scan(cnt[k]);
        incl[k] = x;
        if (lane == 63) s_wsum[k][wave] = x;
    }
    __syncthreads();
    int running = 0;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        int before = running;
#pragma unroll
        for (int w2 = 0; w2 < MAP_BLOCK / 64; w2++) {
            const int s = s_wsum[k][w2];
            if (w2 < wave) before += s;
            running += s;
        }
        excl[k] = before + incl[k] - cnt[k];
    }
    return running;
}

// ABL: the ablation switches of PHZ_MAP_DBG are live (profiling build of the same code); the production instantiation sees dbg == 0 at
// compile time, so none of its tests reach the scalar unit
template <int MAP_BLOCK, int RPT, bool ABL, bool ONE>
__device__ __forceinline__ void map_tile(const MapBatch &bt, const int64_t gtile) {
    const int4 tw = *reinterpret_cast<const int4 *>(bt.tile_w0 + 4 * gtile);      // the pre-pass left the tile's shard here: no search
    const int si = (tw.y >> 16) & 0x1FFF;
    MapArgs a;
    {
        const ShardDev &sh = bt.shards[si];
        a.pos = sh.pos; a.cigar_off = sh.cigar_off; a.cigar = sh.cigar; a.seq_off = sh.seq_off; a.bq = sh.bq; a.shp = &sh;
        if (ONE && !ABL) { a.seq2 = nullptr; a.qual = nullptr; } else { a.seq2 = sh.seq2; a.qual = sh.qual; }
        a.n = sh.n; a.vpos = sh.vpos; a.nv = sh.nv; a.baseq = bt.baseq;
        a.stage = bt.stage; a.side = bt.side; a.slots = bt.slots;
        a.tile_w0 = bt.tile_w0; a.tile_total = bt.tile_total; a.slot_cap = bt.slot_cap; a.ntiles = bt.ntiles; a.dbg = ABL ? bt.dbg : 0;
        a.one = ABL ? ((bt.dbg & 4096) != 0 && sh.bq != nullptr) : ONE;
    }
#define PHZ_STAMP(K) do { if (ABL && (bt.dbg & 2048) && threadIdx.x == 0) bt.prof[8 * gtile + (K)] = wall_clock64(); } while (0)
    PHZ_STAMP(0);
    constexpr int TILE = MAP_BLOCK * RPT;
    constexpr int CIG = TILE * PHZ_CIG_X2 / 2;          // packed CIGAR words staged per tile (max)
    constexpr int CAND = TILE * PHZ_CAND_X4 / 4;         // candidate buffer entries (complex records only)
    __shared__ int32_t s_vpos[MAP_WIN];
    __shared__ uint32_t s_coff[TILE + 1];      // cigar_off slice; reused as per-read output offsets after the walk
    __shared__ uint32_t s_soff[TILE];
    __shared__ int32_t s_pos[TILE];            // record positions for the multi-op walk; after it the same words are s_mask
    uint32_t *s_mask = reinterpret_cast<uint32_t *>(s_pos);      // per read: bit o set <=> candidate with ordinal o is a call (resolve phase on)
    __shared__ uint16_t s_cx[TILE];            // reads that need the general CIGAR walk
    __shared__ uint32_t s_cig[CIG];
    __shared__ uint32_t s_key[CAND];
    __shared__ uint16_t s_var[CAND];
    __shared__ uint16_t s_x0[CAND];
    __shared__ unsigned long long s_ins[CAND_INS];
    __shared__ int s_wsum[RPT][MAP_BLOCK / 64];
    __shared__ int s_ncand, s_ncx, s_nlong, s_nins;
    __shared__ uint32_t s_poison[TILE / 32];   // records the lean walker gave up on after it had appended candidates

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tile = gtile - bt.tile0[si];          // tile index inside the shard
    const int64_t r0 = tile * TILE;
    const int nr = (int)((a.n - r0) < TILE ? (a.n - r0) : TILE);

    // ---- coalesced streaming loads: pos / seq_off / cigar_off slices, the het-SNP window and the tile's CIGAR words into LDS.
    //      Every load a lane needs (the window and the CIGAR words almost always fit one and three rounds of the workgroup) is
    //      REQUESTED before the first LDS store waits for anything: one global round trip at the start of a tile instead of three.
    VarWin vw;
    vw.g = a.vpos; vw.lds = s_vpos; vw.nv = a.nv; vw.w0 = tw.x;
    vw.wlen = tw.y & 0xFFFF;
    const bool complete = (tw.y >> 30) & 1;
    CigWin cw;
    cw.g = a.cigar; cw.lds = s_cig; cw.c_begin = (uint32_t)tw.z; cw.cap = CIG;
    uint32_t cnt_w = (uint32_t)tw.w;
    if (cnt_w > (uint32_t)CIG) cnt_w = CIG;
    int rpos_[RPT];
    uint32_t soff_[RPT], coff_[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int j = k * MAP_BLOCK + tid;
        const bool ok = j < nr;
        rpos_[k] = ok ? a.pos[r0 + j] : 0;
        soff_[k] = ok ? a.seq_off[r0 + j] : 0;
        coff_[k] = a.cigar_off[r0 + (ok ? j : nr)];
    }
    const uint32_t coff_end = a.cigar_off[r0 + nr];
    int v_first = 0x7fffffff;
    if (tid < vw.wlen) { const int idx = vw.w0 + tid; v_first = idx < a.nv ? a.vpos[idx] : 0x7fffffff; }
    uint32_t cg[3] = {0, 0, 0};
#pragma unroll
    for (int u = 0; u < 3; u++) { const uint32_t j = (uint32_t)tid + (uint32_t)u * MAP_BLOCK; if (j < cnt_w) cg[u] = a.cigar[cw.c_begin + j]; }
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int j = k * MAP_BLOCK + tid;
        s_pos[j] = rpos_[k]; s_soff[j] = soff_[k]; s_coff[j] = coff_[k];
    }
    if (tid == 0) { s_coff[TILE] = coff_end; s_ncand = 0; s_ncx = 0; s_nlong = 0; s_nins = 0; }
    if (tid < TILE / 32) s_poison[tid] = 0;
    const int wdepth = window_depth(vw.wlen);
    // INT_MAX beyond the window: lds_lower_bound probes without a bounds test.  Its halving chain enters at step 2^(wdepth-1) and its closing
    // probe reads at most index 2^wdepth - 1 -- but the scans of phase 1a read up to wlen -- so the padding stops at max(wlen, 2^wdepth)
#ifdef PHZ_FILL_ALL
    const int wfill = MAP_WIN;
#else
    const int wfill = (1 << wdepth) > vw.wlen ? (1 << wdepth) : vw.wlen;
#endif
    if (tid < wfill) s_vpos[tid] = v_first;
    for (int j = tid + MAP_BLOCK; j < wfill; j += MAP_BLOCK) {
        const int idx = vw.w0 + j;
        s_vpos[j] = (j < vw.wlen && idx < a.nv) ? a.vpos[idx] : 0x7fffffff;
    }
#pragma unroll
    for (int u = 0; u < 3; u++) { const uint32_t j = (uint32_t)tid + (uint32_t)u * MAP_BLOCK; if (j < cnt_w) s_cig[j] = cg[u]; }
    for (uint32_t j = (uint32_t)tid + 3u * MAP_BLOCK; j < cnt_w; j += MAP_BLOCK) s_cig[j] = a.cigar[cw.c_begin + j];
    __syncthreads();
    PHZ_STAMP(1);
    CandBuf cb;
    cb.key = s_key; cb.var = s_var; cb.x0 = s_x0; cb.ins = s_ins; cb.n = &s_ncand; cb.nins = &s_nins; cb.cap = CAND; cb.w0 = vw.w0;
    const long long cover = (long long)s_pos[nr - 1] + MAP_COVER;     // every het SNP below this is inside the window

    // ---- phase 1a: records made of one aligned run (the common case) never leave registers: an LDS-only,
    //      uniform-trip-count binary search yields (first window index, count) of the het SNPs under the read
    uint32_t c0_[RPT], c1_[RPT];
    int base_[RPT], n_[RPT];
    uint32_t vmask_[RPT], codes_[RPT];
    bool fast_[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int j = k * MAP_BLOCK + tid;
        c0_[k] = s_coff[j]; c1_[k] = s_coff[j + 1];
        base_[k] = 0; n_[k] = 0; vmask_[k] = 0; codes_[k] = 0; fast_[k] = false;
    }
    const bool walk_on = !(a.dbg & 8);
    // Records are position-sorted, so the het SNPs any single-run record of a wave can touch lie in ONE narrow index range
    // [lo, hi) of the window: lo = lower_bound(position of the wave's first record), hi = lower_bound(largest end).  All
    // 2*RPT bounds come out of a single uniform-trip search (lane group g looks for bound g); each record then needs only a
    // scan over hi - lo entries (0 for most waves: nothing under any of their records) instead of its own search.
    int len_[RPT]; bool one_[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int j = k * MAP_BLOCK + tid;
        one_[k] = false; len_[k] = 0;
        const uint32_t d = c0_[k] - cw.c_begin;
        if (j < nr && walk_on && complete && c1_[k] - c0_[k] == 1 && d < (uint32_t)CIG) {
            const uint32_t w = s_cig[d];
            const uint32_t op = w & 15;
            len_[k] = (int)(w >> 4);
            one_[k] = (op == OP_M || op == OP_EQ || op == OP_X) && (long long)rpos_[k] + len_[k] <= cover;
        }
    }
    constexpr int GROUP = 64 / (2 * RPT);
    int target = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int e = wave_max(one_[k] ? rpos_[k] + len_[k] : (int)0x80000000);
        const int first = __builtin_amdgcn_readfirstlane(rpos_[k]);
        if ((lane / GROUP) == 2 * k) target = first;
        if ((lane / GROUP) == 2 * k + 1) target = e;
    }
    const int bound = lds_lower_bound(s_vpos, wdepth, target);
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int j = k * MAP_BLOCK + tid;
        const int lo = __builtin_amdgcn_readlane(bound, 2 * k * GROUP), hi = __builtin_amdgcn_readlane(bound, (2 * k + 1) * GROUP);
        if (one_[k]) {
            const int pos = rpos_[k], end = pos + len_[k];
            if (hi - lo <= 24) {
                int below = 0, c = 0;
                for (int e = lo; e < hi; e++) {
                    const int vp = s_vpos[e];
                    below += vp < pos ? 1 : 0;
                    c += (vp >= pos && vp < end) ? 1 : 0;
                }
                if (c <= 8) { fast_[k] = true; base_[k] = lo + below; n_[k] = c; }
            } else {
                const int base = lds_lower_bound(s_vpos, wdepth, pos);
                int c = 0;
                while (base + c < vw.wlen && s_vpos[base + c] < end && c <= 8) c++;
                if (c <= 8) { fast_[k] = true; base_[k] = base; n_[k] = c; }
            }
        }
        if (j < nr && walk_on && !fast_[k]) {
            // two lists in one array: records of <= 3 ops from the front, longer ones from the back -- the walk's trip count is the
            // longest CIGAR of the wave, so the first wave gets the short records and the stragglers share the last one
            if (s_coff[j + 1] - s_coff[j] > 3u) s_cx[TILE - 1 - atomicAdd(&s_nlong, 1)] = (uint16_t)j;
            else s_cx[atomicAdd(&s_ncx, 1)] = (uint16_t)j;
        }
    }
    // ---- phase 2a, first half: the quality / base bytes under the FIRST het SNP of every fast record are requested now and used
    //      after the multi-op walk below -- the two gathers are dependent global round trips that nothing else in this phase hides
    uint32_t pre_q[RPT], pre_s[RPT];
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        pre_q[k] = 0; pre_s[k] = 0;
        if (fast_[k] && n_[k] > 0 && !(a.dbg & 1)) {
            const uint32_t soff = s_soff[k * MAP_BLOCK + tid];
            const int x = s_vpos[base_[k]] - rpos_[k];
            if (PHZ_ONEPLANE(a)) pre_q[k] = a.bq[(size_t)soff * 4 + x];
            else { pre_q[k] = a.qual[(size_t)soff * 4 + x]; pre_s[k] = PHZ_SEQ_BYTE(a, soff, x); }
        }
    }
    PHZ_STAMP(2);
    __syncthreads();
    PHZ_STAMP(3);
    // ---- phase 1b: spliced / gapped / clipped records, densely re-packed so the divergent walk runs on full waves
    const int nshort = walk_on ? s_ncx : 0;
    const int ncx = (walk_on && !(a.dbg & 256)) ? nshort + s_nlong : 0;       // dbg 256: multi-op records listed but not walked
    const bool lean_on = complete && !(a.dbg & 32);
    // 1b: the lean walker first (one lane per multi-op record, densely packed); what it declines is redone by the general walker
    bool redo_any = false;
    for (int t = tid; t < ncx; t += MAP_BLOCK) {
        const int ts = t < nshort ? t : TILE - 1 - (t - nshort);
        const int j = s_cx[ts];
        const bool done = lean_on && walk_lean(s_vpos, vw.wlen, wdepth, vw.w0, s_cig, cw.c_begin, (uint32_t)CIG, cb, s_poison, j, s_pos[j], s_coff[j], s_coff[j + 1], cover, a.dbg);
        if (!done) { s_cx[ts] = (uint16_t)(j | 0x8000); redo_any = true; }
    }
    const bool redo = __syncthreads_or(redo_any ? 1 : 0) != 0;
    // the lean walk was the last reader of s_pos: its words become the per-record call masks (zeroed here, OR-ed after the next barrier);
    // the few records left to the general walker fetch their position again
#pragma unroll
    for (int k = 0; k < RPT; k++) s_mask[k * MAP_BLOCK + tid] = 0;
    if (redo) {
        for (int t = tid; t < ncx; t += MAP_BLOCK) {
            const int jf = s_cx[t < nshort ? t : TILE - 1 - (t - nshort)];
            if (jf & 0x8000) { const int j = jf & 0x7FFF; walk_read<0>(a, vw, cw, cb, j, r0 + j, a.pos[r0 + j], s_coff[j], s_coff[j + 1], 0, 0, 0); }
        }
    }
    __syncthreads();
    PHZ_STAMP(4);
    const int ncand = s_ncand;
    const bool fb = ncand > CAND;              // candidate buffer overflow: complex records fall back to in-lane work
    // the bytes of this lane's buffered candidate (if it has one) are requested now and used after the arithmetic below
    uint32_t cq = 0, cs = 0; int cx_off = -1;
    if (!fb && tid < ncand && !(a.dbg & 1)) {
        const uint32_t key = s_key[tid];
        if ((key & 0xFF) == 7u) {
            const uint32_t x0 = cand_x0(cb, tid);
            cx_off = x0 != 0xFFFFFFFFu ? (int)x0 : (int)(cand_x1(cb, tid, key) >> 12);
            const uint32_t soff = s_soff[key >> 16];
            if (PHZ_ONEPLANE(a)) cq = a.bq[(size_t)soff * 4 + cx_off];
            else { cq = a.qual[(size_t)soff * 4 + cx_off]; cs = PHZ_SEQ_BYTE(a, soff, cx_off); }
        }
    }
    // ---- phase 2a, second half: resolve the fast records' bases (the first one from the bytes requested above)
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        if (!fast_[k]) continue;
        const int j = k * MAP_BLOCK + tid;
        const uint32_t soff = s_soff[j];
        for (int o = 0; o < n_[k]; o++) {
            const int x = s_vpos[base_[k] + o] - rpos_[k];
            int sy;
            if (a.dbg & 1) sy = x & 3;
            else if (o == 0) {
                if (PHZ_ONEPLANE(a)) sy = sym_from_bq(a, soff, x, pre_q[k]);
                else {
                    const uint32_t q = pre_q[k], sb = (pre_s[k] >> (2 * (x & 3))) & 3;
                    sy = (int)(q & 0x7f) < a.baseq ? 4 : ((q & 0x80) ? (sb == 0 ? 4 : 5) : (int)sb);
                }
            } else sy = masked_base(a, soff, x);
            if (sy != 4) { vmask_[k] |= 1u << o; codes_[k] |= (uint32_t)(sy < 4 ? sy : 4) << (4 * o); }
        }
    }
    int cnt[RPT], off[RPT];
    if (!fb) {
        // ---- phase 2b: resolve the buffered candidates with all lanes gathering at once
        for (int e = tid; e < ncand; e += MAP_BLOCK) {
            uint32_t key = s_key[e];
            const int j = (int)(key >> 16);
            int code = (int)(key & 0xFF);
            if ((key & 0x8000u) && ((s_poison[j >> 5] >> (j & 31)) & 1u)) code = -1;      // lean-origin candidate of a poisoned record
            else if (code == 7) {
                const uint32_t x0 = cand_x0(cb, e);
                const int x = x0 != 0xFFFFFFFFu ? (int)x0 : (int)(cand_x1(cb, e, key) >> 12);
                int sy;
                if (a.dbg & 1) sy = x & 3;
                else if (e == tid && cx_off == x) {
                    if (PHZ_ONEPLANE(a)) sy = sym_from_bq(a, s_soff[j], x, cq);
                    else {
                        const uint32_t sb = (cs >> (2 * (x & 3))) & 3;
                        sy = (int)(cq & 0x7f) < a.baseq ? 4 : ((cq & 0x80) ? (sb == 0 ? 4 : 5) : (int)sb);
                    }
                } else sy = masked_base(a, s_soff[j], x);
                code = sy < 4 ? sy : (sy == 4 ? -1 : 4);
            }
            if (code >= 0) {
                atomicOr(&s_mask[j], 1u << ((key >> 8) & 31));
                s_key[e] = (key & 0xFFFFFF00u) | (uint32_t)code;
            } else {
                s_key[e] = key | 0xFFu;            // not a call
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; k++) cnt[k] = fast_[k] ? __popc(vmask_[k]) : __popc(s_mask[k * MAP_BLOCK + tid]);
    } else {
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int j = k * MAP_BLOCK + tid;
            cnt[k] = fast_[k] ? __popc(vmask_[k])
                              : (j < nr && walk_on ? walk_read<1>(a, vw, cw, cb, j, r0 + j, rpos_[k], c0_[k], c1_[k], s_soff[j], 0, 0) : 0);
        }
    }
    PHZ_STAMP(5);
    // ---- phase 3: per-record counts -> offsets (s_coff is free now)
    const int T = block_scan<MAP_BLOCK, RPT>(cnt, off, s_wsum, lane, wave);
#pragma unroll
    for (int k = 0; k < RPT; k++) s_coff[k * MAP_BLOCK + tid] = (uint32_t)off[k];
    if (tid == 0) {
        a.tile_total[gtile] = T;
        // where the tile's calls are staged: its own slot, or -- a tile denser than a slot -- a stretch of the overflow area behind the slots, taken with one
        // cursor step (the slots are sized for the typical tile, not for the densest of a submission: a deep sample with one dense region would have cost
        // tiles x densest x 16 bytes).  Published through two LDS words that are dead by now (slot indices fit 32 bits: checked by the host).
        uint32_t first = (uint32_t)(gtile * (int64_t)a.slot_cap);
        int room = a.slot_cap;
        if (T > a.slot_cap) {
            const OvfArea ov = *bt.ovf;
            const unsigned long long at = atomicAdd(ov.cursor, (unsigned long long)T);
            const bool fits = at + (unsigned long long)T <= (unsigned long long)ov.cap;
            if (fits) { first = (uint32_t)(ov.base + (long long)at); room = T; }
            ov.tile_first[gtile] = fits ? first : 0xFFFFFFFFu;
        }
        s_ncx = (int)first; s_nlong = room;
    }
    __syncthreads();
    PHZ_STAMP(6);
    if (a.dbg & 2) return;
    const int64_t slot0 = (int64_t)(uint32_t)s_ncx;
    const int slot_n = s_nlong;               // calls the tile's stretch of the staging area holds
    // ---- phase 4: ordered flush into the tile's staging slot
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int j = k * MAP_BLOCK + tid;
        if (fast_[k]) {
            int o2 = off[k];
            for (int o = 0; o < n_[k]; o++) {
                if (!((vmask_[k] >> o) & 1)) continue;
                if (o2 < slot_n) {
                    const int64_t g = slot0 + o2;
                    stage_put(a.stage, a.side, a.slots, g, (uint32_t)(vw.w0 + base_[k] + o), (uint32_t)(s_vpos[base_[k] + o] - rpos_[k]), 0u,
                              (uint32_t)j, (codes_[k] >> (4 * o)) & 15u);
                }
                o2++;
            }
        } else if (fb && cnt[k] > 0) {
            walk_read<2>(a, vw, cw, cb, j, r0 + j, rpos_[k], c0_[k], c1_[k], s_soff[j], slot0 + off[k], slot0 + slot_n);
        }
    }
    if (!fb) {
        for (int e = tid; e < ncand; e += MAP_BLOCK) {
            const uint32_t key = s_key[e];
            if ((key & 0xFF) == 0xFF) continue;
            const int j = (int)(key >> 16);
            const uint32_t ord = (key >> 8) & 31;
            const int o = (int)s_coff[j] + __popc(s_mask[j] & ((1u << ord) - 1));
            if (o < slot_n) {
                const int64_t g = slot0 + o;
                stage_put(a.stage, a.side, a.slots, g, (uint32_t)(vw.w0 + (int)s_var[e]), cand_x0(cb, e), cand_x1(cb, e, key), (uint32_t)j, key & 15u);
            }
        }
    }
    PHZ_STAMP(7);
#undef PHZ_STAMP
}

// Eight waves per SIMD: the kernel's time follows the number of resident tiles (measured by padding the workgroups with unused LDS:
// 12 / 11 / 10 / 8 / 6 tiles per CU = 1.21 / 1.39 / 1.51 / 1.67 / 2.56 ms), so the tile's LDS is kept under 10 KB (16 tiles per CU: tools/occ_probe2.hip
// maps LDS bytes to resident workgroups) and the register allocation is held to 64 VGPRs (the scalar registers that no longer fit spill to
// lanes of a vector register): 1.21 -> 1.09 ms.
#ifndef PHZ_MAP_WAVES
#define PHZ_MAP_WAVES 8
#endif
#if PHZ_MAP_WAVES > 0
#define PHZ_MAP_OCC __attribute__((amdgpu_waves_per_eu(PHZ_MAP_WAVES, PHZ_MAP_WAVES)))
#else
#define PHZ_MAP_OCC
#endif
template <int MAP_BLOCK, int RPT, bool ABL, bool ONE = false>
__global__ __launch_bounds__(MAP_BLOCK) PHZ_MAP_OCC void k_map(MapBatch bt) {
    // (XCD-contiguous tile order -- workgroup i runs on XCD i % 8; every XCD given one contiguous eighth of the tiles -- was measured in
    // round 4 for k_map, k_line, k_tile and k_pairs: no gain anywhere, 1.291 against 1.282 ms here.  Neighbouring tiles share only their
    // het-SNP window, which the memory-side cache serves.)
    map_tile<MAP_BLOCK, RPT, ABL, ONE>(bt, (int64_t)blockIdx.x);
}

// two-level exclusive prefix sum of the per-tile totals: 1024-tile chunks in parallel, then one small pass
__global__ __launch_bounds__(1024) void k_chunk_scan(const int32_t *tile_total, int64_t ntiles, int32_t *tile_pref, int64_t *chunk_sum,
                                                     int32_t *chunk_max) {
    __shared__ int s_w[16], s_m[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    const int v = i < ntiles ? tile_total[i] : 0;
    int x = v, mx = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { int y = __shfl_xor(mx, d); mx = y > mx ? y : mx; }
    if (lane == 63) s_w[wave] = x;
    if (lane == 0) s_m[wave] = mx;
    __syncthreads();
    int before = 0;
    for (int w2 = 0; w2 < wave; w2++) before += s_w[w2];
    if (i < ntiles) tile_pref[i] = before + x - v;
    if (tid == 1023) chunk_sum[blockIdx.x] = before + x;
    if (tid == 0) {
        int m = 0;
        for (int w2 = 0; w2 < 16; w2++) m = s_m[w2] > m ? s_m[w2] : m;
        chunk_max[blockIdx.x] = m;
    }
}

__global__ __launch_bounds__(1024) void k_chunk_base(const int64_t *chunk_sum, const int32_t *chunk_max, int nchunks, int64_t *chunk_base,
                                                     unsigned long long *scal /* [0] total, [1] max tile */) {
    __shared__ long long s_w[16];
    __shared__ long long s_carry;
    __shared__ int s_m[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    int mx = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nchunks; b0 += 1024) {
        const int i = b0 + tid;
        const long long v = i < nchunks ? chunk_sum[i] : 0;
        if (i < nchunks) mx = chunk_max[i] > mx ? chunk_max[i] : mx;
        long long x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            long long y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        long long before = s_carry;
        for (int w2 = 0; w2 < wave; w2++) before += s_w[w2];
        if (i < nchunks) chunk_base[i] = before + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = before + x;
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { int y = __shfl_xor(mx, d); mx = y > mx ? y : mx; }
    if (lane == 0) s_m[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        int m = 0;
        for (int w2 = 0; w2 < 16; w2++) m = s_m[w2] > m ? s_m[w2] : m;
        scal[0] = (unsigned long long)s_carry;
        scal[1] = (unsigned long long)m;
    }
}

struct CompactArgs {
    const uint2 *stage; const uint32_t *side; int64_t slots;
    const ShardDev *shards; const int64_t *tile0; int n_shards;
    const int32_t *tile_total; const int32_t *tile_pref; const int64_t *chunk_base; const int64_t *shard_base;
    const int32_t *tile_w0;
    int slot_cap, tile_reads; int64_t ntiles;
    const uint32_t *tile_ovf;             // first slot of a tile with more than slot_cap calls (overflow area), 0xFFFFFFFF: it did not fit
};

// calls before global tile T over all shards (T < ntiles)
__device__ __forceinline__ int64_t calls_before(const int64_t *chunk_base, const int32_t *tile_pref, int64_t T) {
    return chunk_base[T >> 10] + tile_pref[T];
}

// per shard: offset of its first call in the batch-wide numbering and its number of calls -> scal[2 + s] (scal[0] total, [1] max tile)
__global__ void k_shard_totals(const int64_t *tile0, int n_shards, int64_t ntiles, const int32_t *tile_pref, const int64_t *chunk_base,
                               unsigned long long *scal, int64_t *shard_base) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_shards) return;
    const int64_t a = tile0[s], b = tile0[s + 1];
    const int64_t lo = a < ntiles ? calls_before(chunk_base, tile_pref, a) : (int64_t)scal[0];
    const int64_t hi = b < ntiles ? calls_before(chunk_base, tile_pref, b) : (int64_t)scal[0];
    shard_base[s] = lo;
    scal[2 + s] = (unsigned long long)(hi - lo);
}

// One wave per CT consecutive tiles.  A lane first fetches one tile's bookkeeping (count, shard, destination, first record), so the
// chain of dependent loads a tile needs is paid once per CT tiles; the wave then walks the CT slots as ONE flat list of calls, UNR
// elements per lane in flight, each finding its tile by a lower bound over the wave's prefix in LDS.  One wave per tile was bound
// by that chain (~5 round trips for ~60 calls: 254 us per genome whatever the record size); larger CT amortises it further but
// leaves the tail to the waves that drew the densest tiles (CT 4 / 8 / 16 / 64: 1.50 / 1.49 / 1.50 / 1.52 ms per genome step,
// 0.91 / 0.93 / 0.96 / 1.08 ms on configs[1], whose calls sit in a few tiles).
#ifndef PHZ_CT
#define PHZ_CT 4
#endif
constexpr int CT = PHZ_CT, CT_UNR = 4;
template <bool UNI>
__device__ __forceinline__ void compact_run(const CompactArgs &c, const int *pref, const int64_t *dstv, const int32_t *r0v, const int32_t *siv, const int64_t *slotv,
                                            int64_t T0, int M, int lane, int si0) {
    for (int e0 = lane; e0 < M; e0 += 64 * CT_UNR) {
        int k_[CT_UNR]; uint2 w_[CT_UNR]; int64_t g_[CT_UNR];
#pragma unroll
        for (int u = 0; u < CT_UNR; u++) {
            const int e = e0 + 64 * u;
            int k = 0;
#pragma unroll
            for (int st = CT / 2; st > 0; st >>= 1) k += (pref[k + st] <= e) ? st : 0;     // last k with pref[k] <= e
            k_[u] = k;
            g_[u] = slotv[k] + (e - pref[k]);
            w_[u] = e < M ? c.stage[g_[u]] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < CT_UNR; u++) {
            const int e = e0 + 64 * u;
            if (e >= M) break;
            const int k = k_[u];
            const uint2 w = w_[u];
            const ShardDev &sh = c.shards[UNI ? si0 : siv[k]];
            const int64_t o = dstv[k] + (e - pref[k]);
            if (o < sh.cap) {
                sh.o_read[o] = r0v[k] + (int32_t)(w.y & 0x3FFu);
                sh.o_var[o] = (int32_t)w.x;
                sh.o_code[o] = (uint8_t)((w.y >> 10) & 15u);
                if (sh.o_aux0) {             // the two planes only the mapper's text output reads (NULL for a caller that wants (record, variant, code))
                    uint32_t a0 = w.y >> 16, a1 = 0u;
                    if (w.y & 0x8000u) a0 = c.side[g_[u]];
                    if (w.y & 0x4000u) a1 = c.side[c.slots + g_[u]];
                    sh.o_aux0[o] = a0;
                    sh.o_aux1[o] = a1;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_compact(CompactArgs c) {
    __shared__ int s_pref[4][CT + 1];
    __shared__ int64_t s_dst[4][CT], s_slot[4][CT];
    __shared__ int32_t s_r0[4][CT], s_si[4][CT];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t T0 = ((int64_t)blockIdx.x * 4 + w) * CT;
    const int64_t T = T0 + lane;
    int n = 0, si = -1;
    if (lane < CT && T < c.ntiles) {
        n = c.tile_total[T];
        int64_t first = T * (int64_t)c.slot_cap;
        if (n > c.slot_cap) {
            const uint32_t ov = c.tile_ovf[T];
            if (ov != 0xFFFFFFFFu) first = (int64_t)ov; else n = c.slot_cap;         // (no room in the overflow area: the batch is redone with a larger one)
        }
        s_slot[w][lane] = first;
        si = (c.tile_w0[4 * T + 1] >> 16) & 0x1FFF;
        s_dst[w][lane] = calls_before(c.chunk_base, c.tile_pref, T) - c.shard_base[si];
        s_r0[w][lane] = (int32_t)((T - c.tile0[si]) * c.tile_reads);
        s_si[w][lane] = si;
    }
    const int incl = wave_incl_scan(n);
    if (lane < CT) s_pref[w][lane + 1] = incl;
    if (lane == 0) s_pref[w][0] = 0;
    __syncthreads();
    const int M = __builtin_amdgcn_readlane(incl, CT - 1);
    if (M == 0) return;
    const int si0 = __builtin_amdgcn_readfirstlane(si);
    const bool uni = __ballot(si >= 0 && si != si0) == 0ull;
    if (uni) compact_run<true>(c, s_pref[w], s_dst[w], s_r0[w], s_si[w], s_slot[w], T0, M, lane, si0);
    else compact_run<false>(c, s_pref[w], s_dst[w], s_r0[w], s_si[w], s_slot[w], T0, M, lane, si0);
}

}  // namespace

// All shards of a submission run through ONE grid per stage (pre-pass, k_map, tile scan, per-shard totals, compaction): a whole
// genome is 5 launches and one host wait, with no per-shard ramp-up / tail.
// (Tried and dropped, round 3: k_map writing every call to its final place, the calls before a tile found by decoupled look-back over
// per-tile status words -- no staging, no k_compact.  A tile knows its count only at the END of its work (the count needs the quality
// bytes), so its flush waits for every earlier tile still in flight: 3.01 ms against 1.32 + 0.16 ms for k_map + k_compact.)  A batch whose densest tile overflows its staging slot
// is redone with larger slots (rare: the slot capacity is kept across calls).
static int launch_map_batch(phz_ctx *ctx, int n, const phz_reads *r, const phz_variants *v, int baseq, const phz_calls *out, int64_t *n_calls);

int phz_launch_map_batch(phz_ctx *ctx, int n, const phz_reads *r, const phz_variants *v, int baseq, const phz_calls *out,
                         int64_t *n_calls) {
    const int s = launch_map_batch(ctx, n, r, v, baseq, out, n_calls);
    // map_tab_image claims "these bytes are in ctx->map_tab" (nobody else writes map_tab; its [tab_bytes, +m*8) tail, shard_base, is recomputed by every
    // launch).  A submission that failed anywhere between the upload and its stream wait may not have delivered them: forget the image.
    if (s != PHZ_OK && s != PHZ_E_CAPACITY) { ctx->map_tab_image.clear(); ctx->map_tab_dev = nullptr; memset(ctx->map_ovf_image, 0, sizeof ctx->map_ovf_image); }
    return s;
}

static int launch_map_batch(phz_ctx *ctx, int n, const phz_reads *r, const phz_variants *v, int baseq, const phz_calls *out, int64_t *n_calls) {
    for (int i = 0; i < n; i++) n_calls[i] = 0;
    if (n <= 0) return PHZ_OK;
    int rpt = 2, blk = 128;
    { const char *e = getenv("PHZ_MAP_RPT"); if (e && atoi(e) > 0) rpt = atoi(e); }
    { const char *e = getenv("PHZ_MAP_BLOCK"); if (e && atoi(e) > 0) blk = atoi(e); }
    // (64 x 2 / 64 x 4 / 128 x 4 / 256 x 4 records per tile were swept in rounds 1-3 -- 1.84 ms and worse against 1.28 -- and are no longer built:
    // their tiles cannot be held at eight waves per SIMD)
    if (!((blk == 128 || blk == 256) && rpt == 2)) return phz_fail(ctx, PHZ_E_ARG, "bad PHZ_MAP_BLOCK / PHZ_MAP_RPT");
    const int tile_reads = blk * rpt;
    // live shards (records and variants present) and their tile ranges
    std::vector<int> live;
    for (int i = 0; i < n; i++) {
        if (v[i].n > 0x7fffffff) return phz_fail(ctx, PHZ_E_ARG, "too many variants in one shard");
        if (r[i].n_reads > 0 && v[i].n > 0) live.push_back(i);
    }
    const int m = (int)live.size();
    if (m == 0) return PHZ_OK;
    // host image of the shard table: [ShardDev x m][tile0 x (m+1)], pinned; device copy in ctx->map_tab
    const size_t tab_bytes = (size_t)m * sizeof(ShardDev) + (size_t)(m + 1) * 8;
    if (int s = phz_reserve_host(ctx, ctx->h_shard_tab, tab_bytes)) return s;
    if (int s = phz_reserve(ctx, ctx->map_tab, tab_bytes + (size_t)m * 8 + 64)) return s;          // [table][shard_base x m][overflow-area record]
    ShardDev *hs = (ShardDev *)ctx->h_shard_tab.p;
    int64_t *ht0 = (int64_t *)((char *)ctx->h_shard_tab.p + (size_t)m * sizeof(ShardDev));
    int64_t ntiles = 0;
    for (int k = 0; k < m; k++) {
        const int i = live[(size_t)k];
        ShardDev &d = hs[k];
        d.pos = r[i].pos; d.cigar_off = r[i].cigar_off; d.cigar = r[i].cigar; d.seq_off = r[i].seq_off; d.seq2 = r[i].seq2; d.qual = r[i].qual; d.bq = r[i].bq;
        d.n = r[i].n_reads; d.vpos = v[i].pos; d.nv = (int)v[i].n; d.pad = 0;
        d.o_read = out[i].read_idx; d.o_var = out[i].var_idx; d.o_code = out[i].code; d.o_aux0 = out[i].aux0; d.o_aux1 = out[i].aux1;
        if (!d.o_aux0 || !d.o_aux1) { d.o_aux0 = nullptr; d.o_aux1 = nullptr; }           // both or none
        d.cap = out[i].cap;
        ht0[k] = ntiles;
        ntiles += (r[i].n_reads + tile_reads - 1) / tile_reads;
    }
    ht0[m] = ntiles;
    // every shard of the submission carries the one-byte plane (a caller's choice: phz_reads.bq; PHZ_MAP_ONE_PLANE=1 for the Python host): the ONE instantiation
    // reads it (PHZ_MAP_TWO_PLANES=1: the two-plane instantiation regardless).  Same-box A/B of the two production instantiations on the whole-genome sample:
    // 1.0940 against 1.0928 ms -- the plane buys nothing there, so nothing in the product builds it by default
    bool one_plane = m > 0 && getenv("PHZ_MAP_TWO_PLANES") == nullptr;
    for (int k = 0; k < m; k++) if (!hs[k].bq) one_plane = false;
    if (ntiles >= (1ll << 31)) return phz_fail(ctx, PHZ_E_ARG, "too many tiles in one submission");
    if (m > 0x1FFF) return phz_fail(ctx, PHZ_E_ARG, "more than 8191 shards in one submission");
    const ShardDev *d_shards = (const ShardDev *)ctx->map_tab.p;
    const int64_t *d_tile0 = (const int64_t *)((char *)ctx->map_tab.p + (size_t)m * sizeof(ShardDev));
    int64_t *d_shard_base = (int64_t *)((char *)ctx->map_tab.p + tab_bytes);
    hipStream_t sm = ctx->stream;
    // the table of a repeated submission (same shards, same output buffers) is already on the device
    if (ctx->map_tab_image.size() != tab_bytes || ctx->map_tab_dev != ctx->map_tab.p || memcmp(ctx->map_tab_image.data(), ctx->h_shard_tab.p, tab_bytes) != 0) {
        PHZ_HIP(ctx, hipMemcpyAsync(ctx->map_tab.p, ctx->h_shard_tab.p, tab_bytes, hipMemcpyHostToDevice, sm));
        ctx->map_tab_image.assign((const char *)ctx->h_shard_tab.p, (const char *)ctx->h_shard_tab.p + tab_bytes);
        ctx->map_tab_dev = ctx->map_tab.p;
    }
    if ((int)ctx->map_ev.size() < 2) {
        ctx->map_ev.resize(2, nullptr);
        for (auto &e : ctx->map_ev) if (!e) PHZ_HIP(ctx, hipEventCreate(&e));
    }
    DevBuf *S = ctx->scratch;      // 17: tile_total, 18: packed staging slots
    if (int s = phz_reserve(ctx, ctx->tile_w0, (size_t)ntiles * 16)) return s;
    if (int s = phz_reserve(ctx, S[17], (size_t)ntiles * 4)) return s;
    const int nchunks = (int)((ntiles + 1023) / 1024);
    if (int s = phz_reserve(ctx, ctx->desc, (size_t)ntiles * 4 + (size_t)nchunks * 24 + 64)) return s;
    int32_t *tile_pref = (int32_t *)ctx->desc.p;
    int64_t *chunk_sum = (int64_t *)((char *)ctx->desc.p + (((size_t)ntiles * 4 + 15) & ~(size_t)15));
    int64_t *chunk_base = chunk_sum + nchunks;
    int32_t *chunk_max = (int32_t *)(chunk_base + nchunks);
    if (int s = phz_reserve(ctx, ctx->scalars, (size_t)8 * (m + 3) + 128)) return s;           // [0] total, [1] densest tile, [2, 2 + m) calls per shard, [2 + m] overflow-area cursor, then the OvfArea record
    if (int s = phz_reserve_host(ctx, ctx->h_scalars, (size_t)8 * (m + 3) + 128)) return s;
    if (int s = phz_reserve(ctx, S[19], (size_t)ntiles * 4)) return s;                          // tile_ovf
    unsigned long long *scal = (unsigned long long *)ctx->h_scalars.p;
    // Staging: every tile owns a slot of slot_cap calls (half a tile's records: the typical RNA-seq tile has ~60) and a tile with more takes a stretch of
    // the OVERFLOW AREA behind the slots (one cursor step per such tile).  The area starts at a quarter of the slots' size and is kept at what the densest
    // submission so far needed; a submission that needs more is redone once with exactly that (round 4 sized EVERY slot for the densest tile: 50 GB for a
    // deep sample with one dense region).  PHZ_MAP_SLOT_CAP: another slot size (tests: 8, so that most tiles overflow).
    if (ctx->map_slot_cap <= 0 || ctx->map_tile_reads != tile_reads) {
        ctx->map_slot_cap = tile_reads / 2 < 64 ? 64 : tile_reads / 2; ctx->map_tile_reads = tile_reads;
        const char *e = getenv("PHZ_MAP_SLOT_CAP"); if (e && atoi(e) >= 1) ctx->map_slot_cap = atoi(e);
    }
    float ms_total = 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        const int slot_cap = ctx->map_slot_cap;
        const size_t base_slots = (size_t)ntiles * (size_t)slot_cap;
        if (ctx->map_ovf_cap < (int64_t)(base_slots / 4)) ctx->map_ovf_cap = (int64_t)(base_slots / 4);
        if (ctx->map_ovf_cap < 65536) ctx->map_ovf_cap = 65536;
        const size_t slots = base_slots + (size_t)ctx->map_ovf_cap;
        if (slots >= 0xFFFFFFF0ull) return phz_fail(ctx, PHZ_E_ARG, "K_map staging area beyond 2^32 slots: submit the shards in smaller batches");
        if (int s = phz_reserve(ctx, S[18], slots * 16)) return s;
        MapBatch bt;
        bt.shards = d_shards; bt.tile0 = d_tile0; bt.n_shards = m; bt.baseq = baseq;
        bt.stage = (uint2 *)S[18].p; bt.side = (uint32_t *)((char *)S[18].p + slots * 8); bt.slots = (int64_t)slots;
        bt.tile_w0 = (int32_t *)ctx->tile_w0.p; bt.tile_total = (int32_t *)S[17].p;
        bt.slot_cap = slot_cap; bt.ntiles = ntiles;
        {
            OvfArea *h_ov = (OvfArea *)((char *)ctx->h_scalars.p + (size_t)8 * (m + 3) + 64);          // (pinned; behind the words the read-back fills)
            h_ov->tile_first = (uint32_t *)S[19].p; h_ov->cursor = (unsigned long long *)ctx->scalars.p + (2 + m); h_ov->base = (long long)base_slots; h_ov->cap = (long long)ctx->map_ovf_cap;
            OvfArea *d_ov = (OvfArea *)((char *)ctx->map_tab.p + tab_bytes + (size_t)m * 8);            // in K_map's own buffer: ctx->scalars is shared scratch
            // (uploaded only when it changed -- like the shard table --; the cursor is zeroed by the pre-pass kernel: no extra operation on the stream per step)
            long long img[5] = {(long long)(intptr_t)h_ov->tile_first, (long long)(intptr_t)h_ov->cursor, h_ov->base, h_ov->cap, (long long)(intptr_t)d_ov};
            if (memcmp(img, ctx->map_ovf_image, sizeof img) != 0) {
                PHZ_HIP(ctx, hipMemcpyAsync(d_ov, h_ov, sizeof(OvfArea), hipMemcpyHostToDevice, sm));
                memcpy(ctx->map_ovf_image, img, sizeof img);
            }
            bt.ovf = d_ov;
        }
        { const char *e = getenv("PHZ_MAP_DBG"); bt.dbg = e ? atoi(e) : 0; }
        bt.prof = nullptr;
        if (bt.dbg & 2048) { if (int s = phz_reserve(ctx, S[21], (size_t)ntiles * 64)) return s; bt.prof = (unsigned long long *)S[21].p; }
        hipLaunchKernelGGL(k_tile_window, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, sm, bt, tile_reads);
        PHZ_HIP(ctx, hipEventRecord(ctx->map_ev[0], sm));
        unsigned dyn_lds = 0;          // experiment: dynamic LDS nobody uses, to bound the workgroups per CU (PHZ_MAP_DYNLDS bytes)
        { const char *e = getenv("PHZ_MAP_DYNLDS"); if (e && atoi(e) > 0) dyn_lds = (unsigned)atoi(e); }
#define PHZ_LAUNCH_MAP(B, R) do { if (bt.dbg) hipLaunchKernelGGL((k_map<B, R, true>), dim3((unsigned)ntiles), dim3(B), dyn_lds, sm, bt); \
                                 else if (one_plane) hipLaunchKernelGGL((k_map<B, R, false, true>), dim3((unsigned)ntiles), dim3(B), dyn_lds, sm, bt); \
                                 else hipLaunchKernelGGL((k_map<B, R, false>), dim3((unsigned)ntiles), dim3(B), dyn_lds, sm, bt); } while (0)
        if (blk == 128) PHZ_LAUNCH_MAP(128, 2);
        else PHZ_LAUNCH_MAP(256, 2);
#undef PHZ_LAUNCH_MAP
        PHZ_HIP(ctx, hipEventRecord(ctx->map_ev[1], sm));
        hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)nchunks), dim3(1024), 0, sm, (const int32_t *)S[17].p, ntiles, tile_pref, chunk_sum,
                           chunk_max);
        hipLaunchKernelGGL(k_chunk_base, dim3(1), dim3(1024), 0, sm, (const int64_t *)chunk_sum, (const int32_t *)chunk_max, nchunks, chunk_base,
                           (unsigned long long *)ctx->scalars.p);
        hipLaunchKernelGGL(k_shard_totals, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, sm, d_tile0, m, ntiles, (const int32_t *)tile_pref,
                           (const int64_t *)chunk_base, (unsigned long long *)ctx->scalars.p, d_shard_base);
        CompactArgs c;
        c.stage = bt.stage; c.side = bt.side; c.slots = bt.slots;
        c.shards = d_shards; c.tile0 = d_tile0; c.n_shards = m;
        c.tile_total = (const int32_t *)S[17].p; c.tile_pref = tile_pref; c.chunk_base = chunk_base; c.shard_base = d_shard_base;
        c.tile_w0 = bt.tile_w0;
        c.slot_cap = slot_cap; c.tile_reads = tile_reads; c.ntiles = ntiles; c.tile_ovf = (const uint32_t *)S[19].p;
        hipLaunchKernelGGL(k_compact, dim3((unsigned)((ntiles + 4 * CT - 1) / (4 * CT))), dim3(256), 0, sm, c);
        PHZ_HIP(ctx, hipGetLastError());
        PHZ_HIP(ctx, hipMemcpyAsync(scal, ctx->scalars.p, (size_t)8 * (m + 3), hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipStreamSynchronize(sm));
        float ms = 0;
        PHZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->map_ev[0], ctx->map_ev[1]));
        if (bt.prof) {          // mean time between the stamps of a tile (wall_clock64: 100 MHz), first wave's view
            std::vector<unsigned long long> pr((size_t)ntiles * 8);
            PHZ_HIP(ctx, hipMemcpy(pr.data(), bt.prof, pr.size() * 8, hipMemcpyDeviceToHost));
            double seg[7] = {0, 0, 0, 0, 0, 0, 0}; double life = 0;
            for (int64_t t = 0; t < ntiles; t++) { for (int k = 0; k < 7; k++) seg[k] += (double)(pr[8 * t + k + 1] - pr[8 * t + k]); life += (double)(pr[8 * t + 7] - pr[8 * t]); }
            fprintf(stderr, "[k_map profile] %lld tiles, kernel %.3f ms, tile lifetime %.2f us: stage loads -> LDS %.2f | single-run records + first gathers %.2f | barrier %.2f | "
                            "multi-op walk %.2f | candidate gathers + resolve %.2f | scan %.2f | flush %.2f us\n", (long long)ntiles, ms, life / ntiles / 100.0,
                    seg[0] / ntiles / 100.0, seg[1] / ntiles / 100.0, seg[2] / ntiles / 100.0, seg[3] / ntiles / 100.0, seg[4] / ntiles / 100.0, seg[5] / ntiles / 100.0, seg[6] / ntiles / 100.0);
        }
        ms_total += ms;
        if ((int64_t)scal[2 + m] <= ctx->map_ovf_cap) break;          // every dense tile found room
        if (attempt == 2) return phz_fail(ctx, PHZ_E_HIP, "K_map staging overflow area did not converge");
        ctx->map_ovf_cap = (int64_t)scal[2 + m] + (int64_t)(scal[2 + m] / 16) + 4096;      // the calls of all dense tiles (+ slack for the next submission): redo the batch
    }
    ctx->last_ms[PHZ_T_MAP] = ms_total; ctx->total_ms[PHZ_T_MAP] += ms_total; ctx->launches[PHZ_T_MAP]++;
    int st = PHZ_OK;
    for (int k = 0; k < m; k++) {
        const int i = live[(size_t)k];
        n_calls[i] = (int64_t)scal[2 + k];
        if (n_calls[i] > out[i].cap) st = PHZ_E_CAPACITY;
    }
    return st;
}

int phz_launch_map(phz_ctx *ctx, const phz_reads &r, const phz_variants &v, int baseq, const phz_calls &out,
                   int64_t *n_calls) {
    return phz_launch_map_batch(ctx, 1, &r, &v, baseq, &out, n_calls);
}
