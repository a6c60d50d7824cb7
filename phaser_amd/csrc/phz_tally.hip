// K_tally family: what phaser/phaser.py does between the mapper's call files and the binomial test, for ANY number of
// (chromosome, BAM) shards in one submission (variant and QNAME ids are offset per chromosome into one index space, so a whole
// genome is one call and nothing ever pairs across chromosomes):
//   k_as_hist      :545-553   AS column histogram (host turns it into numpy.percentile's value)
//   k_tile_base               window base of every tile of 1,024 call lines: the variant of its first line, made monotone over the shard's tiles
//   k_line         :1287-1328 process_mapping_result: AS cutoff, allele class, first-appearance index; per TILE the kept lines of every
//                             (window variant, class) -- the band matrix M -- and the tile's QNAMEs as bits of two bitmaps over the QNAME ids
//                             ("seen by a tile", "seen by a second tile")
//   k_colscan                 column sums and column prefixes of M: per-variant line counters, lines per (variant, allele, BAM) read list, and
//                             for every (tile, list) the number of the list's entries in EARLIER tiles -- so that k_tile can put a read-list
//                             entry at its final, line-ordered place (:1318, :917-931, :1086-1115) without a sort
//   k_tile         :636-640, :558-581, :1265-1285  the lines of a QNAME are its mates' records, a few hundred bases apart: nearly every QNAME has
//                             all its lines inside one tile.  A workgroup groups its tile's lines by QNAME in LDS (hash on the id); a group whose
//                             QNAME no other tile has seen is finished in place: sorted, its first ref/alt line, the connectivity-map rank of its
//                             variants, the distinct (variant, class) items and the per-variant distinct-QNAME counters.  Lines of the few QNAMEs
//                             that straddle tiles (or BAMs) are spilled.  Read-list entries go straight to their place: list start + entries in
//                             earlier tiles (k_colscan) + entries in earlier rows of the tile + rank inside the row (wave ballots)
//   k_groups                  the spilled QNAMEs (a per cent of them): gathered into groups through cursors, one thread per group, incl. the
//                             owner BAM of the read_vars list ("last BAM wins", stale-variable quirk)
//   k_pairs        :1602-1632 every QNAME contributes one count to cell (class_a, class_b) of every variant pair it touches -- the nine set
//                             intersections of test_variant_connection for all pairs at once (LDS hash -> global hash)
//   k_edge_*                  edge list in (a, b) order: counting sort by a over the USED hash slots, tiny groups sorted by b; the table is
//                             cleaned slot by slot while it is read (no per-call memset of a 400 MB table)
//   k_rl_sort_dirty           the few read lists with a line outside its tile's variant window (a read spliced over hundreds of het SNPs) are
//                             filled through cursors and sorted afterwards
//   k_components   :1861-1882/:1985-1998 connected components (lock-free union-find)
// Integer work, hand-written kernels and primitives only (phz_sort.h).  Nothing is swept by the number of QNAME ids per call except two
// bitmaps (one bit per id each, cleared per call); the per-QNAME cursor array is touched for spilled QNAMEs only and returns to zero.
#include <cmath>
#include <cstring>
#include "phz_internal.h"
#include "phz_sort.h"
#include "phz_uf.h"

namespace {

constexpr uint64_t KEY_DROPPED = ~0ull;
// Statistics that every workgroup of a kernel adds to live in SPREAD counters: N_SPREAD copies, one memory sector each, picked by the
// workgroup number (atomics on one address are served one after the other, ~8 ns each: ten thousand workgroups x 3 counters on one sector
// would be a quarter of a millisecond of queueing); the host adds the copies up.  spread[c][0] distinct items, [1] pair events, [2] kept lines
constexpr int N_SPREAD = 32, SPREAD_WORDS = 8;
constexpr int AS_LDS_BINS = 8192;       // AS in [-4096, 4096) is histogrammed in LDS

struct LinesDev {
    int64_t n;
    const int32_t *read_idx, *var_idx;
    const uint8_t *code;
    const int32_t *read_qid, *read_as;
    const uint8_t *read_has_as;
    double cutoff;
    int use_cutoff, bam;
    int32_t var_base;          // first variant of the shard's chromosome in the call's variant space
    uint32_t qid_base;         // first QNAME id of the shard's chromosome in the call's id space
    int64_t line_base;         // index of the shard's first line in the call's line space (shards ordered chromosome, BAM)
    const int16_t *read_as16;  // 2-byte AS plane with the has-AS flag folded in (phz.h), or NULL
    int32_t nv_chrom;          // variants of the shard's chromosome: [var_base, var_base + nv_chrom) of the call's variant space
    const double *cut_dev;     // phz_lines.as_cutoff_dev: {cutoff, found, ...} computed on the device (phz_as_cutoff_enqueue), or NULL
};
// AS of record r from whichever form the shard carries; false = the record has no AS tag.  *range = the value lies outside the band the
// histogram accepts (the caller refuses the input)
__device__ __forceinline__ bool as_of(const LinesDev &L, int r, int *a, bool *range) {
    if (L.read_as16) {
        const int x = L.read_as16[r];
        *a = x; *range = x == PHZ_AS16_RANGE || x == -PHZ_AS16_RANGE;
        return x != PHZ_AS16_NONE;
    }
    if (L.read_has_as && !L.read_has_as[r]) { *a = 0; *range = false; return false; }
    const int x = L.read_as[r];
    *a = x; *range = x < -32768 || x >= 32768;
    return true;
}

// All shards of a call go through ONE grid per per-line stage: block b belongs to the shard s with blk0[s] <= b < blk0[s+1]
// (a genome is 22+ shards of well under a million lines each: one launch per shard is mostly ramp-up and tail).
struct LinesTab {
    const LinesDev *L;
    const uint32_t *blk0;      // [n + 1]
    int n;
};
__device__ __forceinline__ int tab_find(const LinesTab &t, uint32_t blk) {
    int lo = 0, hi = t.n - 1;          // largest s with blk0[s] <= blk (shards without blocks share their successor's start)
    while (lo < hi) {
        const int m = (lo + hi + 1) >> 1;
        if (t.blk0[m] <= blk) lo = m; else hi = m - 1;
    }
    return lo;
}
// BAM of a line of the call's line space (shards ordered by line_base)
__device__ __forceinline__ int bam_of_line(const LinesTab &t, uint32_t g) {
    int lo = 0, hi = t.n - 1;
    while (lo < hi) {
        const int m = (lo + hi + 1) >> 1;
        if ((uint64_t)t.L[m].line_base <= g) lo = m; else hi = m - 1;
    }
    return t.L[lo].bam;
}

__global__ __launch_bounds__(256) void k_as_hist(LinesTab T, unsigned long long *hist, unsigned int *out_of_range) {
    const int sh_ = tab_find(T, blockIdx.x);
    const LinesDev L = T.L[sh_];
    const uint32_t bx = blockIdx.x - T.blk0[sh_], gx = T.blk0[sh_ + 1] - T.blk0[sh_];
    __shared__ unsigned int s_h[AS_LDS_BINS];
    for (int j = threadIdx.x; j < AS_LDS_BINS; j += 256) s_h[j] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < L.n; i += (int64_t)gx * 256) {
        const int r = L.read_idx[i];
        int a; bool range;
        if (!as_of(L, r, &a, &range)) continue;
        const int b = a + AS_LDS_BINS / 2;
        if (range) { if (out_of_range) atomicOr(out_of_range, 1u); }    // an alignment score outside int16: the caller refuses the input
        else if ((unsigned)b < (unsigned)AS_LDS_BINS) atomicAdd(&s_h[b], 1u);
        else atomicAdd(&hist[a + 32768], 1ull);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < AS_LDS_BINS; j += 256)
        if (s_h[j]) atomicAdd(&hist[j - AS_LDS_BINS / 2 + 32768], (unsigned long long)s_h[j]);
}

// the occupied bins of the 64 Ki-bin histogram as (bin, count) pairs (alignment scores live in a narrow band: a few dozen bins), so that the
// host reads a few hundred bytes instead of 512 KB.  One thread per bin, one cursor step per workgroup with occupied bins: the pairs come out
// in no particular order (the caller sorts the handful); n_out[0] = number of pairs (may exceed cap: the caller then takes the dense path)
__global__ __launch_bounds__(1024) void k_hist_compact(const unsigned long long *hist, int cap, int32_t *bins, unsigned long long *counts, int32_t *n_out) {
    __shared__ int s_w[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bin = blockIdx.x * 1024 + tid;
    const unsigned long long c = hist[bin];
    const unsigned long long m = __ballot(c != 0ull);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < 16; w++) { const int x = s_w[w]; s_w[w] = tot; tot += x; }
        s_base = tot ? atomicAdd(n_out, tot) : 0;
    }
    __syncthreads();
    if (c) {
        const int at = s_base + s_w[wave] + __popcll(m & (lane ? (~0ull >> (64 - lane)) : 0ull));
        if (at < cap) { bins[at] = bin; counts[at] = c; }
    }
}

// ------------------------------------------------------------------------------------------------ tiles, windows, the per-line pass
// Call lines arrive in mapper order (record, then variant) and records are coordinate-sorted, so a TILE of TL consecutive lines of a shard
// touches a narrow run of variant indices -- its WINDOW [wb, wb + TWT) -- and deeply covered variants repeat hundreds of times in a row.
// k_line and k_tile work on the same tiles (workgroup = tile) and the same windows.
constexpr int TW = 1024;            // variants per LDS window of k_groups
#ifndef PHZ_TALLY_TILE
#define PHZ_TALLY_TILE 1024        // (the emulation tests also build a 256-line variant so that small fixtures straddle tiles)
#endif
constexpr int TL = PHZ_TALLY_TILE; // lines per tile
static_assert(TL == 1024 || TL == 512 || TL == 256, "tile of 256 / 512 / 1024 lines");
constexpr int TWT = 256;           // variants per tile window (1,024 lines span ~100 variants)
constexpr int QW = 16384;          // QNAME ids of the LDS bitmap of k_line (a tile's QNAMEs: the ids handed out while its ~4,000 records went by, and their mates')
#ifndef PHZ_LINE_TB
#define PHZ_LINE_TB 512
#endif
constexpr int LINE_TB = PHZ_LINE_TB < TL ? PHZ_LINE_TB : TL;   // threads per workgroup of k_line

// Window base of every tile: the variant of the tile's first line minus a little room (mate pairs / overlapping records), then made
// non-decreasing over the shard's tiles by a suffix minimum.  (The first line of a tile can sit in the middle of a spliced record, far to the
// right of the records that follow: without the suffix minimum such a tile would push later, lower windows out of order, and k_colscan finds
// the tiles of a variant block by bisection over these bases.)  A line outside its tile's window -- a read spliced over hundreds of het SNPs --
// is a FAR line: global atomics, and its read list is filled through a cursor and sorted afterwards (k_rl_sort_dirty).
// One workgroup per shard.
__global__ __launch_bounds__(256) void k_tile_base(LinesTab TT, int32_t *tile_wb) {
    const LinesDev L = TT.L[blockIdx.x];
    const uint32_t T0 = TT.blk0[blockIdx.x], n = TT.blk0[blockIdx.x + 1] - T0;
    if (n == 0) return;
    __shared__ int s_min[256];
    const uint32_t per = (n + 255u) / 256u;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    int m = 0x7FFFFFFF;
    for (uint32_t t = lo; t < hi; t++) { const int v = L.var_idx[(int64_t)t * TL] + L.var_base; m = v < m ? v : m; }
    s_min[threadIdx.x] = m;
    __syncthreads();
    int after = 0x7FFFFFFF;                                  // minimum over the chunks behind this thread's
    for (int j = threadIdx.x + 1; j < 256; j++) after = s_min[j] < after ? s_min[j] : after;
    for (uint32_t t = hi; t-- > lo;) {
        const int v = L.var_idx[(int64_t)t * TL] + L.var_base;
        after = v < after ? v : after;
        tile_wb[T0 + t] = after - 64 > 0 ? after - 64 : 0;
    }
}

struct LineOut {
    const uint8_t *a0, *a1;
    uint8_t *line_cls;               // [n_lines] 0 ref / 1 alt / 2 other / 255 dropped by the AS cutoff
    uint32_t *line_q;                // [n_lines] QNAME id of the line in the call's id space
    int32_t *var_count;              // [nv*3]   (far lines only: the rest comes from k_colscan)
    unsigned long long *var_first;   // [nv]
    uint32_t *rl_cnt;                // [nv*2*nb] kept ref/alt lines per (variant, allele, BAM)  (far lines only)
    const int32_t *tile_wb;          // [tiles] window base
    uint16_t *tile_m;                // [tiles][TWT*3] kept lines per (window variant, class)
    uint32_t *q_seen, *q_dup;        // one bit per QNAME id: some tile holds a kept line of it / a second tile (or BAM) does too
    uint32_t *rl_dirty, *dirty_list; // one bit per read list: it has a far line; the lists with the bit, once each (cursor: counters[11])
    unsigned long long *counters;    // [12] far lines
    int nb;
    unsigned long long *prof;        // PHZ_TALLY_PROFILE=1: start / end clock of every workgroup
};

__global__ __launch_bounds__(LINE_TB) void k_line(LinesTab T, LineOut O) {
    if (O.prof && threadIdx.x == 0) O.prof[2 * (size_t)blockIdx.x] = wall_clock64();
    const int sh_ = tab_find(T, blockIdx.x);
    const LinesDev L = T.L[sh_];
    const uint32_t bx = blockIdx.x - T.blk0[sh_];
    __shared__ int s_cnt[TWT * 3];
    __shared__ unsigned long long s_first[TWT];
    __shared__ uint32_t s_qb[QW / 32];
    __shared__ unsigned int s_kept, s_far;
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)bx * TL;
    for (int j = tid; j < TWT * 3; j += LINE_TB) s_cnt[j] = 0;
    for (int j = tid; j < TWT; j += LINE_TB) s_first[j] = ~0ull;
    for (int j = tid; j < QW / 32; j += LINE_TB) s_qb[j] = 0u;
    if (tid == 0) { s_kept = 0; s_far = 0; }
    const int wb = O.tile_wb[blockIdx.x];
    // the LDS bitmap covers the QNAME ids around the tile's first line (ids are handed out in coordinate order: a tile's QNAMEs are the
    // ones that first appeared while its records went by, and mates of slightly earlier ones)
    const uint32_t q0 = L.qid_base + (uint32_t)L.read_qid[L.read_idx[i0]];
    const uint32_t qb = q0 > (uint32_t)(QW / 2) ? ((q0 - (uint32_t)(QW / 2)) & ~31u) : 0u;
    unsigned int kept = 0, far = 0;
    // all loads of a lane's lines are requested before the first one is used (two dependent rounds: the line, then its record / variant)
    constexpr int K = TL / LINE_TB;
    int l_r[K], l_v[K], l_as[K]; uint32_t l_q[K]; uint8_t l_code[K], l_a0[K], l_a1[K]; bool l_in[K], l_has[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int64_t i = i0 + tid + LINE_TB * k;
        l_in[k] = i < L.n;
        l_r[k] = l_in[k] ? L.read_idx[i] : 0; l_v[k] = l_in[k] ? L.var_idx[i] + L.var_base : 0; l_code[k] = l_in[k] ? L.code[i] : (uint8_t)4;
    }
    // the cutoff of the shard's BAM: a host value, or the block the device-side percentile left (phz_as_cutoff_enqueue; [1] == 0: no AS tag in the BAM, every line kept)
    const bool use_cut = L.use_cutoff != 0 && (L.cut_dev == nullptr || L.cut_dev[1] != 0.0);
    const double cut = L.cut_dev != nullptr ? L.cut_dev[0] : L.cutoff;
    if (bx == 0 && tid == 0 && L.cut_dev != nullptr && L.cut_dev[2] != 0.0) O.counters[13] = 1ull;      // the histogram saw an AS value outside int16: phz_tally refuses the input at its first wait
#pragma unroll
    for (int k = 0; k < K; k++) {
        l_as[k] = 0; l_has[k] = true;
        if (l_in[k] && use_cut) { bool range; l_has[k] = as_of(L, l_r[k], &l_as[k], &range); }
        l_q[k] = l_in[k] ? L.qid_base + (uint32_t)L.read_qid[l_r[k]] : 0u;
        l_a0[k] = l_in[k] ? O.a0[l_v[k]] : (uint8_t)0; l_a1[k] = l_in[k] ? O.a1[l_v[k]] : (uint8_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (!l_in[k]) continue;
        const int64_t g = L.line_base + i0 + tid + LINE_TB * k;
        const int v = l_v[k];
        const bool keep = !use_cut || (l_has[k] && (double)l_as[k] >= cut);
        if (!keep) { O.line_cls[g] = 255; continue; }
        kept++;
        const uint8_t c = l_code[k];
        // codes 5 / 6 come from the general (indel) mapper, which compared the text with the allele strings itself
        const int cls = c == 5 ? 0 : (c == 6 ? 1 : ((c < 4 && c == l_a0[k]) ? 0 : ((c < 4 && c == l_a1[k]) ? 1 : 2)));
        O.line_cls[g] = (uint8_t)cls;
        O.line_q[g] = l_q[k];
        const unsigned d = (unsigned)(v - wb);
        if (d < (unsigned)TWT) {
            atomicAdd(&s_cnt[d * 3 + cls], 1);
            atomicMin(&s_first[d], (unsigned long long)g);
        } else {                                                 // a far line
            far++;
            atomicAdd(&O.var_count[(int64_t)v * 3 + cls], 1);
            atomicMin(&O.var_first[v], (unsigned long long)g);
            if (cls < 2) {
                const uint32_t e = ((uint32_t)v * 2u + (uint32_t)cls) * (uint32_t)O.nb + (uint32_t)L.bam;
                atomicAdd(&O.rl_cnt[e], 1u);
                const uint32_t bit = 1u << (e & 31u);
                if (!(atomicOr(&O.rl_dirty[e >> 5], bit) & bit)) O.dirty_list[atomicAdd(&O.counters[11], 1ull)] = e;      // the first far line of the list puts it on the work list
            }
        }
        const uint32_t dq = l_q[k] - qb;
        if (dq < (uint32_t)QW) atomicOr(&s_qb[dq >> 5], 1u << (dq & 31u));
        else {                                                   // outside the LDS bitmap (a deep region: mates thousands of ids back): straight to the global ones.  Two such
            const uint32_t bit = 1u << (l_q[k] & 31u);           // lines of one tile make their QNAME look shared -- it then takes the general (spill) path, which is always right
            if (atomicOr(&O.q_seen[l_q[k] >> 5], bit) & bit) atomicOr(&O.q_dup[l_q[k] >> 5], bit);
        }
    }
    if (kept) atomicAdd(&s_kept, kept);
    if (far) atomicAdd(&s_far, far);
    __syncthreads();
    // the tile's row of M, two 16-bit counters per store (a tile holds at most TL <= 1,024 lines)
    {
        uint32_t *row = (uint32_t *)(O.tile_m + (size_t)blockIdx.x * (TWT * 3));
        for (int j = tid; j < TWT * 3 / 2; j += LINE_TB) row[j] = (uint32_t)s_cnt[2 * j] | ((uint32_t)s_cnt[2 * j + 1] << 16);
    }
    for (int j = tid; j < TWT; j += LINE_TB) {
        const unsigned long long f = s_first[j];
        if (f != ~0ull) atomicMin(&O.var_first[wb + j], f);
    }
    // the tile's QNAMEs: one atomic per occupied word; the bits somebody else had set before mark QNAMEs that live in more than one tile
    for (int j = tid; j < QW / 32; j += LINE_TB) {
        const uint32_t bits = s_qb[j];
        if (bits) {
            const uint32_t w = (qb >> 5) + (uint32_t)j;
            const uint32_t dup = atomicOr(&O.q_seen[w], bits) & bits;
            if (dup) atomicOr(&O.q_dup[w], dup);
        }
    }
    if (tid == 0 && s_kept) atomicAdd(&O.counters[16 + (blockIdx.x % N_SPREAD) * SPREAD_WORDS + 2], (unsigned long long)s_kept);
    if (tid == 0 && s_far) atomicAdd(&O.counters[12], (unsigned long long)s_far);
    if (O.prof && tid == 0) O.prof[2 * (size_t)blockIdx.x + 1] = wall_clock64();
}

// Columns of the band matrix M (tiles x variants): for every variant of a block of TWT variants the kept lines per class over all tiles of the
// shard -- the per-variant counters and the sizes of the read lists -- and, tile by tile, the running sum BEFORE the tile (tile_b), i.e. the
// number of entries of list (variant, allele, BAM) that lie in earlier tiles = earlier lines.  One workgroup per (shard, variant block); its
// tiles are found by bisection over the monotone window bases; long tile ranges (deep coverage) are cut into CS_PARTS parts that first add up
// their own stretch.
constexpr int CS_PARTS = 4;
struct ColScan { const int32_t *tile_wb; const uint16_t *tile_m; uint32_t *tile_b; int32_t *var_count; uint32_t *rl_cnt; int nb; };
__global__ __launch_bounds__(TWT * CS_PARTS) void k_colscan(LinesTab TC, LinesTab TT, ColScan O) {
    const int sh_ = tab_find(TC, blockIdx.x);
    const LinesDev L = TC.L[sh_];
    const int blk = (int)(blockIdx.x - TC.blk0[sh_]);
    const int a = L.var_base + blk * TWT;                        // first variant of the block
    const int j = threadIdx.x & (TWT - 1), part = threadIdx.x / TWT;
    const int v = a + j;
    const bool live = blk * TWT + j < L.nv_chrom;
    const uint32_t T0 = TT.blk0[sh_], T1 = TT.blk0[sh_ + 1];
    // tiles whose window [wb, wb + TWT) meets [a, a + TWT): a - TWT < wb < a + TWT
    uint32_t lo, hi;
    { uint32_t x = T0, y = T1; while (x < y) { const uint32_t m = (x + y) >> 1; if (O.tile_wb[m] > a - TWT) y = m; else x = m + 1; } lo = x; }
    { uint32_t x = lo, y = T1; while (x < y) { const uint32_t m = (x + y) >> 1; if (O.tile_wb[m] >= a + TWT) y = m; else x = m + 1; } hi = x; }
    const uint32_t n = hi - lo;
    const bool split = n > 16u;
    const uint32_t per = split ? (n + CS_PARTS - 1) / CS_PARTS : n;
    uint32_t t0 = lo + (uint32_t)part * per; t0 = t0 < hi ? t0 : hi;
    const uint32_t t1 = t0 + per < hi ? t0 + per : hi;
    __shared__ uint32_t s_tot[CS_PARTS][TWT * 3];
    uint32_t off0 = 0, off1 = 0, off2 = 0;
    if (split) {
        uint32_t c0 = 0, c1 = 0, c2 = 0;
#pragma unroll 4
        for (uint32_t t = t0; t < t1; t++) {
            const unsigned d = (unsigned)(v - O.tile_wb[t]);
            if (d < (unsigned)TWT) { const uint16_t *m = O.tile_m + (size_t)t * (TWT * 3) + d * 3; c0 += m[0]; c1 += m[1]; c2 += m[2]; }
        }
        s_tot[part][j * 3] = c0; s_tot[part][j * 3 + 1] = c1; s_tot[part][j * 3 + 2] = c2;
        __syncthreads();
        for (int p = 0; p < part; p++) { off0 += s_tot[p][j * 3]; off1 += s_tot[p][j * 3 + 1]; off2 += s_tot[p][j * 3 + 2]; }
    }
    uint32_t r0 = off0, r1 = off1, r2 = off2;
#pragma unroll 4
    for (uint32_t t = t0; t < t1; t++) {
        const unsigned d = (unsigned)(v - O.tile_wb[t]);
        if (d < (unsigned)TWT) {
            const uint16_t *m = O.tile_m + (size_t)t * (TWT * 3) + d * 3;
            uint32_t *bb = O.tile_b + (size_t)t * (TWT * 2) + d * 2;
            const uint32_t c0 = m[0], c1 = m[1], c2 = m[2];
            bb[0] = r0; bb[1] = r1;
            r0 += c0; r1 += c1; r2 += c2;
        }
    }
    if (!live) return;
    r0 -= off0; r1 -= off1; r2 -= off2;                          // this part's own lines (several BAMs of a chromosome add to the same variant: atomics)
    if (r0) { atomicAdd(&O.var_count[(int64_t)v * 3], (int)r0); atomicAdd(&O.rl_cnt[((uint32_t)v * 2u) * (uint32_t)O.nb + (uint32_t)L.bam], r0); }
    if (r1) { atomicAdd(&O.var_count[(int64_t)v * 3 + 1], (int)r1); atomicAdd(&O.rl_cnt[((uint32_t)v * 2u + 1u) * (uint32_t)O.nb + (uint32_t)L.bam], r1); }
    if (r2) atomicAdd(&O.var_count[(int64_t)v * 3 + 2], (int)r2);
}

// group of every spilled QNAME: its line count (the counter goes back to zero and serves as the fill cursor of k_items_spill)
__global__ __launch_bounds__(256) void k_group_plan(int64_t nt, const uint32_t *touched, uint32_t *qcount, uint32_t *cnt_t) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    const uint32_t q = touched[i];
    cnt_t[i] = qcount[q]; qcount[q] = 0u;                        // the spilled lines of the QNAME, added up by the tiles that hold them
}
// ... and the counter becomes the QNAME's write cursor: it starts at the group's base, k_items takes slots from it, k_groups returns it to zero
__global__ __launch_bounds__(256) void k_group_base(int64_t nt, const uint32_t *touched, const uint32_t *base_t, uint32_t *qcount) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nt) qcount[touched[i]] = base_t[i];
}
// item = variant:28 | class:2 | line:32 (sorts by variant, class, line); read-list entry = line:32 | chromosome-local QNAME id:32
__device__ __forceinline__ uint64_t item_pack(uint32_t v, uint32_t cls, uint32_t g) { return ((uint64_t)v << 34) | ((uint64_t)cls << 32) | g; }
// distinct item of a group = QNAME id:32 | variant:28 << 4 | class << 2 | linked (the QNAME id in the high word keeps the groups apart in k_pairs)
__device__ __forceinline__ uint64_t dist_pack(uint32_t q, uint32_t v, uint32_t cls, uint32_t linked) { return ((uint64_t)q << 32) | ((uint64_t)v << 4) | ((uint64_t)cls << 2) | linked; }

constexpr int TH = 2 * TL;         // LDS hash slots (at most TL distinct QNAMEs: load factor <= 0.5)
#ifndef PHZ_TILE_TB
#define PHZ_TILE_TB 512
#endif
constexpr int TILE_TB = PHZ_TILE_TB < TL ? PHZ_TILE_TB : TL;   // threads per workgroup of k_tile
constexpr int TSP = TH / TILE_TB;  // slots per thread
constexpr int TH_SHIFT = TL == 1024 ? 21 : (TL == 512 ? 22 : 23);
constexpr int NROWS = TL / 64;     // rows of 64 consecutive lines in a tile (a wave holds one row per round)
constexpr uint32_t Q_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t RL_DIRTY = 0xFFFFFFFFu;      // base of a dirty list in the tile's LDS table: its entries go through the cursor

struct TileOut {
    const uint8_t *line_cls; const uint32_t *line_q;
    const uint32_t *q_dup;           // bit per QNAME id: lines in more than one tile / BAM (k_line)
    uint32_t *qcount;                // [nq] spilled lines per QNAME, added up here (all zero between calls; k_group_plan / k_groups return it to zero)
    uint64_t *items;                 // [tiles * TL] distinct items of the complete groups in the slots of their tile, holes = KEY_DROPPED
    uint32_t *sp_q; uint64_t *sp_item;       // spilled lines (cursor: low word of counters[10])
    uint32_t *touched;               // spilled QNAMEs, once each (cursor: high word of counters[10])
    unsigned long long *var_rank; int32_t *var_distinct;
    const int32_t *tile_wb; const uint32_t *tile_b;      // window base; entries of every window list in earlier tiles (k_colscan)
    const uint32_t *rl_start, *rl_dirty;
    int32_t *rl_qid; uint32_t *rl_list;                  // the read lists, written in place
    uint32_t *rl_cursor; uint64_t *rl_tmp;               // dirty lists only: entries in arrival order, line | QNAME id (sorted by k_rl_sort_dirty)
    unsigned long long *counters;
    int nb; int64_t nv;
    unsigned long long *prof;
};

__global__ __launch_bounds__(TILE_TB) void k_tile(LinesTab T, TileOut O) {
    if (O.prof && threadIdx.x == 0) O.prof[2 * (size_t)blockIdx.x] = wall_clock64();
    const int sh_ = tab_find(T, blockIdx.x);
    const LinesDev L = T.L[sh_];
    const uint32_t bx = blockIdx.x - T.blk0[sh_];
    __shared__ uint32_t s_q[TH];                 // QNAME id of the slot
    __shared__ uint32_t s_c[TH];                 // lines of the slot's QNAME in this tile -> write cursor of its group -> end of its group (groups lie in slot order)
    __shared__ uint64_t s_it[TL];                // the tile's kept lines as items, group by group; before that: entries per (row, window list), one byte each
    __shared__ int s_cnt[TWT * 3];
    __shared__ unsigned long long s_rank[TWT];
    __shared__ uint32_t s_rl[TL / 2 > TWT * 2 ? TL / 2 : TWT * 2];   // place of the tile's first entry in every window list (RL_DIRTY: the list goes through its cursor);
                                                 // later: the slots whose QNAMEs this tile puts on the list of spilled QNAMEs (16 bits each)
    __shared__ uint16_t s_grp[TL];               // the occupied slots, densely: group g of the tile lives in slot s_grp[g]
    __shared__ uint32_t s_part[TILE_TB / 64];
    __shared__ uint32_t s_nspill, s_ntouch, s_nkept, s_ngroups;
    __shared__ unsigned long long s_obase;
    uint16_t *s_touch = (uint16_t *)s_rl;
    uint8_t *s_mat = (uint8_t *)s_it;            // [NROWS][TWT * 2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t i0 = (int64_t)bx * TL;
    if (tid == 0) { s_nspill = 0; s_ntouch = 0; }
    for (int j = tid; j < TH; j += TILE_TB) { s_q[j] = Q_EMPTY; s_c[j] = 0u; }
    for (int j = tid; j < TWT * 3; j += TILE_TB) s_cnt[j] = 0;
    for (int j = tid; j < TWT; j += TILE_TB) s_rank[j] = ~0ull;
    for (int j = tid; j < TL; j += TILE_TB) s_it[j] = 0ull;
    const int vbase = O.tile_wb[blockIdx.x];
    // ---- 0. where the tile's entries of every window list start: list start + entries in earlier tiles
    for (int x = tid; x < TWT * 2; x += TILE_TB) {
        const int64_t v = (int64_t)vbase + (x >> 1);
        uint32_t base = RL_DIRTY;
        if (v < O.nv) {
            const uint32_t e = ((uint32_t)v * 2u + (uint32_t)(x & 1)) * (uint32_t)O.nb + (uint32_t)L.bam;
            if (!((O.rl_dirty[e >> 5] >> (e & 31u)) & 1u)) base = O.rl_start[e] + O.tile_b[(size_t)blockIdx.x * (TWT * 2) + x];
        }
        s_rl[x] = base;
    }
    // ---- 1. the tile's lines: QNAME -> slot (count); read-list entry -> rank among the row's entries of its list (wave ballots), the row's
    //         entries per list into s_mat
    constexpr int K = TL / TILE_TB;
    uint32_t l_cls[K], l_q[K], l_v[K], l_slot[K], l_rank[K], l_base[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int64_t i = i0 + tid + TILE_TB * k;
        l_cls[k] = 255u; l_q[k] = 0u; l_v[k] = 0u;
        if (i < L.n) {
            const int64_t g = L.line_base + i;
            l_cls[k] = O.line_cls[g]; l_q[k] = O.line_q[g]; l_v[k] = (uint32_t)(L.var_idx[i] + L.var_base);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        l_slot[k] = Q_EMPTY;
        if (l_cls[k] != 255u) {
            const uint32_t q = l_q[k];
            uint32_t s = (q * 2654435761u) >> TH_SHIFT;
            for (;;) {
                const uint32_t prev = atomicCAS(&s_q[s], Q_EMPTY, q);
                if (prev == Q_EMPTY || prev == q) break;
                s = (s + 1) & (TH - 1);
            }
            atomicAdd(&s_c[s], 1u);
            l_slot[k] = s;
        }
        // every lane gets here: the lanes of a row with the same list key find each other by nine ballots
        const bool isrl = l_cls[k] < 2u;
        const unsigned d = l_v[k] - (unsigned)vbase;
        const uint32_t key = (d * 2u + (l_cls[k] & 1u)) & (uint32_t)(TWT * 2 - 1);
        l_base[k] = (isrl && d < (unsigned)TWT) ? s_rl[key] : RL_DIRTY;      // a far line's list is dirty by construction (k_line flagged it)
        const bool ord = isrl && l_base[k] != RL_DIRTY;
        unsigned long long same = __ballot(ord);
#pragma unroll
        for (int b = 0; b < 9; b++) {
            const bool bit = (key >> b) & 1u;
            const unsigned long long bal = __ballot(ord && bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t inrow = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        l_rank[k] = isrl ? inrow : Q_EMPTY;
        if (ord && inrow == 0u) s_mat[((tid >> 6) + (TILE_TB / 64) * k) * (TWT * 2) + key] = (uint8_t)__popcll(same);
    }
    __syncthreads();
    // ---- 2. read-list entries to their final places (the entries of earlier rows come from s_mat, which step 3 overwrites); meanwhile the
    //         group ranges (exclusive scan of the slot counts)
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (l_rank[k] == Q_EMPTY) continue;
        const int64_t i = i0 + tid + TILE_TB * k;
        const uint32_t g = (uint32_t)(L.line_base + i);
        const uint32_t e = (l_v[k] * 2u + l_cls[k]) * (uint32_t)O.nb + (uint32_t)L.bam;
        if (l_base[k] != RL_DIRTY) {
            const unsigned d = l_v[k] - (unsigned)vbase;
            const uint32_t key = d * 2u + l_cls[k];
            const int row = (tid >> 6) + (TILE_TB / 64) * k;
            uint32_t before = 0;
            for (int r = 0; r < row; r++) before += s_mat[r * (TWT * 2) + key];
            const uint32_t at = l_base[k] + before + l_rank[k];
            O.rl_qid[at] = (int32_t)(l_q[k] - L.qid_base); O.rl_list[at] = e;
        } else {
            const uint32_t at = O.rl_start[e] + atomicAdd(&O.rl_cursor[e], 1u);
            O.rl_tmp[at] = ((uint64_t)g << 32) | (uint32_t)(l_q[k] - L.qid_base); O.rl_list[at] = e;
        }
    }
    {
        // one scan for both: lines (low half) and occupied slots (high half) before every slot
        uint32_t c[TSP], sum = 0;
#pragma unroll
        for (int j = 0; j < TSP; j++) { c[j] = s_c[tid * TSP + j]; sum += c[j] + (c[j] ? 0x10000u : 0u); }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if ((tid & 63) >= d) incl += y; }
        if ((tid & 63) == 63) s_part[tid >> 6] = incl;
        __syncthreads();
        uint32_t base = incl - sum;
        for (int w = 0; w < (tid >> 6); w++) base += s_part[w];
        if (tid == TILE_TB - 1) { s_nkept = (base + sum) & 0xFFFFu; s_ngroups = (base + sum) >> 16; }
#pragma unroll
        for (int j = 0; j < TSP; j++) {
            s_c[tid * TSP + j] = base & 0xFFFFu;
            if (c[j]) s_grp[base >> 16] = (uint16_t)(tid * TSP + j);
            base += c[j] + (c[j] ? 0x10000u : 0u);
        }
    }
    __syncthreads();
    // ---- 3. lines into their groups; whether some other tile holds lines of the QNAMEs of this lane's groups is requested first (used after
    //         the next barrier)
    constexpr int GK = TL / TILE_TB;                            // groups per lane (a tile of TL lines holds at most TL groups)
    const uint32_t ngroups = s_ngroups;
    uint32_t dupw[GK], my_q[GK], my_slot[GK];
#pragma unroll
    for (int j = 0; j < GK; j++) {
        const uint32_t gi = (uint32_t)tid + (uint32_t)TILE_TB * (uint32_t)j;
        my_slot[j] = gi < ngroups ? (uint32_t)s_grp[gi] : Q_EMPTY;
        my_q[j] = my_slot[j] != Q_EMPTY ? s_q[my_slot[j]] : Q_EMPTY;
        dupw[j] = my_q[j] != Q_EMPTY ? O.q_dup[my_q[j] >> 5] : 0u;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (l_slot[k] == Q_EMPTY) continue;
        const int64_t i = i0 + tid + TILE_TB * k;
        const uint32_t g = (uint32_t)(L.line_base + i);
        s_it[atomicAdd(&s_c[l_slot[k]], 1u)] = item_pack(l_v[k], l_cls[k], g);
    }
    __syncthreads();
    // ---- 4. one thread per group.  A group holding every line of its QNAME is finished here (all its lines come from this shard's BAM: that
    //         BAM owns the read_vars list, every ref/alt line is linked; "all" = no other tile has set the QNAME's bit): its distinct items stay at the front of its range, the rest of the
    //         range becomes KEY_DROPPED.  The others hand their lines to the spill list (ranges inside the tile's share from LDS counters)
    uint32_t sp_beg[GK], sp_n[GK], sp_at[GK];
#pragma unroll
    for (int j = 0; j < GK; j++) {
        sp_n[j] = 0; sp_beg[j] = 0; sp_at[j] = 0;
        if (my_q[j] == Q_EMPTY) continue;
        const int slot = (int)my_slot[j];
        const uint32_t q = my_q[j];
        const uint32_t beg = slot ? s_c[slot - 1] : 0u, n = s_c[slot] - beg;        // the cursors stopped at the groups' ends
        uint64_t *it = s_it + beg;
        if ((dupw[j] >> (q & 31u)) & 1u) {
            if (atomicAdd(&O.qcount[q], n) == 0u) s_touch[atomicAdd(&s_ntouch, 1u)] = (uint16_t)slot;   // the first tile to add its lines lists the QNAME
            sp_n[j] = n; sp_beg[j] = beg; sp_at[j] = atomicAdd(&s_nspill, n);
            continue;
        }
        if (n == 1u) {                                         // most QNAMEs: one line, nothing to sort, no pair, no rank
            const uint64_t x = it[0];
            const uint32_t v = (uint32_t)(x >> 34), cls = (uint32_t)(x >> 32) & 3u;
            const unsigned d = v - (unsigned)vbase;
            if (d < (unsigned)TWT) atomicAdd(&s_cnt[d * 3 + cls], 1); else atomicAdd(&O.var_distinct[(int64_t)v * 3 + cls], 1);
            it[0] = dist_pack(q, v, cls, cls < 2u ? 1u : 0u);
            continue;
        }
        for (uint32_t a = 1; a < n; a++) {
            const uint64_t x = it[a];
            uint32_t b = a;
            while (b > 0 && it[b - 1] > x) { it[b] = it[b - 1]; b--; }
            it[b] = x;
        }
        uint32_t first = 0xFFFFFFFFu, vmin = 0xFFFFFFFFu, vmax = 0;
        for (uint32_t a = 0; a < n; a++) {
            const uint64_t x = it[a];
            if (((x >> 32) & 3ull) >= 2ull) continue;
            const uint32_t g = (uint32_t)x, v = (uint32_t)(x >> 34);
            first = g < first ? g : first;
            vmin = v < vmin ? v : vmin; vmax = v > vmax ? v : vmax;
        }
        const bool multi = vmin != 0xFFFFFFFFu && vmin != vmax;
        uint32_t w = 0, a = 0;
        while (a < n) {
            const uint64_t key = it[a] >> 32;                  // variant:28 | class:2
            const uint32_t v = (uint32_t)(key >> 2), cls = (uint32_t)(key & 3ull);
            const uint32_t gl = (uint32_t)it[a];               // the run is sorted by line: its first line
            while (a < n && (it[a] >> 32) == key) a++;
            const uint32_t linked = cls < 2u ? 1u : 0u;
            const unsigned d = v - (unsigned)vbase;
            if (multi && linked) {
                const unsigned long long rk = ((unsigned long long)first << 32) | gl;
                if (d < (unsigned)TWT) atomicMin(&s_rank[d], rk); else atomicMin(&O.var_rank[v], rk);
            }
            if (d < (unsigned)TWT) atomicAdd(&s_cnt[d * 3 + cls], 1); else atomicAdd(&O.var_distinct[(int64_t)v * 3 + cls], 1);
            it[w++] = dist_pack(q, v, cls, linked);
        }
        for (uint32_t k = w; k < n; k++) it[k] = KEY_DROPPED;
    }
    __syncthreads();
    // ---- 5. ONE global cursor step per tile (spilled lines in the low word, listed QNAMEs in the high word) -- while it is under way the
    //         tile's slots of the item array are written in one sweep (groups in place, holes = KEY_DROPPED) -- then the spilled lines go
    //         out and their slots of the item array are crossed out
    unsigned long long cur = 0;
    if (tid == 0) {
        const unsigned long long add = ((unsigned long long)s_ntouch << 32) | s_nspill;
        if (add) cur = atomicAdd(&O.counters[10], add);
    }
    uint64_t *dst = O.items + ((int64_t)blockIdx.x * TL);
    {
        const uint32_t nk = s_nkept;
        for (int j = tid; j < TL; j += TILE_TB) dst[j] = (uint32_t)j < nk ? s_it[j] : KEY_DROPPED;
    }
    for (int j = tid; j < TWT * 3; j += TILE_TB) {
        const int cc = s_cnt[j];
        if (cc) atomicAdd(&O.var_distinct[(int64_t)(vbase + j / 3) * 3 + (j % 3)], cc);
    }
    for (int j = tid; j < TWT; j += TILE_TB) {
        const unsigned long long f = s_rank[j];
        if (f != ~0ull) atomicMin(&O.var_rank[vbase + j], f);
    }
    if (tid == 0) s_obase = cur;
    __syncthreads();
    if (O.prof && tid == 0) O.prof[2 * (size_t)blockIdx.x + 1] = wall_clock64();
    if (s_nspill == 0) return;
    const uint32_t spill_base = (uint32_t)s_obase, touch_base = (uint32_t)(s_obase >> 32);
#pragma unroll
    for (int j = 0; j < GK; j++) {
        if (!sp_n[j]) continue;
        for (uint32_t a = 0; a < sp_n[j]; a++) {
            O.sp_q[spill_base + sp_at[j] + a] = my_q[j]; O.sp_item[spill_base + sp_at[j] + a] = s_it[sp_beg[j] + a];
            dst[sp_beg[j] + a] = KEY_DROPPED;
        }
    }
    for (uint32_t j = tid; j < s_ntouch; j += TILE_TB) O.touched[touch_base + j] = s_q[s_touch[j]];
}

// the spilled lines into the groups of their QNAMEs (ranges from the scan over the spilled QNAMEs' line counts)
__global__ __launch_bounds__(256) void k_items_spill(int64_t n, const uint32_t *sp_q, const uint64_t *sp_item, uint32_t *qcount, uint64_t *items) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) items[atomicAdd(&qcount[sp_q[i]], 1u)] = sp_item[i];
}

// One thread per QNAME group.  Sorts the group (insertion sort: a QNAME owns a handful of lines), then
//   owner  = last BAM holding a ref/alt line of the QNAME (the stale-variable overwrite at phaser.py:578: a later BAM replaces the read_vars list)
//   first  = first ref/alt line of the QNAME over all BAMs
//   linked = ref/alt line of the owner BAM (an entry of the surviving read_vars list)
//   rank   : overlap-dictionary key order (SURVEY.md 8.1 rule 4, phaser.py:1271-1283): for QNAMEs whose surviving list holds >= 2 distinct
//            variants, every variant gets (first << 32 | its first linked line) as a candidate for its smallest key
//   the distinct (variant, class) items (linked = max over the run of equal lines), written back at the front of the group as
//            QNAME id:32 | variant:28 << 4 | class << 2 | linked (the rest of the group becomes KEY_DROPPED), and the per-variant distinct-QNAME counters
struct GroupOut {
    uint32_t *qcount; uint64_t *items; uint32_t *cnt_t; const uint32_t *base_t, *touched;
    unsigned long long *var_rank; int32_t *var_distinct; unsigned long long *counters;      // [0] distinct items
    int single_bam;
};
__global__ __launch_bounds__(256) void k_groups(int64_t nt, LinesTab T, GroupOut O) {
    __shared__ int s_cnt[TW * 3];
    __shared__ unsigned long long s_rank[TW];
    __shared__ int s_vbase;
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    for (int j = tid; j < TW * 3; j += 256) s_cnt[j] = 0;
    for (int j = tid; j < TW; j += 256) s_rank[j] = ~0ull;
    if (tid == 0) { const int64_t i0 = (int64_t)blockIdx.x * 256; s_vbase = (int)(O.items[O.base_t[i0]] >> 34); }
    __syncthreads();
    const int vbase = s_vbase - 256 > 0 ? s_vbase - 256 : 0;
    if (i < nt) {
        const uint32_t b = O.base_t[i], c = O.cnt_t[i];
        O.qcount[O.touched[i]] = 0u;                         // the fill cursor goes back to zero: clean for the next call
        uint64_t *it = O.items + b;
        for (uint32_t a = 1; a < c; a++) {
            const uint64_t x = it[a];
            uint32_t j = a;
            while (j > 0 && it[j - 1] > x) { it[j] = it[j - 1]; j--; }
            it[j] = x;
        }
        int owner = -1; uint32_t first = 0xFFFFFFFFu;
        for (uint32_t a = 0; a < c; a++) {
            const uint64_t x = it[a];
            if (((x >> 32) & 3ull) >= 2ull) continue;
            const uint32_t g = (uint32_t)x;
            first = g < first ? g : first;
            if (!O.single_bam) { const int bm = bam_of_line(T, g); owner = bm > owner ? bm : owner; }
        }
        uint32_t vmin = 0xFFFFFFFFu, vmax = 0;
        for (uint32_t a = 0; a < c; a++) {
            const uint64_t x = it[a];
            if (((x >> 32) & 3ull) >= 2ull) continue;
            if (!O.single_bam && bam_of_line(T, (uint32_t)x) != owner) continue;
            const uint32_t v = (uint32_t)(x >> 34);
            vmin = v < vmin ? v : vmin; vmax = v > vmax ? v : vmax;
        }
        const bool multi = vmin != 0xFFFFFFFFu && vmin != vmax;
        // runs of equal (variant, class): one distinct item each; the variant's first linked line competes for its rank
        uint32_t w = 0, a = 0;
        while (a < c) {
            const uint64_t key = it[a] >> 32;                  // variant:28 | class:2
            const uint32_t v = (uint32_t)(key >> 2), cls = (uint32_t)(key & 3ull);
            uint32_t linked = 0, gl = 0xFFFFFFFFu;
            while (a < c && (it[a] >> 32) == key) {
                const uint32_t g = (uint32_t)it[a];
                if (cls < 2u && (O.single_bam || bam_of_line(T, g) == owner)) { linked = 1u; gl = g < gl ? g : gl; }
                a++;
            }
            if (multi && linked) {
                const unsigned long long rk = ((unsigned long long)first << 32) | gl;
                const unsigned d = (unsigned)((int)v - vbase);
                if (d < (unsigned)TW) atomicMin(&s_rank[d], rk); else atomicMin(&O.var_rank[v], rk);
            }
            {
                const unsigned d = (unsigned)((int)v - vbase);
                if (d < (unsigned)TW) atomicAdd(&s_cnt[d * 3 + cls], 1); else atomicAdd(&O.var_distinct[(int64_t)v * 3 + cls], 1);
            }
            it[w++] = dist_pack(O.touched[i], v, cls, linked);
        }
        for (uint32_t k = w; k < c; k++) it[k] = KEY_DROPPED;
        O.cnt_t[i] = w;
    }
    __syncthreads();
    for (int j = tid; j < TW * 3; j += 256) {
        const int cc = s_cnt[j];
        if (cc) atomicAdd(&O.var_distinct[(int64_t)(vbase + j / 3) * 3 + (j % 3)], cc);
    }
    for (int j = tid; j < TW; j += 256) {
        const unsigned long long f = s_rank[j];
        if (f != ~0ull) atomicMin(&O.var_rank[vbase + j], f);
    }
}

__device__ __forceinline__ uint32_t hash64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (uint32_t)k;
}

constexpr int PH_SLOTS = 1024;     // LDS hash slots per workgroup
constexpr int PH_WORDS = 5;        // LDS counters of a slot: the nine cells as 16-bit halves (a tile holds < 2^16 QNAMEs), linked flag = bit 16 of word 4
constexpr int PH_PROBES = 24;
constexpr int GH_PROBES = 256;     // global probe bound; beyond it the table is declared too small and the pass is redone with a larger one (once
                                   // that is decided nobody probes any more: a full table made every insertion a walk of the whole bound)
// the global table: one 64-byte entry per slot = one memory sector: key (words 0-1, 0 = empty: a pair key (a << 32 | b) has b > a >= 0),
// the nine cells (words 2-10), the linked flag (word 11)
constexpr int GE_WORDS = 16;
constexpr int GE_CELL0 = 2, GE_LINKED = 11;

// counters[]: 2 global hash overflow, 3 kept lines, 4/5 noise, 7 used hash slots, 10 spilled lines | spilled QNAMEs << 32
// -> slot of the pair in the global table, claiming it if new (*claimed); -1 when the probe bound is hit
// Home slot of a pair: a scattering hash.  (A table kept in variant order -- the pairs of variant a from slot 4a on, so that a tile's pairs
// share a few dozen KB -- was measured: k_pairs 0.38 -> 1.65 ms.  The workgroups in flight work on neighbouring tiles, their atomics
// then queue on the few L2 channels that own that stretch of the table.)
__device__ __forceinline__ uint32_t pair_home(uint64_t key, uint32_t gmask) { return hash64(key) & gmask; }
__device__ __forceinline__ int global_slot(uint32_t *tab, uint32_t gmask, uint64_t key, unsigned long long *counters, bool *claimed) {
    uint32_t s = pair_home(key, gmask);
    *claimed = false;
    for (int t = 0; t < GH_PROBES; t++) {
        // a long probe sequence is the sign of a table that is filling up: look whether the pass is lost already (not on every call: the
        // flag shares a memory sector with the used-slot cursor, and a load per insertion tripled the kernel's time)
        if ((t & 15) == 15 && __hip_atomic_load(&counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return -1;
        const unsigned long long prev = atomicCAS((unsigned long long *)(tab + (size_t)s * GE_WORDS), 0ull, (unsigned long long)key);
        if (prev == 0ull) { *claimed = true; return (int)s; }
        if (prev == key) return (int)s;
        s = (s + 1) & gmask;
    }
    atomicAdd(&counters[2], 1ull);
    return -1;
}

// QNAME groups hold their distinct items sorted by (variant, class): every pair of items on different variants adds 1 to cell
// (class_a, class_b) of that variant pair.  One thread per item slot (the pairs of an item with the later items of its group): PAIR_ITEMS
// slots per workgroup.  The slots this workgroup claims in the global table go to the list of used slots.
constexpr int PAIR_ITEMS = 2048;      // item slots per round
#ifndef PHZ_PAIRS_TB
#define PHZ_PAIRS_TB 512
#endif
constexpr int PAIRS_TB = PHZ_PAIRS_TB;   // threads per workgroup of k_pairs
constexpr int PAIR_ROUNDS = 1;        // rounds per workgroup sharing one LDS table and one flush (2: the table overflows, 0.35 -> 0.43 ms)
template <int MODE> __global__ __launch_bounds__(PAIRS_TB) void k_pairs(const uint64_t *items, int64_t m, uint32_t *tab, uint32_t gmask, uint32_t *used, uint64_t *used_key, uint32_t *deg,
                                                                       unsigned long long *counters) {
    __shared__ unsigned long long s_keys[PH_SLOTS];
    __shared__ uint32_t s_vals[PH_SLOTS * PH_WORDS];
    __shared__ uint32_t s_claim[PH_SLOTS];
    __shared__ uint16_t s_claim_l[PH_SLOTS];     // LDS slot of the claim (its key is still there when the list goes out)
    __shared__ unsigned int s_part[PAIRS_TB / 64], s_ipart[PAIRS_TB / 64], s_nclaim;
    __shared__ unsigned long long s_ubase;
    for (int j = threadIdx.x; j < PH_SLOTS; j += PAIRS_TB) s_keys[j] = KEY_DROPPED;
    for (int j = threadIdx.x; j < PH_SLOTS * PH_WORDS; j += PAIRS_TB) s_vals[j] = 0;
    if (threadIdx.x == 0) s_nclaim = 0;
    __syncthreads();
    unsigned int n_event = 0, n_item = 0;
    constexpr int KI = PAIR_ITEMS / PAIRS_TB;
    // A wave holds 64 consecutive items per round (one per lane) plus the 64 after them: the later items of a lane's group are read from the
    // neighbouring lanes (the groups are a handful of items long), no memory access inside the pair loop
    const int lane = threadIdx.x & 63;
    for (int round = 0; round < PAIR_ROUNDS; round++) {
    const int64_t i0 = ((int64_t)blockIdx.x * PAIR_ROUNDS + round) * PAIR_ITEMS;
    if (i0 >= m) break;
    unsigned long long my_item[KI], nx_item[KI];
#pragma unroll
    for (int t = 0; t < KI; t++) {                           // all of the lane's items requested together
        const int64_t i = i0 + threadIdx.x + PAIRS_TB * t;
        my_item[t] = i < m ? items[i] : KEY_DROPPED;
        nx_item[t] = i + 64 < m ? items[i + 64] : KEY_DROPPED;
    }
#pragma unroll
    for (int t = 0; t < KI; t++) {
        const int64_t i = i0 + threadIdx.x + PAIRS_TB * t;
        const unsigned long long k = my_item[t];
        bool active = k != KEY_DROPPED;
        if (active) n_item++;
        if (MODE == 2) continue;
        const uint32_t q = (uint32_t)(k >> 32), v = (uint32_t)(k >> 4) & 0x0FFFFFFFu, cls = (uint32_t)(k >> 2) & 3u, ln = (uint32_t)k & 1u;
        for (int d = 1; __any(active); d++) {
            unsigned long long k2;
            if (d < 64) {
                const int src = lane + d;
                const unsigned long long a_ = __shfl(k, src & 63), b_ = __shfl(nx_item[t], src & 63);
                k2 = src < 64 ? a_ : b_;
            } else k2 = (active && i + d < m) ? items[i + d] : KEY_DROPPED;      // a group of more than 64 distinct items
            bool ev = active;                                  // this lane pairs its item with k2 in this round
            if (ev && (k2 == KEY_DROPPED || (uint32_t)(k2 >> 32) != q)) { active = false; ev = false; }        // the distinct items of a group sit at its front
            const uint32_t v2 = (uint32_t)(k2 >> 4) & 0x0FFFFFFFu;
            if (v2 == v) ev = false;
            bool claimed = false;
            int gs = -1;
            uint64_t ck = 0;
            if (ev) {
                n_event++;
                const uint32_t cls2 = (uint32_t)(k2 >> 2) & 3u, ln2 = (uint32_t)k2 & 1u;
                const uint64_t pk = ((uint64_t)v << 32) | v2;          // v < v2 because the group is sorted
                const int cell = (int)(cls * 3 + cls2);
                const int linked = (int)(ln & ln2);
                uint32_t s = hash64(pk) & (PH_SLOTS - 1);
                bool done = false;
                for (int pr = 0; pr < PH_PROBES; pr++) {
                    const unsigned long long prev = atomicCAS(&s_keys[s], (unsigned long long)KEY_DROPPED, (unsigned long long)pk);
                    if (prev == KEY_DROPPED || prev == pk) {
                        atomicAdd(&s_vals[s * PH_WORDS + (cell >> 1)], 1u << ((cell & 1) * 16));
                        if (linked) atomicOr(&s_vals[s * PH_WORDS + 4], 0x10000u);
                        done = true;
                        break;
                    }
                    s = (s + 1) & (PH_SLOTS - 1);
                }
                // the LDS table is full (a tile whose QNAMEs pair up variants all over the chromosome: QNAME ids shared by unrelated reads of
                // several BAMs): straight to the global table
                if (!done) {
                    ck = pk;
                    gs = global_slot(tab, gmask, pk, counters, &claimed);
                    if (gs >= 0) {
                        atomicAdd(&tab[(size_t)gs * GE_WORDS + GE_CELL0 + cell], 1u);
                        if (linked) atomicOr(&tab[(size_t)gs * GE_WORDS + GE_LINKED], 1u);
                    }
                }
            }
            // the lanes that claimed a slot there take their places in the used-slot list with ONE cursor step per wave and round (one
            // same-address atomic per claim was 48 ms for such a sample); every lane of the wave gets here
            const unsigned long long cm = __ballot(claimed ? 1 : 0);
            if (cm) {
                const int leader = __builtin_ctzll(cm);
                unsigned long long base = 0;
                if (lane == leader) base = atomicAdd(&counters[7], (unsigned long long)__popcll(cm));
                base = __shfl(base, leader);
                if (claimed) {
                    const unsigned long long at = base + (unsigned long long)__popcll(cm & ((1ull << lane) - 1ull));
                    used[at] = (uint32_t)gs; used_key[at] = ck;
                    atomicAdd(&deg[(uint32_t)(ck >> 32)], 1u);
                }
            }
        }
    }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { n_event += __shfl_xor(n_event, d); n_item += __shfl_xor(n_item, d); }
    if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6] = n_event; s_ipart[threadIdx.x >> 6] = n_item; }
    __syncthreads();
    if (MODE != 0) { if (n_item == 12345678u) used[0] = s_vals[threadIdx.x]; return; }
    {
        constexpr int KS = PH_SLOTS / PAIRS_TB;
        uint64_t pk[KS]; uint32_t hs[KS]; unsigned long long prev[KS];
#pragma unroll
        for (int t = 0; t < KS; t++) {                       // the first probe of each of the lane's slots requested together
            pk[t] = s_keys[threadIdx.x + PAIRS_TB * t];
            hs[t] = pair_home(pk[t], gmask); prev[t] = 0ull;
            if (pk[t] != KEY_DROPPED) prev[t] = atomicCAS((unsigned long long *)(tab + (size_t)hs[t] * GE_WORDS), 0ull, (unsigned long long)pk[t]);
        }
#pragma unroll
        for (int t = 0; t < KS; t++) {
            if (pk[t] == KEY_DROPPED) continue;
            const int j = threadIdx.x + PAIRS_TB * t;
            bool claimed = prev[t] == 0ull;
            int gs = (int)hs[t];
            if (!claimed && prev[t] != pk[t]) gs = global_slot(tab, gmask, pk[t], counters, &claimed);      // taken by another pair: probe on
            if (gs < 0) continue;
            if (claimed) { const uint32_t at = atomicAdd(&s_nclaim, 1u); s_claim[at] = (uint32_t)gs; s_claim_l[at] = (uint16_t)j; }
            uint32_t *e = tab + (size_t)gs * GE_WORDS;
#pragma unroll
            for (int c = 0; c < 9; c++) {
                const uint32_t val = (s_vals[j * PH_WORDS + (c >> 1)] >> ((c & 1) * 16)) & 0xFFFFu;
                if (val) atomicAdd(&e[GE_CELL0 + c], val);
            }
            if (s_vals[j * PH_WORDS + 4] & 0x10000u) atomicOr(&e[GE_LINKED], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = 0, ni = 0;
        for (int w = 0; w < PAIRS_TB / 64; w++) { b += s_part[w]; ni += s_ipart[w]; }
        unsigned long long *spread = counters + 16 + (blockIdx.x % N_SPREAD) * SPREAD_WORDS;
        if (b) atomicAdd(&spread[1], b);
        if (ni) atomicAdd(&spread[0], ni);                    // distinct (QNAME, variant, class) items
        s_ubase = s_nclaim ? atomicAdd(&counters[7], (unsigned long long)s_nclaim) : 0ull;      // one global atomic per workgroup
    }
    __syncthreads();
    // the claimed slots go to the used-slot list together with their keys, and every claim counts as one edge of its first variant: the
    // counting sort of the edges by variant starts from finished degrees and never has to look a key up in the table
    for (unsigned j = threadIdx.x; j < s_nclaim; j += PAIRS_TB) {
        const unsigned long long key = s_keys[s_claim_l[j]];
        used[s_ubase + j] = s_claim[j]; used_key[s_ubase + j] = key;
        atomicAdd(&deg[(uint32_t)(key >> 32)], 1u);
    }
}

// ---- edge list in (a, b) order: counting sort of the used table slots by a (the degrees were counted by k_pairs as it claimed the slots);
//      the position of an edge inside its variant's (small) group is the number of group members with a smaller b
constexpr uint32_t EDGE_RANK_MAX = 128;     // groups up to this size are ranked by counting (<= 128 cached loads per edge); larger ones are sorted first
__global__ __launch_bounds__(256) void k_edge_scatter(const uint32_t *used, const uint64_t *used_key, int64_t n_used, const uint32_t *eoff, uint32_t *deg,
                                                      uint32_t *e_a, uint32_t *e_b, uint32_t *e_slot) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_used) return;
    const uint64_t k = used_key[i];
    const uint32_t a = (uint32_t)(k >> 32);
    const uint32_t old = atomicSub(&deg[a], 1u);               // the counters go back to zero
    const uint32_t p = eoff[a] + old - 1;
    e_a[p] = a; e_b[p] = (uint32_t)k; e_slot[p] = used[i];
}
// one thread per variant with more than EDGE_RANK_MAX edges (QNAME ids shared by unrelated reads pair a variant with thousands of others):
// heap sort of its group in place, the keys are distinct
__global__ __launch_bounds__(256) void k_edge_sort_big(int64_t nv, const uint32_t *eoff, uint32_t *e_b, uint32_t *e_slot) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= nv) return;
    const uint32_t lo = eoff[a], hi = eoff[a + 1], d = hi - lo;
    if (d <= EDGE_RANK_MAX) return;
    uint32_t *kb = e_b + lo, *ks = e_slot + lo;
    auto sift = [&](uint32_t root, uint32_t n) {
        const uint32_t xb = kb[root], xs = ks[root];
        for (;;) {
            uint32_t c = 2 * root + 1;
            if (c >= n) break;
            if (c + 1 < n && kb[c + 1] > kb[c]) c++;
            if (kb[c] <= xb) break;
            kb[root] = kb[c]; ks[root] = ks[c]; root = c;
        }
        kb[root] = xb; ks[root] = xs;
    };
    for (uint32_t i = d / 2; i-- > 0;) sift(i, d);
    for (uint32_t n = d - 1; n > 0; n--) {
        const uint32_t tb = kb[0], ts = ks[0];
        kb[0] = kb[n]; ks[0] = ks[n]; kb[n] = tb; ks[n] = ts;
        sift(0, n);
    }
}
// one thread per edge: its place in the (a, b)-ordered list, its table entry (one sector) read, the result columns written, the entry returned
// to "empty" -- the table is left clean slot by slot, so the next call needs no memset of it
__global__ __launch_bounds__(256) void k_edge_out(int64_t ne, const uint32_t *eoff, const uint32_t *e_a, const uint32_t *e_b, const uint32_t *e_slot, uint32_t *tab,
                                                  int32_t *ea, int32_t *eb, int32_t *cells, uint8_t *linked, int32_t *cto, int32_t *stats) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= ne) return;
    uint4 *e = (uint4 *)(tab + (size_t)e_slot[p] * GE_WORDS);
    const uint4 w0 = e[0], w1 = e[1], w2 = e[2];                      // requested before the group is looked at
    const uint32_t a = e_a[p], b = e_b[p];
    const uint32_t lo = eoff[a], hi = eoff[a + 1];
    uint32_t rank = (uint32_t)p - lo;                                 // a sorted group
    if (hi - lo <= EDGE_RANK_MAX) {
        rank = 0;
        for (uint32_t j = lo; j < hi; j++) rank += e_b[j] < b ? 1u : 0u;
    }
    const int64_t i = (int64_t)lo + rank;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    e[0] = z; e[1] = z; e[2] = z;
    ea[i] = (int32_t)a; eb[i] = (int32_t)b;
    const int32_t x[9] = {(int32_t)w0.z, (int32_t)w0.w, (int32_t)w1.x, (int32_t)w1.y, (int32_t)w1.z, (int32_t)w1.w, (int32_t)w2.x, (int32_t)w2.y, (int32_t)w2.z};
#pragma unroll
    for (int c = 0; c < 9; c++) cells[i * 9 + c] = x[c];
    linked[i] = (uint8_t)(w2.w & 1u);
    // test_variant_connection's three sums (phaser.py:1634-1636): same configuration rr+aa, opposite ar+ra, the five "other" cells
    const int32_t cis = x[0] + x[4], trans = x[3] + x[1], oth = x[6] + x[7] + x[2] + x[5] + x[8];
    cto[i * 3] = cis; cto[i * 3 + 1] = trans; cto[i * 3 + 2] = oth;
    // the derived columns of the pair test (:1637-1649), one plane each: same-configuration / opposite counts, supporting =
    // the larger of the two, total, chosen configuration (0 same, 1 opposite, -1 tie)
    stats[i] = cis; stats[ne + i] = trans; stats[2 * ne + i] = cis > trans ? cis : trans; stats[3 * ne + i] = cis + trans + oth;
    stats[4 * ne + i] = cis > trans ? 0 : (cis < trans ? 1 : -1);
}

// ---- read lists with a far line (dirty lists): their entries were placed through a cursor, in arrival order; one workgroup per list puts them
//      into line order (bitonic sort in LDS) and keeps the QNAME ids; a list beyond the LDS stage goes to the host-driven radix sort.
//      counters32[1] = lists left to that sort.  The list's cursor returns to zero (it is persistent, like the per-QNAME one).
constexpr int RL_LDS = 4096;
__global__ __launch_bounds__(256) void k_rl_sort_dirty(const uint32_t *dirty_list, const uint32_t *rl_start, const uint64_t *rl_tmp, int32_t *rl_qid, uint32_t *rl_cursor,
                                                       uint32_t *big_list, uint32_t *counters32) {
    __shared__ unsigned long long s_x[RL_LDS];
    const uint32_t e = dirty_list[blockIdx.x];
    const uint32_t lo = rl_start[e], n = rl_start[e + 1] - lo;
    if (threadIdx.x == 0) rl_cursor[e] = 0u;
    if (n > (uint32_t)RL_LDS) { if (threadIdx.x == 0) big_list[atomicAdd(&counters32[1], 1u)] = e; return; }
    uint32_t m = 64;
    while (m < n) m <<= 1;
    for (uint32_t t = threadIdx.x; t < m; t += 256) s_x[t] = t < n ? rl_tmp[lo + t] : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < m; t += 256) {
                const uint32_t p = t ^ j;
                if (p > t) {
                    const unsigned long long a = s_x[t], b = s_x[p];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) { s_x[t] = b; s_x[p] = a; }
                }
            }
            __syncthreads();
        }
    for (uint32_t t = threadIdx.x; t < n; t += 256) rl_qid[lo + t] = (int32_t)(uint32_t)s_x[t];
}
__global__ __launch_bounds__(256) void k_rl_take_qid(const uint64_t *src, int64_t n, int32_t *dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (int32_t)(uint32_t)src[i];
}

// sequencing-noise counters (phaser.py:610-632): over the variants with ref+alt lines and an "other" share below 5 %, the ref+alt
// lines (match) and the other-allele lines (mismatch); the caller all-reduces them over chromosomes / ranks
__global__ __launch_bounds__(256) void k_noise(const int32_t *var_count, int64_t nv, unsigned long long *out /* [0] match, [1] mismatch */) {
    __shared__ unsigned long long s_m[4], s_x[4];
    unsigned long long m = 0, x = 0;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256) {       // few workgroups: two same-address atomics each
        const long long mm = (long long)var_count[v * 3] + var_count[v * 3 + 1], oo = var_count[v * 3 + 2];
        if (mm > 0 && (double)oo / (double)(oo + mm) < 0.05) { m += (unsigned long long)mm; x += (unsigned long long)oo; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { m += __shfl_xor(m, d); x += __shfl_xor(x, d); }
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_x[threadIdx.x >> 6] = x; }
    __syncthreads();
    if (threadIdx.x == 0) {
        m = s_m[0] + s_m[1] + s_m[2] + s_m[3]; x = s_x[0] + s_x[1] + s_x[2] + s_x[3];
        if (m) atomicAdd(&out[0], m);
        if (x) atomicAdd(&out[1], x);
    }
}

constexpr size_t CNT_BYTES = 128 + (size_t)N_SPREAD * SPREAD_WORDS * 8;
inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

// HIP-event timing of a stage on the ctx stream; stop() waits for the stage and reports HIP errors
struct Timer {
    phz_ctx *c; int slot; hipError_t err;
    Timer(phz_ctx *ctx, int s) : c(ctx), slot(s) { err = hipEventRecord(c->ev0, c->stream); }
    int stop() {
        hipError_t e = hipEventRecord(c->ev1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
        if (err != hipSuccess) e = err;
        if (e != hipSuccess) return phz_fail(c, PHZ_E_HIP, "stage timing / kernel execution", e);
        c->last_ms[slot] = ms; c->total_ms[slot] += ms; c->launches[slot]++;
        return PHZ_OK;
    }
};

int stage_lines(Staging &st, const phz_lines &h, int space, LinesDev *d) {
    d->n = h.n_calls; d->cutoff = h.as_cutoff; d->use_cutoff = h.use_cutoff; d->bam = h.bam_index; d->cut_dev = h.as_cutoff_dev;
    d->var_base = (int32_t)h.var_base; d->qid_base = (uint32_t)h.qid_base; d->line_base = 0;
    if (int s = st.in(h.read_idx, (size_t)h.n_calls, space, &d->read_idx)) return s;
    if (int s = st.in(h.var_idx, (size_t)h.n_calls, space, &d->var_idx)) return s;
    if (int s = st.in(h.code, (size_t)h.n_calls, space, &d->code)) return s;
    if (int s = st.in(h.read_qid, (size_t)h.n_reads, space, &d->read_qid)) return s;
    if (int s = st.in(h.read_as, (size_t)h.n_reads, space, &d->read_as)) return s;
    if (int s = st.in(h.read_has_as, (size_t)h.n_reads, space, &d->read_has_as)) return s;
    if (int s = st.in(h.read_as16, (size_t)h.n_reads, space, &d->read_as16)) return s;
    d->nv_chrom = 0;
    return PHZ_OK;
}


// scratch slots of ctx->scratch used by the tally (0 is the AS histogram, 16.. belong to components / K_map)
enum { T_QBASE = 1, T_TOUCHED, T_CNT_T, T_BASE_T, T_ITEMS, T_COUNTERS, T_GKEYS, T_USEDKEY, T_DEG, T_EOFF, T_EB, T_ESLOT, T_SCAN_TMP, T_USED, T_MISC, T_EA = 19 };
// results and the read-list buffers live in their own buffers (ctx->tally_buf)
enum { R_CNT = 0, R_FIRST, R_DIST, R_RANK, R_CLS, R_EA, R_EB, R_CELLS, R_LINKED, R_CTO, R_STATS, R_RLCNT, R_RLSTART, R_RLFILL, R_RLTMP, R_RLLIST, R_RLQID, R_A0, R_A1,
       R_LINEQ, R_SORTK, R_SORTV0, R_SORTV1, R_SORTCNT, R_SPQ, R_SPITEM, R_TILEWB, R_TILEM, R_TILEB, R_QSEEN, R_QDUP, R_RLDIRTY, R_COUNT };

}  // namespace

// k_as_hist is grid-stride with one LDS histogram per block, flushed with global atomics at the end: few, long-running blocks
static unsigned as_hist_blocks(const LinesDev &l) { const unsigned g = nblk((l.n + 15) / 16); return g > 512u ? 512u : (g ? g : 1u); }

// Device image of a shard table for one batched stage: [LinesDev x n][blk0 x (n+1)], blocks per shard from `blocks_of`.  Every
// table of a call gets its own slice of ctx->shard_tab (`slot`), so stages with different block sizes can be in flight together.
template <class F>
static int upload_tab(phz_ctx *ctx, const LinesDev *L, int n, F blocks_of, LinesTab *out, std::vector<uint32_t> *blk0_host, int slot = 0,
                      int n_slots = 1) {
    const size_t one = ((size_t)n * sizeof(LinesDev) + (size_t)(n + 1) * 4 + 63) & ~(size_t)63;
    if (ctx->tab_pending) { PHZ_HIP(ctx, hipEventSynchronize(ctx->tab_ev)); ctx->tab_pending = false; }      // the previous table has left the pinned image
    if (int s = phz_reserve_host(ctx, ctx->h_shard_tab, one * (size_t)n_slots)) return s;
    if (int s = phz_reserve(ctx, ctx->shard_tab, one * (size_t)n_slots)) return s;
    char *h = (char *)ctx->h_shard_tab.p + one * (size_t)slot;
    char *d = (char *)ctx->shard_tab.p + one * (size_t)slot;
    memcpy(h, L, (size_t)n * sizeof(LinesDev));
    uint32_t *b0 = (uint32_t *)(h + (size_t)n * sizeof(LinesDev));
    blk0_host->assign((size_t)n + 1, 0u);
    uint64_t acc = 0;
    for (int i = 0; i < n; i++) {
        b0[i] = (uint32_t)acc; (*blk0_host)[(size_t)i] = (uint32_t)acc;
        acc += L[i].n > 0 ? (uint64_t)blocks_of(L[i]) : 0u;
        if (acc >= (1ull << 31)) return phz_fail(ctx, PHZ_E_ARG, "too many blocks in one batched stage");
    }
    b0[n] = (uint32_t)acc; (*blk0_host)[(size_t)n] = (uint32_t)acc;
    PHZ_HIP(ctx, hipMemcpyAsync(d, h, one, hipMemcpyHostToDevice, ctx->stream));
    if (!ctx->tab_ev) PHZ_HIP(ctx, hipEventCreateWithFlags(&ctx->tab_ev, hipEventDisableTiming));
    PHZ_HIP(ctx, hipEventRecord(ctx->tab_ev, ctx->stream)); ctx->tab_pending = true;
    out->L = (const LinesDev *)d; out->blk0 = (const uint32_t *)(d + (size_t)n * sizeof(LinesDev)); out->n = n;
    return PHZ_OK;
}

extern "C" int phz_as_histogram(phz_ctx *ctx, const phz_lines *shard, int64_t *hist, int space) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !shard || !hist) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    Staging st(ctx);
    LinesDev L;
    if (int s = stage_lines(st, *shard, space, &L)) return s;
    unsigned long long *dh = nullptr;
    if (space == PHZ_DEVICE) dh = (unsigned long long *)hist;
    else {
        if (int s = phz_reserve(ctx, ctx->scratch[0], PHZ_AS_BINS * 8)) return s;
        dh = (unsigned long long *)ctx->scratch[0].p;
        PHZ_HIP(ctx, hipMemcpyAsync(dh, hist, PHZ_AS_BINS * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    Timer t(ctx, PHZ_T_ASHIST);
    if (L.n > 0) {
        LinesTab T;
        std::vector<uint32_t> grids;
        if (int s2 = upload_tab(ctx, &L, 1, as_hist_blocks, &T, &grids)) return s2;
        hipLaunchKernelGGL(k_as_hist, dim3(grids.back()), dim3(256), 0, ctx->stream, T, dh, (unsigned int *)nullptr);
    }
    PHZ_HIP(ctx, hipGetLastError());
    if (int s = t.stop()) return s;
    if (space == PHZ_HOST) PHZ_HIP(ctx, hipMemcpyAsync(hist, dh, PHZ_AS_BINS * 8, hipMemcpyDeviceToHost, ctx->stream));
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PHZ_OK;
}

// AS histograms of several device-resident shards accumulated into one device histogram, one host wait
extern "C" int phz_as_histogram_batch(phz_ctx *ctx, const phz_lines *shards, int n_shards, int64_t *hist) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || (!shards && n_shards) || !hist || n_shards < 0) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    Timer t(ctx, PHZ_T_ASHIST);
    Staging st(ctx);
    std::vector<LinesDev> L((size_t)n_shards);
    for (int i = 0; i < n_shards; i++)
        if (int s = stage_lines(st, shards[i], PHZ_DEVICE, &L[(size_t)i])) return s;
    if (n_shards > 0) {
        LinesTab T;
        std::vector<uint32_t> grids;
        if (int s2 = upload_tab(ctx, L.data(), n_shards, as_hist_blocks, &T, &grids)) return s2;
        if (int s2 = phz_reserve(ctx, ctx->scratch[0], 64)) return s2;
        PHZ_HIP(ctx, hipMemsetAsync(ctx->scratch[0].p, 0, 4, ctx->stream));
        if (grids.back() > 0) hipLaunchKernelGGL(k_as_hist, dim3(grids.back()), dim3(256), 0, ctx->stream, T, (unsigned long long *)hist, (unsigned int *)ctx->scratch[0].p);
    }
    PHZ_HIP(ctx, hipGetLastError());
    unsigned int oor = 0;
    if (n_shards > 0) PHZ_HIP(ctx, hipMemcpyAsync(&oor, ctx->scratch[0].p, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (int s = t.stop()) return s;
    if (oor) return phz_fail(ctx, PHZ_E_UNSUPPORTED, "AS value outside int16");
    return PHZ_OK;
}

// ... and for one rank that needs no all-reduce: the histogram never leaves the device, the host gets its occupied bins
extern "C" int phz_as_histogram_sparse(phz_ctx *ctx, const phz_lines *shards, int n_shards, int cap, int32_t *bins, int64_t *counts, int32_t *n_bins) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || (!shards && n_shards) || n_shards < 0 || cap < 1 || cap > 4096 || !bins || !counts || !n_bins) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    Timer t(ctx, PHZ_T_ASHIST);
    Staging st(ctx);
    std::vector<LinesDev> L((size_t)n_shards);
    for (int i = 0; i < n_shards; i++)
        if (int s = stage_lines(st, shards[i], PHZ_DEVICE, &L[(size_t)i])) return s;
    // scratch[0]: [hist 64 Ki x 8][out-of-range flag, pair count][bins cap x 4][counts cap x 8]; the host image of the tail in h_scalars
    const size_t tail = 16 + (size_t)cap * 12;
    if (int s2 = phz_reserve(ctx, ctx->scratch[0], PHZ_AS_BINS * 8 + tail)) return s2;
    if (int s2 = phz_reserve_host(ctx, ctx->h_scalars, tail)) return s2;
    char *d = (char *)ctx->scratch[0].p;
    unsigned long long *hist = (unsigned long long *)d;
    unsigned int *flags = (unsigned int *)(d + PHZ_AS_BINS * 8);
    int32_t *d_bins = (int32_t *)(d + PHZ_AS_BINS * 8 + 16);
    unsigned long long *d_counts = (unsigned long long *)(d + PHZ_AS_BINS * 8 + 16 + (size_t)cap * 4);
    PHZ_HIP(ctx, hipMemsetAsync(d, 0, PHZ_AS_BINS * 8 + 16, ctx->stream));
    if (n_shards > 0) {
        LinesTab T;
        std::vector<uint32_t> grids;
        if (int s2 = upload_tab(ctx, L.data(), n_shards, as_hist_blocks, &T, &grids)) return s2;
        if (grids.back() > 0) hipLaunchKernelGGL(k_as_hist, dim3(grids.back()), dim3(256), 0, ctx->stream, T, hist, flags);
    }
    hipLaunchKernelGGL(k_hist_compact, dim3(PHZ_AS_BINS / 1024), dim3(1024), 0, ctx->stream, (const unsigned long long *)hist, cap, d_bins, d_counts, (int32_t *)(flags + 1));
    PHZ_HIP(ctx, hipGetLastError());
    PHZ_HIP(ctx, hipMemcpyAsync(ctx->h_scalars.p, flags, tail, hipMemcpyDeviceToHost, ctx->stream));
    if (int s = t.stop()) return s;
    const unsigned int *hf = (const unsigned int *)ctx->h_scalars.p;
    if (hf[0]) return phz_fail(ctx, PHZ_E_UNSUPPORTED, "AS value outside int16");
    *n_bins = (int32_t)hf[1];
    if ((int)hf[1] > cap) return PHZ_E_CAPACITY;
    {   // the pairs arrive workgroup by workgroup in no particular order: into bin order (a few dozen of them)
        const int32_t *hb = (const int32_t *)((const char *)ctx->h_scalars.p + 16);
        const int64_t *hc = (const int64_t *)((const char *)ctx->h_scalars.p + 16 + (size_t)cap * 4);
        std::vector<int> order(hf[1]);
        for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return hb[a] < hb[b]; });
        for (size_t i = 0; i < order.size(); i++) { bins[i] = hb[order[i]]; counts[i] = hc[order[i]]; }
    }
    return PHZ_OK;
}

// the AS column of a shard as the 2-byte plane (phz_lines.read_as16): what a producer that only has the 4-byte column + flag calls once per shard
__global__ __launch_bounds__(256) void k_as16(const int32_t *aln, const uint8_t *has, int64_t n, int16_t *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int a = aln[i];
    out[i] = (has && !has[i]) ? (int16_t)PHZ_AS16_NONE : (int16_t)(a >= PHZ_AS16_RANGE ? PHZ_AS16_RANGE : (a <= -PHZ_AS16_RANGE ? -PHZ_AS16_RANGE : a));
}
extern "C" int phz_as_plane(phz_ctx *ctx, const int32_t *aln, const uint8_t *has_as, int64_t n, int16_t *out) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || n < 0 || (n && (!aln || !out))) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    if (n) hipLaunchKernelGGL(k_as16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, aln, has_as, n, out);
    PHZ_HIP(ctx, hipGetLastError());
    return PHZ_OK;          // (no wait: the consumers -- phz_as_cutoff, phz_tally -- run on the same stream)
}

// numpy.percentile(scores, q) (the default "linear" method: numpy/lib/_function_base_impl.py _quantile / _lerp; the reference's call at
// phaser.py:551) over the multiset {bin - 32768 repeated count times}, from the occupied bins in ascending order: the same float64 operations on the
// two neighbouring order statistics (engine.percentile_from_band is the Python twin, pinned against numpy in the tests)
static double percentile_of_bins(const int32_t *bins, const int64_t *counts, int nb, double q) {
    int64_t n = 0;
    for (int i = 0; i < nb; i++) n += counts[i];
    const double quant = q / 100.0;
    const double virt = (double)(n - 1) * quant;
    int64_t prev = (int64_t)floor(virt);
    const double gamma = virt - (double)prev;
    int64_t nxt = prev + 1;
    if (virt >= (double)(n - 1)) prev = nxt = n - 1;
    if (virt < 0) prev = nxt = 0;
    auto value_at = [&](int64_t k) {                 // k-th smallest score (0-based)
        int64_t c = 0;
        for (int i = 0; i < nb; i++) { c += counts[i]; if (c > k) return (int64_t)bins[i] - 32768; }
        return (int64_t)bins[nb - 1] - 32768;
    };
    const int64_t a = value_at(prev), b = value_at(nxt), diff = b - a;
    double out = (double)a + (double)diff * gamma;
    if (gamma >= 0.5) out = (double)b - (double)diff * (1 - gamma);
    return out;
}

// AS histogram of the shards of one BAM + the percentile, in one call and one host wait: *found = 0 when no record carries an alignment score
extern "C" int phz_as_cutoff(phz_ctx *ctx, const phz_lines *shards, int n_shards, double q_percent, double *cutoff, int32_t *found) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !cutoff || !found) return PHZ_E_ARG;
    constexpr int CAP = 4096;
    std::vector<int32_t> bins(CAP); std::vector<int64_t> counts(CAP);
    int32_t nb = 0;
    if (int s = phz_as_histogram_sparse(ctx, shards, n_shards, CAP, bins.data(), counts.data(), &nb)) return s;
    int64_t n = 0;
    for (int i = 0; i < nb; i++) n += counts[i];
    *found = n > 0 ? 1 : 0; *cutoff = 0.0;
    if (n > 0) *cutoff = percentile_of_bins(bins.data(), counts.data(), nb, q_percent);
    return PHZ_OK;
}

// numpy.percentile on the device: one workgroup over the 64 Ki-bin histogram (bin b = AS b - 32768).  Thread t adds up bins [64 t, 64 t + 64), the partial sums are
// scanned by the waves, the two order statistics the linear method interpolates between are found by the threads whose ranges hold them; thread 0 then performs
// percentile_of_bins' float64 operations one by one (contraction off: an fma would round once where numpy rounds twice).
__global__ __launch_bounds__(1024) void k_as_percentile(const unsigned long long *hist, const unsigned int *flags, double quant, double *out) {
#pragma clang fp contract(off)
    __shared__ unsigned long long s_incl[1024];
    __shared__ unsigned long long s_wsum[16];
    __shared__ long long s_val[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long mine = 0;
    for (int b = 0; b < 64; b++) mine += hist[tid * 64 + b];
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    unsigned long long before = 0;
    for (int w = 0; w < wave; w++) before += s_wsum[w];
    incl += before;
    s_incl[tid] = incl;
    __syncthreads();
    const long long n = (long long)s_incl[1023];
    if (n <= 0) { if (tid == 0) { out[0] = 0.0; out[1] = 0.0; out[2] = flags[0] ? 1.0 : 0.0; out[3] = 0.0; } return; }
    const double virt = (double)(n - 1) * quant;
    long long prev = (long long)floor(virt);
    const double gamma = virt - (double)prev;
    long long nxt = prev + 1;
    if (virt >= (double)(n - 1)) prev = nxt = n - 1;
    if (virt < 0) prev = nxt = 0;
    const unsigned long long excl = incl - mine;
    for (int which = 0; which < 2; which++) {
        const unsigned long long k = (unsigned long long)(which ? nxt : prev);
        if (k >= excl && k < incl) {          // the k-th smallest score lies in this thread's bins
            unsigned long long c = excl;
            for (int b = 0; b < 64; b++) { c += hist[tid * 64 + b]; if (c > k) { s_val[which] = (long long)(tid * 64 + b) - 32768; break; } }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const long long a = s_val[0], b = s_val[1], diff = b - a;
        double r = (double)a + (double)diff * gamma;
        if (gamma >= 0.5) r = (double)b - (double)diff * (1 - gamma);
        out[0] = r; out[1] = 1.0; out[2] = flags[0] ? 1.0 : 0.0; out[3] = (double)n;
    }
}

extern "C" int phz_as_cutoff_enqueue(phz_ctx *ctx, const phz_lines *shards, int n_shards, double q_percent, double *dev_out) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || (!shards && n_shards) || n_shards < 0 || !dev_out) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    Staging st(ctx);
    std::vector<LinesDev> L((size_t)n_shards);
    for (int i = 0; i < n_shards; i++)
        if (int s = stage_lines(st, shards[i], PHZ_DEVICE, &L[(size_t)i])) return s;
    if (int s2 = phz_reserve(ctx, ctx->scratch[0], PHZ_AS_BINS * 8 + 16)) return s2;
    char *d = (char *)ctx->scratch[0].p;
    unsigned long long *hist = (unsigned long long *)d;
    unsigned int *flags = (unsigned int *)(d + PHZ_AS_BINS * 8);
    PHZ_HIP(ctx, hipMemsetAsync(d, 0, PHZ_AS_BINS * 8 + 16, ctx->stream));
    if (n_shards > 0) {
        LinesTab T;
        std::vector<uint32_t> grids;
        if (int s2 = upload_tab(ctx, L.data(), n_shards, as_hist_blocks, &T, &grids)) return s2;
        if (grids.back() > 0) hipLaunchKernelGGL(k_as_hist, dim3(grids.back()), dim3(256), 0, ctx->stream, T, hist, flags);
    }
    static_assert(PHZ_AS_BINS == 65536, "k_as_percentile: 1024 threads x 64 bins");
    hipLaunchKernelGGL(k_as_percentile, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned long long *)hist, (const unsigned int *)flags, q_percent / 100.0, dev_out);
    PHZ_HIP(ctx, hipGetLastError());
    return PHZ_OK;
}

extern "C" int phz_tally(phz_ctx *ctx, const phz_lines *shards, int n_shards, int64_t nv, const uint8_t *a0, const uint8_t *a1,
                         int64_t n_qid, int n_bams, phz_tally_sizes *sizes, int space) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || (!shards && n_shards) || !sizes || nv < 0 || n_qid < 0 || n_shards < 0 || n_bams < 1) return PHZ_E_ARG;
    if (nv >= (1ll << 28)) return phz_fail(ctx, PHZ_E_ARG, "more than 2^28 variants in one call");
    if (n_qid >= (1ll << 31)) return phz_fail(ctx, PHZ_E_ARG, "more than 2^31 QNAME ids in one call");
    if (2 * nv * n_bams >= (1ll << 31)) return phz_fail(ctx, PHZ_E_ARG, "variants x BAMs exceeds the read-list key space");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    memset(sizes, 0, sizeof(*sizes));
    Staging st(ctx);
    std::vector<LinesDev> L((size_t)n_shards);
    int64_t total = 0;
    for (int b = 0; b < n_shards; b++) {
        if (shards[b].bam_index < 0 || shards[b].bam_index >= n_bams) return phz_fail(ctx, PHZ_E_ARG, "bam_index outside [0, n_bams)");
        if (shards[b].var_base < 0 || shards[b].qid_base < 0) return phz_fail(ctx, PHZ_E_ARG, "negative base");
        if (int s = stage_lines(st, shards[b], space, &L[b])) return s;
        L[b].line_base = total;
        total += L[b].n;
    }
    if (total >= (1ll << 32) - 16) return phz_fail(ctx, PHZ_E_ARG, "more than 2^32 call lines in one call");
    // variants of every shard's chromosome: up to the next chromosome's first variant (shards of a chromosome share var_base)
    for (int b = 0; b < n_shards; b++) {
        int64_t end = nv;
        for (int c = 0; c < n_shards; c++)
            if (L[c].var_base > L[b].var_base && L[c].var_base < end) end = L[c].var_base;
        if (L[b].var_base > nv) return phz_fail(ctx, PHZ_E_ARG, "var_base beyond the variant space");
        L[b].nv_chrom = (int32_t)(end - L[b].var_base);
    }
    const size_t NV = (size_t)(nv ? nv : 1), NQ = (size_t)(n_qid ? n_qid : 1), TOT = (size_t)(total ? total : 1);
    const size_t NRL = NV * 2 * (size_t)n_bams;
    const size_t QWORDS = (NQ + 31) / 32 + (size_t)QW / 32, RLWORDS = (NRL + 31) / 32;      // (+ the LDS window of k_line, which may reach past the last id)
    if (ctx->tally_buf.size() < (size_t)R_COUNT) ctx->tally_buf.resize(R_COUNT);
    DevBuf *R = ctx->tally_buf.data();
    DevBuf *S = ctx->scratch;
    const uint8_t *d_a0, *d_a1;
    if (space == PHZ_DEVICE) { d_a0 = a0; d_a1 = a1; }
    else {
        if (int s = phz_reserve(ctx, R[R_A0], NV)) return s;
        if (int s = phz_reserve(ctx, R[R_A1], NV)) return s;
        if (nv) {
            PHZ_HIP(ctx, hipMemcpyAsync(R[R_A0].p, a0, (size_t)nv, hipMemcpyHostToDevice, ctx->stream));
            PHZ_HIP(ctx, hipMemcpyAsync(R[R_A1].p, a1, (size_t)nv, hipMemcpyHostToDevice, ctx->stream));
        }
        d_a0 = (const uint8_t *)R[R_A0].p; d_a1 = (const uint8_t *)R[R_A1].p;
    }
#define RSV(buf, bytes) do { if (int s_ = phz_reserve(ctx, buf, (bytes))) return s_; } while (0)
    // tiles of the per-line stages (k_line and k_tile: TL lines each) and variant blocks of k_colscan (TWT variants each), over the shard table
    LinesTab TT, TC;
    TT.L = nullptr; TT.blk0 = nullptr; TT.n = 0; TC = TT;
    std::vector<uint32_t> gt, gc;
    if (n_shards > 0) {
        if (int s2 = upload_tab(ctx, L.data(), n_shards, [](const LinesDev &l) { return (unsigned)((l.n + TL - 1) / TL); }, &TT, &gt, 0, 2)) return s2;
        if (int s2 = upload_tab(ctx, L.data(), n_shards, [](const LinesDev &l) { return (unsigned)((l.nv_chrom + TWT - 1) / TWT); }, &TC, &gc, 1, 2)) return s2;
    }
    const unsigned grid_t = n_shards > 0 ? gt.back() : 0u, grid_c = n_shards > 0 ? gc.back() : 0u;
    const size_t NT = grid_t ? grid_t : 1;
    RSV(R[R_CNT], NV * 12); RSV(R[R_FIRST], NV * 8); RSV(R[R_DIST], NV * 12); RSV(R[R_RANK], NV * 8); RSV(R[R_CLS], TOT); RSV(R[R_LINEQ], TOT * 4);
    RSV(R[R_RLCNT], NRL * 4); RSV(R[R_RLSTART], (NRL + 1) * 4); RSV(R[R_RLTMP], TOT * 8); RSV(R[R_RLLIST], TOT * 4); RSV(R[R_RLQID], TOT * 4);
    RSV(R[R_SPQ], TOT * 4); RSV(R[R_SPITEM], TOT * 8);
    RSV(R[R_TILEWB], NT * 4); RSV(R[R_TILEM], NT * (TWT * 3) * 2); RSV(R[R_TILEB], NT * (TWT * 2) * 4);
    RSV(R[R_QSEEN], QWORDS * 4); RSV(R[R_QDUP], QWORDS * 4); RSV(R[R_RLDIRTY], RLWORDS * 4);
    RSV(S[T_TOUCHED], TOT * 4); RSV(S[T_CNT_T], (TOT + 1) * 4); RSV(S[T_BASE_T], (TOT + 1) * 4); RSV(S[T_ITEMS], (2 * TOT + (size_t)(n_shards + 1) * TL) * 8); RSV(S[T_COUNTERS], CNT_BYTES);
    RSV(S[T_DEG], NV * 4); RSV(S[T_EOFF], (NV + 1) * 4); RSV(S[T_MISC], std::max(NRL, (size_t)1) * 8 + 64);
    hipStream_t sm = ctx->stream;
    {   // two arrays are persistent and all zero between calls: the spilled-line counter / fill cursor per QNAME (k_group_plan and k_groups return it to
        // zero) and the fill cursor per read list (k_rl_sort_dirty does).  They are cleared only when (re)allocated -- or after a call that failed half way
        const size_t before = ctx->tally_qcount.cap, before_rl = R[R_RLFILL].cap;
        RSV(ctx->tally_qcount, NQ * 4); RSV(R[R_RLFILL], NRL * 4);
        if (ctx->tally_qcount.cap != before || ctx->tally_dirty) PHZ_HIP(ctx, hipMemsetAsync(ctx->tally_qcount.p, 0, ctx->tally_qcount.cap, sm));
        if (R[R_RLFILL].cap != before_rl || ctx->tally_dirty) PHZ_HIP(ctx, hipMemsetAsync(R[R_RLFILL].p, 0, R[R_RLFILL].cap, sm));
    }
    int32_t *d_cnt = (int32_t *)R[R_CNT].p, *d_dist = (int32_t *)R[R_DIST].p;
    unsigned long long *d_first = (unsigned long long *)R[R_FIRST].p, *d_rank = (unsigned long long *)R[R_RANK].p;
    uint8_t *d_cls = (uint8_t *)R[R_CLS].p;
    uint32_t *line_q = (uint32_t *)R[R_LINEQ].p;
    uint32_t *rl_cnt = (uint32_t *)R[R_RLCNT].p, *rl_start = (uint32_t *)R[R_RLSTART].p, *rl_fill = (uint32_t *)R[R_RLFILL].p, *rl_list = (uint32_t *)R[R_RLLIST].p;
    uint64_t *rl_tmp = (uint64_t *)R[R_RLTMP].p;
    int32_t *rl_qid = (int32_t *)R[R_RLQID].p;
    uint32_t *qcount = (uint32_t *)ctx->tally_qcount.p;
    uint32_t *touched = (uint32_t *)S[T_TOUCHED].p, *cnt_t = (uint32_t *)S[T_CNT_T].p, *base_t = (uint32_t *)S[T_BASE_T].p;
    uint64_t *items = (uint64_t *)S[T_ITEMS].p;
    unsigned long long *counters = (unsigned long long *)S[T_COUNTERS].p;      // see k_pairs
    uint32_t *counters32 = (uint32_t *)(counters + 8);
    uint32_t *deg = (uint32_t *)S[T_DEG].p, *eoff = (uint32_t *)S[T_EOFF].p;
    uint32_t *dirty_list = (uint32_t *)S[T_MISC].p, *big_list = dirty_list + NRL;
    int32_t *tile_wb = (int32_t *)R[R_TILEWB].p; uint16_t *tile_m = (uint16_t *)R[R_TILEM].p; uint32_t *tile_b = (uint32_t *)R[R_TILEB].p;
    uint32_t *q_seen = (uint32_t *)R[R_QSEEN].p, *q_dup = (uint32_t *)R[R_QDUP].p, *rl_dirty = (uint32_t *)R[R_RLDIRTY].p;

    Timer timer(ctx, PHZ_T_TALLY);
    ctx->tally_dirty = true;           // cleared again when the call completes
    PHZ_HIP(ctx, hipMemsetAsync(d_cnt, 0, NV * 12, sm));
    PHZ_HIP(ctx, hipMemsetAsync(d_dist, 0, NV * 12, sm));
    PHZ_HIP(ctx, hipMemsetAsync(rl_cnt, 0, NRL * 4, sm));
    PHZ_HIP(ctx, hipMemsetAsync(d_rank, 0xff, NV * 8, sm));
    PHZ_HIP(ctx, hipMemsetAsync(counters, 0, CNT_BYTES, sm));
    PHZ_HIP(ctx, hipMemsetAsync(d_first, 0xff, NV * 8, sm));         // unsigned max for atomicMin == -1 as int64 ("none")
    PHZ_HIP(ctx, hipMemsetAsync(q_seen, 0, QWORDS * 4, sm));
    PHZ_HIP(ctx, hipMemsetAsync(q_dup, 0, QWORDS * 4, sm));
    PHZ_HIP(ctx, hipMemsetAsync(rl_dirty, 0, RLWORDS * 4, sm));

    const int single_bam = n_bams <= 1 ? 1 : 0;     // one BAM: every QNAME's read_vars list is owned by that BAM
    LineOut O;
    O.a0 = d_a0; O.a1 = d_a1; O.line_cls = d_cls; O.line_q = line_q; O.var_count = d_cnt; O.var_first = d_first; O.rl_cnt = rl_cnt;
    O.tile_wb = tile_wb; O.tile_m = tile_m; O.q_seen = q_seen; O.q_dup = q_dup; O.rl_dirty = rl_dirty; O.dirty_list = dirty_list;
    O.counters = counters; O.nb = n_bams; O.prof = nullptr;
    const bool profiling = getenv("PHZ_TALLY_PROFILE") != nullptr;
    const unsigned grid_l = grid_t;
    if (profiling) { RSV(S[20], (size_t)(grid_l + grid_t + 2) * 16); O.prof = (unsigned long long *)S[20].p; PHZ_HIP(ctx, hipMemsetAsync(S[20].p, 0, (size_t)(grid_l + grid_t + 2) * 16, sm)); }
    if (grid_t) {
        hipLaunchKernelGGL(k_tile_base, dim3((unsigned)n_shards), dim3(256), 0, sm, TT, tile_wb);
        hipLaunchKernelGGL(k_line, dim3(grid_l), dim3(LINE_TB), 0, sm, TT, O);
        ColScan CS; CS.tile_wb = tile_wb; CS.tile_m = tile_m; CS.tile_b = tile_b; CS.var_count = d_cnt; CS.rl_cnt = rl_cnt; CS.nb = n_bams;
        if (grid_c) hipLaunchKernelGGL(k_colscan, dim3(grid_c), dim3(TWT * CS_PARTS), 0, sm, TC, TT, CS);
    }
    if (nv) hipLaunchKernelGGL(k_noise, dim3(std::min(nblk(nv), 256u)), dim3(256), 0, sm, (const int32_t *)d_cnt, nv, counters + 4);
    if (int s = gscan_excl<uint32_t, uint32_t>(ctx, rl_cnt, rl_start, (int64_t)NRL, S[T_SCAN_TMP])) return s;
    uint32_t *sp_q = (uint32_t *)R[R_SPQ].p; uint64_t *sp_item = (uint64_t *)R[R_SPITEM].p;
    if (grid_t) {
        TileOut TO;
        TO.line_cls = d_cls; TO.line_q = line_q; TO.q_dup = q_dup; TO.qcount = qcount; TO.items = items; TO.sp_q = sp_q; TO.sp_item = sp_item; TO.touched = touched;
        TO.var_rank = d_rank; TO.var_distinct = d_dist; TO.tile_wb = tile_wb; TO.tile_b = tile_b; TO.rl_start = rl_start; TO.rl_dirty = rl_dirty;
        TO.rl_qid = rl_qid; TO.rl_list = rl_list; TO.rl_cursor = rl_fill; TO.rl_tmp = rl_tmp; TO.counters = counters; TO.nb = n_bams; TO.nv = nv;
        TO.prof = profiling ? (unsigned long long *)S[20].p + 2 * (size_t)grid_l : nullptr;
        hipLaunchKernelGGL(k_tile, dim3(grid_t), dim3(TILE_TB), 0, sm, TT, TO);
    }
    PHZ_HIP(ctx, hipGetLastError());
    // (read-backs go to page-locked memory: a copy into a pageable vector is a blocking, staged copy -- PhzMail in phz_internal.h)
    if (int s = phz_reserve_host(ctx, ctx->mail_host, CNT_BYTES + 64)) return s;
    std::vector<unsigned long long> h_cnt(CNT_BYTES / 8, 0ull);
    unsigned long long *h_counters = h_cnt.data();
    auto spread_sum = [&](int k) { unsigned long long t = 0; for (int c = 0; c < N_SPREAD; c++) t += h_counters[16 + c * SPREAD_WORDS + k]; return t; };
    PHZ_HIP(ctx, hipMemcpyAsync(ctx->mail_host.p, counters, CNT_BYTES, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    memcpy(h_counters, ctx->mail_host.p, CNT_BYTES);
    if (h_counters[13]) return phz_fail(ctx, PHZ_E_UNSUPPORTED, "AS value outside int16");          // (reported by the device-side percentile of a BAM: phz_as_cutoff_enqueue)
    const int64_t n_kept = (int64_t)spread_sum(2);
    const int64_t n_dirty = (int64_t)h_counters[11];
    ctx->counters[PHZ_C_FAR_LINES] += (int64_t)h_counters[12]; ctx->counters[PHZ_C_DIRTY_LISTS] += n_dirty;
    if (profiling) {
        std::vector<unsigned long long> pr((size_t)(grid_l + grid_t) * 2);
        PHZ_HIP(ctx, hipMemcpy(pr.data(), S[20].p, pr.size() * 8, hipMemcpyDeviceToHost));
        for (int which = 0; which < 2; which++) {
            const size_t b0 = which ? grid_l : 0, nb_ = which ? grid_t : grid_l;
            if (!nb_) continue;
            unsigned long long t0 = ~0ull, t1 = 0; double sum = 0; unsigned long long mx = 0; std::vector<unsigned long long> d;
            for (size_t b = 0; b < nb_; b++) { const unsigned long long a = pr[2 * (b0 + b)], e = pr[2 * (b0 + b) + 1]; t0 = std::min(t0, a); t1 = std::max(t1, e); sum += (double)(e - a); mx = std::max(mx, e - a); d.push_back(e - a); }
            std::sort(d.begin(), d.end());
            // start-time profile: how many workgroups had started by 10 %, 50 %, 90 % of the kernel's span
            size_t s10 = 0, s50 = 0, s90 = 0; const double span = (double)(t1 - t0);
            for (size_t b = 0; b < nb_; b++) { const double st = (double)(pr[2 * (b0 + b)] - t0) / span; s10 += st <= 0.1; s50 += st <= 0.5; s90 += st <= 0.9; }
            fprintf(stderr, "[tally profile] %s: %zu workgroups, span %.1f us, lifetime avg %.2f us median %.2f p99 %.2f max %.2f us, sum %.1f ms -> avg %.0f resident; started by 10/50/90%% of the span: %zu %zu %zu\n",
                    which ? "k_tile" : "k_line", nb_, span / 100.0, sum / nb_ / 100.0, d[nb_ / 2] / 100.0, d[nb_ * 99 / 100] / 100.0, mx / 100.0, sum / 1e5, sum / span, s10, s50, s90);
        }
        fprintf(stderr, "[tally profile] far lines %llu, dirty read lists %lld\n", h_counters[12], (long long)n_dirty);
    }
    const int64_t n_complete = (int64_t)grid_t * TL;     // item slots of the tiles (groups finished inside their tile, holes in between)
    const int64_t nt = (int64_t)(h_counters[10] >> 32);  // QNAMEs whose lines straddle tiles: one group each, built from the spilled lines
    const int64_t n_spill = (int64_t)(h_counters[10] & 0xFFFFFFFFull);
    if (nt) {
        hipLaunchKernelGGL(k_group_plan, dim3(nblk(nt)), dim3(256), 0, sm, nt, (const uint32_t *)touched, qcount, cnt_t);
        if (int s = gscan_excl<uint32_t, uint32_t>(ctx, cnt_t, base_t, nt, S[T_SCAN_TMP])) return s;
        hipLaunchKernelGGL(k_group_base, dim3(nblk(nt)), dim3(256), 0, sm, nt, (const uint32_t *)touched, (const uint32_t *)base_t, qcount);
        hipLaunchKernelGGL(k_items_spill, dim3(nblk(n_spill)), dim3(256), 0, sm, n_spill, (const uint32_t *)sp_q, (const uint64_t *)sp_item, qcount, items + n_complete);
        GroupOut G; G.qcount = qcount; G.items = items + n_complete; G.cnt_t = cnt_t; G.base_t = base_t; G.touched = touched; G.var_rank = d_rank; G.var_distinct = d_dist;
        G.counters = counters; G.single_bam = single_bam;
        hipLaunchKernelGGL(k_groups, dim3(nblk(nt)), dim3(256), 0, sm, nt, TT, G);
    }
    // the read lists with far lines into line order
    PHZ_HIP(ctx, hipMemsetAsync(counters32, 0, 16, sm));
    if (n_dirty) hipLaunchKernelGGL(k_rl_sort_dirty, dim3((unsigned)n_dirty), dim3(256), 0, sm, (const uint32_t *)dirty_list, (const uint32_t *)rl_start, (const uint64_t *)rl_tmp, rl_qid, rl_fill,
                                    big_list, counters32);
    // variant pairs.  The table lives in the ctx, sized from the variant count and kept clean by k_edge_final; a pass that overflows it is redone
    // with a larger one
    uint64_t cap = 1 << 16;
    while (cap < 4 * (uint64_t)NV && cap < (1ull << 30)) cap <<= 1;
    if (ctx->tally_table_cap > cap) cap = ctx->tally_table_cap;          // a sample that needed a larger table keeps it (no overflow + redo per call)
    int64_t ne = 0;
    uint32_t h_c32[4] = {0, 0, 0, 0};
    uint32_t h_tail[2] = {0, 0};
    for (int attempt = 0;; attempt++) {
        const size_t old_k = S[T_GKEYS].cap;
        RSV(S[T_GKEYS], cap * GE_WORDS * 4); RSV(S[T_USED], cap * 4); RSV(S[T_USEDKEY], cap * 8);
        uint32_t *tab = (uint32_t *)S[T_GKEYS].p;
        if (S[T_GKEYS].cap != old_k || ctx->tally_table_dirty || attempt > 0) PHZ_HIP(ctx, hipMemsetAsync(tab, 0, S[T_GKEYS].cap, sm));
        ctx->tally_table_dirty = true;
        PHZ_HIP(ctx, hipMemsetAsync(counters, 0, 24, sm));              // overflow
        PHZ_HIP(ctx, hipMemsetAsync(counters + 16, 0, CNT_BYTES - 128, sm));      // spread statistics
        PHZ_HIP(ctx, hipMemsetAsync(counters + 7, 0, 8, sm));           // used slots
        PHZ_HIP(ctx, hipMemsetAsync(deg, 0, NV * 4, sm));               // edges per first variant, counted while the slots are claimed
        const int64_t m_items = n_complete + n_spill;                // the tiles' slots, then one slot per spilled line
        if (m_items && getenv("PHZ_TALLY_DEBUG")) {
            hipLaunchKernelGGL(k_pairs<1>, dim3((unsigned)((m_items + PAIR_ITEMS * PAIR_ROUNDS - 1) / (PAIR_ITEMS * PAIR_ROUNDS))), dim3(PAIRS_TB), 0, sm, (const uint64_t *)items, m_items, tab, (uint32_t)(cap - 1), (uint32_t *)S[T_USED].p, (uint64_t *)S[T_USEDKEY].p, deg, counters);
            hipLaunchKernelGGL(k_pairs<2>, dim3((unsigned)((m_items + PAIR_ITEMS * PAIR_ROUNDS - 1) / (PAIR_ITEMS * PAIR_ROUNDS))), dim3(PAIRS_TB), 0, sm, (const uint64_t *)items, m_items, tab, (uint32_t)(cap - 1), (uint32_t *)S[T_USED].p, (uint64_t *)S[T_USEDKEY].p, deg, counters);
        }
        if (m_items) hipLaunchKernelGGL(k_pairs<0>, dim3((unsigned)((m_items + PAIR_ITEMS * PAIR_ROUNDS - 1) / (PAIR_ITEMS * PAIR_ROUNDS))), dim3(PAIRS_TB), 0, sm, (const uint64_t *)items, m_items, tab,
                                        (uint32_t)(cap - 1), (uint32_t *)S[T_USED].p, (uint64_t *)S[T_USEDKEY].p, deg, counters);
        PHZ_HIP(ctx, hipGetLastError());
        {
            PhzMail mail(ctx);
            const int m0 = mail.add(counters, CNT_BYTES), m1 = mail.add(rl_start + NRL, 4);       // (counters32 is part of the counter block)
            if (int s = mail.send()) return s;
            PHZ_HIP(ctx, hipStreamSynchronize(sm));
            memcpy(h_counters, mail.at<char>(m0), CNT_BYTES);
            memcpy(h_c32, (const char *)mail.at<char>(m0) + 64, 16);
            h_tail[1] = *mail.at<uint32_t>(m1);
        }
        if (h_counters[2] == 0) { ne = (int64_t)h_counters[7]; ctx->tally_table_cap = cap; break; }
        if (attempt == 4 || cap >= (1ull << 31)) return phz_fail(ctx, PHZ_E_NOMEM, "variant-pair table did not converge");
        cap <<= 2;
    }
    // the dirty read lists beyond the LDS stage: through the device radix sort
    if (h_c32[1]) {
        std::vector<uint32_t> big(h_c32[1]), rs((size_t)NRL + 1);
        PHZ_HIP(ctx, hipMemcpy(big.data(), big_list, big.size() * 4, hipMemcpyDeviceToHost));
        PHZ_HIP(ctx, hipMemcpy(rs.data(), rl_start, rs.size() * 4, hipMemcpyDeviceToHost));
        size_t longest = 0;
        for (uint32_t e : big) longest = std::max(longest, (size_t)(rs[e + 1] - rs[e]));
        RSV(R[R_SORTK], longest * 8); RSV(R[R_SORTV0], longest * 4); RSV(R[R_SORTV1], longest * 4);
        const int hi_bit = 32 + bits_for((uint64_t)(total > 1 ? total - 1 : 1));
        for (uint32_t e : big) {
            const int64_t n = (int64_t)rs[e + 1] - rs[e];
            int where = 0;
            if (int s = radix_sort_pairs<uint64_t, uint32_t>(ctx, rl_tmp + rs[e], (uint64_t *)R[R_SORTK].p, (uint32_t *)R[R_SORTV0].p, (uint32_t *)R[R_SORTV1].p, n, 32, hi_bit,
                                                            R[R_SORTCNT], S[T_SCAN_TMP], &where)) return s;
            hipLaunchKernelGGL(k_rl_take_qid, dim3(nblk(n)), dim3(256), 0, sm, (const uint64_t *)(where ? (uint64_t *)R[R_SORTK].p : rl_tmp + rs[e]), n, rl_qid + rs[e]);
        }
    }
    const size_t NE = (size_t)(ne ? ne : 1);
    RSV(R[R_EA], NE * 4); RSV(R[R_EB], NE * 4); RSV(R[R_CELLS], NE * 36); RSV(R[R_LINKED], NE); RSV(R[R_CTO], NE * 12); RSV(R[R_STATS], NE * 20);
    RSV(S[T_EB], NE * 4); RSV(S[T_ESLOT], NE * 4); RSV(S[T_EA], NE * 4);
    if (ne > 0) {
        uint32_t *tab = (uint32_t *)S[T_GKEYS].p;
        if (int s = gscan_excl<uint32_t, uint32_t>(ctx, deg, eoff, nv, S[T_SCAN_TMP])) return s;
        hipLaunchKernelGGL(k_edge_scatter, dim3(nblk(ne)), dim3(256), 0, sm, (const uint32_t *)S[T_USED].p, (const uint64_t *)S[T_USEDKEY].p, ne, (const uint32_t *)eoff, deg,
                           (uint32_t *)S[T_EA].p, (uint32_t *)S[T_EB].p, (uint32_t *)S[T_ESLOT].p);
        hipLaunchKernelGGL(k_edge_sort_big, dim3(nblk(nv)), dim3(256), 0, sm, nv, (const uint32_t *)eoff, (uint32_t *)S[T_EB].p, (uint32_t *)S[T_ESLOT].p);
        hipLaunchKernelGGL(k_edge_out, dim3(nblk(ne)), dim3(256), 0, sm, ne, (const uint32_t *)eoff, (const uint32_t *)S[T_EA].p, (const uint32_t *)S[T_EB].p,
                           (const uint32_t *)S[T_ESLOT].p, tab, (int32_t *)R[R_EA].p, (int32_t *)R[R_EB].p, (int32_t *)R[R_CELLS].p, (uint8_t *)R[R_LINKED].p,
                           (int32_t *)R[R_CTO].p, (int32_t *)R[R_STATS].p);
    }
    PHZ_HIP(ctx, hipGetLastError());
    if (int s = timer.stop()) return s;
#undef RSV
    ctx->tally_dirty = false; ctx->tally_table_dirty = false;
    ctx->tally_gen++;
    auto &T = ctx->tally;
    T.nv = nv; T.nb = n_bams; T.n_lines = total; T.n_kept = n_kept; T.n_edges = ne; T.n_rl = (int64_t)h_tail[1];
    T.var_count = d_cnt; T.var_distinct = d_dist; T.var_first = (int64_t *)d_first; T.var_rank = (uint64_t *)d_rank; T.line_cls = d_cls;
    T.ea = (int32_t *)R[R_EA].p; T.eb = (int32_t *)R[R_EB].p; T.cells = (int32_t *)R[R_CELLS].p; T.linked = (uint8_t *)R[R_LINKED].p;
    T.cto = (int32_t *)R[R_CTO].p; T.stats = (int32_t *)R[R_STATS].p;
    T.rl_start = rl_start; T.rl_qid = rl_qid; T.rl_list = rl_list;
    sizes->n_lines = total; sizes->n_kept = T.n_kept; sizes->n_edges = ne; sizes->n_read_list = T.n_rl;
    sizes->n_items = (int64_t)spread_sum(0); sizes->pair_events = (int64_t)spread_sum(1);
    sizes->noise_match = (int64_t)h_counters[4]; sizes->noise_mismatch = (int64_t)h_counters[5];
    ctx->counters[PHZ_C_LINES] += total; ctx->counters[PHZ_C_ITEMS] += sizes->n_items; ctx->counters[PHZ_C_PAIR_EVENTS] += sizes->pair_events;
    ctx->counters[PHZ_C_EDGES] += ne;
    return PHZ_OK;
}

// copy the results of the last phz_tally into the caller's arrays (NULL members are skipped); one host wait
extern "C" int phz_tally_fetch(phz_ctx *ctx, const phz_tally_out *out, int space) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !out) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    auto &T = ctx->tally;
    const hipMemcpyKind kind = space == PHZ_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    hipStream_t sm = ctx->stream;
    auto cp = [&](void *dst, const void *src, size_t bytes) -> int {
        if (!dst || !bytes) return PHZ_OK;
        PHZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, kind, sm));
        return PHZ_OK;
    };
    const size_t nv = (size_t)T.nv, ne = (size_t)T.n_edges;
    if (int s = cp(out->var_count, T.var_count, nv * 12)) return s;
    if (int s = cp(out->var_first, T.var_first, nv * 8)) return s;
    if (int s = cp(out->var_distinct, T.var_distinct, nv * 12)) return s;
    if (int s = cp(out->var_rank, T.var_rank, nv * 8)) return s;
    if (int s = cp(out->line_cls, T.line_cls, (size_t)T.n_lines)) return s;
    if (int s = cp(out->edge_a, T.ea, ne * 4)) return s;
    if (int s = cp(out->edge_b, T.eb, ne * 4)) return s;
    if (int s = cp(out->edge_cells, T.cells, ne * 36)) return s;
    if (int s = cp(out->edge_linked, T.linked, ne)) return s;
    if (int s = cp(out->edge_cto, T.cto, ne * 12)) return s;
    if (int s = cp(out->edge_stats, T.stats, ne * 20)) return s;
    if (int s = cp(out->rl_start, T.rl_start, (nv * 2 * (size_t)T.nb + 1) * 4)) return s;
    if (int s = cp(out->rl_qid, T.rl_qid, (size_t)T.n_rl * 4)) return s;
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    return PHZ_OK;
}

// edge_a == edge_b == NULL: the edges of the last phz_tally, still resident in HBM (n_edges must match); keep[] lives in `space`
extern "C" int phz_components(phz_ctx *ctx, int64_t nv, int64_t n_edges, const int32_t *edge_a, const int32_t *edge_b,
                              const uint8_t *keep, int32_t *label, int space) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || nv < 0 || n_edges < 0 || (!label && nv)) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    Staging st(ctx);
    const int32_t *ea, *eb; const uint8_t *kp; int32_t *lab;
    if (!edge_a && !edge_b && n_edges) {
        if (n_edges != ctx->tally.n_edges || nv != ctx->tally.nv) return phz_fail(ctx, PHZ_E_ARG, "no resident edge list of that size");
        ea = ctx->tally.ea; eb = ctx->tally.eb;
    } else {
        if (int s = st.in(edge_a, (size_t)n_edges, space, &ea)) return s;
        if (int s = st.in(edge_b, (size_t)n_edges, space, &eb)) return s;
    }
    if (int s = st.in(keep, (size_t)n_edges, space, &kp)) return s;
    if (int s = st.out(label, (size_t)nv, space, &lab)) return s;
    if (int s = phz_reserve(ctx, ctx->scratch[16], (size_t)(nv ? nv : 1) * 4)) return s;
    int32_t *parent = (int32_t *)ctx->scratch[16].p;
    Timer t(ctx, PHZ_T_COMPONENTS);
    if (nv) hipLaunchKernelGGL(k_uf_init, dim3(nblk(nv)), dim3(256), 0, ctx->stream, parent, nv);
    if (n_edges) hipLaunchKernelGGL(k_uf_hook, dim3(nblk(n_edges)), dim3(256), 0, ctx->stream, parent, ea, eb, kp, n_edges);
    if (nv) hipLaunchKernelGGL(k_uf_flatten, dim3(nblk(nv)), dim3(256), 0, ctx->stream, parent, lab, nv);
    PHZ_HIP(ctx, hipGetLastError());
    if (int s = t.stop()) return s;
    if (space == PHZ_HOST && nv) PHZ_HIP(ctx, hipMemcpyAsync(label, lab, (size_t)nv * 4, hipMemcpyDeviceToHost, ctx->stream));
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PHZ_OK;
}
