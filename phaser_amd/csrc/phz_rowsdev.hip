// Device row stage: stages T7-O2 of the phasing path on the GPU, from the results phz_tally left in HBM to the finished text of the
// five output files (phaser/phaser.py line numbers):
//   pair test bookkeeping          :1594-1654  distinct (supporting, total) arguments -> the host evaluates scipy's binom.cdf once per
//                                              distinct pair (the reference's own third-party call) and hands back value + repr text
//   pruning + components           :686-726, :1861-1882, :1985-1998
//   ordering rules                 SURVEY.md 8.1 rules 2, 4, 5 (hand-written radix sorts, phz_sort.h)
//   block phasing                  :2107-2324  phase_v3: flood fill (resolve_phase), weak-point split, 2^(n-1) brute force per fragment
//                                              (lanes of a wave share the configurations), left-to-right stitching
//   haplotype read sets            :917-931, :1086-1115  distinct QNAMEs per (block, haplotype[, BAM]) and the first-appearance labels
//   rows                           :691-695 variant_connections, :737-749 allelic_counts, :865-1172 haplotypes / haplotypic_counts /
//                                  allele_config, :1180-1239 singletons
// Every file is produced in two passes over its rows (byte counts -> scan -> write) by the SAME row function instantiated with a
// counting sink and a writing sink, so the two passes cannot disagree.  Integer / byte work: no MFMA; the stage streams the tally's
// arrays and the string pools and writes ~1 GB of text per genome.
#include <string.h>

#include <chrono>
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include <type_traits>
#include "phz_internal.h"
#include "phz_sort.h"
#include "phz_text.h"
#include "phz_uf.h"

namespace {

constexpr int PS_SLOTS = 1 << 14;              // default size of the hash set of the distinct (total, supporting) arguments of the binomial test (phz_rowsdev::ps_slots)
constexpr unsigned long long PS_EMPTY = ~0ull;
#ifndef PHZ_STAT_N
#define PHZ_STAT_N 512          // the emulation tests also build a variant with a tiny value: every block then takes the paths of a block beyond the table
#endif
constexpr int STAT_N = PHZ_STAT_N;             // largest block (variants) covered by the gwStat text table and by the LDS piece arrays of k_seg_big; larger blocks:
                                               // gwStat text formatted by the host inside the run (kind 3), piece arrays in the global pool (k_seg_big<.., HUGE>)
constexpr int PH_NMAX = 256, PH_EMAX = 2048;   // largest component (variants / kept pairs) k_phase_general takes; larger ones go to the host
constexpr int PH_BRUTE_MAX = 22;               // largest fragment brute-forced on the device (2^21 configurations shared by 64 lanes)
#ifdef PHZ_PHASE_PROFILE      // build flag: cycle counts of the slowest component's sections, printed by phase_all
#define PH_TICK() __builtin_readcyclecounter()
#else
#define PH_TICK() 0ull
#endif
constexpr int PH_DP_MAXD = 10, PH_DP_MINLEN = 8;   // fragments of >= 8 variants whose pairs span <= 10 variants: dynamic programme instead of brute force
constexpr int SEG_SMALL = 32, SEG_MID = 256, SEG_LDS = 2048;       // read-set sizes: one thread / one wave (512-slot table) / one workgroup (4096 slots, global pool beyond)
constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr unsigned long long NONE64 = ~0ull;

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

// separator-joined string pool: item i = b[off[i] .. off[i+1] - 1)
struct PoolD {
    const uint32_t *off;
    const char *b;
    __device__ __forceinline__ uint32_t len(int64_t i) const { return off[i + 1] - off[i] - 1u; }
    __device__ __forceinline__ const char *at(int64_t i) const { return b + off[i]; }
};

__device__ __forceinline__ int ndigits(unsigned long long v) {
    int d = 1;
    while (v >= 10ull) { v /= 10ull; d++; }
    return d;
}

#ifndef PHZ_ROW_WAVE_MIN
#define PHZ_ROW_WAVE_MIN 16          // block rows of more variants than this are formatted by a wave each (the emulation tests also build a variant with 0)
#endif
constexpr int ROW_WAVE_MIN = PHZ_ROW_WAVE_MIN;

// ---- sinks: the same row function counts bytes or writes them
struct SCount {
    static constexpr bool writing = false;
    uint32_t n = 0;
    __device__ __forceinline__ void ch(char) { n++; }
    template <int N> __device__ __forceinline__ void lit(const char (&)[N]) { n += (uint32_t)(N - 1); }
    __device__ __forceinline__ void pool(const PoolD &P, int64_t i) { n += P.len(i); }
    __device__ __forceinline__ void raw(const char *, uint32_t l) { n += l; }
    __device__ __forceinline__ void num(long long v) { n += v < 0 ? 1u + (uint32_t)ndigits((unsigned long long)(-v)) : (uint32_t)ndigits((unsigned long long)v); }
    __device__ __forceinline__ void skip(uint32_t k) { n += k; }
    __device__ __forceinline__ unsigned long long where() const { return 0; }
    // elements t = 0 .. cnt-1 with inc(t) true, joined by sep (0 = nothing between them); put(t, sink) emits element t
    template <class INC, class PUT> __device__ __forceinline__ void join(uint32_t cnt, char sep, INC inc, PUT put) {
        bool first = true;
        for (uint32_t t = 0; t < cnt; t++) { if (!inc(t)) continue; if (!first && sep) n++; first = false; put(t, *this); }
    }
};
struct SWrite {
    static constexpr bool writing = true;
    char *p0, *p;
    unsigned long long base;         // file offset of p0
    __device__ __forceinline__ void ch(char c) { *p++ = c; }
    template <int N> __device__ __forceinline__ void lit(const char (&s)[N]) { for (int i = 0; i < N - 1; i++) p[i] = s[i]; p += N - 1; }
    __device__ __forceinline__ void raw(const char *s, uint32_t l) {        // exactly l bytes, eight at a time (gfx950 serves unaligned 8-byte accesses to
        uint32_t k = 0;                                                     // global memory and LDS; byte by byte this copy was most of the row kernels' time)
        for (; k + 8 <= l; k += 8) { unsigned long long w; __builtin_memcpy(&w, s + k, 8); __builtin_memcpy(p + k, &w, 8); }
        if (l & 4u) { uint32_t w; __builtin_memcpy(&w, s + k, 4); __builtin_memcpy(p + k, &w, 4); k += 4; }
        if (l & 2u) { uint16_t w; __builtin_memcpy(&w, s + k, 2); __builtin_memcpy(p + k, &w, 2); k += 2; }
        if (l & 1u) p[k] = s[k];
        p += l;
    }
    __device__ __forceinline__ void pool(const PoolD &P, int64_t i) { raw(P.at(i), P.len(i)); }
    __device__ __forceinline__ void num(long long v) {
        unsigned long long u = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
        if (v < 0) *p++ = '-';
        const int d = ndigits(u);
        for (int k = d - 1; k >= 0; k--) { p[k] = (char)('0' + (int)(u % 10ull)); u /= 10ull; }
        p += d;
    }
    __device__ __forceinline__ void skip(uint32_t k) { p += k; }            // bytes somebody else writes (label text)
    __device__ __forceinline__ unsigned long long where() const { return base + (unsigned long long)(p - p0); }
    template <class INC, class PUT> __device__ __forceinline__ void join(uint32_t cnt, char sep, INC inc, PUT put) {
        bool first = true;
        for (uint32_t t = 0; t < cnt; t++) { if (!inc(t)) continue; if (!first && sep) *p++ = sep; first = false; put(t, *this); }
    }
};

// ---- the same sinks for ONE ROW FORMATTED BY A WAVE (rows of blocks with dozens to hundreds of variants: one thread walking them was the tail of the
// row kernels).  All 64 lanes run the row function together on the same row; the scalar pieces are identical on every lane (lane 0 stores them), join()
// spreads the elements over the lanes: lengths by a counting sub-sink, positions by a wave scan, then every lane writes its element in place.
struct WCount {
    static constexpr bool writing = false;
    uint32_t n = 0;
    __device__ __forceinline__ void ch(char) { n++; }
    template <int N> __device__ __forceinline__ void lit(const char (&)[N]) { n += (uint32_t)(N - 1); }
    __device__ __forceinline__ void pool(const PoolD &P, int64_t i) { n += P.len(i); }
    __device__ __forceinline__ void raw(const char *, uint32_t l) { n += l; }
    __device__ __forceinline__ void num(long long v) { SCount c; c.num(v); n += c.n; }
    __device__ __forceinline__ void skip(uint32_t k) { n += k; }
    __device__ __forceinline__ unsigned long long where() const { return 0; }
    template <class INC, class PUT> __device__ __forceinline__ void join(uint32_t cnt, char sep, INC inc, PUT put) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t bytes = 0, members = 0;
        for (uint32_t t0 = 0; t0 < cnt; t0 += 64) {
            const uint32_t t = t0 + lane;
            const bool in = t < cnt && inc(t);
            SCount c;
            if (in) put(t, c);
            uint32_t l = c.n, k = in ? 1u : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { l += __shfl_xor(l, d); k += __shfl_xor(k, d); }
            bytes += l; members += k;
        }
        n += bytes + ((sep && members > 1) ? members - 1 : 0u);
    }
};
struct WWrite {
    static constexpr bool writing = true;
    char *p0, *p;
    unsigned long long base;
    __device__ __forceinline__ bool first_lane() const { return (threadIdx.x & 63u) == 0u; }
    __device__ __forceinline__ void ch(char c) { if (first_lane()) *p = c; p++; }
    template <int N> __device__ __forceinline__ void lit(const char (&s)[N]) { if (first_lane()) for (int i = 0; i < N - 1; i++) p[i] = s[i]; p += N - 1; }
    __device__ __forceinline__ void raw(const char *s, uint32_t l) { if (first_lane()) { SWrite w; w.p0 = w.p = p; w.base = 0; w.raw(s, l); } p += l; }
    __device__ __forceinline__ void pool(const PoolD &P, int64_t i) { raw(P.at(i), P.len(i)); }
    __device__ __forceinline__ void num(long long v) { SWrite w; w.p0 = w.p = p; w.base = 0; if (first_lane()) w.num(v); else { SCount c; c.num(v); w.p += c.n; } p = w.p; }
    __device__ __forceinline__ void skip(uint32_t k) { p += k; }
    __device__ __forceinline__ unsigned long long where() const { return base + (unsigned long long)(p - p0); }
    template <class INC, class PUT> __device__ __forceinline__ void join(uint32_t cnt, char sep, INC inc, PUT put) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t members = 0;                                   // elements placed so far (uniform)
        for (uint32_t t0 = 0; t0 < cnt; t0 += 64) {
            const uint32_t t = t0 + lane;
            const bool in = t < cnt && inc(t);
            SCount c;
            if (in) put(t, c);
            const uint32_t l = c.n, k = in ? 1u : 0u;
            uint32_t il = l, ik = k;                            // inclusive scans over the lanes
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t yl = __shfl_up(il, d), yk = __shfl_up(ik, d); if (lane >= (uint32_t)d) { il += yl; ik += yk; } }
            const uint32_t before_k = members + ik - k;         // elements before this one in the whole section
            if (in) {
                char *q = p + (il - l) + (sep ? (before_k > 0 ? (ik - k) + (members > 0 ? 1u : 0u) : 0u) : 0u);
                // separators written before this chunk's elements: one per earlier element of the chunk, plus one in front of the chunk's first
                // element when elements were placed before the chunk
                if (sep && before_k > 0) q[-1] = sep;
                SWrite w; w.p0 = w.p = q; w.base = base + (unsigned long long)(q - p0);
                put(t, w);
            }
            const uint32_t tl = __shfl(il, 63), tk = __shfl(ik, 63);
            p += tl + (sep ? (tk > 0 ? (members > 0 ? tk : tk - 1u) : 0u) : 0u);
            members += tk;
        }
    }
};

// Per block member (index into mem_s), built once per pass by k_mem_rec: everything the block rows need of the variant in ONE 32-byte load -- the
// row functions walked mem_s -> v_alle -> pool offsets -> bytes per variant and section, a chain of dependent loads that was their whole run time
struct __attribute__((aligned(16))) MemRec {
    uint32_t uid_off, rsid_off, a_off[2];       // a_off / a_len / ph[h]: the allele of haplotype h (A = 0, B = 1) and its VCF phase index
    uint16_t uid_len, rsid_len, a_len[2];
    uint8_t black, ref_a, ref_b; int8_t ph[2]; uint8_t wide, pad[2];
};
static_assert(sizeof(MemRec) == 32, "MemRec is two 16-byte loads");

// ---- everything the row functions read (device pointers)
struct RD {
    int64_t nv, ne, nblocks, n_linked, n_keys;
    int nchrom, nb, unique_ids, unphased_vars;
    const uint16_t *vchrom; const int32_t *pos;
    PoolD uid, rsid, alle, maft, chromn, bamn, statt, statx, pvt;
    const double *mafv; const uint8_t *is_ref; const int8_t *phase_idx; const uint8_t *black; const uint8_t *bam_excl;
    // tally
    const int32_t *var_count, *var_distinct, *ea, *eb, *cis, *trans, *sup, *tot, *cfgv;
    const uint32_t *rl_start; const int32_t *rl_qid; const uint32_t *rl_list;
    // derived
    const uint32_t *eorder; const int32_t *va, *vb; const uint32_t *e_slot;
    const uint32_t *key_g;
    const uint32_t *mem_s, *blk_mstart, *blk_len; const int32_t *blk_of; const uint8_t *v_alle;
    const uint32_t *blk_sup, *blk_tot, *blk_cnt, *seg_ns, *single_n;
    const uint8_t *blk_conc, *blk_cormode, *blk_statkind; const uint32_t *blk_statidx; const int32_t *blk_maxmaf;
    const uint32_t *its, *labels; unsigned long long *piece_dst;
    const unsigned long long *cfg_base; const uint32_t *cfg_chunk;
    const uint32_t *cfg_pl, *cfg_pb; const unsigned long long *cfg_ps, *cfg_bbase;      // allele_config byte offsets in closed form (k_cfg_prefix)
    const MemRec *mrec; const uint32_t *lab_e, *lab_skip; int64_t nmem;       // lab_*[(h * nb + bam) * nmem + member]: read list and room for its label text
    // --output_read_ids 1: QNAME pool over all chromosomes, first item of every chromosome, "first of its QNAME" flags per read-list entry for the
    // (block, haplotype, BAM) segments (isf0) and for the single (variant, allele, BAM) lists (isf2)
    int read_ids; PoolD qn; const long long *qbase; const uint8_t *isf0, *isf2;
};

// strings of a block member through its record; a string of 64 KB or more (a structural variant's allele) does not fit the record's 16-bit lengths:
// such a member is marked wide and read through the pools
template <class S> __device__ __forceinline__ void put_uid(const RD &D, const MemRec &r, int64_t m, S &s) {
    if (r.wide) s.pool(D.uid, D.mem_s[m]); else s.raw(D.uid.b + r.uid_off, r.uid_len);
}
template <class S> __device__ __forceinline__ void put_rsid(const RD &D, const MemRec &r, int64_t m, S &s) {
    if (r.wide) s.pool(D.rsid, D.mem_s[m]); else s.raw(D.rsid.b + r.rsid_off, r.rsid_len);
}
template <class S> __device__ __forceinline__ void put_alle(const RD &D, const MemRec &r, int64_t m, int h, S &s) {
    if (r.wide) { const int64_t g = D.mem_s[m]; s.pool(D.alle, 2 * g + (D.v_alle[g] ^ h)); } else s.raw(D.alle.b + r.a_off[h], r.a_len[h]);
}
__device__ __forceinline__ int8_t hap_phase(const RD &D, int b, int h, uint32_t t) {        // VCF phase index of haplotype h's allele at the block's t-th variant
    const uint32_t g = D.mem_s[D.blk_mstart[b] + t];
    return D.phase_idx[2 * (int64_t)g + (D.v_alle[g] ^ h)];
}
__device__ __forceinline__ char phase_char(int8_t x) { return x < 0 ? '-' : (char)('0' + x); }

template <class S> __device__ __forceinline__ void put_stat(const RD &D, int b, S &s) {     // gwStat: int 1, 0.5, or repr(float) from the table
    const uint8_t k = D.blk_statkind[b];
    if (k == 1) s.ch('1');
    else if (k == 2) s.lit("0.5");
    else if (k == 3) s.pool(D.statx, D.blk_statidx[b]);          // a block of more known phases than the table covers: text laid out by the host during the run
    else s.pool(D.statt, D.blk_statidx[b]);
}

// ---------------------------------------------------------------------------------------------- row functions
// variant_connections.txt (:691-695); row = position in the (rank a, rank b) order of the tested pairs
struct RowConn {
    __device__ static bool big(const RD &, int64_t) { return false; }
    template <class S> __device__ static void emit(const RD &D, int64_t r, S &s) {
        const uint32_t e = D.eorder[r];
        const int a = D.va[e], b = D.vb[e];
        const int sup = D.sup[e], tot = D.tot[e];
        s.pool(D.uid, a); s.ch('\t'); s.pool(D.uid, b); s.ch('\t'); s.num(sup); s.ch('\t'); s.num(tot); s.ch('\t');
        if (sup == 0) s.ch('0');
        else if (tot - sup > 0) s.pool(D.pvt, D.e_slot[e]);
        else s.ch('1');
        s.ch('\t');
        const int8_t pa0 = D.phase_idx[2 * (int64_t)a], pb0 = D.phase_idx[2 * (int64_t)b];
        const int cis = D.cis[e], trans = D.trans[e];
        if (pa0 >= 0 && pb0 >= 0 && cis != trans) {
            const int8_t pa = cis > trans ? pa0 : D.phase_idx[2 * (int64_t)a + 1];
            s.ch(pa == pb0 ? '1' : '0');
        } else s.ch('.');
        s.ch('\n');
    }
};

// allelic_counts.txt (:737-749); row = first-appearance key
struct RowAllelic {
    __device__ static bool big(const RD &, int64_t) { return false; }
    template <class S> __device__ static void emit(const RD &D, int64_t r, S &s) {
        const int64_t g = D.key_g[r];
        const long long r0 = D.var_distinct[3 * g], r1 = D.var_distinct[3 * g + 1];
        if (r0 + r1 <= 0) return;
        s.pool(D.chromn, D.vchrom[g]); s.ch('\t'); s.num(D.pos[g]); s.ch('\t'); s.pool(D.uid, g); s.ch('\t'); s.pool(D.alle, 2 * g); s.ch('\t');
        s.pool(D.alle, 2 * g + 1); s.ch('\t'); s.num(r0); s.ch('\t'); s.num(r1); s.ch('\t'); s.num(r0 + r1); s.ch('\n');
    }
};

__device__ __forceinline__ bool single_live(const RD &D, int64_t g) {
    return (long long)D.var_count[3 * g] + D.var_count[3 * g + 1] != 0 && D.blk_of[g] < 0;
}
template <class S> __device__ __forceinline__ void put_vcf_phase(const RD &D, int64_t g, S &s, bool slash) {
    if (D.phase_idx[2 * g] >= 0) { s.num(D.phase_idx[2 * g]); s.ch('|'); s.num(D.phase_idx[2 * g + 1]); }
    else if (slash) s.lit("0/1");
    else s.lit("-|-");
}

// singleton rows of haplotypic_counts.txt (:1180-1239); row = key * nb + bam
struct RowSingleAse {
    __device__ static bool big(const RD &, int64_t) { return false; }
    template <class S> __device__ static void emit(const RD &D, int64_t r, S &s) {
        const int64_t g = D.key_g[r / D.nb]; const int b = (int)(r % D.nb);
        if (!D.unphased_vars || !single_live(D, g)) return;
        if (D.black && D.black[g]) return;
        if (D.bam_excl && D.bam_excl[b]) return;
        const long long n0 = D.nb == 1 ? D.var_distinct[3 * g] : D.single_n[(2 * g) * D.nb + b];
        const long long n1 = D.nb == 1 ? D.var_distinct[3 * g + 1] : D.single_n[(2 * g + 1) * D.nb + b];
        if (n0 + n1 <= 0) return;
        s.pool(D.chromn, D.vchrom[g]); s.ch('\t'); s.num(D.pos[g]); s.ch('\t'); s.num(D.pos[g]); s.ch('\t'); s.pool(D.uid, g);
        s.lit("\t1\t\t0\t"); s.pool(D.alle, 2 * g); s.ch('\t'); s.pool(D.alle, 2 * g + 1); s.ch('\t');
        s.num(n0); s.ch('\t'); s.num(n1); s.ch('\t'); s.num(n0 + n1); s.ch('\t');
        put_vcf_phase(D, g, s, true);
        s.lit("\t1\t");
        if (D.read_ids) {          // set(haplo_reads[allele][bam]) of :1196-1204 as QNAME strings, first-appearance order (the canonical form of the set)
            const long long q0 = D.qbase[D.vchrom[g]];
            for (int k = 0; k < 2; k++) {
                const int64_t e = (2 * g + k) * D.nb + b;
                bool first = true;
                for (uint32_t p = D.rl_start[e]; p < D.rl_start[e + 1]; p++) {
                    if (!D.isf2[p]) continue;
                    if (!first) s.ch(',');
                    first = false;
                    s.pool(D.qn, q0 + D.rl_qid[p]);
                }
                s.ch('\t');
            }
        }
        s.pool(D.maft, g); s.ch('\t'); s.pool(D.bamn, b); s.lit("\t\t\n");
    }
};

// singleton rows of haplotypes.txt; row = key
struct RowSingleHap {
    __device__ static bool big(const RD &, int64_t) { return false; }
    template <class S> __device__ static void emit(const RD &D, int64_t r, S &s) {
        const int64_t g = D.key_g[r];
        if (!D.unphased_vars || !single_live(D, g)) return;
        const long long d0 = D.var_distinct[3 * g], d1 = D.var_distinct[3 * g + 1];
        s.pool(D.chromn, D.vchrom[g]); s.ch('\t'); s.num((long long)D.pos[g] - 1); s.ch('\t'); s.num(D.pos[g]); s.lit("\t1\t1\t");
        s.pool(D.unique_ids ? D.uid : D.rsid, g); s.ch('\t'); s.pool(D.alle, 2 * g); s.ch('|'); s.pool(D.alle, 2 * g + 1); s.ch('\t');
        s.num(d0); s.ch('\t'); s.num(d1); s.ch('\t'); s.num(d0 + d1); s.lit("\t0\t0\t");
        put_vcf_phase(D, g, s, false); s.lit("\tnan\t"); put_vcf_phase(D, g, s, false); s.lit("\tnan\n");
    }
};

// haplotypes.txt, one row per block (:865-1043)
struct RowHap {
    __device__ static bool big(const RD &D, int64_t b) { return D.blk_len[b] > (uint32_t)ROW_WAVE_MIN; }      // formatted by a wave (k_row_wave_*)
    template <class S> __device__ static void emit(const RD &D, int64_t b, S &s) {
        const uint32_t m0 = D.blk_mstart[b], n = D.blk_len[b];
        const int64_t g0 = D.mem_s[m0], g1 = D.mem_s[m0 + n - 1];
        const int minpos = D.pos[g0], maxpos = D.pos[g1];
        s.pool(D.chromn, D.vchrom[g0]); s.ch('\t'); s.num(minpos); s.ch('\t'); s.num(maxpos); s.ch('\t'); s.num(maxpos - minpos); s.ch('\t');
        s.num(n); s.ch('\t');
        const MemRec *R = D.mrec + m0;
        auto all = [&](uint32_t) { return true; };
        s.join(n, ',', all, [&](uint32_t t, auto &o) { const MemRec r = R[t]; if (D.unique_ids) put_uid(D, r, m0 + t, o); else put_rsid(D, r, m0 + t, o); });
        s.ch('\t');
        for (int h = 0; h < 2; h++) {
            if (h) s.ch('|');
            s.join(n, ',', all, [&](uint32_t t, auto &o) { const MemRec r = R[t]; put_alle(D, r, m0 + t, h, o); });
        }
        const long long c0 = D.blk_cnt[2 * b], c1 = D.blk_cnt[2 * b + 1];
        s.ch('\t'); s.num(c0); s.ch('\t'); s.num(c1); s.ch('\t'); s.num(c0 + c1); s.ch('\t');
        s.num(D.blk_sup[b]); s.lit(".0\t"); s.num(D.blk_tot[b]); s.lit(".0\t");
        for (int h = 0; h < 2; h++) { if (h) s.ch('|'); s.join(n, (char)0, all, [&](uint32_t t, auto &o) { o.ch(phase_char(R[t].ph[h])); }); }
        s.ch('\t'); s.num(D.blk_conc[b]); s.ch('\t');
        const uint8_t cm = D.blk_cormode[b];
        for (int h = 0; h < 2; h++) {
            if (h) s.ch('|');
            s.join(n, (char)0, all, [&](uint32_t t, auto &o) { o.ch(cm == 0 ? phase_char(R[t].ph[h]) : (char)('0' + ((cm == 1 ? 0 : 1) ^ h))); });
        }
        s.ch('\t'); put_stat(D, (int)b, s); s.ch('\n');
    }
};

// haplotypic_counts.txt, one row per (block, BAM) (:1048-1125); the label lists at the end of the row are written by k_label_write:
// this function leaves room for them and records where each (variant, allele, BAM) list starts
struct RowAse {
    __device__ static bool big(const RD &D, int64_t r) { return D.blk_len[r / D.nb] > (uint32_t)ROW_WAVE_MIN; }
    template <class S> __device__ static void emit(const RD &D, int64_t r, S &s) {
        const int64_t b = r / D.nb; const int bb = (int)(r % D.nb);
        if (D.bam_excl && D.bam_excl[bb]) return;
        const long long ns0 = D.seg_ns[(2 * b) * D.nb + bb], ns1 = D.seg_ns[(2 * b + 1) * D.nb + bb];
        if (ns0 + ns1 <= 0) return;
        const uint32_t m0 = D.blk_mstart[b], n = D.blk_len[b];
        const int64_t g0 = D.mem_s[m0], g1 = D.mem_s[m0 + n - 1];
        s.pool(D.chromn, D.vchrom[g0]); s.ch('\t'); s.num(D.pos[g0]); s.ch('\t'); s.num(D.pos[g1]); s.ch('\t');
        const MemRec *R = D.mrec + m0;
        long long nblack = 0;
        if (D.black) for (uint32_t t = 0; t < n; t++) nblack += R[t].black ? 1 : 0;
        const long long used = (long long)n - nblack;
        auto live = [&](uint32_t t) { return R[t].black == 0; };
        s.join(n, ',', live, [&](uint32_t t, auto &o) { const MemRec r = R[t]; put_uid(D, r, m0 + t, o); });
        s.ch('\t'); s.num(used); s.ch('\t');
        if (D.black) s.join(n, ',', [&](uint32_t t) { return R[t].black != 0; }, [&](uint32_t t, auto &o) { const MemRec r = R[t]; put_uid(D, r, m0 + t, o); });
        s.ch('\t'); s.num(nblack); s.ch('\t');
        for (int h = 0; h < 2; h++) {
            s.join(n, ',', live, [&](uint32_t t, auto &o) { const MemRec r = R[t]; put_alle(D, r, m0 + t, h, o); });
            s.ch('\t');
        }
        s.num(ns0); s.ch('\t'); s.num(ns1); s.ch('\t'); s.num(ns0 + ns1); s.ch('\t');
        const uint8_t cm = D.blk_cormode[b];
        const int c00 = cm == 0 ? (int)R[0].ph[0] : (cm == 1 ? 0 : 1);
        if (c00 == 0) s.lit("0|1"); else if (c00 == 1) s.lit("1|0"); else s.lit("0/1");
        s.ch('\t'); put_stat(D, (int)b, s); s.ch('\t');
        if (D.read_ids) {          // hap_a_reads / hap_b_reads of :1120-1123: the haplotype's read set in this BAM as QNAME strings, first-appearance order.
            // A plain loop over the read-list entries (identical on every lane of a wave sink: lane 0 stores): a debugging option, not a hot row
            const long long q0 = D.qbase[D.vchrom[g0]];
            for (int h = 0; h < 2; h++) {
                const size_t lab0 = (size_t)(h * D.nb + bb) * (size_t)D.nmem + m0;
                bool first = true;
                for (uint32_t t = 0; t < n; t++) {
                    if (R[t].black) continue;
                    const uint32_t e = D.lab_e[lab0 + t];
                    for (uint32_t p = D.rl_start[e]; p < D.rl_start[e + 1]; p++) {
                        if (!D.isf0[p]) continue;
                        if (!first) s.ch(',');
                        first = false;
                        s.pool(D.qn, q0 + D.rl_qid[p]);
                    }
                }
                s.ch('\t');
            }
        }
        s.pool(D.maft, D.blk_maxmaf[b]); s.ch('\t'); s.pool(D.bamn, bb); s.ch('\t');
        for (int h = 0; h < 2; h++) {
            const size_t lab0 = (size_t)(h * D.nb + bb) * (size_t)D.nmem + m0;
            s.join(n, ';', live, [&](uint32_t t, auto &o) {
                if (std::remove_reference_t<decltype(o)>::writing) D.piece_dst[D.lab_e[lab0 + t]] = o.where();
                const uint32_t room = D.lab_skip[lab0 + t];
                if (room) o.skip(room);                             // every label is followed by one separator byte except the list's last
            });
            s.ch(h == 0 ? '\t' : '\n');
        }
    }
};

// allele_config.txt (:1160-1172); row = cfg_base[block] + i * (n - 1) + (j with i skipped)
struct RowCfg {
    __device__ static bool big(const RD &, int64_t) { return false; }
    template <class S> __device__ static void emit(const RD &D, int64_t r, S &s) {
        // largest b with cfg_base[b] <= r: the block of the row's 256-row chunk is known, and a block has >= 2 rows
        int64_t lo = D.cfg_chunk[r >> 8], hi = lo + 257 < D.nblocks ? lo + 257 : D.nblocks;
        while (hi - lo > 1) { const int64_t m = (lo + hi) >> 1; if (D.cfg_base[m] <= (unsigned long long)r) lo = m; else hi = m; }
        const int64_t b = lo;
        const uint32_t m0 = D.blk_mstart[b], n = D.blk_len[b];
        const uint32_t idx = (uint32_t)((unsigned long long)r - D.cfg_base[b]);
        const uint32_t i = idx / (n - 1); uint32_t j = idx % (n - 1);
        if (j >= i) j++;
        const MemRec xa = D.mrec[m0 + i], xb = D.mrec[m0 + j];
        put_uid(D, xa, m0 + i, s); s.ch('\t'); put_rsid(D, xa, m0 + i, s); s.ch('\t');
        put_uid(D, xb, m0 + j, s); s.ch('\t'); put_rsid(D, xb, m0 + j, s);
        if ((xa.ref_a != 0) == (xb.ref_b != 0)) s.lit("\ttrans\n"); else s.lit("\tcis\n");
    }
};

__global__ __launch_bounds__(256) void k_mem_rec(RD D, MemRec *out) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= D.nmem) return;
    const int64_t g = D.mem_s[m];
    const int va = D.v_alle[g];
    MemRec r;
    uint32_t longest = D.uid.len(g) > D.rsid.len(g) ? D.uid.len(g) : D.rsid.len(g);
    r.uid_off = D.uid.off[g]; r.uid_len = (uint16_t)D.uid.len(g); r.rsid_off = D.rsid.off[g]; r.rsid_len = (uint16_t)D.rsid.len(g);
    for (int h = 0; h < 2; h++) {
        const int64_t a = 2 * g + (va ^ h);
        r.a_off[h] = D.alle.off[a]; r.a_len[h] = (uint16_t)D.alle.len(a); r.ph[h] = D.phase_idx[a];
        longest = D.alle.len(a) > longest ? D.alle.len(a) : longest;
    }
    r.wide = longest >= 65536u ? 1 : 0;
    r.black = (D.black && D.black[g]) ? 1 : 0;
    r.ref_a = D.is_ref[2 * g + va]; r.ref_b = D.is_ref[2 * g + (va ^ 1)];
    r.pad[0] = r.pad[1] = 0;
    out[m] = r;
}
// read list of (haplotype, BAM, block member) -- as soon as the blocks are known: the read-set kernels walk a block's members through it
__global__ __launch_bounds__(256) void k_lab_e(int64_t nmem, int nb, const uint32_t *mem_s, const uint8_t *v_alle, uint32_t *lab_e) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nmem * 2 * nb) return;
    const int64_t m = i % nmem; const int hb = (int)(i / nmem), h = hb / nb, bb = hb % nb;
    const int64_t g = mem_s[m];
    lab_e[i] = (uint32_t)((2 * g + (v_alle[g] ^ h)) * nb + bb);
}
// ... and the room its label text takes in the member's row of haplotypic_counts
__global__ __launch_bounds__(256) void k_mem_lab(RD D, uint32_t *lab_e, uint32_t *lab_skip) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= D.nmem * 2 * D.nb) return;
    const uint32_t e = lab_e[i];
    const uint32_t lo = D.rl_start[e], hi = D.rl_start[e + 1];
    lab_skip[i] = hi > lo ? D.its[hi] - D.its[lo] - 1u : 0u;
}
// block of the first row of every 256-row chunk of allele_config (one search of the whole block table per chunk instead of per row)
__global__ __launch_bounds__(256) void k_cfg_chunks(int64_t nchunks, int64_t nblocks, const unsigned long long *cfg_base, uint32_t *cfg_chunk) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunks) return;
    const unsigned long long r = (unsigned long long)c << 8;
    int64_t lo = 0, hi = nblocks;
    while (hi - lo > 1) { const int64_t m = (lo + hi) >> 1; if (cfg_base[m] <= r) lo = m; else hi = m; }
    cfg_chunk[c] = (uint32_t)lo;
}

// ---- allele_config.txt without a length per row and a scan over its 11 M rows: the byte offset of row (i, j) of a block follows from three
// prefix arrays over the block's members.  A row is uid_i \t rsid_i \t uid_j \t rsid_j + "\ttrans\n" (7) or "\tcis\n" (5); with L = len(uid) +
// len(rsid), A = the member's haplotype-A allele is the reference, B = its haplotype-B allele is, a row is "trans" iff A_i == B_j:
//   len(i, j)  = L_i + L_j + 3 + (A_i == B_j ? 7 : 5)
//   rows (i, j') before (i, j) in group i: cnt = j - [i < j] of them; their bytes = cnt (L_i + 8) + (PL[j] - [i < j] L_i) + 2 (E - [i < j and B_i == A_i]),
//                                           E = A_i ? PB[j] : j - PB[j]   (PL / PB: prefix of L / of B over the members before j)
//   group i as a whole: S_i = (n - 1)(L_i + 8) + (TL - L_i) + 2 ((A_i ? NB : n - NB) - [B_i == A_i]);  PS[i] = S_0 + .. + S_{i-1};  block bytes = PS[n]
__device__ __forceinline__ uint32_t cfg_member_len(const RD &D, const MemRec &r, int64_t m) {
    if (!r.wide) return (uint32_t)r.uid_len + (uint32_t)r.rsid_len;
    const int64_t g = D.mem_s[m];
    return D.uid.len(g) + D.rsid.len(g);
}
__global__ __launch_bounds__(256) void k_cfg_prefix(RD D, int64_t nblocks, uint32_t *pl, uint32_t *pb, unsigned long long *ps, unsigned long long *blk_bytes) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t m0 = D.blk_mstart[b], n = D.blk_len[b];
    if (n > (uint32_t)ROW_WAVE_MIN) return;                  // a wave each (k_cfg_prefix_wave): one thread walking hundreds of members twice was the launch's tail
    unsigned long long TL = 0; uint32_t NB = 0;
    for (uint32_t t = 0; t < n; t++) {
        const MemRec r = D.mrec[m0 + t];
        pl[m0 + t] = (uint32_t)TL; pb[m0 + t] = NB;
        TL += cfg_member_len(D, r, m0 + t); NB += r.ref_b != 0 ? 1u : 0u;
    }
    unsigned long long run = 0;
    for (uint32_t t = 0; t < n; t++) {
        const MemRec r = D.mrec[m0 + t];
        const unsigned long long L = cfg_member_len(D, r, m0 + t);
        const bool A = r.ref_a != 0, B = r.ref_b != 0;
        ps[m0 + t] = run;
        run += (unsigned long long)(n - 1) * (L + 8ull) + (TL - L) + 2ull * ((unsigned long long)(A ? NB : n - NB) - (B == A ? 1ull : 0ull));
    }
    blk_bytes[b] = run;
}
__device__ __forceinline__ unsigned long long wave_incl_u64(unsigned long long x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(x, d); if (lane >= d) x += y; }
    return x;
}
// the same for the blocks of more than ROW_WAVE_MIN members, one wave per block (the list k_big_blocks made): lanes take members 64 at a time, prefixes by wave scans
__global__ __launch_bounds__(64) void k_cfg_prefix_wave(RD D, const uint32_t *big_blk, uint32_t *pl, uint32_t *pb, unsigned long long *ps, unsigned long long *blk_bytes) {
    const int64_t b = big_blk[blockIdx.x];
    const int lane = threadIdx.x;
    const uint32_t m0 = D.blk_mstart[b], n = D.blk_len[b];
    unsigned long long TL = 0, NB = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += 64) {
        const uint32_t t = t0 + (uint32_t)lane;
        unsigned long long L = 0, B = 0;
        if (t < n) { const MemRec r = D.mrec[m0 + t]; L = cfg_member_len(D, r, m0 + t); B = r.ref_b != 0 ? 1ull : 0ull; }
        const unsigned long long iL = wave_incl_u64(L, lane), iB = wave_incl_u64(B, lane);
        if (t < n) { pl[m0 + t] = (uint32_t)(TL + iL - L); pb[m0 + t] = (uint32_t)(NB + iB - B); }
        TL += __shfl(iL, 63); NB += __shfl(iB, 63);
    }
    unsigned long long run = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += 64) {
        const uint32_t t = t0 + (uint32_t)lane;
        unsigned long long S = 0;
        if (t < n) {
            const MemRec r = D.mrec[m0 + t];
            const unsigned long long L = cfg_member_len(D, r, m0 + t);
            const bool A = r.ref_a != 0, B = r.ref_b != 0;
            S = (unsigned long long)(n - 1) * (L + 8ull) + (TL - L) + 2ull * ((A ? NB : (unsigned long long)n - NB) - (B == A ? 1ull : 0ull));
        }
        const unsigned long long iS = wave_incl_u64(S, lane);
        if (t < n) ps[m0 + t] = run + iS - S;
        run += __shfl(iS, 63);
    }
    if (lane == 0) blk_bytes[b] = run;
}
// the writer: ROWS consecutive rows per workgroup, formatted into LDS at their place in the file and copied out with aligned 16-byte stores (as k_row_write)
template <int ROWS, int STAGE> __global__ __launch_bounds__(ROWS) void k_cfg_write(RD D, int64_t nrows, char *out) {
    __shared__ __attribute__((aligned(16))) char s_buf[STAGE];
    __shared__ unsigned long long s_b[2];
    const int64_t r0 = (int64_t)blockIdx.x * ROWS;
    const int64_t r1 = r0 + ROWS < nrows ? r0 + ROWS : nrows;
    const int64_t r = r0 + threadIdx.x;
    const bool live = r < r1;
    unsigned long long off = 0; uint32_t len = 0, ma = 0, mb = 0;
    MemRec xa, xb;
    bool trans = false;
    if (live) {
        int64_t lo = D.cfg_chunk[r >> 8], hi = lo + 257 < D.nblocks ? lo + 257 : D.nblocks;
        while (hi - lo > 1) { const int64_t m = (lo + hi) >> 1; if (D.cfg_base[m] <= (unsigned long long)r) lo = m; else hi = m; }
        const int64_t b = lo;
        const uint32_t m0 = D.blk_mstart[b], n = D.blk_len[b];
        const uint32_t idx = (uint32_t)((unsigned long long)r - D.cfg_base[b]);
        const uint32_t i = idx / (n - 1); uint32_t j = idx % (n - 1);
        if (j >= i) j++;
        ma = m0 + i; mb = m0 + j;
        xa = D.mrec[ma]; xb = D.mrec[mb];
        const unsigned long long La = cfg_member_len(D, xa, ma), Lb = cfg_member_len(D, xb, mb);
        const bool A = xa.ref_a != 0, Bi = xa.ref_b != 0, Bj = xb.ref_b != 0, lt = i < j;
        const unsigned long long cnt = (unsigned long long)j - (lt ? 1ull : 0ull);
        const unsigned long long PLj = D.cfg_pl[mb], PBj = D.cfg_pb[mb];
        const unsigned long long E = A ? PBj : (unsigned long long)j - PBj;
        const unsigned long long W = cnt * (La + 8ull) + (PLj - (lt ? La : 0ull)) + 2ull * (E - ((lt && Bi == A) ? 1ull : 0ull));
        trans = A == Bj;
        off = D.cfg_bbase[b] + D.cfg_ps[ma] + W;
        len = (uint32_t)(La + Lb + 3ull + (trans ? 7ull : 5ull));
        if (threadIdx.x == 0) s_b[0] = off;
        if (r == r1 - 1) s_b[1] = off + len;
    }
    __syncthreads();
    const unsigned long long b0 = s_b[0], b1 = s_b[1];
    const unsigned mis = (unsigned)((unsigned long long)(out + b0) & 15ull);
    const bool staged = (b1 - b0) + mis <= (unsigned long long)STAGE;
    if (live) {
        SWrite s;
        s.p0 = s.p = staged ? s_buf + mis + (unsigned)(off - b0) : out + off; s.base = off;
        put_uid(D, xa, ma, s); s.ch('\t'); put_rsid(D, xa, ma, s); s.ch('\t');
        put_uid(D, xb, mb, s); s.ch('\t'); put_rsid(D, xb, mb, s);
        if (trans) s.lit("\ttrans\n"); else s.lit("\tcis\n");
    }
    if (!staged) return;
    __syncthreads();
    const unsigned total = mis + (unsigned)(b1 - b0);
    char *g = out + b0 - mis;                       // 16-byte aligned
    for (unsigned c = threadIdx.x * 16u; c < total; c += ROWS * 16u) {
        if (c >= mis && c + 16u <= total) *(uint4 *)(g + c) = *(const uint4 *)(s_buf + c);
        else for (unsigned k = c < mis ? mis : c; k < c + 16u && k < total; k++) g[k] = s_buf[k];
    }
}
template <class ROW> __global__ __launch_bounds__(256) void k_row_len(RD D, int64_t nrows, uint32_t *len) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows || ROW::big(D, r)) return;
    SCount s;
    ROW::emit(D, r, s);
    len[r] = s.n;
}
// rows of big blocks: one wave per row (block index from the list of big blocks; per_blk rows per block)
template <class ROW> __global__ __launch_bounds__(64) void k_row_wave_len(RD D, const uint32_t *big_blk, int per_blk, uint32_t *len) {
    const int64_t r = (int64_t)big_blk[blockIdx.x / per_blk] * per_blk + (blockIdx.x % per_blk);
    WCount s;
    ROW::emit(D, r, s);
    if (threadIdx.x == 0) len[r] = s.n;
}
template <class ROW> __global__ __launch_bounds__(64) void k_row_wave_write(RD D, const uint32_t *big_blk, int per_blk, const unsigned long long *off, char *out) {
    const int64_t r = (int64_t)big_blk[blockIdx.x / per_blk] * per_blk + (blockIdx.x % per_blk);
    if (off[r + 1] == off[r]) return;
    WWrite s; s.p0 = s.p = out + off[r]; s.base = off[r];
    ROW::emit(D, r, s);
}
__global__ __launch_bounds__(256) void k_big_blocks(int64_t nblocks, const uint32_t *blk_len, uint32_t *big_blk, uint32_t *count) {
    __shared__ uint32_t s_n, s_base;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const bool big = b < nblocks && blk_len[b] > (uint32_t)ROW_WAVE_MIN;
    uint32_t at = 0;
    if (big) at = atomicAdd(&s_n, 1u);
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_n ? atomicAdd(count, s_n) : 0u;
    __syncthreads();
    if (big) big_blk[s_base + at] = (uint32_t)b;
}
// One workgroup writes ROWS consecutive rows.  The rows are formatted into LDS at the position they have in the file (shifted so that LDS
// byte i and file byte i agree modulo 16) and copied out with aligned 16-byte stores: a thread writing its row byte by byte to global memory
// reaches ~0.2 TB/s (680 MB of allele_config rows: 3 ms), the staged copy streams.  A row range that does not fit the stage (rows with
// thousands of read labels) is written directly.
// The stage is sized per file (rows x typical row length): it decides how many workgroups a CU holds, and the row functions are chains of
// dependent loads that only occupancy hides.
#ifndef PHZ_HAP_ROWS
#define PHZ_HAP_ROWS 128
#endif
#ifndef PHZ_HAP_STAGE
#define PHZ_HAP_STAGE (32 * 1024)
#endif
#ifndef PHZ_ASE_ROWS
#define PHZ_ASE_ROWS 64
#endif
#ifndef PHZ_ASE_STAGE
#define PHZ_ASE_STAGE (16 * 1024)
#endif
template <class ROW, int ROWS, int ROW_STAGE> __global__ __launch_bounds__(ROWS) void k_row_write(RD D, int64_t nrows, const unsigned long long *off, char *out) {
    __shared__ __attribute__((aligned(16))) char s_buf[ROW_STAGE];
    const int64_t r0 = (int64_t)blockIdx.x * ROWS;
    const int64_t r1 = r0 + ROWS < nrows ? r0 + ROWS : nrows;
    const int64_t r = r0 + threadIdx.x;
    const unsigned long long b0 = off[r0], b1 = off[r1];
    if (b1 == b0) return;
    const unsigned mis = (unsigned)((unsigned long long)(out + b0) & 15ull);
    if ((b1 - b0) + mis <= (unsigned long long)ROW_STAGE) {
        if (r < r1 && off[r + 1] > off[r] && !ROW::big(D, r)) {      // (a big row's bytes are written by k_row_wave_write afterwards)
            SWrite s; s.p0 = s.p = s_buf + mis + (unsigned)(off[r] - b0); s.base = off[r];
            ROW::emit(D, r, s);
        }
        __syncthreads();
        const unsigned total = mis + (unsigned)(b1 - b0);
        char *g = out + b0 - mis;                       // 16-byte aligned
        for (unsigned c = threadIdx.x * 16u; c < total; c += ROWS * 16u) {
            if (c >= mis && c + 16u <= total) *(uint4 *)(g + c) = *(const uint4 *)(s_buf + c);
            else for (unsigned k = c < mis ? mis : c; k < c + 16u && k < total; k++) g[k] = s_buf[k];
        }
    } else if (r < r1 && off[r + 1] > off[r] && !ROW::big(D, r)) {
        SWrite s; s.p0 = s.p = out + off[r]; s.base = off[r];
        ROW::emit(D, r, s);
    }
}

// label text of haplotypic_counts: one thread per read-list entry; width of a label's text + its separator (the input transform of the scan that places them)
struct LabelWidth { template <class T> __device__ static __forceinline__ T f(T x) { return (T)((uint32_t)ndigits((unsigned long long)x) + 1u); } };
__global__ __launch_bounds__(256) void k_label_write(RD D, int64_t n_rl, char *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rl) return;
    const uint32_t e = D.rl_list[i];
    const unsigned long long dst = D.piece_dst[e];
    if (dst == NONE64) return;
    const uint32_t lo = D.rl_start[e], hi = D.rl_start[e + 1];
    char *p = out + dst + (D.its[i] - D.its[lo]);
    unsigned long long u = D.labels[i];
    const int d = ndigits(u);
    for (int k = d - 1; k >= 0; k--) { p[k] = (char)('0' + (int)(u % 10ull)); u /= 10ull; }
    if (i + 1 < (int64_t)hi) p[d] = ',';
}

// ---------------------------------------------------------------------------------------------- pair test bookkeeping
__device__ __forceinline__ uint32_t ps_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (uint32_t)k;
}
__device__ __forceinline__ bool pair_tested(const uint8_t *linked, const int32_t *sup, const int32_t *tot, int64_t e) {
    return linked[e] && sup[e] > 0 && tot[e] - sup[e] > 0;
}
// flags[0]: table full
__global__ __launch_bounds__(256) void k_pair_keys(int64_t ne, const uint8_t *linked, const int32_t *sup, const int32_t *tot, unsigned long long *hkeys, uint32_t mask, uint32_t *flags) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= ne || !pair_tested(linked, sup, tot, e)) return;
    const unsigned long long key = ((unsigned long long)(uint32_t)tot[e] << 32) | (uint32_t)sup[e];
    uint32_t s = ps_hash(key) & mask;
    // a probe sequence this long means the table is (nearly) full: the host redoes the stage with a larger one (phz_rowsdev_set_pair_slots)
    for (int t = 0; t < 2048; t++) {
        const unsigned long long cur = hkeys[s];
        if (cur == key) return;
        if (cur == PS_EMPTY) {
            const unsigned long long prev = atomicCAS(&hkeys[s], PS_EMPTY, key);
            if (prev == PS_EMPTY || prev == key) return;
        }
        s = (s + 1) & mask;
        if ((t & 63) == 63 && ((volatile uint32_t *)flags)[0]) return;          // somebody has given up already
    }
    atomicOr(&flags[0], 1u);
}
// verdict per pair (:1645-1652, :686-700): p = 0 without supporting reads, 1 without conflicting ones, else the table; counters[0] linked pairs,
// [1] linked pairs dropped
__global__ __launch_bounds__(256) void k_keep(int64_t ne, const uint8_t *linked, const int32_t *sup, const int32_t *tot, const unsigned long long *hkeys, uint32_t mask,
                                              const double *slot_pv, double threshold, uint8_t *keep, uint32_t *e_slot, uint32_t *deg, const int32_t *ea,
                                              const int32_t *eb, unsigned long long *counters) {
    __shared__ unsigned int s_c[8];
    unsigned int n_linked = 0, n_dropped = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < ne; e += (int64_t)gridDim.x * 256) {
        uint32_t slot = NONE32;
        bool kept = false;
        if (linked[e]) {
            double pv = 1.0;
            if (sup[e] == 0) pv = 0.0;
            else if (tot[e] - sup[e] > 0) {
                const unsigned long long key = ((unsigned long long)(uint32_t)tot[e] << 32) | (uint32_t)sup[e];
                uint32_t s = ps_hash(key) & mask;
                while (hkeys[s] != key) s = (s + 1) & mask;
                slot = s; pv = slot_pv[s];
            }
            kept = !(pv < threshold);
            n_linked++; n_dropped += kept ? 0u : 1u;
        }
        keep[e] = kept ? 1 : 0; e_slot[e] = slot;
        if (kept) { deg[ea[e]] = 1u; deg[eb[e]] = 1u; }          // membership flag of the surviving graph (any writer stores the same value)
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { n_linked += __shfl_xor(n_linked, d); n_dropped += __shfl_xor(n_dropped, d); }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = n_linked; s_c[4 + (threadIdx.x >> 6)] = n_dropped; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int a = s_c[0] + s_c[1] + s_c[2] + s_c[3], b = s_c[4] + s_c[5] + s_c[6] + s_c[7];
        if (a) atomicAdd(&counters[0], (unsigned long long)a);
        if (b) atomicAdd(&counters[1], (unsigned long long)b);
    }
}

// ---------------------------------------------------------------------------------------------- ordering stage
__global__ __launch_bounds__(256) void k_iota_rank(int64_t nv, const unsigned long long *var_rank, unsigned long long *key, uint32_t *val) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < nv) { const unsigned long long r = var_rank[v]; key[v] = (r & 0xFFFFFFFF00000000ull) | (uint32_t)((uint32_t)r - (uint32_t)(r >> 32)); val[v] = (uint32_t)v; }   // (first, gap)
}
__global__ __launch_bounds__(256) void k_invert(int64_t n, const uint32_t *perm, uint32_t *inv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[perm[i]] = (uint32_t)i;
}
// pairs oriented by first appearance (:667-678 enumerates from the earlier key); sort key = (rank index of a, rank index of b), untested pairs last
__global__ __launch_bounds__(256) void k_edge_keys(int64_t ne, const uint8_t *linked, const int32_t *ea, const int32_t *eb, const uint32_t *ridx, int32_t *va, int32_t *vb,
                                                   unsigned long long *key, uint32_t *val) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    int a = ea[e], b = eb[e];
    val[e] = (uint32_t)e;
    if (!linked[e]) { va[e] = a; vb[e] = b; key[e] = ~0ull; return; }
    if (ridx[b] < ridx[a]) { const int t = a; a = b; b = t; }
    va[e] = a; vb[e] = b;
    key[e] = ((unsigned long long)ridx[a] << 32) | ridx[b];
}
__global__ __launch_bounds__(256) void k_flag_members(int64_t nv, const uint32_t *deg, const int32_t *label, uint32_t *is_mem, uint32_t *is_root) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const bool m = deg[v] > 0;
    is_mem[v] = m ? 1u : 0u; is_root[v] = (m && label[v] == (int32_t)v) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_compact_members(int64_t nv, const uint32_t *deg, const uint32_t *mem_pos, const int32_t *label, uint32_t *key, uint32_t *val) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nv || deg[v] == 0) return;
    key[mem_pos[v]] = (uint32_t)label[v]; val[mem_pos[v]] = (uint32_t)v;
}
// group starts of a sorted key array: start[id_of[key]] = first position of the key
__global__ __launch_bounds__(256) void k_group_starts(int64_t n, const uint32_t *sorted_key, const uint32_t *id_of, uint32_t *start) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    if (t == 0 || sorted_key[t] != sorted_key[t - 1]) start[id_of ? id_of[sorted_key[t]] : sorted_key[t]] = (uint32_t)t;
}
__global__ __launch_bounds__(256) void k_comp_min(int64_t ncomp, const uint32_t *cstart, const uint32_t *mem_s, const uint32_t *ridx, uint32_t *key, uint32_t *val) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncomp) return;
    uint32_t m = NONE32;
    for (uint32_t t = cstart[c]; t < cstart[c + 1]; t++) m = min(m, ridx[mem_s[t]]);
    key[c] = m; val[c] = (uint32_t)c;
}
__global__ __launch_bounds__(256) void k_flag_u8(int64_t n, const uint8_t *f, uint32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = f[i] ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_compact_kept(int64_t ne, const uint8_t *keep, const uint32_t *kpos, const int32_t *ea, const int32_t *label, const uint32_t *cid,
                                                      uint32_t *key, uint32_t *val) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= ne || !keep[e]) return;
    key[kpos[e]] = cid[label[ea[e]]]; val[kpos[e]] = (uint32_t)e;
}
struct ShardTab { const long long *lo, *hi; const int32_t *bam; int n; };
// first-appearance keys (rule 2): covered variants by (BAM of the first kept line, line)
// ... and, on the way, the largest distance between the two halves of a connectivity-map rank (first ref/alt line of the QNAME, the variant's first
// linked line in it: a few hundred lines apart at most in practice), which decides how many radix passes the rank sort needs for its low half
// the same key in 32 bits when (first line, gap) fit: first << gap_bits | gap (a 32-bit sort moves half the bytes per pass and needs fewer passes)
__global__ __launch_bounds__(256) void k_iota_rank32(int64_t nv, const unsigned long long *var_rank, int gap_bits, uint32_t line_mask, uint32_t *key, uint32_t *val) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < nv) { const unsigned long long r = var_rank[v]; key[v] = (((uint32_t)(r >> 32) & line_mask) << gap_bits) | (uint32_t)((uint32_t)r - (uint32_t)(r >> 32)); val[v] = (uint32_t)v; }
}
__global__ __launch_bounds__(256) void k_flag_keys(int64_t nv, const long long *var_first, uint32_t *is_key, const unsigned long long *var_rank, unsigned long long *max_gap) {
    __shared__ uint32_t s_m[4];
    uint32_t m = 0;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256) {
        is_key[v] = var_first[v] >= 0 ? 1u : 0u;
        const unsigned long long r = var_rank[v];
        const uint32_t gap = (uint32_t)r - (uint32_t)(r >> 32);
        m = gap > m ? gap : m;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t y = __shfl_xor(m, d); m = y > m ? y : m; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) m = s_m[w] > m ? s_m[w] : m;
        if (m) atomicMax(max_gap, (unsigned long long)m);
    }
}
__global__ __launch_bounds__(256) void k_compact_keys(int64_t nv, const long long *var_first, const uint32_t *kpos, ShardTab T, unsigned long long *key, uint32_t *val) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const long long f = var_first[v];
    if (f < 0) return;
    int bam = 0;
    for (int t = 0; t < T.n; t++) if (f >= T.lo[t] && f < T.hi[t]) bam = T.bam[t];
    key[kpos[v]] = ((unsigned long long)(uint32_t)bam << 32) | (unsigned long long)f;
    val[kpos[v]] = (uint32_t)v;
}
// Rows of every file are chromosome-major (BAM x chromosome-major for the key files): the rows of a segment are found from the first row of
// each segment (one thread per row compares with its predecessor; same-address atomics on two dozen counters cost 15 ms per genome)
__global__ __launch_bounds__(256) void k_conn_starts(int64_t n_linked, const uint32_t *eorder, const int32_t *va, const uint16_t *vchrom, uint32_t *start) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_linked) return;
    const uint32_t s = vchrom[va[eorder[r]]];
    if (r == 0 || vchrom[va[eorder[r - 1]]] != s) start[s] = (uint32_t)r;
}
// (BAM, first line) in 32 bits when they fit: bam << line_bits | line
__global__ __launch_bounds__(256) void k_compact_keys32(int64_t nv, const long long *var_first, const uint32_t *kpos, ShardTab T, int line_bits, uint32_t *key, uint32_t *val) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const long long f = var_first[v];
    if (f < 0) return;
    int bam = 0;
    for (int t = 0; t < T.n; t++) if (f >= T.lo[t] && f < T.hi[t]) bam = T.bam[t];
    key[kpos[v]] = ((uint32_t)bam << line_bits) | (uint32_t)f;
    val[kpos[v]] = (uint32_t)v;
}
__global__ __launch_bounds__(256) void k_key_starts32(int64_t nkeys, const uint32_t *key_sorted, int line_bits, const uint32_t *key_g, const uint16_t *vchrom, int nchrom, uint32_t *start) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nkeys) return;
    const uint32_t s = (key_sorted[r] >> line_bits) * (uint32_t)nchrom + vchrom[key_g[r]];
    if (r == 0 || (key_sorted[r - 1] >> line_bits) * (uint32_t)nchrom + vchrom[key_g[r - 1]] != s) start[s] = (uint32_t)r;
}
__global__ __launch_bounds__(256) void k_key_starts(int64_t nkeys, const unsigned long long *key_sorted, const uint32_t *key_g, const uint16_t *vchrom, int nchrom, uint32_t *start) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nkeys) return;
    const uint32_t s = (uint32_t)(key_sorted[r] >> 32) * (uint32_t)nchrom + vchrom[key_g[r]];
    if (r == 0 || (uint32_t)(key_sorted[r - 1] >> 32) * (uint32_t)nchrom + vchrom[key_g[r - 1]] != s) start[s] = (uint32_t)r;
}
__global__ __launch_bounds__(256) void k_block_starts(int64_t nblocks, const uint32_t *blk_mstart, const uint32_t *mem_s, const uint16_t *vchrom, uint32_t *start) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t s = vchrom[mem_s[blk_mstart[b]]];
    if (b == 0 || vchrom[mem_s[blk_mstart[b - 1]]] != s) start[s] = (uint32_t)b;
}
// first rows -> row counts per segment (a segment without rows starts where its successor starts)
__global__ void k_starts_to_counts(uint32_t *start, int nseg, uint32_t total, uint32_t *count) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t next = total;
    for (int s = nseg - 1; s >= 0; s--) {
        if (start[s] == NONE32) start[s] = next;
        count[s] = next - start[s];
        next = start[s];
    }
}
// per chromosome: blocks, block variants, allele_config rows, from the chromosomes' first blocks
__global__ void k_block_counts(const uint32_t *start, const uint32_t *count, int nchrom, const uint32_t *blk_voff, const unsigned long long *cfg_base, uint32_t *blkvars,
                               unsigned long long *cfgrows) {
    if (threadIdx.x || blockIdx.x) return;
    for (int c = 0; c < nchrom; c++) {
        const uint32_t b0 = start[c], b1 = b0 + count[c];
        blkvars[c] = blk_voff[b1] - blk_voff[b0]; cfgrows[c] = cfg_base[b1] - cfg_base[b0];
    }
}
// (few workgroups, one same-address atomic each: one per wave of 64 elements was 35 us for 186,000 blocks)
__global__ __launch_bounds__(256) void k_max_u32(const uint32_t *x, int64_t n, unsigned long long *out) {
    __shared__ uint32_t s_m[4];
    uint32_t v = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const uint32_t y = x[i]; v = y > v ? y : v; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) v = s_m[w] > v ? s_m[w] : v;
        if (v) atomicMax(out, (unsigned long long)v);
    }
}

// ---------------------------------------------------------------------------------------------- block phasing (phase_v3)
struct PH {
    const uint32_t *cstart, *mem_s, *estart, *ekeep, *eloc;
    const int32_t *ea, *eb, *cfgv;
    uint8_t *alle_of; int32_t *sub_of; uint32_t *nsub;
    uint32_t *complex_list, *exc_list; uint32_t *counters;       // [0] complex components, [1] exceptions, [2] unsupported (the reference loops forever / 2^30 configurations)
    int max_block_size;
};

// component-local indices of every kept pair: eloc[t] = i | j << 12 | (configuration + 1) << 24 (NONE32 for components of 4096+ variants)
__global__ __launch_bounds__(256) void k_edge_local(int64_t nkeep, int64_t ncomp, PH P, uint32_t *eloc) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nkeep) return;
    int64_t lo = 0, hi = ncomp;                            // component of the pair: largest c with estart[c] <= t
    while (hi - lo > 1) { const int64_t m = (lo + hi) >> 1; if (P.estart[m] <= (uint32_t)t) lo = m; else hi = m; }
    const uint32_t m0 = P.cstart[lo], n = P.cstart[lo + 1] - m0;
    if (n >= 4096) { eloc[t] = NONE32; return; }
    const uint32_t e = P.ekeep[t];
    const uint32_t ga = (uint32_t)P.ea[e], gb = (uint32_t)P.eb[e];
    uint32_t a = 0, b = n;
    while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (P.mem_s[m0 + m] <= ga) a = m; else b = m; }
    const uint32_t i = a;
    a = 0; b = n;
    while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (P.mem_s[m0 + m] <= gb) a = m; else b = m; }
    eloc[t] = i | (a << 12) | ((uint32_t)(P.cfgv[e] + 1) << 24);
}

// One thread per component: the cases resolve_phase (:2172-2207) settles.  Two variants joined by one pair with a verdict: the flood goes from
// (variant 0, allele 0) to (variant 1, allele 0) for a same-configuration pair and to (variant 1, allele 1) for an opposite one, block "00" /
// "01".  Up to 32 variants: the flood over the allele graph as a 64-bit mask; accepted when it reaches exactly one allele of every variant.
// Everything else (conflicts, larger components, the quirk cases of the string construction) goes to k_phase_general.
__global__ __launch_bounds__(256) void k_phase_pair(int64_t ncomp, PH P) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncomp) return;
    const uint32_t m0 = P.cstart[c], n = P.cstart[c + 1] - m0, e0 = P.estart[c], ne = P.estart[c + 1] - e0;
    if (n == 2 && ne == 1) {
        const int k = P.cfgv[P.ekeep[e0]];
        if (k == 0 || k == 1) {
            P.alle_of[m0] = 0; P.alle_of[m0 + 1] = (uint8_t)k; P.sub_of[m0] = 0; P.sub_of[m0 + 1] = 0; P.nsub[c] = 1;
            return;
        }
    }
    if (n <= 32 && ne <= 16) {            // (more pairs than that: a wave is quicker than one thread re-reading them until nothing changes)
        unsigned long long reach = 1ull;
        for (bool changed = true; changed;) {
            changed = false;
            for (uint32_t t = e0; t < e0 + ne; t++) {
                const uint32_t x = P.eloc[t];
                const int k = (int)(x >> 24) - 1;
                if (k < 0) continue;
                const uint32_t i = x & 0xFFFu, j = (x >> 12) & 0xFFFu;
                for (uint32_t a = 0; a < 2; a++) {
                    const uint32_t u = 2 * i + a, w = 2 * j + (a ^ (uint32_t)k);
                    const bool bu = (reach >> u) & 1ull, bw = (reach >> w) & 1ull;
                    if (bu && !bw) { reach |= 1ull << w; changed = true; }
                    else if (bw && !bu) { reach |= 1ull << u; changed = true; }
                }
            }
        }
        const unsigned long long even = n == 32 ? 0x5555555555555555ull : (0x5555555555555555ull & ((1ull << (2 * n)) - 1ull));
        const unsigned long long lo = reach & even, hi = (reach >> 1) & even;
        if ((lo ^ hi) == even) {                              // exactly one allele of every variant: |reach| = n, the string has n characters
            for (uint32_t t = 0; t < n; t++) { P.alle_of[m0 + t] = (uint8_t)((hi >> (2 * t)) & 1ull); P.sub_of[m0 + t] = 0; }
            P.nsub[c] = 1;
            return;
        }
    }
    P.complex_list[atomicAdd(&P.counters[0], 1u)] = (uint32_t)c;
}

__device__ __forceinline__ char flipc(char c) { return c == '-' ? '-' : (c == '0' ? '1' : '0'); }

// One wave per component.  Lane-parallel: loading the pairs, the flood fills, the 2^(n-1) brute force; everything sequential in the
// reference (weak-point selection, stitching) runs on lane 0 between wave barriers.
__device__ void phase_general_one(const PH &P, const uint32_t c) {
    __shared__ uint8_t s_i[PH_EMAX], s_j[PH_EMAX];
    __shared__ int8_t s_k[PH_EMAX];
    __shared__ uint32_t s_m[2][32];                                        // pairs of the fragment under brute force as bit masks by index distance
    __shared__ uint8_t s_mark[2 * PH_NMAX + 2];
    __shared__ int32_t s_weak[PH_NMAX + 4];
    __shared__ uint16_t s_bounds[PH_NMAX + 4];
    __shared__ char s_p0[PH_NMAX + 4];                                     // configuration of haplotype A per fragment, concatenated
    __shared__ uint16_t s_p0off[PH_NMAX + 4];
    __shared__ char s_cur[4 * PH_NMAX], s_cand[4 * PH_NMAX], s_fin[4 * PH_NMAX];
    __shared__ uint16_t s_finlen[PH_NMAX + 4];
    __shared__ int s_nf, s_status, s_nfin, s_st[8];
    __shared__ int32_t s_dp[2][1 << PH_DP_MAXD];                          // dynamic programme over the last D alleles: best score, number of ways (capped at 2),
    __shared__ uint8_t s_dc[2][1 << PH_DP_MAXD];
    __shared__ uint32_t s_ch[PH_BRUTE_MAX][(1 << PH_DP_MAXD) / 32];         // and the predecessor taken, one bit per (step, state)
    const int lane = threadIdx.x;
    const uint32_t m0 = P.cstart[c], e0 = P.estart[c];
    const int n = (int)(P.cstart[c + 1] - m0), E = (int)(P.estart[c + 1] - e0);
    if (n > PH_NMAX || E > PH_EMAX) {
        if (lane == 0) P.exc_list[atomicAdd(&P.counters[1], 1u)] = c;
        return;
    }
    for (int t = lane; t < E; t += 64) {
        const uint32_t x = P.eloc[e0 + t];
        s_i[t] = (uint8_t)(x & 0xFFFu); s_j[t] = (uint8_t)((x >> 12) & 0xFFFu); s_k[t] = (int8_t)((int)(x >> 24) - 1);
    }
    if (lane == 0) { s_status = 0; s_nfin = 0; }
    __syncthreads();

    // flood fill from (variant lo, allele 0) over the pairs inside [lo, hi) (all pairs when lo = 0, hi = n): resolve_phase :2172-2207.
    // Returns the number of allele nodes reached (the reference accepts iff it equals hi - lo); marks stay in s_mark.
    auto flood = [&](int lo, int hi) -> int {
        for (int t = lane; t < 2 * n; t += 64) s_mark[t] = 0;
        __syncthreads();
        if (lane == 0) s_mark[2 * lo] = 1;
        __syncthreads();
        for (;;) {
            int changed = 0;
            for (int t = lane; t < E; t += 64) {
                const int k = s_k[t];
                if (k < 0) continue;
                const int i = s_i[t], j = s_j[t];
                if (i < lo || i >= hi || j < lo || j >= hi) continue;
                for (int a = 0; a < 2; a++) {
                    const int u = 2 * i + a, w = 2 * j + (a ^ k);
                    const int mu = s_mark[u], mw = s_mark[w];
                    if (mu && !mw) { s_mark[w] = 1; changed = 1; }
                    else if (mw && !mu) { s_mark[u] = 1; changed = 1; }
                }
            }
            const int any = __any(changed);
            __syncthreads();
            if (!any) break;
        }
        int cnt = 0;
        for (int t = lane; t < 2 * n; t += 64) cnt += s_mark[t];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        return cnt;
    };
    // configuration string of [lo, hi) from the marks (a variant without a reached allele is skipped: quirk kept); lane 0 only
    auto marks_to_string = [&](int lo, int hi, char *dst) -> int {
        int w = 0;
        for (int i = lo; i < hi; i++) {
            if (s_mark[2 * i]) dst[w++] = '0';
            else if (s_mark[2 * i + 1]) dst[w++] = '1';
        }
        return w;
    };
    const unsigned long long tk0 = PH_TICK();
    unsigned long long tk1 = tk0, tk2 = tk0, tk3 = tk0;
    const int reached = flood(0, n);
    tk1 = PH_TICK();
    if (reached == n) {
        if (lane == 0) { s_finlen[0] = (uint16_t)marks_to_string(0, n, s_fin); s_nfin = 1; }
    } else {
        const int xmax = P.max_block_size == 0 ? n : P.max_block_size;
        // split_by_weak (:2271-2324)
        // weak[p] = pairs spanning the cut before variant p.  Everything below is wave-uniform and lives in registers: the cut set is a 256-bit
        // mask, a level's candidate cuts come from ballots, and only the levels that occur are visited.  (Done by one lane over LDS arrays -- one
        // pass over all positions per level -- this was 0.3 ms for a component of 96 variants: the tail of the whole launch.)
        for (int p = lane; p <= n + 1; p += 64) s_weak[p] = 0;
        __syncthreads();
        for (int t = lane; t < E; t += 64) {
            const int i = s_i[t] < s_j[t] ? s_i[t] : s_j[t], j = s_i[t] < s_j[t] ? s_j[t] : s_i[t];
            if (i == j) continue;
            atomicAdd(&s_weak[i + 1], 1); atomicSub(&s_weak[j + 1], 1);             // a pair (i, j) spans the cut positions p in (i, j]
        }
        __syncthreads();
        {   // running sum over positions 0 .. n: four consecutive positions per lane + wave scan
            int v[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { const int q = 4 * lane + k; v[k] = q <= n ? s_weak[q] : 0; sum += v[k]; }
            int incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(incl, d); if (lane >= d) incl += y; }
            int run = incl - sum;
#pragma unroll
            for (int k = 0; k < 4; k++) { const int q = 4 * lane + k; run += v[k]; if (q <= n) s_weak[q] = run; }
        }
        __syncthreads();
        int wv[4];                                               // weak value of the candidate positions 64k + lane (2 <= p < n - 1), -1 elsewhere
#pragma unroll
        for (int k = 0; k < 4; k++) { const int q = 64 * k + lane; wv[k] = (q >= 2 && q < n - 1) ? s_weak[q] : -1; }
        unsigned long long cut[4] = {0ull, 0ull, 0ull, 0ull};
        auto is_cut = [&](int q) -> bool { return (cut[(q >> 6) & 3] >> (q & 63)) & 1ull; };
        int biggest = n, level = 1, status = 0;
        bool first_round = true;
        while (biggest > xmax || first_round) {
            unsigned long long cm[4];
#pragma unroll
            for (int k = 0; k < 4; k++) cm[k] = __ballot(wv[k] == level);
            if (cm[0] | cm[1] | cm[2] | cm[3]) {
#pragma unroll
                for (int k = 0; k < 4; k++) {                      // ascending positions: a cut made here blocks its right neighbour (:2296-2300)
                    unsigned long long m = cm[k];
                    while (m) {
                        const int q = 64 * k + __builtin_ctzll(m);
                        m &= m - 1ull;
                        if (!is_cut(q - 1) && !is_cut(q + 1)) cut[k] |= 1ull << (q & 63);
                    }
                }
                int prev = 0, big = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    unsigned long long m = cut[k];
                    while (m) {
                        const int q = 64 * k + __builtin_ctzll(m);
                        m &= m - 1ull;
                        big = q - prev > big ? q - prev : big; prev = q;
                    }
                }
                biggest = n - prev > big ? n - prev : big;
            }
            first_round = false;
            int nl = 0x7FFFFFFF;                                 // the next level that occurs
#pragma unroll
            for (int k = 0; k < 4; k++) if (wv[k] > level && wv[k] < nl) nl = wv[k];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const int y = __shfl_xor(nl, d); nl = y < nl ? y : nl; }
            if (nl == 0x7FFFFFFF) { if (biggest > xmax) status = 1; break; }          // no level left: the reference never leaves this loop
            level = nl;
        }
        if (lane == 0) {
            int nbd = 0;
            s_bounds[nbd++] = 0;
            for (int k = 0; k < 4; k++) {
                unsigned long long m = cut[k];
                while (m) { s_bounds[nbd++] = (uint16_t)(64 * k + __builtin_ctzll(m)); m &= m - 1ull; }
            }
            s_bounds[nbd++] = (uint16_t)n;
            s_nf = nbd - 1;
            if (status) s_status = 1;
        }
        __syncthreads();
        tk2 = PH_TICK();
        if (s_status) { if (lane == 0) atomicAdd(&P.counters[2], 1u); return; }
        const int nf = s_nf;
        // sub_block_phase without a given configuration, per fragment (:2209-2258)
        if (lane == 0) s_p0off[0] = 0;
        __syncthreads();
        for (int f = 0; f < nf; f++) {
            const int base = s_bounds[f], len = (int)s_bounds[f + 1] - base;
            bool done = false;
            // Pairs of the fragment as bit masks by index distance d = j - i: bit i of s_m[0][d] / s_m[1][d] = a same-configuration /
            // opposite pair (i, i + d) -- for the flood fill of a fragment of up to 32 variants and for the search over its configurations
            const bool small = len <= 32;
            if (small) {
                for (int t = lane; t < 2 * 32; t += 64) s_m[t >> 5][t & 31] = 0u;
                __syncthreads();
                for (int t = lane; t < E; t += 64) {
                    const int k = s_k[t];
                    if (k < 0) continue;
                    int i = (int)s_i[t] - base, j = (int)s_j[t] - base;
                    if (i < 0 || j < 0 || i >= len || j >= len) continue;
                    if (i > j) { const int x = i; i = j; j = x; }
                    atomicOr(&s_m[k][j - i], 1u << i);
                }
                __syncthreads();
            }
            if (nf > 1) {
                if (small) {
                    // flood fill on bit masks (r0 / r1 = variants reached with allele 0 / 1): lane d spreads along the pairs at distance d, the lanes'
                    // results are OR-ed; a same-configuration pair keeps the allele, an opposite pair swaps it
                    const int sh = lane & 31;
                    const uint32_t a0 = (lane >= 1 && lane < 32) ? s_m[0][sh] : 0u, a1 = (lane >= 1 && lane < 32) ? s_m[1][sh] : 0u;
                    uint32_t r0 = 1u, r1 = 0u;
                    for (;;) {
                        uint32_t n0 = r0 | ((r0 & a0) << sh) | ((r0 >> sh) & a0) | ((r1 & a1) << sh) | ((r1 >> sh) & a1);
                        uint32_t n1 = r1 | ((r1 & a0) << sh) | ((r1 >> sh) & a0) | ((r0 & a1) << sh) | ((r0 >> sh) & a1);
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) { n0 |= __shfl_xor(n0, d); n1 |= __shfl_xor(n1, d); }
                        if (n0 == r0 && n1 == r1) break;
                        r0 = n0; r1 = n1;
                    }
                    if (__popc(r0) + __popc(r1) == len) {          // the reference accepts iff that many allele nodes were reached (:2199)
                        const uint32_t any = r0 | r1;                // a variant without a reached allele is skipped: quirk kept
                        if (lane < len && ((any >> lane) & 1u)) s_p0[s_p0off[f] + __popc(any & ((1u << lane) - 1u))] = ((r0 >> lane) & 1u) ? '0' : '1';
                        if (lane == 0) s_p0off[f + 1] = (uint16_t)(s_p0off[f] + __popc(any));
                        done = true;
                    }
                } else {
                    const int r = flood(base, base + len);
                    if (r == len) {
                        if (lane == 0) s_p0off[f + 1] = (uint16_t)(s_p0off[f] + marks_to_string(base, base + len, s_p0 + s_p0off[f]));
                        done = true;
                    }
                }
            }
            if (!done) {
                if (len > PH_BRUTE_MAX) {                  // the host takes it (the reference's limit is its patience)
                    if (lane == 0) P.exc_list[atomicAdd(&P.counters[1], 1u)] = c;
                    return;
                }
                // A configuration is the bit vector b (bit t = allele of variant t, bit 0 = 0); its consistent pairs at distance d are
                // popc(~(b ^ b >> d) & same) + popc((b ^ b >> d) & opposite): no memory access per configuration.  (The reference scores
                // 2^(n-1) configurations pair by pair, :2236-2258; only the maximum, the number of configurations reaching it and the unique
                // winner matter, so the order of enumeration is free.)
                // The score is a sum over pairs (i, i + d): when no pair of the fragment reaches further than D <= PH_DP_MAXD variants, the maximum
                // over the 2^(len-1) configurations, the number of configurations reaching it (1 or "several") and the winner follow from a dynamic
                // programme over the last D alleles -- len * 2^D steps instead of 2^(len-1) * len.  (A fragment of 20 variants, brute force on
                // one wave, was the 0.77 ms tail of the whole phasing launch.)
                int D = 0;
                for (int d = 1; d < len && d < 32; d++) if (s_m[0][d] | s_m[1][d]) D = d;
                uint32_t tt, cc;
                if (len >= PH_DP_MINLEN && D <= PH_DP_MAXD) {
                    const int De = D > 0 ? D : 1;
                    const int NS = 1 << De;
                    const int32_t NEG = -(1 << 28);
                    for (int sp = lane; sp < NS; sp += 64) { s_dp[0][sp] = sp == 0 ? 0 : NEG; s_dc[0][sp] = sp == 0 ? 1 : 0; }
                    __syncthreads();
                    int cur = 0;
                    for (int t = 1; t < len; t++) {
                        // pairs (t - d, t) as masks over the state bits: bit d - 1 = allele of variant t - d
                        const int dd = lane + 1;
                        const bool inr = dd <= De && t - dd >= 0;
                        const uint32_t A = (uint32_t)__ballot(inr && ((s_m[0][dd & 31] >> ((t - dd) & 31)) & 1u));
                        const uint32_t B = (uint32_t)__ballot(inr && ((s_m[1][dd & 31] >> ((t - dd) & 31)) & 1u));
                        for (int s0 = 0; s0 < NS; s0 += 64) {
                            const int sp = s0 + lane;
                            const bool valid = sp < NS;
                            int32_t best = NEG; uint32_t cnt = 0; bool take1 = false;
                            if (valid) {
                                const uint32_t X = (sp & 1) ? ~0u : 0u;
                                const uint32_t p0 = (uint32_t)sp >> 1, p1 = p0 | (1u << (De - 1));
                                const int32_t d0 = s_dp[cur][p0], d1 = s_dp[cur][p1];
                                const int32_t c0 = d0 < 0 ? NEG : d0 + __popc(~(p0 ^ X) & A) + __popc((p0 ^ X) & B);
                                const int32_t c1 = d1 < 0 ? NEG : d1 + __popc(~(p1 ^ X) & A) + __popc((p1 ^ X) & B);
                                best = c0 > c1 ? c0 : c1;
                                if (best >= 0) cnt = (c0 == best ? s_dc[cur][p0] : 0u) + (c1 == best ? s_dc[cur][p1] : 0u);
                                cnt = cnt > 2u ? 2u : cnt;
                                take1 = c1 > c0;
                            }
                            const unsigned long long chb = __ballot(valid && take1);
                            if (valid) { s_dp[cur ^ 1][sp] = best; s_dc[cur ^ 1][sp] = (uint8_t)cnt; }
                            if (lane == 0) { s_ch[t][s0 >> 5] = (uint32_t)chb; if (s0 + 32 < NS) s_ch[t][(s0 >> 5) + 1] = (uint32_t)(chb >> 32); }
                        }
                        __syncthreads();
                        cur ^= 1;
                    }
                    int32_t best_s = NEG; uint32_t ties = 0, arg = NONE32;
                    for (int sp = lane; sp < NS; sp += 64) {
                        const int32_t v = s_dp[cur][sp];
                        if (v > best_s) { best_s = v; ties = s_dc[cur][sp]; arg = (uint32_t)sp; }
                        else if (v == best_s && v >= 0) ties += s_dc[cur][sp];
                    }
                    int32_t gmax = best_s;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) { const int32_t o = __shfl_xor(gmax, d); gmax = o > gmax ? o : gmax; }
                    tt = best_s == gmax ? ties : 0u;
                    uint32_t aa = best_s == gmax ? arg : NONE32;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) { tt += __shfl_xor(tt, d); const uint32_t o = __shfl_xor(aa, d); aa = o < aa ? o : aa; }
                    cc = 0;
                    if (tt == 1) {                                   // the winner, back through the recorded predecessors (every lane walks the same path)
                        uint32_t st = aa;
                        for (int t = len - 1; t >= 1; t--) {
                            cc |= (st & 1u) << (t - 1);
                            const uint32_t hb = (s_ch[t][st >> 5] >> (st & 31u)) & 1u;
                            st = (st >> 1) | (hb << (De - 1));
                        }
                    }
                } else {
                uint32_t m0[PH_BRUTE_MAX], m1[PH_BRUTE_MAX];
#pragma unroll
                for (int d = 1; d < PH_BRUTE_MAX; d++) { m0[d] = s_m[0][d]; m1[d] = s_m[1][d]; }
                const uint32_t ncode = 1u << (len - 1);
                int best_s = -1; uint32_t best_c = 0; uint32_t ties = 0;
                for (uint32_t code = (uint32_t)lane; code < ncode; code += 64) {
                    const uint32_t bv = code << 1;
                    int sc = 0;
#pragma unroll
                    for (int d = 1; d < PH_BRUTE_MAX; d++) {
                        const uint32_t x = bv ^ (bv >> d);
                        sc += __popc(~x & m0[d]) + __popc(x & m1[d]);
                    }
                    if (sc > best_s) { best_s = sc; best_c = code; ties = 1; }
                    else if (sc == best_s) ties++;
                }
                int gmax = best_s;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(gmax, d); gmax = o > gmax ? o : gmax; }
                tt = best_s == gmax ? ties : 0u; cc = best_s == gmax ? best_c : NONE32;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { tt += __shfl_xor(tt, d); const uint32_t o = __shfl_xor(cc, d); cc = o < cc ? o : cc; }
                }
                if (lane == 0) {
                    char *dst = s_p0 + s_p0off[f];
                    if (tt == 1) { dst[0] = '0'; for (int k = 1; k < len; k++) dst[k] = ((cc >> (k - 1)) & 1u) ? '1' : '0'; }
                    else for (int k = 0; k < len; k++) dst[k] = '-';              // a tie for the best score: all '-' (:2255-2258)
                    s_p0off[f + 1] = (uint16_t)(s_p0off[f] + len);
                }
            }
            __syncthreads();
        }
        tk3 = PH_TICK();
        // stitching, left to right (:2138-2160); haplotype B is always the flip of haplotype A, so only A is carried.  The control flow is
        // wave-uniform (state in LDS, written by lane 0): the scores of the joint configurations are summed over the pairs by all lanes
        if (lane == 0) {
            const int curlen = (int)s_p0off[1];
            for (int k = 0; k < curlen; k++) s_cur[k] = s_p0[k];
            s_st[0] = curlen; s_st[1] = 0; s_st[2] = 0; s_st[3] = 0;        // |cur|, start, strings in fin, characters in fin
        }
        __syncthreads();
        for (int f = 1; f < nf; f++) {
            const char *nxt = s_p0 + s_p0off[f];
            const int nlen = (int)s_p0off[f + 1] - (int)s_p0off[f];
            const int curlen = s_st[0], start = s_st[1];
            const int used = curlen + nlen;                     // (|cur A| + |cur B| + |next A| + |next B| + 1) / 2
            const int hi = n < start + used ? n : start + used;
            const int idx_n = hi - start > 0 ? hi - start : 0;
            // candidates: cur A + next A, cur A + next B; the other two joint configurations are their complements (skipped, :2228-2234).
            // All string work is spread over the lanes (a component of 250 variants stitches a hundred fragments: done by one lane, these loops were
            // the tail of the launch); the decisions are wave-uniform
            {
                int cur_not = 0, nxt_not = 0;                       // a string of '-' (or an empty one) equals its own flip
                for (int k = lane; k < curlen; k += 64) { const char ch = s_cur[k]; if (ch != '-') cur_not = 1; s_cand[k] = ch; s_cand[used + k] = ch; }
                for (int k = lane; k < nlen; k += 64) { const char ch = nxt[k]; if (ch != '-') nxt_not = 1; s_cand[curlen + k] = ch; s_cand[used + curlen + k] = flipc(ch); }
                const int both = __any(cur_not) && __any(nxt_not);
                if (lane == 0) s_st[4] = both ? 2 : 1;
            }
            __syncthreads();
            const int ncand = s_st[4];
            // supporting allele pairs of a joint configuration over variants start.. (:2236-2249): twice the number of consistent pairs
            int sc[2] = {0, 0};
            const int L = idx_n < used ? idx_n : used;
            for (int t = lane; t < E; t += 64) {
                const int k = s_k[t];
                if (k < 0) continue;
                const int i = (int)s_i[t] - start, j = (int)s_j[t] - start;
                if (i < 0 || j < 0 || i >= L || j >= L) continue;
                for (int q = 0; q < ncand; q++) {
                    const char ci = s_cand[q * used + i], cj = s_cand[q * used + j];
                    if (ci != '-' && cj != '-' && (cj - '0') == ((ci - '0') ^ k)) sc[q] += 2;
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { sc[0] += __shfl_xor(sc[0], d); sc[1] += __shfl_xor(sc[1], d); }
            {
                int win = -1;
                if (ncand == 1) win = 0;
                else if (sc[0] > sc[1]) win = 0;
                else if (sc[1] > sc[0]) win = 1;
                bool has_dash;
                int outlen;
                if (win >= 0) {
                    outlen = used;
                    int dash = 0;
                    for (int k = lane; k < used; k += 64) if (s_cand[win * used + k] == '-') dash = 1;
                    has_dash = __any(dash) != 0;
                } else { outlen = idx_n; has_dash = idx_n > 0; }               // all '-' (length idx_n); empty string holds no '-'
                const int finpos = s_st[3], nfin_now = s_st[2];
                __syncthreads();                                               // everybody has read the state before lane 0 changes it
                if (has_dash) {
                    for (int k = lane; k < curlen; k += 64) s_fin[finpos + k] = s_cur[k];
                    __syncthreads();
                    for (int k = lane; k < nlen; k += 64) s_cur[k] = nxt[k];
                    if (lane == 0) {
                        s_finlen[nfin_now] = (uint16_t)curlen; s_st[2] = nfin_now + 1; s_st[3] = finpos + curlen;
                        s_st[1] = used;                                          // ASSIGNED, not advanced (:2152)
                        s_st[0] = nlen;
                    }
                } else {
                    for (int k = lane; k < outlen; k += 64) s_cur[k] = win >= 0 ? s_cand[win * used + k] : '-';
                    if (lane == 0) s_st[0] = outlen;
                }
            }
            __syncthreads();
        }
        if (lane == 0) {
            const int curlen = s_st[0], finpos = s_st[3];
            for (int k = 0; k < curlen; k++) s_fin[finpos + k] = s_cur[k];
            s_finlen[s_st[2]] = (uint16_t)curlen;
            s_nfin = s_st[2] + 1;
        }
    }
    __syncthreads();
    // final sub-blocks (:2162-2169): strings laid end to end over the variants; one that starts with '-' is dropped
    if (lane == 0) {
        for (int t = 0; t < n; t++) P.sub_of[m0 + t] = -1;
        int vi = 0, fp = 0, ns = 0;
        for (int q = 0; q < s_nfin; q++) {
            const int len = s_finlen[q];
            if (len > 0 && s_fin[fp] != '-') {
                int put = 0;
                for (int t = 0; t < len; t++) {
                    const int li = vi + t;
                    if (li < n) { P.sub_of[m0 + li] = (int32_t)ns; P.alle_of[m0 + li] = (uint8_t)(s_fin[fp + t] == '1' ? 1 : 0); put++; }
                }
                if (put > 0) ns++;
            }
            vi += len; fp += len;
        }
        P.nsub[c] = (uint32_t)ns;
#ifdef PHZ_PHASE_PROFILE
        const unsigned long long tk4 = PH_TICK();
        const uint32_t tot = (uint32_t)(tk4 - tk0);
        if (atomicMax(&P.counters[4], tot) < tot) { P.counters[5] = (uint32_t)(tk1 - tk0); P.counters[6] = (uint32_t)(tk2 - tk1); P.counters[7] = (uint32_t)(tk3 - tk2); P.counters[9] = (uint32_t)(tk4 - tk3); P.counters[10] = (uint32_t)n; P.counters[11] = (uint32_t)E; P.counters[12] = (uint32_t)s_nf; }
#endif
    }
    (void)tk1; (void)tk2; (void)tk3;
}
// One wave per component of the list k_phase_pair left (counters[0] of them), taken in ticket order by a fixed number of waves: the host launches this
// kernel right behind k_phase_pair without reading the count back (PH_TICKET: counters word of the ticket)
constexpr int PH_TICKET = 16;
__global__ __launch_bounds__(64) void k_phase_general(PH P) {
    // (tickets in batches: same-address atomics are served one after the other, ~8 ns each -- one per component was the length of the launch)
    constexpr uint32_t BATCH = 4;
    __shared__ uint32_t s_t;
    const uint32_t n = P.counters[0];
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(&P.counters[PH_TICKET], BATCH);
        __syncthreads();
        const uint32_t t0 = s_t;
        if (t0 >= n) return;
        for (uint32_t t = t0; t < t0 + BATCH && t < n; t++) {
            phase_general_one(P, P.complex_list[t]);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------- blocks
// blocks of the components in block order (rule 4): component r of the order owns blocks [blk_base[r], blk_base[r + 1])
struct BK {
    const uint32_t *corder, *blk_base, *cstart, *mem_s;
    const int32_t *sub_of; const uint8_t *alle_of;
    uint32_t *blk_mstart, *blk_len; int32_t *blk_of; uint8_t *v_alle;
};
__global__ __launch_bounds__(256) void k_blocks(int64_t ncomp, BK B) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= ncomp) return;
    const uint32_t c = B.corder[r], base = B.blk_base[r];
    const uint32_t m0 = B.cstart[c], n = B.cstart[c + 1] - m0;
    int cur = -1; uint32_t run0 = 0;
    for (uint32_t t = 0; t <= n; t++) {
        const int s = t < n ? (int)B.sub_of[m0 + t] : -2;
        if (s != cur) {
            if (cur >= 0) {
                const uint32_t len = t - run0;
                B.blk_mstart[base + (uint32_t)cur] = m0 + run0; B.blk_len[base + (uint32_t)cur] = len;
            }
            cur = s; run0 = t;
        }
        if (t < n && s >= 0) { const uint32_t v = B.mem_s[m0 + t]; B.blk_of[v] = (int32_t)(base + (uint32_t)s); B.v_alle[v] = B.alle_of[m0 + t]; }
    }
}
__global__ __launch_bounds__(256) void k_gather_nsub(int64_t ncomp, const uint32_t *corder, const uint32_t *nsub, uint32_t *out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < ncomp) out[r] = nsub[corder[r]];
}
// allele pairs supporting / inside each final block (:876-895; ordered pairs halved = pairs)
__global__ __launch_bounds__(256) void k_blk_edges(int64_t nkeep, const uint32_t *ekeep, const int32_t *ea, const int32_t *eb, const int32_t *cfgv, const int32_t *blk_of,
                                                   const uint8_t *v_alle, uint32_t *blk_sup, uint32_t *blk_tot) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // the kept pairs come component by component, so the lanes of a wave count for a handful of blocks: one pair of atomics per block and wave
    // (one per pair meant up to 64 same-address atomics per instruction)
    int ba = -1; bool sup = false;
    if (t < nkeep) {
        const uint32_t e = ekeep[t];
        const int k = cfgv[e];
        if (k >= 0) {
            const int a = ea[e], b = eb[e];
            const int x = blk_of[a];
            if (x >= 0 && x == blk_of[b]) {
                ba = x;
                const int want = k == 0 ? v_alle[a] : 1 - v_alle[a];
                sup = (int)v_alle[b] == want;
            }
        }
    }
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(ba >= 0);
    const unsigned long long supm = __ballot(sup);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int key = __shfl(ba, leader);
        const unsigned long long m = __ballot(ba == key) & todo;
        if (lane == leader) {
            atomicAdd(&blk_tot[key], (uint32_t)__popcll(m));
            const uint32_t ns = (uint32_t)__popcll(m & supm);
            if (ns) atomicAdd(&blk_sup[key], ns);
        }
        todo &= ~m;
    }
}
// per block: annotated phase, concordance, genome-wide phase by majority (:945-980), gwStat, the variant whose maf text is printed,
// rows of allele_config
struct BS {
    const uint32_t *mem_s, *blk_mstart, *blk_len; const uint8_t *v_alle; const int8_t *phase_idx; const double *mafv;
    uint8_t *conc, *cormode, *statkind; uint32_t *statidx; int32_t *maxmaf; double *stat; unsigned long long *cfg_rows;
    double *big_stat; uint32_t *big_stat_n;        // gwStat of the blocks whose text the table does not hold (slot = the block's statidx); count
    int gw_phase_method;                    // 1: MAF-weighted genome-wide phase (:982-1025)
};
__global__ __launch_bounds__(256) void k_blk_stats(int64_t nblocks, BS S) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t m0 = S.blk_mstart[b], n = S.blk_len[b];
    int nknown = 0, ksum = 0, first_known = -1;
    bool all_equal = true, any_nan = false;
    int64_t best = S.mem_s[m0];
    for (uint32_t t = 0; t < n; t++) {
        const int64_t g = S.mem_s[m0 + t];
        const int p = S.phase_idx[2 * g + S.v_alle[g]];
        if (p >= 0) {
            if (first_known < 0) first_known = p; else if (p != first_known) all_equal = false;
            nknown++; ksum += p;
        } else any_nan = true;
        if (S.mafv[g] > S.mafv[best]) best = g;                      // first maximal element, like max()
    }
    uint8_t cm = 0, kind = 2; uint32_t idx = 0; double stat = 0.5;
    if (nknown > 0) {
        if (!any_nan && all_equal) { kind = 1; stat = 1.0; }
        else {
            bool by_mean = true;
            if (S.gw_phase_method == 1) {
                // phase_support (:993-1000): the MAFs of the variants with annotated phase 0 / 1 on haplotype A, added up in block order
                double w0 = 0, w1 = 0;
                for (uint32_t t = 0; t < n; t++) {
                    const int64_t g = S.mem_s[m0 + t];
                    const int p = S.phase_idx[2 * g + S.v_alle[g]];
                    if (p == 0) w0 += S.mafv[g]; else if (p == 1) w1 += S.mafv[g];
                }
                const double sw = w0 + w1;
                if (sw > 0) {
                    by_mean = false;
                    stat = (w0 >= w1 ? w0 : w1) / sw;
                    if (w0 > w1) cm = 1; else if (w1 > w0) cm = 2;
                    kind = 3;
                }
            }
            if (by_mean) {
                const double m = (double)ksum / (double)nknown;
                if (m < 0.5) cm = 1; else if (m > 0.5) cm = 2;
                const double other = 1 - m;
                stat = m >= other ? m : other;
                if (nknown <= STAT_N) { kind = 0; idx = (uint32_t)nknown * (STAT_N + 1) + (uint32_t)ksum; }
                else kind = 3;
            }
            if (kind == 3) { idx = atomicAdd(S.big_stat_n, 1u); S.big_stat[idx] = stat; }          // text formatted by the host inside the run
        }
    }
    S.conc[b] = all_equal ? 1 : 0; S.cormode[b] = cm; S.statkind[b] = kind; S.statidx[b] = idx; S.maxmaf[b] = (int32_t)best; S.stat[b] = stat;
    S.cfg_rows[b] = (unsigned long long)n * (n - 1);
}

// ---------------------------------------------------------------------------------------------- read sets of the haplotypes
// A segment is a list of read-list pieces; its items (QNAME ids, in line order) get the number of their QNAME by first appearance
// and the segment its number of distinct QNAMEs.
//   MODE 0: segment (block, haplotype, BAM): the haplotype's allele lists of the block's non-blacklisted variants in that BAM; labels written
//   MODE 1: segment (block, haplotype): all variants, all BAMs; count only (reads_hap_a / reads_hap_b of haplotypes.txt)
//   MODE 2: segment = one (variant, allele, BAM) list; count only (singleton rows when there are several BAMs)
struct SG {
    int64_t nseg; int nb;
    const uint32_t *mem_s, *blk_mstart, *blk_len; const uint8_t *v_alle, *black;
    const uint32_t *rl_start; const int32_t *rl_qid;
    uint32_t *labels, *ns;
    uint32_t *big_list, *big_list2; uint32_t *counters;         // [0] segments left to the wave kernel, [1] pool slots used, [2] pool overflow, [3] segments left to the workgroup kernel
    uint32_t *huge_list, *huge_count;                             // segments of more than STAT_N pieces (a block of more than STAT_N variants): k_seg_big<.., HUGE>, piece arrays in the pool
    uint32_t *tickets;                                            // [3] next list entry of the mid / large / huge kernel
    uint32_t *overflow;                                           // pool overflow seen by ANY mode of this attempt
    uint32_t *pool; uint32_t pool_cap;
    const uint32_t *lab_e; int64_t nmem;        // read list of (haplotype, BAM, block member): one load instead of mem_s -> v_alle -> index arithmetic
    uint8_t *isf;                                // --output_read_ids 1 (modes 0 and 2): item p is the FIRST of its QNAME in its segment (else nullptr)
};
template <int MODE> __device__ __forceinline__ uint32_t seg_pieces(const SG &G, int64_t seg) {      // number of pieces (some may be empty / skipped)
    if (MODE == 2) return 1;
    const int64_t b = MODE == 0 ? seg / (2 * G.nb) : seg / 2;
    return G.blk_len[b];
}
template <int MODE> __device__ __forceinline__ bool seg_piece(const SG &G, int64_t seg, uint32_t t, uint32_t *lo, uint32_t *hi) {
    if (MODE == 2) { *lo = G.rl_start[seg]; *hi = G.rl_start[seg + 1]; return true; }
    const int64_t b = MODE == 0 ? seg / (2 * G.nb) : seg / 2;
    const int h = MODE == 0 ? (int)((seg / G.nb) & 1) : (int)(seg & 1);
    const uint32_t m = G.blk_mstart[b] + t;
    if (MODE == 0) {
        if (G.black && G.black[G.mem_s[m]]) return false;
        const int bb = (int)(seg % G.nb);
        const uint32_t e = G.lab_e[(size_t)(h * G.nb + bb) * (size_t)G.nmem + m];
        *lo = G.rl_start[e]; *hi = G.rl_start[e + 1];
    } else {
        const uint32_t e0 = G.lab_e[(size_t)(h * G.nb) * (size_t)G.nmem + m];         // the lists of the member's allele over all BAMs are adjacent
        *lo = G.rl_start[e0]; *hi = G.rl_start[e0 + G.nb];
    }
    return true;
}

template <int MODE> __global__ __launch_bounds__(64) void k_seg_small(SG G) {
    __shared__ int32_t s_q[64][SEG_SMALL + 1];
    __shared__ uint8_t s_l[64][SEG_SMALL + 4];
    const int64_t seg = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int tid = threadIdx.x;
    const bool live = seg < G.nseg;
    const uint32_t np = live ? seg_pieces<MODE>(G, seg) : 0u;
    uint32_t M = 0;
    for (uint32_t t = 0; t < np; t++) { uint32_t lo, hi; if (seg_piece<MODE>(G, seg, t, &lo, &hi)) M += hi - lo; }
    if (live && M == 0) G.ns[seg] = 0;
    // the segments left to the wave / workgroup kernels take their places in the two lists with one cursor step per wave and list (82,000 of a
    // genome's 370,000 segments: one same-address atomic each was most of this kernel's time)
    {
        const bool huge = np > (uint32_t)STAT_N && M > (uint32_t)SEG_SMALL;              // its pieces do not fit the LDS arrays of k_seg_big
        if (huge) G.huge_list[atomicAdd(G.huge_count, 1u)] = (uint32_t)seg;                 // (rare: one atomic each)
        const unsigned long long m2 = __ballot(!huge && M > (uint32_t)SEG_MID), m1 = __ballot(!huge && M > (uint32_t)SEG_SMALL && M <= (uint32_t)SEG_MID);
        const unsigned long long below = tid ? (~0ull >> (64 - tid)) : 0ull;
        if (m2) {
            uint32_t base = 0;
            const int leader = __builtin_ctzll(m2);
            if (tid == leader) base = atomicAdd(&G.counters[3], (uint32_t)__popcll(m2));
            base = __shfl(base, leader);
            if ((m2 >> tid) & 1ull) G.big_list2[base + (uint32_t)__popcll(m2 & below)] = (uint32_t)seg;
        }
        if (m1) {
            uint32_t base = 0;
            const int leader = __builtin_ctzll(m1);
            if (tid == leader) base = atomicAdd(&G.counters[0], (uint32_t)__popcll(m1));
            base = __shfl(base, leader);
            if ((m1 >> tid) & 1ull) G.big_list[base + (uint32_t)__popcll(m1 & below)] = (uint32_t)seg;
        }
    }
    if (!live || M == 0 || M > (uint32_t)SEG_SMALL) return;
    uint32_t idx = 0;
    for (uint32_t t = 0; t < np; t++) {
        uint32_t lo, hi;
        if (!seg_piece<MODE>(G, seg, t, &lo, &hi)) continue;
        for (uint32_t p = lo; p < hi; p++) s_q[tid][idx++] = G.rl_qid[p];
    }
    // labels by first appearance, in registers: every QNAME id against all earlier ones with selects (fully unrolled, no memory access; the loop
    // over LDS it replaces -- up to 500 dependent reads per thread -- took 23 us per wave)
    int32_t qv[SEG_SMALL];
#pragma unroll
    for (int i = 0; i < SEG_SMALL; i++) qv[i] = (uint32_t)i < M ? s_q[tid][i] : -(i + 2);       // padding never matches (QNAME ids are >= 0)
    uint32_t cnt = 0;
    uint32_t lab[SEG_SMALL];
#pragma unroll
    for (int i = 0; i < SEG_SMALL; i++) {
        uint32_t l = cnt;                                    // label if new
        bool seen = false;
#pragma unroll
        for (int j = i - 1; j >= 0; j--) { const bool eq = qv[j] == qv[i]; l = eq ? lab[j] : l; seen = seen || eq; }
        lab[i] = l;
        if (!seen && (uint32_t)i < M) cnt++;
    }
#pragma unroll
    for (int i = 0; i < SEG_SMALL; i++) s_l[tid][i] = (uint8_t)lab[i];
    G.ns[seg] = cnt;
    if (MODE == 0 || (MODE == 2 && G.isf)) {
        // an item is the first of its QNAME exactly when its label is one more than every label before it (labels count first appearances)
        idx = 0;
        uint32_t seen_n = 0;
        for (uint32_t t = 0; t < np; t++) {
            uint32_t lo, hi;
            if (!seg_piece<MODE>(G, seg, t, &lo, &hi)) continue;
            for (uint32_t p = lo; p < hi; p++) {
                const uint32_t l = s_l[tid][idx++];
                if (MODE == 0) G.labels[p] = l;
                if (G.isf) { const bool f = l == seen_n; G.isf[p] = f ? 1 : 0; seen_n += f ? 1u : 0u; }
            }
        }
    }
}

__device__ __forceinline__ uint32_t seg_hash(uint32_t q) { q ^= q >> 16; q *= 0x7feb352du; q ^= q >> 15; q *= 0x846ca68bu; q ^= q >> 16; return q; }

// one workgroup per segment with more than SEG_SMALL items: open-addressing table keyed by QNAME id holding the first position (then
// the number) of the QNAME -- in LDS up to SEG_LDS items, in a slice of the global pool beyond
#ifndef PHZ_SEG_THREADS
#define PHZ_SEG_THREADS 512
#endif
template <int MODE, int SLOTS, int THREADS, bool HUGE> __device__ void seg_big_one(const SG &G, const int64_t seg) {
    __shared__ uint32_t s_tab[3 * SLOTS];
    __shared__ uint32_t s_pref_l[HUGE ? 2 : STAT_N + 2], s_lo_l[HUGE ? 2 : STAT_N + 2];
    __shared__ uint32_t s_w[THREADS / 64 > 4 ? THREADS / 64 : 4];
    __shared__ uint32_t s_carry, s_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t np = seg_pieces<MODE>(G, seg);
    uint32_t *s_pref = s_pref_l, *s_lo = s_lo_l;
    if (HUGE) {          // piece starts and their prefix live in a slice of the global pool (same overflow protocol as the tables: grow and redo)
        if (tid == 0) {
            const uint32_t need = 2u * (np + 2u);
            const uint32_t off = atomicAdd(&G.counters[1], need);
            s_off = off;
            if ((unsigned long long)off + need > (unsigned long long)G.pool_cap) { atomicOr(&G.counters[2], 1u); atomicOr(G.overflow, 1u); s_off = NONE32; }
        }
        __syncthreads();
        if (s_off == NONE32) return;
        s_pref = G.pool + s_off; s_lo = s_pref + (np + 2u);
        __syncthreads();                                          // s_off is reused below
    }
    // the pieces of the segment (one read list each) looked up by all threads -- a chain of dependent loads per piece: one thread doing
    // them one after the other was the tail of the launch --, then their exclusive prefix by the first wave
    for (uint32_t t = tid; t < np; t += THREADS) {
        uint32_t lo = 0, hi = 0;
        const bool ok = seg_piece<MODE>(G, seg, t, &lo, &hi);
        s_lo[t] = lo; s_pref[t] = ok ? hi - lo : 0u;
    }
    if (tid == 0) { s_carry = 0; s_off = 0; }
    __syncthreads();
    if (wave == 0) {
        const uint32_t chunk = (np + 63u) / 64u, beg = (uint32_t)lane * chunk;
        uint32_t sum = 0;
        for (uint32_t t = beg; t < beg + chunk && t < np; t++) sum += s_pref[t];
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
        uint32_t run = incl - sum;
        for (uint32_t t = beg; t < beg + chunk && t < np; t++) { const uint32_t x = s_pref[t]; s_pref[t] = run; run += x; }
        if (lane == 63) s_pref[np] = incl;
    }
    __syncthreads();
    const uint32_t M = s_pref[np];
    uint32_t cap = 64;
    while (cap < 2 * M) cap <<= 1;                          // <= SLOTS as long as the table is in LDS
    uint32_t *keys = s_tab;
    if (2 * M > (uint32_t)SLOTS) {
        if (tid == 0) {
            const uint32_t off = atomicAdd(&G.counters[1], 3 * cap);
            s_off = off;
            if ((unsigned long long)off + 3ull * cap > (unsigned long long)G.pool_cap) { atomicOr(&G.counters[2], 1u); atomicOr(G.overflow, 1u); s_off = NONE32; }
        }
        __syncthreads();
        if (s_off == NONE32) return;
        keys = G.pool + s_off;
    }
    uint32_t *first = keys + cap, *rank = keys + 2 * cap;
    const uint32_t mask = cap - 1;
    for (uint32_t s = tid; s < cap; s += THREADS) { keys[s] = 0; first[s] = NONE32; }
    __syncthreads();
    auto item_pos = [&](uint32_t idx) -> uint32_t {
        uint32_t lo = 0, hi = np;                              // largest t with s_pref[t] <= idx (pieces of length 0 share a start: the last one wins, it is the non-empty one)
        while (hi - lo > 1) { const uint32_t m = (lo + hi) >> 1; if (s_pref[m] <= idx) lo = m; else hi = m; }
        return s_lo[lo] + (idx - s_pref[lo]);
    };
    auto slot_of = [&](uint32_t q) -> uint32_t {
        uint32_t s = seg_hash(q) & mask;
        while (keys[s] != q + 1u) s = (s + 1) & mask;
        return s;
    };
    for (uint32_t idx = tid; idx < M; idx += THREADS) {
        const uint32_t q = (uint32_t)G.rl_qid[item_pos(idx)];
        uint32_t s = seg_hash(q) & mask;
        for (;;) {
            const uint32_t prev = atomicCAS(&keys[s], 0u, q + 1u);
            if (prev == 0u || prev == q + 1u) { atomicMin(&first[s], idx); break; }
            s = (s + 1) & mask;
        }
    }
    __syncthreads();
    for (uint32_t base = 0; base < M; base += THREADS) {
        const uint32_t idx = base + tid;
        uint32_t slot = 0; bool isf = false;
        if (idx < M) { slot = slot_of((uint32_t)G.rl_qid[item_pos(idx)]); isf = first[slot] == idx; }
        const unsigned long long bal = __ballot(isf ? 1 : 0);
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = s_carry;
        for (int w = 0; w < wave; w++) before += s_w[w];
        if (isf) rank[slot] = before + (uint32_t)__popcll(bal & (lane ? (~0ull >> (64 - lane)) : 0ull));
        __syncthreads();
        if (tid == 0) { uint32_t add = 0; for (int w = 0; w < THREADS / 64; w++) add += s_w[w]; s_carry += add; }
        __syncthreads();
    }
    if (tid == 0) G.ns[seg] = s_carry;
    if (MODE == 0 || (MODE == 2 && G.isf))
        for (uint32_t idx = tid; idx < M; idx += THREADS) {
            const uint32_t p = item_pos(idx);
            const uint32_t slot = slot_of((uint32_t)G.rl_qid[p]);
            if (MODE == 0) G.labels[p] = rank[slot];
            if (G.isf) G.isf[p] = first[slot] == idx ? 1 : 0;
        }
}
// The segments of a list (mid: one wave each, large / huge: one workgroup each) taken in ticket order by a fixed number of workgroups; the list's length is
// read on the device (k_seg_small counted it in the launch before), so the host enqueues the three kernels without a wait in between.
// counters: [0] mid, [3] large, *huge_count; tickets: G.tickets[0..2]
template <int MODE, int SLOTS, int THREADS, bool HUGE = false> __global__ __launch_bounds__(THREADS) void k_seg_big(SG G) {
    __shared__ uint32_t s_t;
    const uint32_t n = HUGE ? *G.huge_count : (THREADS == 64 ? G.counters[0] : G.counters[3]);
    uint32_t *ticket = G.tickets + (HUGE ? 2 : (THREADS == 64 ? 0 : 1));
    const uint32_t *list = HUGE ? G.huge_list : (THREADS == 64 ? G.big_list : G.big_list2);
    // (tickets in batches: a genome has 80,000 wave-sized segments, and same-address atomics are served one after the other at ~8 ns each)
    constexpr uint32_t BATCH = HUGE ? 1 : (THREADS == 64 ? 16 : 2);
    for (;;) {
        __syncthreads();                                          // (s_t of the previous round has been read; the previous segment's LDS is dead)
        if (threadIdx.x == 0) s_t = atomicAdd(ticket, BATCH);
        __syncthreads();
        const uint32_t t0 = s_t;
        if (t0 >= n) return;
        for (uint32_t t = t0; t < t0 + BATCH && t < n; t++) {
            seg_big_one<MODE, SLOTS, THREADS, HUGE>(G, (int64_t)list[t]);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------- small helpers of the orchestration
// seg_off[s] = off[sum of count[0..s)]: byte offsets of the per-chromosome (per BAM x chromosome) segments of a file
// (one workgroup: every thread sums a contiguous chunk of the counts, the chunks' prefix comes from LDS -- one thread walking the segments with
//  a dependent load each was 8 us per file)
template <class C> __device__ __forceinline__ void seg_offsets_block(const C *count, unsigned long long mult, int nseg, const unsigned long long *off, unsigned long long *seg_off) {
    __shared__ unsigned long long s_part[256];
    const int tid = threadIdx.x;
    const int chunk = (nseg + 1 + 255) / 256, beg = tid * chunk, end = beg + chunk < nseg + 1 ? beg + chunk : nseg + 1;
    unsigned long long sum = 0;
    for (int j = beg; j < end; j++) if (j < nseg) sum += (unsigned long long)count[j] * mult;
    s_part[tid] = sum;
    __syncthreads();
    unsigned long long rows = 0;
    for (int t = 0; t < tid; t++) rows += s_part[t];
    for (int j = beg; j < end; j++) { seg_off[j] = off[rows]; if (j < nseg) rows += (unsigned long long)count[j] * mult; }
}
__global__ __launch_bounds__(256) void k_seg_offsets(const uint32_t *count, const uint32_t *mult, int nseg, const unsigned long long *off, unsigned long long *seg_off) {
    if (blockIdx.x) return;
    seg_offsets_block<uint32_t>(count, mult ? (unsigned long long)*mult : 1ull, nseg, off, seg_off);
}
__global__ __launch_bounds__(256) void k_seg_offsets64(const unsigned long long *count, int nseg, const unsigned long long *off, unsigned long long *seg_off) {
    if (blockIdx.x) return;
    seg_offsets_block<unsigned long long>(count, 1ull, nseg, off, seg_off);
}
__global__ __launch_bounds__(256) void k_count_nonzero(const uint32_t *len, int64_t n, unsigned long long *counter) {
    __shared__ unsigned int s_c[4];
    unsigned int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c += len[i] != 0 ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && (s_c[0] + s_c[1] + s_c[2] + s_c[3])) atomicAdd(counter, (unsigned long long)(s_c[0] + s_c[1] + s_c[2] + s_c[3]));
}
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t *p, int64_t n, uint32_t v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
// per-block arrays of write_vcf in block order, chromosome-local variant indices
struct VB {
    const uint32_t *mem_s, *blk_mstart, *blk_len, *blk_voff; const uint8_t *v_alle, *cormode; const int8_t *phase_idx; const int32_t *maxmaf;
    const uint16_t *vchrom; const long long *chrom_v0;
    int32_t *o_var, *o_maxmaf; uint8_t *o_hap; int8_t *o_cor;
};
__global__ __launch_bounds__(256) void k_vcf_blocks(int64_t nblocks, VB V) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t m0 = V.blk_mstart[b], n = V.blk_len[b], o = V.blk_voff[b];
    const long long v0 = V.chrom_v0[V.vchrom[V.mem_s[m0]]];
    const uint8_t cm = V.cormode[b];
    V.o_maxmaf[b] = (int32_t)(V.maxmaf[b] - v0);
    for (uint32_t t = 0; t < n; t++) {
        const int64_t g = V.mem_s[m0 + t];
        const uint8_t a = V.v_alle[g];
        V.o_var[o + t] = (int32_t)(g - v0); V.o_hap[o + t] = a;
        V.o_cor[2 * (size_t)(o + t)] = cm == 0 ? V.phase_idx[2 * g + a] : (int8_t)(cm == 1 ? 0 : 1);
        V.o_cor[2 * (size_t)(o + t) + 1] = cm == 0 ? V.phase_idx[2 * g + (a ^ 1)] : (int8_t)(cm == 1 ? 1 : 0);
    }
}
__global__ __launch_bounds__(256) void k_vchrom(int64_t nv, const long long *chrom_v0, int nchrom, uint16_t *vchrom) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    int lo = 0, hi = nchrom;
    while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (chrom_v0[m] <= v) lo = m; else hi = m; }
    vchrom[v] = (uint16_t)lo;
}

}  // namespace

// ================================================================================================ host side
struct phz_rowsdev {
    int64_t nv = 0;
    int nchrom = 0;
    bool has_black = false;
    std::vector<long long> chrom_v0;
    // static tables in HBM
    DevBuf d_chrom_v0, d_vchrom, d_pos, d_maf, d_isref, d_phase, d_black;
    DevBuf p_off[6], p_txt[6];          // pools: 0 uid, 1 rsid, 2 allele, 3 maf text, 4 chromosome names, 5 gwStat table
    // per pass
    DevBuf hkeys, flags, up_dev;        // up_dev: the per-pass uploads (p-values and their text by slot, BAM names, shard line ranges) as pieces of one block
    DevBuf up_host;                      // ... and its page-locked host image
    DevBuf keep, e_slot, deg, parent, label, f_a, f_b, f_c, f_d, mem_pos, cid, kpos, keypos;
    DevBuf k64a, k64b, k32a, k32b, v32a, v32b, sort_cnt, scan_tmp;
    DevBuf ridx, va, vb, eorder, mem_s, cstart, corder, ekeep, estart, key_g;
    DevBuf cnt64, cnt32, chrom_cnt, seg_start, key64s, eloc;     // chrom_cnt: uint32 [conn rows | blocks | block vars | keys per (bam, chrom)], then uint64 cfg rows
    DevBuf alle_of, sub_of, nsub, complex_list, exc_list, nsub_o, blk_base;
    DevBuf blk_mstart, blk_len, blk_of, v_alle, blk_sup, blk_tot, conc, cormode, statkind, statidx, maxmaf, stat, cfg_rows, cfg_base, cfg_chunk, cfg_pl, cfg_pb, cfg_ps, cfg_bytes, cfg_bbase, blk_voff, mrec, lab_e, lab_skip, big_blk;
    DevBuf labels, seg_ns, blk_cnt, single_n, big_list, big_list2, huge_list, big_stat, px_off, px_txt, pool, tl, its, piece_dst, rowlen;
    DevBuf off[PHZ_TXT_COUNT], seg_off_d[PHZ_TXT_COUNT], text[PHZ_TXT_COUNT];
    DevBuf o_var, o_maxmaf, o_hap, o_cor;
    DevBuf qn_off, qn_txt, qn_base, isf0, isf2;      // --output_read_ids 1
    // results (host)
    int64_t bytes[PHZ_TXT_COUNT] = {0};
    std::vector<int64_t> seg_off[PHZ_TXT_COUNT];
    std::vector<int64_t> chrom_blocks, chrom_blk_vars;
    std::vector<int32_t> chrom_first_bam;          // first BAM in which the chromosome has a kept call line (-1: none): its place in the block order (phaser.py:558-581, :1299)
    int64_t n_blocks = 0, n_blk_vars = 0;
    bool keys_ready = false, have_vcf = false;
    uint64_t pre_gen = 0;               // ... of WHICH resident tally (phz_ctx::tally_gen): another tally between the two stages discards them
    bool pre_done = false; unsigned long long pre_max_gap = 0;      // the p-value-independent ordering sorts were enqueued by phz_rowsdev_pair_keys (for the resident tally)
    // ... and, when the caller has told the handle the (chromosome, BAM) shards of the tally (phz_rowsdev_set_shards / phz_tally_pairs), the first-appearance keys too:
    // covered variants compacted, sorted by (BAM, first line), their per-(BAM, chromosome) segment starts and counts -- all p-value-independent
    bool have_shards = false, pre_keys = false; int64_t pre_nkeys = 0;
    std::vector<long long> sh_lo, sh_hi; std::vector<int32_t> sh_bam;
    DevBuf sh_dev, sh_host;
    hipEvent_t txt_ev[PHZ_TXT_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // "the writer of file f has finished" (copy-as-written)
    int64_t ps_slots = PS_SLOTS;        // slots of the pair-key hash set (a power of two; grown by the host when a pass reports PHZ_E_CAPACITY)
    std::vector<DevBuf *> all() {
        std::vector<DevBuf *> v = {&d_chrom_v0, &d_vchrom, &d_pos, &d_maf, &d_isref, &d_phase, &d_black, &hkeys, &flags, &up_dev, &keep, &e_slot, &deg, &parent, &label, &f_a, &f_b, &f_c, &f_d, &mem_pos, &cid, &kpos, &keypos,
                                   &k64a, &k64b, &k32a, &k32b, &v32a, &v32b, &sort_cnt, &scan_tmp, &ridx, &va, &vb, &eorder, &mem_s, &cstart, &corder, &ekeep, &estart,
                                   &key_g, &cnt64, &cnt32, &chrom_cnt, &seg_start, &key64s, &eloc, &alle_of, &sub_of, &nsub, &complex_list, &exc_list, &nsub_o, &blk_base, &blk_mstart, &blk_len,
                                   &blk_of, &v_alle, &blk_sup, &blk_tot, &conc, &cormode, &statkind, &statidx, &maxmaf, &stat, &cfg_rows, &cfg_base, &cfg_chunk, &cfg_pl, &cfg_pb, &cfg_ps, &cfg_bytes, &cfg_bbase, &blk_voff, &mrec, &lab_e, &lab_skip, &big_blk, &labels,
                                   &seg_ns, &blk_cnt, &single_n, &big_list, &big_list2, &huge_list, &big_stat, &px_off, &px_txt, &pool, &tl, &its, &piece_dst, &rowlen, &o_var, &o_maxmaf, &o_hap, &o_cor, &qn_off, &qn_txt, &qn_base, &isf0, &isf2, &sh_dev};
        for (int i = 0; i < 6; i++) { v.push_back(&p_off[i]); v.push_back(&p_txt[i]); }
        for (int i = 0; i < PHZ_TXT_COUNT; i++) { v.push_back(&off[i]); v.push_back(&seg_off_d[i]); v.push_back(&text[i]); }
        return v;
    }
};

namespace {

int up(phz_ctx *ctx, DevBuf &b, const void *src, size_t bytes) {
    if (int s = phz_reserve(ctx, b, bytes ? bytes : 1)) return s;
    if (bytes) PHZ_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PHZ_OK;
}
template <class T> T *P(DevBuf &b) { return (T *)b.p; }

// GPU time of the stage = sum over the sync-free sections between two host waits (the host work in between -- scipy, exceptions -- is not GPU time)
struct Sections {
    phz_ctx *c; double ms = 0; bool open = false;
    // PHZ_ROWS_TRACE=1: per host wait, how long the host spent enqueueing since the previous wait, how long it then blocked, and the GPU time of the section
    bool trace = getenv("PHZ_ROWS_TRACE") != nullptr; int nwait = 0;
    std::chrono::steady_clock::time_point t_prev = std::chrono::steady_clock::now();
    explicit Sections(phz_ctx *ctx) : c(ctx) {}
    void begin() { if (!open) { (void)hipEventRecord(c->ev0, c->stream); open = true; } }
    int wait(const char *what = "") {   // host wait: closes the section
        begin();
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipEventRecord(c->ev1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
        float x = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&x, c->ev0, c->ev1);
        if (e != hipSuccess) return phz_fail(c, PHZ_E_HIP, "device row stage", e);
        ms += x; open = false;
        if (trace) {
            const auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "[rows trace] wait %2d %-28s host enqueue %7.1f us, blocked %7.1f us, GPU section %7.1f us\n", nwait++, what,
                    std::chrono::duration<double, std::micro>(t0 - t_prev).count(), std::chrono::duration<double, std::micro>(t1 - t0).count(), (double)x * 1e3);
            t_prev = t1;
        }
        return PHZ_OK;
    }
};

// sort `n` (key, value) pairs living in (ka, va) with scratch (kb, vb); the sorted values land in `dst_val` (and the keys in dst_key if given): the last
// pass of the one-launch-per-pass sort writes there directly
template <class K>
int sort_into(phz_ctx *ctx, phz_rowsdev *h, DevBuf &ka, DevBuf &kb, int64_t n, const int (*ranges)[2], int nranges, uint32_t *dst_val, K *dst_key) {
    K *k0 = P<K>(ka), *k1 = P<K>(kb); uint32_t *v0 = P<uint32_t>(h->v32a), *v1 = P<uint32_t>(h->v32b);
    int where = 0;
    if (int s = radix_sort_ranges<K, uint32_t>(ctx, k0, k1, v0, v1, n, ranges, nranges, h->sort_cnt, h->scan_tmp, &where, dst_val, dst_key)) return s;
    if (where != 2 && n > 0) {           // nothing to sort (n <= 1 or no significant bits): the input order is the result
        const K *ks = where ? k1 : k0; const uint32_t *vs = where ? v1 : v0;
        PHZ_HIP(ctx, hipMemcpyAsync(dst_val, vs, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (dst_key) PHZ_HIP(ctx, hipMemcpyAsync(dst_key, ks, (size_t)n * sizeof(K), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return PHZ_OK;
}

// phase_v3 over all components: the two-variant fast path, the general kernel, the host for what exceeds the kernel's limits.
// cstart / estart: CSR of members (mem_s: variant ids, ascending inside a component) and kept pairs (ekeep: pair ids into ea / eb / cfgv).
// phase_enqueue puts both kernels on the stream WITHOUT a host wait (k_phase_general takes its components in ticket order and reads their number on the
// device); the caller reads cnt32[0..2] (complex components, exceptions, "cannot be split" flag) with its next wait and hands them to phase_exceptions.
constexpr unsigned PH_GRID = 4096;     // waves of k_phase_general (a genome has ~20,000 complex components; each wave takes the next one until the list is empty)
int phase_enqueue(phz_ctx *ctx, DevBuf &cstart, DevBuf &mem_s, DevBuf &estart, DevBuf &ekeep, const int32_t *ea, const int32_t *eb, const int32_t *cfgv,
                  int64_t ncomp, int64_t nmem, int64_t nkeep, int max_block_size, DevBuf &alle_of, DevBuf &sub_of, DevBuf &nsub, DevBuf &complex_list,
                  DevBuf &exc_list, DevBuf &eloc, uint32_t *cnt32) {
    hipStream_t sm = ctx->stream;
    if (int s = phz_reserve(ctx, alle_of, (size_t)(nmem + 1))) return s;
    if (int s = phz_reserve(ctx, sub_of, (size_t)(nmem + 1) * 4)) return s;
    if (int s = phz_reserve(ctx, nsub, (size_t)(ncomp + 1) * 4)) return s;
    if (int s = phz_reserve(ctx, complex_list, (size_t)(ncomp + 1) * 4)) return s;
    if (int s = phz_reserve(ctx, exc_list, (size_t)(2 * ncomp + 2) * 4)) return s;
    if (int s = phz_reserve(ctx, eloc, (size_t)(nkeep + 1) * 4)) return s;
    PHZ_HIP(ctx, hipMemsetAsync(cnt32, 0, 32, sm));
    PHZ_HIP(ctx, hipMemsetAsync(cnt32 + PH_TICKET, 0, 4, sm));
    if (!ncomp) return PHZ_OK;
    PH ph; ph.cstart = P<uint32_t>(cstart); ph.mem_s = P<uint32_t>(mem_s); ph.estart = P<uint32_t>(estart); ph.ekeep = P<uint32_t>(ekeep);
    ph.ea = ea; ph.eb = eb; ph.cfgv = cfgv; ph.alle_of = P<uint8_t>(alle_of); ph.sub_of = P<int32_t>(sub_of); ph.nsub = P<uint32_t>(nsub);
    ph.complex_list = P<uint32_t>(complex_list); ph.exc_list = P<uint32_t>(exc_list); ph.counters = cnt32; ph.max_block_size = max_block_size;
    ph.eloc = P<uint32_t>(eloc);
    if (nkeep) hipLaunchKernelGGL(k_edge_local, dim3(nblk(nkeep)), dim3(256), 0, sm, nkeep, ncomp, ph, P<uint32_t>(eloc));
    hipLaunchKernelGGL(k_phase_pair, dim3(nblk(ncomp)), dim3(256), 0, sm, ncomp, ph);
    hipLaunchKernelGGL(k_phase_general, dim3((unsigned)std::min<int64_t>(ncomp, (int64_t)PH_GRID)), dim3(64), 0, sm, ph);
    PHZ_HIP(ctx, hipGetLastError());
    return PHZ_OK;
}
// components beyond the kernel's limits (more than PH_NMAX variants / PH_EMAX pairs / a brute-force fragment of more than PH_BRUTE_MAX variants):
// phase_v3 on the host (the same routine the host row stage runs), results written back.  h_c32 = cnt32[0..2] as read by the caller AFTER the kernels.
int phase_exceptions(phz_ctx *ctx, const uint32_t *h_c32, DevBuf &cstart, DevBuf &mem_s, DevBuf &estart, DevBuf &ekeep, const int32_t *ea, const int32_t *eb, const int32_t *cfgv,
                     int64_t ncomp, int64_t nmem, int64_t nkeep, int64_t ne, int max_block_size, DevBuf &alle_of, DevBuf &sub_of, DevBuf &nsub, DevBuf &exc_list) {
    if (h_c32[2]) return phz_fail(ctx, PHZ_E_UNSUPPORTED, "a haplotype block cannot be split to --max_block_size (the reference does not terminate on it)");
    if (!h_c32[1]) return PHZ_OK;
    std::vector<uint32_t> exc(h_c32[1]), cs((size_t)ncomp + 1), es((size_t)ncomp + 1), mem((size_t)nmem), ek((size_t)nkeep);
    std::vector<int32_t> hea((size_t)ne), heb((size_t)ne), hcf((size_t)ne);
    PHZ_HIP(ctx, hipMemcpy(exc.data(), exc_list.p, exc.size() * 4, hipMemcpyDeviceToHost));
    PHZ_HIP(ctx, hipMemcpy(cs.data(), cstart.p, cs.size() * 4, hipMemcpyDeviceToHost));
    PHZ_HIP(ctx, hipMemcpy(es.data(), estart.p, es.size() * 4, hipMemcpyDeviceToHost));
    PHZ_HIP(ctx, hipMemcpy(mem.data(), mem_s.p, mem.size() * 4, hipMemcpyDeviceToHost));
    PHZ_HIP(ctx, hipMemcpy(ek.data(), ekeep.p, ek.size() * 4, hipMemcpyDeviceToHost));
    PHZ_HIP(ctx, hipMemcpy(hea.data(), ea, hea.size() * 4, hipMemcpyDeviceToHost));
    PHZ_HIP(ctx, hipMemcpy(heb.data(), eb, heb.size() * 4, hipMemcpyDeviceToHost));
    PHZ_HIP(ctx, hipMemcpy(hcf.data(), cfgv, hcf.size() * 4, hipMemcpyDeviceToHost));
    std::sort(exc.begin(), exc.end());
    exc.erase(std::unique(exc.begin(), exc.end()), exc.end());
    for (uint32_t c : exc) {
        const uint32_t m0 = cs[c], n = cs[c + 1] - m0, e0 = es[c], E = es[c + 1] - e0;
        std::vector<int32_t> ei(E), ej(E), sf(n + 1), sl(n + 1); std::vector<int8_t> ec(E); std::vector<char> cfg((size_t)n + 1);
        for (uint32_t t = 0; t < E; t++) {
            const uint32_t e = ek[e0 + t];
            ei[t] = (int32_t)(std::lower_bound(mem.begin() + m0, mem.begin() + m0 + n, (uint32_t)hea[e]) - (mem.begin() + m0));
            ej[t] = (int32_t)(std::lower_bound(mem.begin() + m0, mem.begin() + m0 + n, (uint32_t)heb[e]) - (mem.begin() + m0));
            ec[t] = (int8_t)hcf[e];
        }
        int32_t nsubs = 0;
        const int st = phz_phase_block((int32_t)n, (int64_t)E, ei.data(), ej.data(), ec.data(), max_block_size, sf.data(), sl.data(), cfg.data(), &nsubs);
        if (st != PHZ_OK) return phz_fail(ctx, st, "block phasing of a large component on the host");
        std::vector<int32_t> so(n, -1); std::vector<uint8_t> ao(n, 0);
        uint32_t ns = 0; size_t w = 0;
        for (int32_t k = 0; k < nsubs; k++) {
            if (sl[k] <= 0) continue;
            for (int32_t t = 0; t < sl[k]; t++) { so[(size_t)sf[k] + t] = (int32_t)ns; ao[(size_t)sf[k] + t] = (uint8_t)(cfg[w++] == '1'); }
            ns++;
        }
        PHZ_HIP(ctx, hipMemcpy(P<int32_t>(sub_of) + m0, so.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        PHZ_HIP(ctx, hipMemcpy(P<uint8_t>(alle_of) + m0, ao.data(), (size_t)n, hipMemcpyHostToDevice));
        PHZ_HIP(ctx, hipMemcpy(P<uint32_t>(nsub) + c, &ns, 4, hipMemcpyHostToDevice));
    }
    return PHZ_OK;
}

// Ordering rules 3 / 4 / 6 of SURVEY 8.1 that do NOT depend on the pair tests' p-values: the rank index of every variant (first appearance in the connectivity map:
// sort by (first line of the QNAME, distance to the variant's line)) and the order of the tested pairs by (rank a, rank b).  Enqueued by phz_rowsdev_pair_keys
// right after it has read the pair keys back, so that these ~14 sort passes run while the host evaluates scipy on the keys (they were the first thing
// phz_rowsdev_run did after that round trip), or by phz_rowsdev_run itself (PHZ_ROWS_NO_PRESTAGE=1, or a caller that skipped the first stage's result).
int order_variants_and_pairs(phz_ctx *ctx, phz_rowsdev *h, unsigned long long max_gap) {
    auto &T = ctx->tally;
    hipStream_t sm = ctx->stream;
    const int64_t nv = T.nv, ne = T.n_edges, n_lines = T.n_lines;
    const size_t NV = (size_t)(nv ? nv : 1), NE = (size_t)(ne ? ne : 1), NS = std::max(NV, NE);
#define RSV(buf, bytes) do { if (int s_ = phz_reserve(ctx, h->buf, (bytes))) return s_; } while (0)
    RSV(k64a, NS * 8); RSV(k64b, NS * 8); RSV(k32a, NS * 4); RSV(k32b, NS * 4); RSV(v32a, NS * 4); RSV(v32b, NS * 4);
    RSV(ridx, NV * 4); RSV(va, NE * 4); RSV(vb, NE * 4); RSV(eorder, NE * 4);
#undef RSV
    const int bv = bits_for((uint64_t)(nv > 1 ? nv - 1 : 1));
    int bl = bits_for((uint64_t)(n_lines > 1 ? n_lines - 1 : 1));
    if (const char *fb = getenv("PHZ_ROWS_FAKE_LINE_BITS")) bl = std::max(bl, std::min(32, atoi(fb)));      // tests: the key layout of a BAM with > 2^31 call lines
    if (nv) {
        const int gb = bits_for((uint64_t)(max_gap ? max_gap : 1));
        const uint32_t *rank_order = P<uint32_t>(h->k32a);
        if (gb + bl <= 32 && getenv("PHZ_ROWS_SORT64") == nullptr) {        // (first line, gap) in one 32-bit key: half the bytes per pass, one range
            hipLaunchKernelGGL(k_iota_rank32, dim3(nblk(nv)), dim3(256), 0, sm, nv, (const unsigned long long *)T.var_rank, gb, (uint32_t)((1ull << bl) - 1ull), P<uint32_t>(h->k32a), P<uint32_t>(h->v32a));
            const int rg[1][2] = {{0, gb + bl}};
            if (int s = sort_into<uint32_t>(ctx, h, h->k32a, h->k32b, nv, rg, 1, P<uint32_t>(h->k64a), nullptr)) return s;      // k64a (as uint32): variants in rank order
            rank_order = P<uint32_t>(h->k64a);
        } else {
            hipLaunchKernelGGL(k_iota_rank, dim3(nblk(nv)), dim3(256), 0, sm, nv, (const unsigned long long *)T.var_rank, P<unsigned long long>(h->k64a), P<uint32_t>(h->v32a));
            const int rg[2][2] = {{0, gb}, {32, 32 + bl}};      // (first line of the QNAME, distance to the variant's line)
            if (int s = sort_into<unsigned long long>(ctx, h, h->k64a, h->k64b, nv, rg, 2, P<uint32_t>(h->k32a), nullptr)) return s;      // k32a: variants in rank order
        }
        hipLaunchKernelGGL(k_invert, dim3(nblk(nv)), dim3(256), 0, sm, nv, rank_order, P<uint32_t>(h->ridx));
    }
    if (ne) {
        hipLaunchKernelGGL(k_edge_keys, dim3(nblk(ne)), dim3(256), 0, sm, ne, (const uint8_t *)T.linked, (const int32_t *)T.ea, (const int32_t *)T.eb,
                           (const uint32_t *)h->ridx.p, P<int32_t>(h->va), P<int32_t>(h->vb), P<unsigned long long>(h->k64a), P<uint32_t>(h->v32a));
        const int rg[2][2] = {{0, bv}, {32, 32 + bv + 1}};
        if (int s = sort_into<unsigned long long>(ctx, h, h->k64a, h->k64b, ne, rg, 2, P<uint32_t>(h->eorder), nullptr)) return s;
    }
    PHZ_HIP(ctx, hipGetLastError());
    return PHZ_OK;
}

// First-appearance keys (ordering rule 5 of SURVEY 8.1: allelic_counts and the singleton rows list the covered variants by the BAM and line of their first kept call):
// compaction, sort by (BAM of the first line, line), segment starts per (BAM, chromosome) and their counts.  p-value-independent: enqueued by the first stage when the
// handle knows the tally's shards, by phz_rowsdev_run otherwise.  ss_keys / cc_keys: the key segments of seg_start / chrom_cnt (cleared by the caller).
int key_stage(phz_ctx *ctx, phz_rowsdev *h, int64_t nkeys, const long long *d_sh_lo, const long long *d_sh_hi, const int32_t *d_sh_bam, int n_shards, uint32_t *ss_keys, uint32_t *cc_keys) {
    auto &T = ctx->tally;
    hipStream_t sm = ctx->stream;
    const int64_t nv = T.nv, n_lines = T.n_lines;
    const int nb = T.nb, nchrom = h->nchrom;
    const size_t NV = (size_t)(nv ? nv : 1), NE = (size_t)(T.n_edges ? T.n_edges : 1), NS = std::max(NV, NE);
#define RSV(buf, bytes) do { if (int s_ = phz_reserve(ctx, h->buf, (bytes))) return s_; } while (0)
    RSV(k64a, NS * 8); RSV(k64b, NS * 8); RSV(k32a, NS * 4); RSV(k32b, NS * 4); RSV(v32a, NS * 4); RSV(v32b, NS * 4);
    RSV(key_g, (size_t)(nkeys + 1) * 4);
    int bl = bits_for((uint64_t)(n_lines > 1 ? n_lines - 1 : 1));
    if (const char *fb = getenv("PHZ_ROWS_FAKE_LINE_BITS")) bl = std::max(bl, std::min(32, atoi(fb)));      // tests: the key layout of a BAM with > 2^31 call lines
    if (nkeys) {
        ShardTab ST; ST.lo = d_sh_lo; ST.hi = d_sh_hi; ST.bam = d_sh_bam; ST.n = n_shards;
        const int bb = nb > 1 ? bits_for((uint64_t)(nb - 1)) : 0;
        RSV(key64s, (size_t)(nkeys + 1) * 8);
        if (bl < 32 && bb + bl <= 32 && getenv("PHZ_ROWS_SORT64") == nullptr) {        // (BAM, first line) in one 32-bit key; bl == 32 (one BAM of > 2^31 lines) would make the kernels shift a 32-bit word by 32
            hipLaunchKernelGGL(k_compact_keys32, dim3(nblk(nv)), dim3(256), 0, sm, nv, (const long long *)T.var_first, (const uint32_t *)h->keypos.p, ST, bl,
                               P<uint32_t>(h->k32a), P<uint32_t>(h->v32a));
            const int rg[1][2] = {{0, bb + bl}};
            if (int s = sort_into<uint32_t>(ctx, h, h->k32a, h->k32b, nkeys, rg, 1, P<uint32_t>(h->key_g), P<uint32_t>(h->key64s))) return s;
            hipLaunchKernelGGL(k_key_starts32, dim3(nblk(nkeys)), dim3(256), 0, sm, nkeys, (const uint32_t *)h->key64s.p, bl, (const uint32_t *)h->key_g.p,
                               (const uint16_t *)h->d_vchrom.p, nchrom, ss_keys);
        } else {
            hipLaunchKernelGGL(k_compact_keys, dim3(nblk(nv)), dim3(256), 0, sm, nv, (const long long *)T.var_first, (const uint32_t *)h->keypos.p, ST,
                               P<unsigned long long>(h->k64a), P<uint32_t>(h->v32a));
            const int rg[2][2] = {{0, bits_for((uint64_t)(n_lines > 1 ? n_lines - 1 : 1))}, {32, 32 + (nb > 1 ? bits_for((uint64_t)(nb - 1)) : 0)}};
            if (int s = sort_into<unsigned long long>(ctx, h, h->k64a, h->k64b, nkeys, rg, 2, P<uint32_t>(h->key_g), P<unsigned long long>(h->key64s))) return s;
            hipLaunchKernelGGL(k_key_starts, dim3(nblk(nkeys)), dim3(256), 0, sm, nkeys, (const unsigned long long *)h->key64s.p, (const uint32_t *)h->key_g.p,
                               (const uint16_t *)h->d_vchrom.p, nchrom, ss_keys);
        }
    }
#undef RSV
    hipLaunchKernelGGL(k_starts_to_counts, dim3(1), dim3(1), 0, sm, ss_keys, nb * nchrom, (uint32_t)nkeys, cc_keys);
    PHZ_HIP(ctx, hipGetLastError());
    return PHZ_OK;
}

// layout of the per-chromosome counters / segment starts (chrom_cnt, seg_start): uint32 [conn | blocks | block vars | keys per (BAM, chromosome)], then uint64 cfg rows per chromosome
inline size_t n_chrom_counters(int nchrom, int nb) { return (size_t)nchrom * 3 + (size_t)nb * nchrom; }

}  // namespace

// The (chromosome, BAM) shards of the phz_tally call whose results the next phz_rowsdev_pair_keys will read: line range and BAM of every shard, in line order (what
// phz_rowsdev_opts.shard_* carry to phz_rowsdev_run).  With them the first stage also enqueues the first-appearance keys (they need no p-value).
extern "C" int phz_rowsdev_set_shards(phz_rowsdev *h, int32_t n_shards, const int64_t *line_lo, const int64_t *line_hi, const int32_t *bam) {
    if (!h || n_shards < 0 || (n_shards && (!line_lo || !line_hi || !bam))) return PHZ_E_ARG;
    h->sh_lo.assign(line_lo, line_lo + n_shards); h->sh_hi.assign(line_hi, line_hi + n_shards); h->sh_bam.assign(bam, bam + n_shards);
    h->have_shards = true; h->pre_keys = false;
    return PHZ_OK;
}

extern "C" int phz_rowsdev_create(phz_ctx *ctx, const phz_rowsdev_tables *t, phz_rowsdev **out) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !t || !out || t->nv < 0 || t->n_chroms < 0 || t->n_chroms > 65535) return PHZ_E_ARG;
    if (t->nv >= (1ll << 28)) return phz_fail(ctx, PHZ_E_ARG, "more than 2^28 variants");
    *out = nullptr;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    phz_rowsdev *h = new (std::nothrow) phz_rowsdev();
    if (!h) return PHZ_E_NOMEM;
    const size_t nv = (size_t)t->nv;
    h->nv = t->nv; h->nchrom = t->n_chroms;
    h->chrom_v0.assign(t->chrom_v0, t->chrom_v0 + t->n_chroms + 1);
    int st = PHZ_OK;
    auto U = [&](DevBuf &b, const void *src, size_t bytes) { if (st == PHZ_OK) st = up(ctx, b, src, bytes); };
    U(h->d_chrom_v0, h->chrom_v0.data(), (size_t)(t->n_chroms + 1) * 8);
    U(h->d_pos, t->pos, nv * 4); U(h->d_maf, t->maf, nv * 8); U(h->d_isref, t->is_ref, 2 * nv); U(h->d_phase, t->phase_idx, 2 * nv);
    h->has_black = t->blacklisted != nullptr;
    if (h->has_black) U(h->d_black, t->blacklisted, nv);
    const uint32_t *offs[5] = {t->uid_off, t->rsid_off, t->allele_off, t->maf_off, t->chrom_name_off};
    const char *txts[5] = {t->uid, t->rsid, t->allele, t->maf_txt, t->chrom_names};
    const size_t cnts[5] = {nv, nv, 2 * nv, nv, (size_t)t->n_chroms};
    for (int i = 0; i < 5; i++) { U(h->p_off[i], offs[i], (cnts[i] + 1) * 4); U(h->p_txt[i], txts[i], (size_t)offs[i][cnts[i]]); }
    // gwStat text of a block whose annotated phases are not all equal (:968-980): m = mean of the known phase indices, printed max(m, 1 - m)
    // as str(numpy.float64).  Entry [nknown * (STAT_N + 1) + ksum]; the same float64 operations as numpy.mean of 0/1 values and 1 - m
    {
        std::string txt; std::vector<uint32_t> off((size_t)(STAT_N + 1) * (STAT_N + 1) + 1);
        size_t w = 0;
        for (int nk = 0; nk <= STAT_N; nk++)
            for (int ks = 0; ks <= STAT_N; ks++) {
                off[w++] = (uint32_t)txt.size();
                if (nk > 0 && ks <= nk) { const double m = (double)ks / (double)nk, o = 1 - m; phztext::put_pyfloat(txt, m >= o ? m : o); }
                txt += '\n';
            }
        off[w] = (uint32_t)txt.size();
        U(h->p_off[5], off.data(), off.size() * 4); U(h->p_txt[5], txt.data(), txt.size());
        if (st == PHZ_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = PHZ_E_HIP;     // txt / off die with this scope
    }
    if (st == PHZ_OK) st = phz_reserve(ctx, h->d_vchrom, nv * 2 + 2);
    if (st == PHZ_OK && nv) hipLaunchKernelGGL(k_vchrom, dim3(nblk(t->nv)), dim3(256), 0, ctx->stream, t->nv, (const long long *)h->d_chrom_v0.p, t->n_chroms, P<uint16_t>(h->d_vchrom));
    if (st == PHZ_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = phz_fail(ctx, PHZ_E_HIP, "phz_rowsdev_create");
    if (st != PHZ_OK) { phz_rowsdev_destroy(h); return st; }
    *out = h;
    return PHZ_OK;
}

extern "C" void phz_rowsdev_destroy(phz_rowsdev *h) {
    if (!h) return;
    for (DevBuf *b : h->all()) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
    if (h->up_host.p) (void)hipHostFree(h->up_host.p);
    if (h->sh_host.p) (void)hipHostFree(h->sh_host.p);
    for (hipEvent_t e : h->txt_ev) if (e) (void)hipEventDestroy(e);
    delete h;
}

// Stage 1: the distinct (total, supporting) argument pairs of the binomial test over the tested pairs of the last phz_tally.  keys_host
// [phz_rowsdev_pair_slots(h)] receives the hash set as it lives on the device: slot s holds (total << 32 | supporting) or all ones when empty.  The caller
// evaluates the p-value of every occupied slot (the reference's scipy call) and passes values and their text to phz_rowsdev_run by slot.
// Size of the pair-key hash set (a power of two in [16, 2^28]; anything below the default 2^16 is for tests); keys_host / slot_pv / slot_txt_off of the two stages are sized by it.
extern "C" int phz_rowsdev_set_pair_slots(phz_rowsdev *h, int64_t n_slots) {
    if (!h || n_slots < 16 || n_slots > (1ll << 28) || (n_slots & (n_slots - 1))) return PHZ_E_ARG;
    h->ps_slots = n_slots; h->keys_ready = false;
    return PHZ_OK;
}
extern "C" int64_t phz_rowsdev_pair_slots(const phz_rowsdev *h) { return h ? h->ps_slots : 0; }

extern "C" int phz_rowsdev_pair_keys(phz_ctx *ctx, phz_rowsdev *h, uint64_t *keys_host) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !h || !keys_host) return PHZ_E_ARG;
    auto &T = ctx->tally;
    if (T.nv != h->nv) return phz_fail(ctx, PHZ_E_ARG, "phz_rowsdev: the resident tally does not belong to these variant tables");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t slots = (size_t)h->ps_slots;
    if (int s = phz_reserve(ctx, h->hkeys, slots * 8)) return s;
    if (int s = phz_reserve(ctx, h->flags, 64)) return s;
    hipStream_t sm = ctx->stream;
    PHZ_HIP(ctx, hipMemsetAsync(h->hkeys.p, 0xff, slots * 8, sm));
    PHZ_HIP(ctx, hipMemsetAsync(h->flags.p, 0, 64, sm));
    const int64_t ne = T.n_edges, nv = T.nv;
    if (ne > 0) hipLaunchKernelGGL(k_pair_keys, dim3(nblk(ne)), dim3(256), 0, sm, ne, (const uint8_t *)T.linked, (const int32_t *)(T.stats + 2 * ne),
                                   (const int32_t *)(T.stats + 3 * ne), P<unsigned long long>(h->hkeys), (uint32_t)(slots - 1), P<uint32_t>(h->flags));
    PHZ_HIP(ctx, hipGetLastError());
    // the first p-value-independent step of the row stage rides on this call's host wait: which variants are covered (the keys of allelic_counts / the singleton
    // rows) and the largest (QNAME line, variant line) gap, which decides the key width of the rank sort
    h->pre_done = false; h->pre_keys = false;
    const bool pre = getenv("PHZ_ROWS_NO_PRESTAGE") == nullptr && nv > 0;
    const bool pre_k = pre && h->have_shards && getenv("PHZ_ROWS_NO_PREKEYS") == nullptr;
    unsigned long long h_gap = 0; uint32_t h_nkeys = 0;
    if (pre) {
        const size_t NV = (size_t)nv, NE = (size_t)(ne ? ne : 1);
        if (int s = phz_reserve(ctx, h->cnt64, 64)) return s;
        if (int s = phz_reserve(ctx, h->f_d, NV * 4)) return s;
        if (int s = phz_reserve(ctx, h->keypos, (NV + 1) * 4)) return s;
        if (int s = phz_reserve(ctx, h->f_a, std::max(NV, NE) * 4)) return s;          // (the sizes phz_rowsdev_run asks for: its own reservations must not move these buffers)
        PHZ_HIP(ctx, hipMemsetAsync(h->cnt64.p, 0, 64, sm));
        hipLaunchKernelGGL(k_flag_keys, dim3(std::min(nblk(nv), 512u)), dim3(256), 0, sm, nv, (const long long *)T.var_first, P<uint32_t>(h->f_d), (const unsigned long long *)T.var_rank,
                           P<unsigned long long>(h->cnt64) + 2);
        if (int s = gscan_excl<uint32_t, uint32_t>(ctx, P<uint32_t>(h->f_d), P<uint32_t>(h->keypos), nv, h->scan_tmp)) return s;
    }
    uint32_t fl = 0;
    {
        PhzMail mail(ctx);
        const int m_gap = pre ? mail.add(P<unsigned long long>(h->cnt64) + 2, 8) : -1, m_fl = mail.add(h->flags.p, 4);
        const int m_nk = pre_k ? mail.add(P<uint32_t>(h->keypos) + nv, 4) : -1;
        if (int s = mail.send()) return s;
        PHZ_HIP(ctx, hipMemcpyAsync(keys_host, h->hkeys.p, slots * 8, hipMemcpyDeviceToHost, sm));          // (copied before the verdict on the table is known -- one wait instead of two;
        PHZ_HIP(ctx, hipStreamSynchronize(sm));                                                             //  the caller hands over page-locked memory)
        if (pre) h_gap = *mail.at<unsigned long long>(m_gap);
        if (pre_k) h_nkeys = *mail.at<uint32_t>(m_nk);
        fl = *mail.at<uint32_t>(m_fl);
    }
    if (fl & 1u) return phz_fail(ctx, PHZ_E_CAPACITY, "the distinct (supporting, total) read-count pairs do not fit the pair-key table: grow it (phz_rowsdev_set_pair_slots) and call again");
    if (pre) {
        // ... and the ordering sorts that need no p-value are on the stream before this call returns: they run while the caller evaluates scipy on the keys
        if (int s = order_variants_and_pairs(ctx, h, h_gap)) return s;
        h->pre_done = true; h->pre_max_gap = h_gap; h->pre_gen = ctx->tally_gen;
        if (pre_k) {
            // ... and the first-appearance keys: the per-chromosome counter / segment-start tables are cleared HERE (phz_rowsdev_run leaves them alone when it finds the keys done)
            const int nchrom = h->nchrom, nb = T.nb, nsh = (int)h->sh_bam.size();
            const size_t n_cc = n_chrom_counters(nchrom, nb);
            if (int s = phz_reserve(ctx, h->chrom_cnt, n_cc * 4 + 8 + (size_t)nchrom * 8)) return s;
            if (int s = phz_reserve(ctx, h->seg_start, (n_cc + 1) * 4)) return s;
            PHZ_HIP(ctx, hipMemsetAsync(h->chrom_cnt.p, 0, h->chrom_cnt.cap, sm));
            PHZ_HIP(ctx, hipMemsetAsync(h->seg_start.p, 0xff, (n_cc + 1) * 4, sm));
            const size_t one = ((size_t)nsh * 8 + 255) & ~(size_t)255;
            if (int s = phz_reserve_host(ctx, h->sh_host, 3 * one + 256)) return s;          // (page-locked, the handle's own: the copy below may still be queued when the caller comes back)
            if (int s = phz_reserve(ctx, h->sh_dev, 3 * one + 256)) return s;
            char *hp = (char *)h->sh_host.p;
            for (int i = 0; i < nsh; i++) { ((long long *)hp)[i] = h->sh_lo[(size_t)i]; ((long long *)(hp + one))[i] = h->sh_hi[(size_t)i]; ((int32_t *)(hp + 2 * one))[i] = h->sh_bam[(size_t)i]; }
            PHZ_HIP(ctx, hipMemcpyAsync(h->sh_dev.p, hp, 3 * one, hipMemcpyHostToDevice, sm));
            const char *dp = (const char *)h->sh_dev.p;
            uint32_t *cc = P<uint32_t>(h->chrom_cnt), *ss = P<uint32_t>(h->seg_start);
            if (int s = key_stage(ctx, h, (int64_t)h_nkeys, (const long long *)dp, (const long long *)(dp + one), (const int32_t *)(dp + 2 * one), nsh,
                                  ss + 3 * (size_t)nchrom, cc + 3 * (size_t)nchrom)) return s;
            h->pre_keys = true; h->pre_nkeys = (int64_t)h_nkeys;
        }
    }
    h->keys_ready = true;
    return PHZ_OK;
}

// phz_tally and stage 1 behind each other inside ONE native call: between the tally's last kernel and k_pair_keys there was nothing but the caller's glue (the return,
// a few Python frames, the next call's prologue: ~0.1 ms of idle GPU per pass).  *pair_status = what phz_rowsdev_pair_keys answered (PHZ_E_CAPACITY: grow the
// table and call phz_rowsdev_pair_keys alone); the tally is complete and resident whatever it says.
extern "C" int phz_tally_pairs(phz_ctx *ctx, const phz_lines *shards, int n_shards, int64_t nv, const uint8_t *a0, const uint8_t *a1, int64_t n_qid, int n_bams,
                               phz_tally_sizes *sizes, int space, phz_rowsdev *h, uint64_t *keys_host, int32_t *pair_status) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !h || !keys_host || !pair_status) return PHZ_E_ARG;
    *pair_status = PHZ_E_ARG;
    if (int s = phz_tally(ctx, shards, n_shards, nv, a0, a1, n_qid, n_bams, sizes, space)) return s;
    {   // the shards of this tally, in line order: the first stage can then put the first-appearance keys on the stream as well
        std::vector<int64_t> lo((size_t)n_shards), hi((size_t)n_shards); std::vector<int32_t> bam((size_t)n_shards);
        int64_t total = 0;
        for (int i = 0; i < n_shards; i++) { lo[(size_t)i] = total; total += shards[i].n_calls; hi[(size_t)i] = total; bam[(size_t)i] = shards[i].bam_index; }
        if (int s = phz_rowsdev_set_shards(h, n_shards, lo.data(), hi.data(), bam.data())) return s;
    }
    *pair_status = phz_rowsdev_pair_keys(ctx, h, keys_host);
    return PHZ_OK;
}

// Stage 2: everything else, up to the finished text in HBM.
extern "C" int phz_rowsdev_run(phz_ctx *ctx, phz_rowsdev *h, const phz_rowsdev_opts *o, const double *slot_pv, const uint32_t *slot_txt_off,
                               const char *slot_txt, phz_rowsdev_result *res) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !h || !o || !slot_pv || !slot_txt_off || !slot_txt || !res) return PHZ_E_ARG;
    auto &T = ctx->tally;
    if (!h->keys_ready) return phz_fail(ctx, PHZ_E_ARG, "phz_rowsdev_run without phz_rowsdev_pair_keys");
    h->keys_ready = false;
    if (h->pre_done && h->pre_gen != ctx->tally_gen) return phz_fail(ctx, PHZ_E_ARG, "phz_rowsdev_run: the resident tally changed since phz_rowsdev_pair_keys (call it again)");
    // the first-appearance keys of the first stage count only for exactly the shards this call names
    bool pre_keys = h->pre_keys && h->pre_done && (int)h->sh_bam.size() == o->n_shards;
    for (int i = 0; pre_keys && i < o->n_shards; i++)
        pre_keys = h->sh_lo[(size_t)i] == (long long)o->shard_line_lo[i] && h->sh_hi[(size_t)i] == (long long)o->shard_line_hi[i] && h->sh_bam[(size_t)i] == o->shard_bam[i];
    h->pre_keys = false;
    if (T.nv != h->nv || T.nb != o->n_bams) return phz_fail(ctx, PHZ_E_ARG, "phz_rowsdev: the resident tally does not match (variants / BAMs)");
    if (o->gw_phase_method != 0 && o->gw_phase_method != 1) return phz_fail(ctx, PHZ_E_ARG, "device row stage: gw_phase_method must be 0 or 1");
    const bool read_ids = o->output_read_ids != 0;
    if (read_ids && (!o->qname_off || !o->qname || !o->qname_base)) return phz_fail(ctx, PHZ_E_ARG, "device row stage: --output_read_ids 1 needs the QNAME pool (qname_off / qname / qname_base)");
    if (!T.rl_list && T.n_rl) return phz_fail(ctx, PHZ_E_ARG, "phz_rowsdev: the resident tally carries no list index per read-list entry");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    memset(res, 0, sizeof(*res));
    hipStream_t sm = ctx->stream;
    const int64_t nv = T.nv, ne = T.n_edges, n_rl = T.n_rl, n_lines = T.n_lines;
    const int nb = T.nb, nchrom = h->nchrom;
    const size_t NV = (size_t)(nv ? nv : 1), NE = (size_t)(ne ? ne : 1), NRL = NV * 2 * (size_t)nb, NR = (size_t)(n_rl ? n_rl : 1);
    const int32_t *cis = T.stats, *trans = T.stats + ne, *sup = T.stats + 2 * ne, *tot = T.stats + 3 * ne, *cfgv = T.stats + 4 * ne;
    Sections sec(ctx);
    sec.begin();
    // ---- per-pass uploads
    const size_t ps_slots = (size_t)h->ps_slots;
    // ONE copy for all of them: gathered in a page-locked block (a hipMemcpyAsync from the caller's pageable arrays is staged synchronously, 10-20 us each
    // -- nine of them were a third of this stage's first section, which is bound by what the host can enqueue), 256-byte aligned pieces of one device block
    const void *up_src[9] = {slot_pv, slot_txt_off, slot_txt, o->bam_name_off, o->bam_names, o->bam_excluded, o->shard_line_lo, o->shard_line_hi, o->shard_bam};
    const size_t up_n[9] = {ps_slots * 8, (ps_slots + 1) * 4, (size_t)slot_txt_off[ps_slots], (size_t)(nb + 1) * 4, (size_t)o->bam_name_off[nb], o->bam_excluded ? (size_t)nb : 0,
                            (size_t)o->n_shards * 8, (size_t)o->n_shards * 8, (size_t)o->n_shards * 4};
    size_t up_off[10]; up_off[0] = 0;
    for (int i = 0; i < 9; i++) up_off[i + 1] = up_off[i] + ((up_n[i] + 255) & ~(size_t)255);
    if (int s = phz_reserve_host(ctx, h->up_host, up_off[9] + 256)) return s;
    if (int s = phz_reserve(ctx, h->up_dev, up_off[9] + 256)) return s;
    for (int i = 0; i < 9; i++) if (up_n[i]) memcpy((char *)h->up_host.p + up_off[i], up_src[i], up_n[i]);
    PHZ_HIP(ctx, hipMemcpyAsync(h->up_dev.p, h->up_host.p, up_off[9], hipMemcpyHostToDevice, sm));
    const char *upd = (const char *)h->up_dev.p;
    const double *d_slot_pv = (const double *)(upd + up_off[0]); const uint32_t *d_pv_off = (const uint32_t *)(upd + up_off[1]); const char *d_pv_txt = upd + up_off[2];
    const uint32_t *d_bam_off = (const uint32_t *)(upd + up_off[3]); const char *d_bam_txt = upd + up_off[4]; const uint8_t *d_bam_excl = (const uint8_t *)(upd + up_off[5]);
    const long long *d_sh_lo = (const long long *)(upd + up_off[6]), *d_sh_hi = (const long long *)(upd + up_off[7]); const int32_t *d_sh_bam = (const int32_t *)(upd + up_off[8]);
    if (read_ids) {          // (a debugging option: plain uploads)
        const size_t nq = (size_t)o->qname_base[nchrom];
        if (int s = up(ctx, h->qn_off, o->qname_off, (nq + 1) * 4)) return s;
        if (int s = up(ctx, h->qn_txt, o->qname, (size_t)o->qname_off[nq])) return s;
        if (int s = up(ctx, h->qn_base, o->qname_base, (size_t)(nchrom + 1) * 8)) return s;
        PHZ_HIP(ctx, hipStreamSynchronize(sm));          // the caller's arrays are pageable
    }
    // ---- pruning (:686-700) + components
#define RSV(buf, bytes) do { if (int s_ = phz_reserve(ctx, h->buf, (bytes))) return s_; } while (0)
    RSV(keep, NE); RSV(e_slot, NE * 4); RSV(deg, NV * 4); RSV(parent, NV * 4); RSV(label, NV * 4);
    RSV(cnt64, 64); RSV(cnt32, 512);          // cnt32: [0..15] counters of the stages, [16] ticket of k_phase_general, [20] pool overflow of the read-set stage, [32 + 16 m ..] counters and tickets of its mode m
    const size_t n_cc = (size_t)nchrom * 3 + (size_t)nb * nchrom;          // uint32 counters per chromosome, then uint64 cfg rows per chromosome
    RSV(chrom_cnt, n_cc * 4 + 8 + (size_t)nchrom * 8);
    uint32_t *cc_conn = P<uint32_t>(h->chrom_cnt), *cc_blocks = cc_conn + nchrom, *cc_blkvars = cc_blocks + nchrom, *cc_keys = cc_blkvars + nchrom;
    unsigned long long *cc_cfg = (unsigned long long *)((char *)h->chrom_cnt.p + ((n_cc * 4 + 7) & ~(size_t)7));
    if (!pre_keys) PHZ_HIP(ctx, hipMemsetAsync(h->chrom_cnt.p, 0, h->chrom_cnt.cap, sm));          // (pre_keys: cleared by the first stage, the key counters are in)
    RSV(seg_start, (n_cc + 1) * 4);            // first row of every segment, same layout as the counters
    uint32_t *ss_conn = P<uint32_t>(h->seg_start), *ss_blocks = ss_conn + nchrom, *ss_keys = ss_blocks + 2 * nchrom;
    if (!pre_keys) PHZ_HIP(ctx, hipMemsetAsync(h->seg_start.p, 0xff, (n_cc + 1) * 4, sm));
    PHZ_HIP(ctx, hipMemsetAsync(h->deg.p, 0, NV * 4, sm));
    PHZ_HIP(ctx, hipMemsetAsync(h->cnt64.p, 0, 64, sm));
    PHZ_HIP(ctx, hipMemsetAsync(h->cnt32.p, 0, 512, sm));
    unsigned long long *cnt64 = P<unsigned long long>(h->cnt64);
    uint32_t *cnt32 = P<uint32_t>(h->cnt32);
    if (ne) hipLaunchKernelGGL(k_keep, dim3(std::min(nblk(ne), 2048u)), dim3(256), 0, sm, ne, (const uint8_t *)T.linked, sup, tot, (const unsigned long long *)h->hkeys.p, (uint32_t)(ps_slots - 1),
                               d_slot_pv, o->cc_threshold, P<uint8_t>(h->keep), P<uint32_t>(h->e_slot), P<uint32_t>(h->deg), (const int32_t *)T.ea,
                               (const int32_t *)T.eb, cnt64);
    if (nv) hipLaunchKernelGGL(k_uf_init, dim3(nblk(nv)), dim3(256), 0, sm, P<int32_t>(h->parent), nv);
    if (ne) hipLaunchKernelGGL(k_uf_hook, dim3(nblk(ne)), dim3(256), 0, sm, P<int32_t>(h->parent), (const int32_t *)T.ea, (const int32_t *)T.eb, (const uint8_t *)h->keep.p, ne);
    if (nv) hipLaunchKernelGGL(k_uf_flatten, dim3(nblk(nv)), dim3(256), 0, sm, P<int32_t>(h->parent), P<int32_t>(h->label), nv);
    // ---- sizes of the compacted lists: members, components, kept pairs, first-appearance keys
    RSV(f_a, std::max(NV, NE) * 4); RSV(f_b, NV * 4); RSV(f_c, NE * 4); RSV(f_d, NV * 4);
    RSV(mem_pos, (NV + 1) * 4); RSV(cid, (NV + 1) * 4); RSV(kpos, (NE + 1) * 4); RSV(keypos, (NV + 1) * 4);
    if (nv) hipLaunchKernelGGL(k_flag_members, dim3(nblk(nv)), dim3(256), 0, sm, nv, (const uint32_t *)h->deg.p, (const int32_t *)h->label.p, P<uint32_t>(h->f_a), P<uint32_t>(h->f_b));
    if (ne) hipLaunchKernelGGL(k_flag_u8, dim3(nblk(ne)), dim3(256), 0, sm, ne, (const uint8_t *)h->keep.p, P<uint32_t>(h->f_c));
    if (nv && !h->pre_done) hipLaunchKernelGGL(k_flag_keys, dim3(std::min(nblk(nv), 512u)), dim3(256), 0, sm, nv, (const long long *)T.var_first, P<uint32_t>(h->f_d), (const unsigned long long *)T.var_rank, cnt64 + 2);
    if (int s = gscan_excl<uint32_t, uint32_t>(ctx, P<uint32_t>(h->f_a), P<uint32_t>(h->mem_pos), nv, h->scan_tmp)) return s;
    if (int s = gscan_excl<uint32_t, uint32_t>(ctx, P<uint32_t>(h->f_b), P<uint32_t>(h->cid), nv, h->scan_tmp)) return s;
    if (int s = gscan_excl<uint32_t, uint32_t>(ctx, P<uint32_t>(h->f_c), P<uint32_t>(h->kpos), ne, h->scan_tmp)) return s;
    if (!h->pre_done) { if (int s = gscan_excl<uint32_t, uint32_t>(ctx, P<uint32_t>(h->f_d), P<uint32_t>(h->keypos), nv, h->scan_tmp)) return s; }
    PHZ_HIP(ctx, hipGetLastError());
    uint32_t h_n[4] = {0, 0, 0, 0}; unsigned long long h_c64[8] = {0};
    {
        PhzMail mail(ctx);
        const int m[5] = {mail.add(P<uint32_t>(h->mem_pos) + nv, 4), mail.add(P<uint32_t>(h->cid) + nv, 4), mail.add(P<uint32_t>(h->kpos) + ne, 4),
                          mail.add(P<uint32_t>(h->keypos) + nv, 4), mail.add(cnt64, 24)};
        if (int s = mail.send()) return s;
        if (int s = sec.wait("sizes: members/comps/kept/keys")) return s;
        for (int i = 0; i < 4; i++) h_n[i] = *mail.at<uint32_t>(m[i]);
        memcpy(h_c64, mail.at<char>(m[4]), 24);
    }
    if (h->pre_done) h_c64[2] = h->pre_max_gap;          // (measured by the first stage; this run's counters were cleared after it)
    const int64_t nmem = h_n[0], ncomp = h_n[1], nkeep = h_n[2], nkeys = h_n[3], n_linked = (int64_t)h_c64[0];
    res->dropped = (int64_t)h_c64[1];
    sec.begin();
    // ---- ordering stage: rank index of every variant, pair order, members by component, component order, kept pairs by component, keys
    const int bv = bits_for((uint64_t)(nv > 1 ? nv - 1 : 1));
    int bl = bits_for((uint64_t)(n_lines > 1 ? n_lines - 1 : 1));
    if (const char *fb = getenv("PHZ_ROWS_FAKE_LINE_BITS")) bl = std::max(bl, std::min(32, atoi(fb)));      // tests: the key layout of a BAM with > 2^31 call lines
    const size_t NS = std::max(NV, NE);
    RSV(k64a, NS * 8); RSV(k64b, NS * 8); RSV(k32a, NS * 4); RSV(k32b, NS * 4); RSV(v32a, NS * 4); RSV(v32b, NS * 4);
    RSV(ridx, NV * 4); RSV(va, NE * 4); RSV(vb, NE * 4); RSV(eorder, NE * 4);
    RSV(mem_s, (size_t)(nmem + 1) * 4); RSV(cstart, (size_t)(ncomp + 2) * 4); RSV(corder, (size_t)(ncomp + 1) * 4); RSV(ekeep, (size_t)(nkeep + 1) * 4);
    RSV(estart, (size_t)(ncomp + 2) * 4); RSV(key_g, (size_t)(nkeys + 1) * 4);
    if (!h->pre_done) { if (int s = order_variants_and_pairs(ctx, h, h_c64[2])) return s; }
    h->pre_done = false;
    if (ne) {
        if (n_linked) hipLaunchKernelGGL(k_conn_starts, dim3(nblk(n_linked)), dim3(256), 0, sm, n_linked, (const uint32_t *)h->eorder.p, (const int32_t *)h->va.p,
                                         (const uint16_t *)h->d_vchrom.p, ss_conn);
    }
    hipLaunchKernelGGL(k_starts_to_counts, dim3(1), dim3(1), 0, sm, ss_conn, nchrom, (uint32_t)n_linked, cc_conn);
    {
    }
    if (nmem) {
        hipLaunchKernelGGL(k_compact_members, dim3(nblk(nv)), dim3(256), 0, sm, nv, (const uint32_t *)h->deg.p, (const uint32_t *)h->mem_pos.p, (const int32_t *)h->label.p,
                           P<uint32_t>(h->k32a), P<uint32_t>(h->v32a));
        const int rg[1][2] = {{0, bv}};
        if (int s = sort_into<uint32_t>(ctx, h, h->k32a, h->k32b, nmem, rg, 1, P<uint32_t>(h->mem_s), P<uint32_t>(h->f_a))) return s;         // f_a: sorted labels
        hipLaunchKernelGGL(k_group_starts, dim3(nblk(nmem)), dim3(256), 0, sm, nmem, (const uint32_t *)h->f_a.p, (const uint32_t *)h->cid.p, P<uint32_t>(h->cstart));
        hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(256), 0, sm, P<uint32_t>(h->cstart) + ncomp, (int64_t)1, (uint32_t)nmem);
        hipLaunchKernelGGL(k_comp_min, dim3(nblk(ncomp)), dim3(256), 0, sm, ncomp, (const uint32_t *)h->cstart.p, (const uint32_t *)h->mem_s.p, (const uint32_t *)h->ridx.p,
                           P<uint32_t>(h->k32a), P<uint32_t>(h->v32a));
        if (int s = sort_into<uint32_t>(ctx, h, h->k32a, h->k32b, ncomp, rg, 1, P<uint32_t>(h->corder), nullptr)) return s;
        hipLaunchKernelGGL(k_compact_kept, dim3(nblk(ne)), dim3(256), 0, sm, ne, (const uint8_t *)h->keep.p, (const uint32_t *)h->kpos.p, (const int32_t *)T.ea,
                           (const int32_t *)h->label.p, (const uint32_t *)h->cid.p, P<uint32_t>(h->k32a), P<uint32_t>(h->v32a));
        const int rgc[1][2] = {{0, bits_for((uint64_t)(ncomp > 1 ? ncomp - 1 : 1))}};
        if (int s = sort_into<uint32_t>(ctx, h, h->k32a, h->k32b, nkeep, rgc, 1, P<uint32_t>(h->ekeep), P<uint32_t>(h->f_a))) return s;        // f_a: component of every kept pair
        hipLaunchKernelGGL(k_group_starts, dim3(nblk(nkeep)), dim3(256), 0, sm, nkeep, (const uint32_t *)h->f_a.p, (const uint32_t *)nullptr, P<uint32_t>(h->estart));
        hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(256), 0, sm, P<uint32_t>(h->estart) + ncomp, (int64_t)1, (uint32_t)nkeep);
    }
    if (!pre_keys) {          // (else: enqueued by the first stage, under the caller's p-value evaluation)
        if (int s = key_stage(ctx, h, nkeys, d_sh_lo, d_sh_hi, d_sh_bam, o->n_shards, ss_keys, cc_keys)) return s;
    } else if (nkeys != h->pre_nkeys) return phz_fail(ctx, PHZ_E_ARG, "phz_rowsdev_run: the covered variants changed since phz_rowsdev_pair_keys");
    // ---- block phasing: both kernels behind each other, then -- speculating that no component needs the host (a handful per genome at most) -- the block
    //      numbering, all before ONE host wait; exceptions are phased on the host and the numbering is redone
    if (int s = phase_enqueue(ctx, h->cstart, h->mem_s, h->estart, h->ekeep, T.ea, T.eb, cfgv, ncomp, nmem, nkeep, o->max_block_size, h->alle_of, h->sub_of, h->nsub,
                              h->complex_list, h->exc_list, h->eloc, cnt32)) return s;
    // ---- blocks in block order; per-block statistics
    const size_t NBK = (size_t)(nmem / 2 + 2);
    RSV(nsub_o, (size_t)(ncomp + 1) * 4); RSV(blk_base, (size_t)(ncomp + 2) * 4);
    RSV(blk_mstart, NBK * 4); RSV(blk_len, NBK * 4); RSV(blk_of, NV * 4); RSV(v_alle, NV); RSV(blk_sup, NBK * 4); RSV(blk_tot, NBK * 4);
    RSV(conc, NBK); RSV(cormode, NBK); RSV(statkind, NBK); RSV(statidx, NBK * 4); RSV(maxmaf, NBK * 4); RSV(stat, NBK * 8); RSV(cfg_rows, NBK * 8);
    RSV(cfg_base, (NBK + 1) * 8); RSV(blk_voff, (NBK + 1) * 4);
    PHZ_HIP(ctx, hipMemsetAsync(h->blk_of.p, 0xff, NV * 4, sm));
    PHZ_HIP(ctx, hipMemsetAsync(h->v_alle.p, 0, NV, sm));
    PHZ_HIP(ctx, hipMemsetAsync(h->blk_sup.p, 0, NBK * 4, sm));
    PHZ_HIP(ctx, hipMemsetAsync(h->blk_tot.p, 0, NBK * 4, sm));
    int64_t nblocks = 0;
    unsigned long long h_cfg_total = 0;
    std::vector<uint32_t> h_cc(n_cc, 0u); std::vector<unsigned long long> h_cfgc((size_t)nchrom, 0ull);
    if (ncomp) {
        uint32_t nb32 = 0, h_c32[4] = {0, 0, 0, 0};
        for (int round = 0; round < 2; round++) {
            hipLaunchKernelGGL(k_gather_nsub, dim3(nblk(ncomp)), dim3(256), 0, sm, ncomp, (const uint32_t *)h->corder.p, (const uint32_t *)h->nsub.p, P<uint32_t>(h->nsub_o));
            if (int s = gscan_excl<uint32_t, uint32_t>(ctx, P<uint32_t>(h->nsub_o), P<uint32_t>(h->blk_base), ncomp, h->scan_tmp)) return s;
            PHZ_HIP(ctx, hipGetLastError());
            {
                PhzMail mail(ctx);
                const int m_nb = mail.add(P<uint32_t>(h->blk_base) + ncomp, 4), m_c = mail.add(cnt32, 12);
                if (int s = mail.send()) return s;
                if (int s = sec.wait(round ? "blocks: count (after exceptions)" : "phase + blocks: count")) return s;
                nb32 = *mail.at<uint32_t>(m_nb);
                if (round == 0) memcpy(h_c32, mail.at<char>(m_c), 12);
            }
            sec.begin();
            if (round == 1 || !(h_c32[1] || h_c32[2])) break;
            if (int s = phase_exceptions(ctx, h_c32, h->cstart, h->mem_s, h->estart, h->ekeep, T.ea, T.eb, cfgv, ncomp, nmem, nkeep, ne, o->max_block_size, h->alle_of, h->sub_of,
                                         h->nsub, h->exc_list)) return s;
        }
        res->n_complex = h_c32[0]; res->n_exceptions = h_c32[1];
        nblocks = nb32;
        BK bk; bk.corder = P<uint32_t>(h->corder); bk.blk_base = P<uint32_t>(h->blk_base); bk.cstart = P<uint32_t>(h->cstart); bk.mem_s = P<uint32_t>(h->mem_s);
        bk.sub_of = P<int32_t>(h->sub_of); bk.alle_of = P<uint8_t>(h->alle_of); bk.blk_mstart = P<uint32_t>(h->blk_mstart); bk.blk_len = P<uint32_t>(h->blk_len);
        bk.blk_of = P<int32_t>(h->blk_of); bk.v_alle = P<uint8_t>(h->v_alle);
        hipLaunchKernelGGL(k_blocks, dim3(nblk(ncomp)), dim3(256), 0, sm, ncomp, bk);
        if (nkeep) hipLaunchKernelGGL(k_blk_edges, dim3(nblk(nkeep)), dim3(256), 0, sm, nkeep, (const uint32_t *)h->ekeep.p, (const int32_t *)T.ea, (const int32_t *)T.eb, cfgv,
                                      (const int32_t *)h->blk_of.p, (const uint8_t *)h->v_alle.p, P<uint32_t>(h->blk_sup), P<uint32_t>(h->blk_tot));
    }
    if (nblocks) {
        BS bs; bs.mem_s = P<uint32_t>(h->mem_s); bs.blk_mstart = P<uint32_t>(h->blk_mstart); bs.blk_len = P<uint32_t>(h->blk_len); bs.v_alle = P<uint8_t>(h->v_alle);
        bs.phase_idx = P<int8_t>(h->d_phase); bs.mafv = P<double>(h->d_maf); bs.conc = P<uint8_t>(h->conc); bs.cormode = P<uint8_t>(h->cormode); bs.statkind = P<uint8_t>(h->statkind);
        bs.statidx = P<uint32_t>(h->statidx); bs.maxmaf = P<int32_t>(h->maxmaf); bs.stat = P<double>(h->stat); bs.cfg_rows = P<unsigned long long>(h->cfg_rows);
        RSV(big_stat, (o->gw_phase_method == 1 ? (size_t)nblocks + 2 : (size_t)(nmem / (STAT_N + 1) + 2)) * 8);      // method 0: only blocks beyond the table (more than STAT_N members)
        bs.big_stat = P<double>(h->big_stat); bs.big_stat_n = cnt32 + 13; bs.gw_phase_method = o->gw_phase_method;
        hipLaunchKernelGGL(k_blk_stats, dim3(nblk(nblocks)), dim3(256), 0, sm, nblocks, bs);
        hipLaunchKernelGGL(k_block_starts, dim3(nblk(nblocks)), dim3(256), 0, sm, nblocks, (const uint32_t *)h->blk_mstart.p, (const uint32_t *)h->mem_s.p,
                           (const uint16_t *)h->d_vchrom.p, ss_blocks);
    }
    if (int s = gscan_excl<unsigned long long, unsigned long long>(ctx, P<unsigned long long>(h->cfg_rows), P<unsigned long long>(h->cfg_base), nblocks, h->scan_tmp)) return s;
    if (int s = gscan_excl<uint32_t, uint32_t>(ctx, P<uint32_t>(h->blk_len), P<uint32_t>(h->blk_voff), nblocks, h->scan_tmp)) return s;
    hipLaunchKernelGGL(k_starts_to_counts, dim3(1), dim3(1), 0, sm, ss_blocks, nchrom, (uint32_t)nblocks, cc_blocks);
    hipLaunchKernelGGL(k_block_counts, dim3(1), dim3(1), 0, sm, (const uint32_t *)ss_blocks, (const uint32_t *)cc_blocks, nchrom, (const uint32_t *)h->blk_voff.p,
                       (const unsigned long long *)h->cfg_base.p, cc_blkvars, cc_cfg);
    // (phased variants and the blocks whose gwStat text the table does not hold are read with the next wait, below)
    // ---- read sets of the haplotypes: labels + distinct counts.  No host wait inside: k_seg_small classifies the segments and counts the lists, the wave /
    //      workgroup kernels take their lists in ticket order and read the counts on the device; the pool-overflow flag is read with the next wait (rare:
    //      the pool is grown and the stage redone)
    const bool need_all = nb > 1 || h->has_black;
    RSV(labels, NR * 4); RSV(seg_ns, (size_t)(nblocks + 1) * 2 * nb * 4); RSV(big_list, std::max((size_t)(nblocks + 1) * 2 * nb, NRL) * 4 + 4); RSV(big_list2, std::max((size_t)(nblocks + 1) * 2 * nb, NRL) * 4 + 4);
    if (need_all) RSV(blk_cnt, (size_t)(nblocks + 1) * 2 * 4);
    if (nb > 1 || read_ids) RSV(single_n, NRL * 4);
    if (read_ids) { RSV(isf0, NR); RSV(isf2, NR); }
    if (h->pool.cap == 0) RSV(pool, (size_t)12 << 20);
    RSV(lab_e, (size_t)(nmem + 1) * 2 * nb * 4);
    RSV(huge_list, ((size_t)(nmem / (STAT_N + 1) + 2) * 2 * (size_t)nb + 16) * 4);
    RSV(its, (NR + 1) * 4); RSV(piece_dst, NRL * 8);
    RSV(big_blk, (size_t)(nblocks + 1) * 4);
    uint32_t h_nbig = 0, h_phased = 0, h_nbs = 0;
    for (int attempt = 0;; attempt++) {
        PHZ_HIP(ctx, hipMemsetAsync(h->labels.p, 0, NR * 4, sm));
        if (read_ids) { PHZ_HIP(ctx, hipMemsetAsync(h->isf0.p, 0, NR, sm)); PHZ_HIP(ctx, hipMemsetAsync(h->isf2.p, 0, NR, sm)); }
        if (nmem) hipLaunchKernelGGL(k_lab_e, dim3(nblk(nmem * 2 * nb)), dim3(256), 0, sm, nmem, nb, (const uint32_t *)h->mem_s.p, (const uint8_t *)h->v_alle.p, P<uint32_t>(h->lab_e));
        SG sg; sg.nb = nb; sg.lab_e = P<uint32_t>(h->lab_e); sg.nmem = nmem; sg.mem_s = P<uint32_t>(h->mem_s); sg.blk_mstart = P<uint32_t>(h->blk_mstart); sg.blk_len = P<uint32_t>(h->blk_len); sg.v_alle = P<uint8_t>(h->v_alle);
        sg.black = h->has_black ? P<uint8_t>(h->d_black) : nullptr; sg.rl_start = T.rl_start; sg.rl_qid = T.rl_qid; sg.labels = P<uint32_t>(h->labels);
        sg.big_list = P<uint32_t>(h->big_list); sg.big_list2 = P<uint32_t>(h->big_list2); sg.overflow = cnt32 + 20;
        sg.huge_list = P<uint32_t>(h->huge_list);
        sg.pool = P<uint32_t>(h->pool); sg.pool_cap = (uint32_t)std::min<size_t>(h->pool.cap / 4, 0xFFFFFFF0u);
        PHZ_HIP(ctx, hipMemsetAsync(cnt32 + 20, 0, 4, sm));
        PHZ_HIP(ctx, hipMemsetAsync(cnt32 + 32, 0, 3 * 64, sm));          // every mode has its own counters: [0..3] list lengths / pool cursor, [4] huge segments, [5..7] tickets -- nothing is
        for (int mode = 0; mode < 3; mode++) {                             // overwritten, so all of it is read back at the ONE wait below
            if (mode == 1 && !need_all) continue;
            if (mode == 2 && nb <= 1 && !read_ids) continue;
            sg.isf = !read_ids ? nullptr : (mode == 0 ? P<uint8_t>(h->isf0) : (mode == 2 ? P<uint8_t>(h->isf2) : nullptr));
            sg.nseg = mode == 0 ? nblocks * 2 * nb : (mode == 1 ? nblocks * 2 : (int64_t)NRL);
            sg.ns = mode == 0 ? P<uint32_t>(h->seg_ns) : (mode == 1 ? P<uint32_t>(h->blk_cnt) : P<uint32_t>(h->single_n));
            if (sg.nseg == 0) continue;
            uint32_t *mc = cnt32 + 32 + 16 * mode;
            sg.counters = mc; sg.huge_count = mc + 4; sg.tickets = mc + 5;
            const unsigned g = (unsigned)((sg.nseg + 63) / 64);
            // fixed numbers of workgroups for the list kernels: what a chip holds at once (10 KB / 52 KB of LDS each); a short list leaves most of them with nothing but one ticket
            const unsigned g_mid = (unsigned)std::min<int64_t>(sg.nseg, 4096), g_large = (unsigned)std::min<int64_t>(sg.nseg, 768), g_huge = (unsigned)std::min<int64_t>(sg.nseg, 64);
            if (mode == 0) {
                hipLaunchKernelGGL(k_seg_small<0>, dim3(g), dim3(64), 0, sm, sg);
                hipLaunchKernelGGL((k_seg_big<0, 512, 64>), dim3(g_mid), dim3(64), 0, sm, sg);
                hipLaunchKernelGGL((k_seg_big<0, 4096, PHZ_SEG_THREADS>), dim3(g_large), dim3(PHZ_SEG_THREADS), 0, sm, sg);
                hipLaunchKernelGGL((k_seg_big<0, 4096, PHZ_SEG_THREADS, true>), dim3(g_huge), dim3(PHZ_SEG_THREADS), 0, sm, sg);
            } else if (mode == 1) {
                hipLaunchKernelGGL(k_seg_small<1>, dim3(g), dim3(64), 0, sm, sg);
                hipLaunchKernelGGL((k_seg_big<1, 512, 64>), dim3(g_mid), dim3(64), 0, sm, sg);
                hipLaunchKernelGGL((k_seg_big<1, 4096, PHZ_SEG_THREADS>), dim3(g_large), dim3(PHZ_SEG_THREADS), 0, sm, sg);
                hipLaunchKernelGGL((k_seg_big<1, 4096, PHZ_SEG_THREADS, true>), dim3(g_huge), dim3(PHZ_SEG_THREADS), 0, sm, sg);
            } else {           // (mode 2 has one piece per segment: never huge)
                hipLaunchKernelGGL(k_seg_small<2>, dim3(g), dim3(64), 0, sm, sg);
                hipLaunchKernelGGL((k_seg_big<2, 512, 64>), dim3(g_mid), dim3(64), 0, sm, sg);
                hipLaunchKernelGGL((k_seg_big<2, 4096, PHZ_SEG_THREADS>), dim3(g_large), dim3(PHZ_SEG_THREADS), 0, sm, sg);
            }
        }
        // ---- text: byte counts -> scan -> write, file by file.  its = scan of the labels' text widths (digits + one separator), the widths computed as the
        //      scan loads the labels (no width array)
        if (int s = gscan_excl<uint32_t, uint32_t, LabelWidth>(ctx, P<uint32_t>(h->labels), P<uint32_t>(h->its), n_rl, h->scan_tmp)) return s;
        PHZ_HIP(ctx, hipMemsetAsync(h->piece_dst.p, 0xff, NRL * 8, sm));
        if (attempt) PHZ_HIP(ctx, hipMemsetAsync(cnt32 + 12, 0, 4, sm));
        if (nblocks) hipLaunchKernelGGL(k_big_blocks, dim3(nblk(nblocks)), dim3(256), 0, sm, nblocks, (const uint32_t *)h->blk_len.p, P<uint32_t>(h->big_blk), cnt32 + 12);
        PHZ_HIP(ctx, hipGetLastError());
        PhzMail mail(ctx);
        const int m_c = mail.add(cnt32, 512), m_ph = mail.add(P<uint32_t>(h->blk_voff) + nblocks, 4), m_cfg = mail.add(P<unsigned long long>(h->cfg_base) + nblocks, 8),
                  m_cc = mail.add(h->chrom_cnt.p, n_cc * 4), m_cfgc = mail.add(cc_cfg, (size_t)nchrom * 8);
        if (int s = mail.send()) return s;
        if (int s = sec.wait("stats + read sets + text sizes")) return s;
        sec.begin();
        const uint32_t *hc = mail.at<uint32_t>(m_c);
        h_nbig = hc[12]; h_nbs = hc[13]; h_phased = *mail.at<uint32_t>(m_ph); h_cfg_total = *mail.at<unsigned long long>(m_cfg);
        memcpy(h_cc.data(), mail.at<char>(m_cc), n_cc * 4); memcpy(h_cfgc.data(), mail.at<char>(m_cfgc), (size_t)nchrom * 8);
        if (!hc[20]) {
            res->n_big_segments = 0;
            for (int mode = 0; mode < 3; mode++) res->n_big_segments += (int64_t)hc[32 + 16 * mode] + hc[32 + 16 * mode + 3] + hc[32 + 16 * mode + 4];
            break;
        }
        if (attempt == 3) return phz_fail(ctx, PHZ_E_NOMEM, "read-set table pool did not converge");
        if (int s = phz_reserve(ctx, h->pool, h->pool.cap * 4)) return s;           // a segment of more than SEG_LDS reads needs 24 B per read: grow and redo
    }
    res->phased = (int64_t)h_phased;
    // gwStat text the table does not hold -- blocks with more known phases than it covers (:968-980), and with --gw_phase_method 1 every block phased by MAF weight (:1002) --: repr() of the float64 the device computed
    if (h_nbs) {
        std::string xt; std::vector<uint32_t> xo((size_t)h_nbs + 1, 0u); std::vector<double> bsv((size_t)h_nbs);
        PHZ_HIP(ctx, hipMemcpy(bsv.data(), h->big_stat.p, bsv.size() * 8, hipMemcpyDeviceToHost));
        xt.reserve((size_t)h_nbs * 20);
        for (uint32_t i = 0; i < h_nbs; i++) {
            xo[i] = (uint32_t)xt.size();
            phztext::put_pyfloat(xt, bsv[i]);
            xt += '\n';
        }
        xo[h_nbs] = (uint32_t)xt.size();
        if (int s = up(ctx, h->px_off, xo.data(), xo.size() * 4)) return s;
        if (int s = up(ctx, h->px_txt, xt.data(), xt.size())) return s;
        PHZ_HIP(ctx, hipStreamSynchronize(sm));            // xo / xt die with this scope
    }
    const uint32_t *blk_cnt = need_all ? P<uint32_t>(h->blk_cnt) : P<uint32_t>(h->seg_ns);     // one BAM, nothing blacklisted: the two read sets coincide
    if (n_rl * 12 >= (1ll << 32)) return phz_fail(ctx, PHZ_E_UNSUPPORTED, "device row stage: read-label text beyond 4 GiB");
    RD D; memset(&D, 0, sizeof(D));
    D.nv = nv; D.ne = ne; D.nblocks = nblocks; D.n_linked = n_linked; D.n_keys = nkeys; D.nchrom = nchrom; D.nb = nb; D.unique_ids = o->unique_ids; D.unphased_vars = o->unphased_vars;
    D.vchrom = P<uint16_t>(h->d_vchrom); D.pos = P<int32_t>(h->d_pos);
    auto pool = [&](int i) { PoolD p; p.off = P<uint32_t>(h->p_off[i]); p.b = P<char>(h->p_txt[i]); return p; };
    D.uid = pool(0); D.rsid = pool(1); D.alle = pool(2); D.maft = pool(3); D.chromn = pool(4); D.statt = pool(5);
    D.statx.off = P<uint32_t>(h->px_off); D.statx.b = P<char>(h->px_txt);
    D.bamn.off = d_bam_off; D.bamn.b = d_bam_txt; D.pvt.off = d_pv_off; D.pvt.b = d_pv_txt;
    D.mafv = P<double>(h->d_maf); D.is_ref = P<uint8_t>(h->d_isref); D.phase_idx = P<int8_t>(h->d_phase); D.black = h->has_black ? P<uint8_t>(h->d_black) : nullptr;
    D.bam_excl = o->bam_excluded ? d_bam_excl : nullptr;
    D.var_count = T.var_count; D.var_distinct = T.var_distinct; D.ea = T.ea; D.eb = T.eb; D.cis = cis; D.trans = trans; D.sup = sup; D.tot = tot; D.cfgv = cfgv;
    D.rl_start = T.rl_start; D.rl_qid = T.rl_qid; D.rl_list = T.rl_list;
    D.eorder = P<uint32_t>(h->eorder); D.va = P<int32_t>(h->va); D.vb = P<int32_t>(h->vb); D.e_slot = P<uint32_t>(h->e_slot); D.key_g = P<uint32_t>(h->key_g);
    D.mem_s = P<uint32_t>(h->mem_s); D.blk_mstart = P<uint32_t>(h->blk_mstart); D.blk_len = P<uint32_t>(h->blk_len); D.blk_of = P<int32_t>(h->blk_of); D.v_alle = P<uint8_t>(h->v_alle);
    D.blk_sup = P<uint32_t>(h->blk_sup); D.blk_tot = P<uint32_t>(h->blk_tot); D.blk_cnt = blk_cnt; D.seg_ns = P<uint32_t>(h->seg_ns); D.single_n = nb > 1 ? P<uint32_t>(h->single_n) : nullptr;
    D.blk_conc = P<uint8_t>(h->conc); D.blk_cormode = P<uint8_t>(h->cormode); D.blk_statkind = P<uint8_t>(h->statkind); D.blk_statidx = P<uint32_t>(h->statidx);
    D.blk_maxmaf = P<int32_t>(h->maxmaf); D.its = P<uint32_t>(h->its); D.labels = P<uint32_t>(h->labels); D.piece_dst = P<unsigned long long>(h->piece_dst);
    D.cfg_base = P<unsigned long long>(h->cfg_base);
    D.nmem = nmem;
    D.read_ids = read_ids ? 1 : 0;
    if (read_ids) { D.qn.off = P<uint32_t>(h->qn_off); D.qn.b = P<char>(h->qn_txt); D.qbase = P<long long>(h->qn_base); D.isf0 = P<uint8_t>(h->isf0); D.isf2 = P<uint8_t>(h->isf2); }
    RSV(mrec, (size_t)(nmem + 1) * sizeof(MemRec)); RSV(lab_e, (size_t)(nmem + 1) * 2 * nb * 4); RSV(lab_skip, (size_t)(nmem + 1) * 2 * nb * 4);
    D.mrec = P<MemRec>(h->mrec); D.lab_e = P<uint32_t>(h->lab_e); D.lab_skip = P<uint32_t>(h->lab_skip);
    if (nmem) {
        hipLaunchKernelGGL(k_mem_rec, dim3(nblk(nmem)), dim3(256), 0, sm, D, P<MemRec>(h->mrec));
        hipLaunchKernelGGL(k_mem_lab, dim3(nblk(nmem * 2 * nb)), dim3(256), 0, sm, D, P<uint32_t>(h->lab_e), P<uint32_t>(h->lab_skip));
    }
    {
        const int64_t nchunks = ((int64_t)h_cfg_total + 255) / 256;
        RSV(cfg_chunk, (size_t)(nchunks + 1) * 4);
        if (nchunks) hipLaunchKernelGGL(k_cfg_chunks, dim3(nblk(nchunks)), dim3(256), 0, sm, nchunks, nblocks, (const unsigned long long *)h->cfg_base.p, P<uint32_t>(h->cfg_chunk));
        D.cfg_chunk = P<uint32_t>(h->cfg_chunk);
    }
    // allele_config offsets in closed form: prefix arrays per block member, bytes per block, one scan over the blocks (not over the rows)
    RSV(cfg_pl, (size_t)(nmem + 1) * 4); RSV(cfg_pb, (size_t)(nmem + 1) * 4); RSV(cfg_ps, (size_t)(nmem + 1) * 8); RSV(cfg_bytes, (size_t)(nblocks + 1) * 8); RSV(cfg_bbase, (size_t)(nblocks + 2) * 8);
    D.cfg_pl = P<uint32_t>(h->cfg_pl); D.cfg_pb = P<uint32_t>(h->cfg_pb); D.cfg_ps = P<unsigned long long>(h->cfg_ps); D.cfg_bbase = P<unsigned long long>(h->cfg_bbase);
    if (nblocks) hipLaunchKernelGGL(k_cfg_prefix, dim3(nblk(nblocks)), dim3(256), 0, sm, D, nblocks, P<uint32_t>(h->cfg_pl), P<uint32_t>(h->cfg_pb), P<unsigned long long>(h->cfg_ps),
                                    P<unsigned long long>(h->cfg_bytes));
    if (h_nbig) hipLaunchKernelGGL(k_cfg_prefix_wave, dim3(h_nbig), dim3(64), 0, sm, D, (const uint32_t *)h->big_blk.p, P<uint32_t>(h->cfg_pl), P<uint32_t>(h->cfg_pb),
                                   P<unsigned long long>(h->cfg_ps), P<unsigned long long>(h->cfg_bytes));
    if (int s = gscan_excl<unsigned long long, unsigned long long>(ctx, P<unsigned long long>(h->cfg_bytes), P<unsigned long long>(h->cfg_bbase), nblocks, h->scan_tmp)) return s;
    const int64_t rows[PHZ_TXT_COUNT] = {n_linked, nblocks, nblocks * nb, (int64_t)h_cfg_total, nkeys, nkeys * nb, nkeys};
    int64_t max_rows = 1;
    for (int f = 0; f < PHZ_TXT_COUNT; f++) max_rows = std::max(max_rows, rows[f]);
    RSV(rowlen, (size_t)max_rows * 4);
    // per-file segment tables: rows per chromosome (per BAM x chromosome for the key files), prefix-summed on the device into byte offsets
    const int nseg[PHZ_TXT_COUNT] = {nchrom, nchrom, nchrom, nchrom, nb * nchrom, nb * nchrom, nb * nchrom};
    uint32_t *d_nb = cnt32 + 8;
    hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(256), 0, sm, d_nb, (int64_t)1, (uint32_t)nb);
    for (int f = 0; f < PHZ_TXT_COUNT; f++) {
        RSV(seg_off_d[f], (size_t)(nseg[f] + 1) * 8);
        if (f == PHZ_TXT_CFG) {        // offsets in closed form: per-chromosome byte offsets = the byte base of the chromosome's first block
            hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(256), 0, sm, (const uint32_t *)cc_blocks, (const uint32_t *)nullptr, nseg[f], (const unsigned long long *)h->cfg_bbase.p,
                               P<unsigned long long>(h->seg_off_d[f]));
            continue;
        }
        RSV(off[f], (size_t)(rows[f] + 1) * 8);
        const unsigned g = nblk(rows[f]);
        uint32_t *len = P<uint32_t>(h->rowlen);
        if (rows[f]) switch (f) {
            case PHZ_TXT_CONN: hipLaunchKernelGGL(k_row_len<RowConn>, dim3(g), dim3(256), 0, sm, D, rows[f], len); break;
            case PHZ_TXT_HAP: hipLaunchKernelGGL(k_row_len<RowHap>, dim3(g), dim3(256), 0, sm, D, rows[f], len);
                if (h_nbig) hipLaunchKernelGGL(k_row_wave_len<RowHap>, dim3(h_nbig), dim3(64), 0, sm, D, (const uint32_t *)h->big_blk.p, 1, len);
                break;
            case PHZ_TXT_ASE: hipLaunchKernelGGL(k_row_len<RowAse>, dim3(g), dim3(256), 0, sm, D, rows[f], len);
                if (h_nbig) hipLaunchKernelGGL(k_row_wave_len<RowAse>, dim3(h_nbig * (unsigned)nb), dim3(64), 0, sm, D, (const uint32_t *)h->big_blk.p, nb, len);
                break;
            case PHZ_TXT_ALLELIC: hipLaunchKernelGGL(k_row_len<RowAllelic>, dim3(g), dim3(256), 0, sm, D, rows[f], len); break;
            case PHZ_TXT_SINGLE_ASE: hipLaunchKernelGGL(k_row_len<RowSingleAse>, dim3(g), dim3(256), 0, sm, D, rows[f], len); break;
            default: hipLaunchKernelGGL(k_row_len<RowSingleHap>, dim3(g), dim3(256), 0, sm, D, rows[f], len); break;
        }
        if (f == PHZ_TXT_ALLELIC && rows[f]) hipLaunchKernelGGL(k_count_nonzero, dim3(g < 512u ? g : 512u), dim3(256), 0, sm, (const uint32_t *)len, rows[f], cnt64 + 4);
        if (int s = gscan_excl<uint32_t, unsigned long long>(ctx, len, P<unsigned long long>(h->off[f]), rows[f], h->scan_tmp)) return s;
        unsigned long long *so = P<unsigned long long>(h->seg_off_d[f]);
        const unsigned long long *of = P<unsigned long long>(h->off[f]);
        switch (f) {
            case PHZ_TXT_CONN: hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(256), 0, sm, (const uint32_t *)cc_conn, (const uint32_t *)nullptr, nseg[f], of, so); break;
            case PHZ_TXT_HAP: hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(256), 0, sm, (const uint32_t *)cc_blocks, (const uint32_t *)nullptr, nseg[f], of, so); break;
            case PHZ_TXT_ASE: hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(256), 0, sm, (const uint32_t *)cc_blocks, (const uint32_t *)d_nb, nseg[f], of, so); break;
            case PHZ_TXT_SINGLE_ASE: hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(256), 0, sm, (const uint32_t *)cc_keys, (const uint32_t *)d_nb, nseg[f], of, so); break;
            default: hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(256), 0, sm, (const uint32_t *)cc_keys, (const uint32_t *)nullptr, nseg[f], of, so); break;
        }
    }
    PHZ_HIP(ctx, hipGetLastError());
    std::vector<unsigned long long> h_so[PHZ_TXT_COUNT];
    {
        PhzMail mail(ctx);
        int m_so[PHZ_TXT_COUNT];
        const int m_az = mail.add(cnt64 + 4, 8);
        for (int f = 0; f < PHZ_TXT_COUNT; f++) m_so[f] = mail.add(h->seg_off_d[f].p, ((size_t)nseg[f] + 1) * 8);
        if (int s = mail.send()) return s;
        if (int s = sec.wait("text: byte offsets")) return s;
        h_c64[4] = *mail.at<unsigned long long>(m_az);
        for (int f = 0; f < PHZ_TXT_COUNT; f++) { const unsigned long long *q = mail.at<unsigned long long>(m_so[f]); h_so[f].assign(q, q + (size_t)nseg[f] + 1); }
    }
    sec.begin();
    // Where the text goes on the host, when the caller gave a page-locked region that holds all of it (phz_rowsdev_opts.host_text): every file at a 4 KB boundary, copied
    // on the ctx's copy stream AS SOON AS ITS WRITER HAS FINISHED -- largest file first (allele_config is half of the bytes), so that the link is busy from the first
    // finished file on while the other writers still run.  (The files are all written in the last fifth of a pass: what this hides is that fifth, not the pass.)
    bool to_host = o->host_text != nullptr && o->host_text_cap > 0;
    struct CopyGuard {          // whatever way this call ends, no copy into the caller's region is still in flight when it returns
        phz_ctx *c; bool armed = false;
        ~CopyGuard() { if (armed && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream); }
    } copy_guard{ctx};
    int64_t host_need = 0;
    for (int f = 0; f < PHZ_TXT_COUNT; f++) {
        h->bytes[f] = (int64_t)h_so[f].back();
        h->seg_off[f].assign(h_so[f].begin(), h_so[f].end());
        res->bytes[f] = h->bytes[f]; res->seg_off[f] = h->seg_off[f].data();
        res->host_off[f] = -1;
        host_need = ((host_need + 4095) & ~(int64_t)4095) + h->bytes[f];
    }
    if (to_host && host_need > o->host_text_cap) to_host = false;
    if (to_host) {
        if (!ctx->copy_stream) PHZ_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        int64_t at = 0;
        for (int f = 0; f < PHZ_TXT_COUNT; f++) { at = (at + 4095) & ~(int64_t)4095; res->host_off[f] = at; at += h->bytes[f]; }
        for (int f = 0; f < PHZ_TXT_COUNT; f++) if (!h->txt_ev[f]) PHZ_HIP(ctx, hipEventCreateWithFlags(&h->txt_ev[f], hipEventDisableTiming));
    }
    static const int write_order[PHZ_TXT_COUNT] = {PHZ_TXT_CFG, PHZ_TXT_ASE, PHZ_TXT_ALLELIC, PHZ_TXT_CONN, PHZ_TXT_HAP, PHZ_TXT_SINGLE_ASE, PHZ_TXT_SINGLE_HAP};
    for (int wo = 0; wo < PHZ_TXT_COUNT; wo++) {
        const int f = write_order[wo];
        RSV(text[f], (size_t)h->bytes[f] + 16);
        const unsigned g = nblk(rows[f]);
        const unsigned long long *of = P<unsigned long long>(h->off[f]);
        char *out = P<char>(h->text[f]);
        if (rows[f]) switch (f) {
            case PHZ_TXT_CONN: hipLaunchKernelGGL((k_row_write<RowConn, 256, 24 * 1024>), dim3((unsigned)((rows[f] + 255) / 256)), dim3(256), 0, sm, D, rows[f], of, out); break;
            case PHZ_TXT_HAP: hipLaunchKernelGGL((k_row_write<RowHap, PHZ_HAP_ROWS, PHZ_HAP_STAGE>), dim3((unsigned)((rows[f] + PHZ_HAP_ROWS - 1) / PHZ_HAP_ROWS)), dim3(PHZ_HAP_ROWS), 0, sm, D, rows[f], of, out);
                if (h_nbig) hipLaunchKernelGGL(k_row_wave_write<RowHap>, dim3(h_nbig), dim3(64), 0, sm, D, (const uint32_t *)h->big_blk.p, 1, of, out);
                break;
            case PHZ_TXT_ASE: hipLaunchKernelGGL((k_row_write<RowAse, PHZ_ASE_ROWS, PHZ_ASE_STAGE>), dim3((unsigned)((rows[f] + PHZ_ASE_ROWS - 1) / PHZ_ASE_ROWS)), dim3(PHZ_ASE_ROWS), 0, sm, D, rows[f], of, out);
                if (h_nbig) hipLaunchKernelGGL(k_row_wave_write<RowAse>, dim3(h_nbig * (unsigned)nb), dim3(64), 0, sm, D, (const uint32_t *)h->big_blk.p, nb, of, out);
                break;
            case PHZ_TXT_CFG: hipLaunchKernelGGL((k_cfg_write<128, 12 * 1024>), dim3((unsigned)((rows[f] + 127) / 128)), dim3(128), 0, sm, D, rows[f], out); break;
            case PHZ_TXT_ALLELIC: hipLaunchKernelGGL((k_row_write<RowAllelic, 256, 24 * 1024>), dim3((unsigned)((rows[f] + 255) / 256)), dim3(256), 0, sm, D, rows[f], of, out); break;
            case PHZ_TXT_SINGLE_ASE: hipLaunchKernelGGL((k_row_write<RowSingleAse, 256, 32 * 1024>), dim3((unsigned)((rows[f] + 255) / 256)), dim3(256), 0, sm, D, rows[f], of, out); break;
            default: hipLaunchKernelGGL((k_row_write<RowSingleHap, 256, 32 * 1024>), dim3((unsigned)((rows[f] + 255) / 256)), dim3(256), 0, sm, D, rows[f], of, out); break;
        }
        if (f == PHZ_TXT_ASE && n_rl && rows[f]) hipLaunchKernelGGL(k_label_write, dim3(nblk(n_rl)), dim3(256), 0, sm, D, n_rl, out);
        if (to_host && h->bytes[f]) {
            PHZ_HIP(ctx, hipEventRecord(h->txt_ev[f], sm));
            PHZ_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, h->txt_ev[f], 0));
            copy_guard.armed = true;
            PHZ_HIP(ctx, hipMemcpyAsync((char *)o->host_text + res->host_off[f], h->text[f].p, (size_t)h->bytes[f], hipMemcpyDeviceToHost, ctx->copy_stream));
        }
    }
    // ---- per-block arrays for write_vcf
    h->have_vcf = false;
    h->n_blocks = nblocks; h->n_blk_vars = res->phased;
    if (o->want_vcf && nblocks) {
        RSV(o_var, (size_t)(res->phased + 1) * 4); RSV(o_maxmaf, (size_t)(nblocks + 1) * 4); RSV(o_hap, (size_t)(res->phased + 1)); RSV(o_cor, (size_t)(res->phased + 1) * 2);
        VB vbk; vbk.mem_s = P<uint32_t>(h->mem_s); vbk.blk_mstart = P<uint32_t>(h->blk_mstart); vbk.blk_len = P<uint32_t>(h->blk_len); vbk.blk_voff = P<uint32_t>(h->blk_voff);
        vbk.v_alle = P<uint8_t>(h->v_alle); vbk.cormode = P<uint8_t>(h->cormode); vbk.phase_idx = P<int8_t>(h->d_phase); vbk.maxmaf = P<int32_t>(h->maxmaf);
        vbk.vchrom = P<uint16_t>(h->d_vchrom); vbk.chrom_v0 = (const long long *)h->d_chrom_v0.p; vbk.o_var = P<int32_t>(h->o_var); vbk.o_maxmaf = P<int32_t>(h->o_maxmaf);
        vbk.o_hap = P<uint8_t>(h->o_hap); vbk.o_cor = P<int8_t>(h->o_cor);
        hipLaunchKernelGGL(k_vcf_blocks, dim3(nblk(nblocks)), dim3(256), 0, sm, nblocks, vbk);
        h->have_vcf = true;
    }
    PHZ_HIP(ctx, hipGetLastError());
    if (int s = sec.wait("end")) return s;
    if (to_host) { copy_guard.armed = false; PHZ_HIP(ctx, hipStreamSynchronize(ctx->copy_stream)); }          // the text is in the caller's region
#undef RSV
    h->chrom_blocks.assign((size_t)nchrom, 0); h->chrom_blk_vars.assign((size_t)nchrom, 0);
    for (int c = 0; c < nchrom; c++) { h->chrom_blocks[(size_t)c] = h_cc[(size_t)nchrom + c]; h->chrom_blk_vars[(size_t)c] = h_cc[(size_t)2 * nchrom + c]; }
    res->chrom_blocks = h->chrom_blocks.data(); res->chrom_blk_vars = h->chrom_blk_vars.data();
    // the reference orders the chromosomes of the block files by the first BAM whose call file has a kept line on them (read_vars is keyed by the `chrom` that
    // process_mapping_result returns, "" for a file without kept lines: phaser.py:1299, :573-574), VCF order inside a BAM.  A chromosome's covered variants are
    // counted per (BAM of the first kept line, chromosome): its first BAM is the first of those counters that is not zero.
    h->chrom_first_bam.assign((size_t)nchrom, -1);
    for (int c = 0; c < nchrom; c++)
        for (int b = 0; b < nb; b++)
            if (h_cc[(size_t)3 * nchrom + (size_t)b * nchrom + c]) { h->chrom_first_bam[(size_t)c] = b; break; }
    res->chrom_first_bam = h->chrom_first_bam.data();
    res->n_blocks = nblocks; res->n_blk_vars = res->phased; res->n_components = ncomp; res->n_linked = n_linked;
    res->allelic_rows = (int64_t)h_c64[4];
    res->gpu_ms = sec.ms;
    ctx->last_ms[PHZ_T_ROWS] = (float)sec.ms; ctx->total_ms[PHZ_T_ROWS] += sec.ms; ctx->launches[PHZ_T_ROWS]++;
    return PHZ_OK;
}

// Self-test entry of the device radix sort (phz_sort.h): sorts n host (key, value) pairs by the given bit ranges with the one-launch-per-pass sort
// (three_launch = 0) or the three-launch passes (1) and returns keys and values in sorted order.  Used by the tests (GPU and emulation) against a
// stable host sort; no reference counterpart.
extern "C" int phz_selftest_sort(phz_ctx *ctx, int key_bytes, const void *keys, const uint32_t *vals, int64_t n, const int32_t *ranges, int nranges, int three_launch,
                                 void *keys_out, uint32_t *vals_out) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || (key_bytes != 4 && key_bytes != 8) || n < 0 || nranges < 1 || nranges > 4 || !ranges || (n && (!keys || !vals || !keys_out || !vals_out))) return PHZ_E_ARG;
    if (!n) return PHZ_OK;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d[6];
    auto fin = [&](int code) { for (DevBuf &b : d) if (b.p) (void)hipFree(b.p); return code; };
    int st = PHZ_OK;
    const size_t kb = (size_t)n * (size_t)key_bytes, vb = (size_t)n * 4;
    if ((st = up(ctx, d[0], keys, kb)) || (st = phz_reserve(ctx, d[1], kb)) || (st = up(ctx, d[2], vals, vb)) || (st = phz_reserve(ctx, d[3], vb))) return fin(st);
    int rg[4][2];
    for (int r = 0; r < nranges; r++) { rg[r][0] = ranges[2 * r]; rg[r][1] = ranges[2 * r + 1]; }
    int where = 0;
    if (key_bytes == 4) st = radix_sort_ranges<uint32_t, uint32_t>(ctx, P<uint32_t>(d[0]), P<uint32_t>(d[1]), P<uint32_t>(d[2]), P<uint32_t>(d[3]), n, rg, nranges, d[4], d[5], &where, nullptr, nullptr, !three_launch);
    else st = radix_sort_ranges<unsigned long long, uint32_t>(ctx, P<unsigned long long>(d[0]), P<unsigned long long>(d[1]), P<uint32_t>(d[2]), P<uint32_t>(d[3]), n, rg, nranges, d[4], d[5], &where, nullptr, nullptr, !three_launch);
    if (st != PHZ_OK) return fin(st);
    hipError_t e = hipMemcpyAsync(keys_out, where ? d[1].p : d[0].p, kb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(vals_out, where ? d[3].p : d[2].p, vb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    return fin(e == hipSuccess ? PHZ_OK : phz_fail(ctx, PHZ_E_HIP, "phz_selftest_sort", e));
}

// SURVEY.md 8(b) `phz_hap_counts`: the number of DISTINCT reads (QNAMEs) in every (variant, allele, BAM) read list of the resident tally --
// len(set(dict_variant_reads[v]['haplo_reads'][allele][bam])), what the haplotype-count loops of the reference evaluate per variant (phaser.py:1196-1204; the
// union over a block's variants is phz_rowsdev_run's read-set stage).  counts[(variant * 2 + allele) * n_bams + bam], host or device array.  The lists are
// deduplicated by the read-set kernels of the row stage (one thread / wave / workgroup per list by its length).
extern "C" int phz_hap_counts(phz_ctx *ctx, int32_t *counts, int64_t n_counts, int space) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || (space != PHZ_HOST && space != PHZ_DEVICE)) return PHZ_E_ARG;
    auto &T = ctx->tally;
    const int64_t nseg = T.nv * 2 * (int64_t)T.nb;
    if (n_counts != nseg || (nseg && !counts)) return phz_fail(ctx, PHZ_E_ARG, "phz_hap_counts: counts must hold variants x 2 x BAMs of the resident tally");
    if (!nseg) return PHZ_OK;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t sm = ctx->stream;
    DevBuf ns, l1, l2, cnt, pool;
    auto fin = [&](int code) { for (DevBuf *b : {&ns, &l1, &l2, &cnt, &pool}) if (b->p) (void)hipFree(b->p); return code; };
    int st = PHZ_OK;
    if ((st = phz_reserve(ctx, ns, (size_t)nseg * 4)) || (st = phz_reserve(ctx, l1, (size_t)nseg * 4 + 4)) || (st = phz_reserve(ctx, l2, (size_t)nseg * 4 + 4)) ||
        (st = phz_reserve(ctx, cnt, 256)) || (st = phz_reserve(ctx, pool, (size_t)4 << 20))) return fin(st);
    for (int attempt = 0;; attempt++) {
        hipError_t e = hipMemsetAsync(cnt.p, 0, 256, sm);
        if (e != hipSuccess) return fin(phz_fail(ctx, PHZ_E_HIP, "phz_hap_counts", e));
        uint32_t *c = P<uint32_t>(cnt);
        SG sg; memset(&sg, 0, sizeof(sg));
        sg.nseg = nseg; sg.nb = T.nb; sg.rl_start = T.rl_start; sg.rl_qid = T.rl_qid; sg.ns = P<uint32_t>(ns);
        sg.big_list = P<uint32_t>(l1); sg.big_list2 = P<uint32_t>(l2); sg.huge_list = P<uint32_t>(l1);      // (a list is one piece: never huge)
        sg.counters = c; sg.huge_count = c + 4; sg.tickets = c + 5; sg.overflow = c + 20;
        sg.pool = P<uint32_t>(pool); sg.pool_cap = (uint32_t)std::min<size_t>(pool.cap / 4, 0xFFFFFFF0u);
        hipLaunchKernelGGL(k_seg_small<2>, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, sm, sg);
        hipLaunchKernelGGL((k_seg_big<2, 512, 64>), dim3((unsigned)std::min<int64_t>(nseg, 4096)), dim3(64), 0, sm, sg);
        hipLaunchKernelGGL((k_seg_big<2, 4096, PHZ_SEG_THREADS>), dim3((unsigned)std::min<int64_t>(nseg, 768)), dim3(PHZ_SEG_THREADS), 0, sm, sg);
        uint32_t ovf = 0;
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&ovf, c + 20, 4, hipMemcpyDeviceToHost, sm);
        if (e == hipSuccess) e = hipStreamSynchronize(sm);
        if (e != hipSuccess) return fin(phz_fail(ctx, PHZ_E_HIP, "phz_hap_counts", e));
        if (!ovf) break;
        if (attempt == 3) return fin(phz_fail(ctx, PHZ_E_NOMEM, "phz_hap_counts: read-set table pool did not converge"));
        if ((st = phz_reserve(ctx, pool, pool.cap * 4))) return fin(st);           // a list of more than SEG_LDS reads keeps its table in the pool: grow and redo
    }
    hipError_t e = hipMemcpyAsync(counts, ns.p, (size_t)nseg * 4, space == PHZ_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, sm);
    if (e == hipSuccess) e = hipStreamSynchronize(sm);
    return fin(e == hipSuccess ? PHZ_OK : phz_fail(ctx, PHZ_E_HIP, "phz_hap_counts", e));
}

// copy one finished text (PHZ_TXT_*) to host memory (page-locked memory gives the full PCIe rate)
extern "C" int phz_rowsdev_fetch_text(phz_ctx *ctx, phz_rowsdev *h, int which, void *dst, int64_t bytes) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !h || which < 0 || which >= PHZ_TXT_COUNT || bytes != h->bytes[which] || (!dst && bytes)) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes) PHZ_HIP(ctx, hipMemcpyAsync(dst, h->text[which].p, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PHZ_OK;
}

extern "C" const void *phz_rowsdev_text_ptr(phz_rowsdev *h, int which) {
    return (h && which >= 0 && which < PHZ_TXT_COUNT) ? h->text[which].p : nullptr;
}

// per-block arrays of the last run (want_vcf): block order = file order; variant indices are chromosome-local
extern "C" int phz_rowsdev_fetch_blocks(phz_ctx *ctx, phz_rowsdev *h, int32_t *blk_size, int32_t *blk_var, uint8_t *blk_hap, int8_t *blk_cor, double *blk_stat,
                                        uint8_t *blk_stat_int, int32_t *blk_maxmaf) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !h || !h->have_vcf) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nbk = (size_t)h->n_blocks, nvr = (size_t)h->n_blk_vars;
    hipStream_t sm = ctx->stream;
    std::vector<uint8_t> kind(nbk);
    if (blk_size) PHZ_HIP(ctx, hipMemcpyAsync(blk_size, h->blk_len.p, nbk * 4, hipMemcpyDeviceToHost, sm));
    if (blk_var) PHZ_HIP(ctx, hipMemcpyAsync(blk_var, h->o_var.p, nvr * 4, hipMemcpyDeviceToHost, sm));
    if (blk_hap) PHZ_HIP(ctx, hipMemcpyAsync(blk_hap, h->o_hap.p, nvr, hipMemcpyDeviceToHost, sm));
    if (blk_cor) PHZ_HIP(ctx, hipMemcpyAsync(blk_cor, h->o_cor.p, nvr * 2, hipMemcpyDeviceToHost, sm));
    if (blk_stat) PHZ_HIP(ctx, hipMemcpyAsync(blk_stat, h->stat.p, nbk * 8, hipMemcpyDeviceToHost, sm));
    if (blk_maxmaf) PHZ_HIP(ctx, hipMemcpyAsync(blk_maxmaf, h->o_maxmaf.p, nbk * 4, hipMemcpyDeviceToHost, sm));
    if (blk_stat_int) PHZ_HIP(ctx, hipMemcpyAsync(kind.data(), h->statkind.p, nbk, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    if (blk_stat_int) for (size_t b = 0; b < nbk; b++) blk_stat_int[b] = kind[b] == 1 ? 1 : 0;
    return PHZ_OK;
}

// phase_v3 (phaser/phaser.py:2107-2170, the worker of `parallelize(phase_v3, ...)` at :808) for a batch of connected components on the
// GPU.  Component c holds the position-sorted variants [comp_start[c], comp_start[c+1]) and the pairs [pair_start[c], pair_start[c+1]);
// pair_i / pair_j are LOCAL variant indices inside the component, pair_cfg 0 same configuration / 1 opposite / -1 tie.  Outputs per
// variant: sub_of = ordinal of its final block inside the component (-1: in none), alle_of = its allele on haplotype A; n_sub per component.
extern "C" int phz_phase_components(phz_ctx *ctx, int64_t n_comp, const uint32_t *comp_start, const uint32_t *pair_start, const int32_t *pair_i, const int32_t *pair_j,
                                    const int8_t *pair_cfg, int32_t max_block_size, int32_t *sub_of, uint8_t *alle_of, uint32_t *n_sub) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || n_comp < 0 || (n_comp && (!comp_start || !pair_start || !sub_of || !alle_of || !n_sub))) return PHZ_E_ARG;
    if (!n_comp) return PHZ_OK;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nmem = comp_start[n_comp], ne = pair_start[n_comp];
    if (ne && (!pair_i || !pair_j || !pair_cfg)) return PHZ_E_ARG;
    std::vector<uint32_t> mem((size_t)nmem + 1), ek((size_t)ne + 1);
    std::vector<int32_t> ea((size_t)ne + 1), eb((size_t)ne + 1), cf((size_t)ne + 1);
    for (int64_t t = 0; t < nmem; t++) mem[(size_t)t] = (uint32_t)t;
    for (int64_t c = 0; c < n_comp; c++) {
        if (comp_start[c + 1] <= comp_start[c] || pair_start[c + 1] < pair_start[c]) return phz_fail(ctx, PHZ_E_ARG, "phz_phase_components: empty component or offsets not ascending");
        const int64_t n = (int64_t)comp_start[c + 1] - comp_start[c];
        for (uint32_t e = pair_start[c]; e < pair_start[c + 1]; e++) {
            if (pair_i[e] < 0 || pair_j[e] < 0 || pair_i[e] >= n || pair_j[e] >= n || pair_i[e] == pair_j[e]) return phz_fail(ctx, PHZ_E_ARG, "phz_phase_components: pair outside its component");
            ek[e] = e; ea[e] = (int32_t)(comp_start[c] + (uint32_t)pair_i[e]); eb[e] = (int32_t)(comp_start[c] + (uint32_t)pair_j[e]); cf[e] = pair_cfg[e];
        }
    }
    enum { B_CS, B_MEM, B_ES, B_EK, B_EA, B_EB, B_CF, B_CNT, B_AL, B_SUB, B_NSUB, B_CX, B_EX, B_EL, B_N };
    DevBuf d[B_N];
    int st = PHZ_OK;
    auto fin = [&](int code) { for (DevBuf &b : d) if (b.p) (void)hipFree(b.p); return code; };
    auto U = [&](DevBuf &b, const void *src, size_t bytes) { if (st == PHZ_OK) st = up(ctx, b, src, bytes); };
    U(d[B_CS], comp_start, (size_t)(n_comp + 1) * 4); U(d[B_MEM], mem.data(), (size_t)nmem * 4); U(d[B_ES], pair_start, (size_t)(n_comp + 1) * 4);
    U(d[B_EK], ek.data(), (size_t)ne * 4); U(d[B_EA], ea.data(), (size_t)ne * 4); U(d[B_EB], eb.data(), (size_t)ne * 4); U(d[B_CF], cf.data(), (size_t)ne * 4);
    if (st == PHZ_OK) st = phz_reserve(ctx, d[B_CNT], 128);
    if (st != PHZ_OK) return fin(st);
    Sections sec(ctx);
    sec.begin();
    st = phase_enqueue(ctx, d[B_CS], d[B_MEM], d[B_ES], d[B_EK], P<int32_t>(d[B_EA]), P<int32_t>(d[B_EB]), P<int32_t>(d[B_CF]), n_comp, nmem, ne, max_block_size,
                       d[B_AL], d[B_SUB], d[B_NSUB], d[B_CX], d[B_EX], d[B_EL], P<uint32_t>(d[B_CNT]));
    if (st != PHZ_OK) return fin(st);
    uint32_t h_c32[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(h_c32, d[B_CNT].p, 12, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return fin(phz_fail(ctx, PHZ_E_HIP, "phz_phase_components"));
    st = sec.wait("phase components");
    if (st == PHZ_OK) st = phase_exceptions(ctx, h_c32, d[B_CS], d[B_MEM], d[B_ES], d[B_EK], P<int32_t>(d[B_EA]), P<int32_t>(d[B_EB]), P<int32_t>(d[B_CF]), n_comp, nmem, ne, ne,
                                            max_block_size, d[B_AL], d[B_SUB], d[B_NSUB], d[B_EX]);
    if (st != PHZ_OK) return fin(st);
    hipError_t e = hipMemcpyAsync(sub_of, d[B_SUB].p, (size_t)nmem * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(alle_of, d[B_AL].p, (size_t)nmem, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(n_sub, d[B_NSUB].p, (size_t)n_comp * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fin(phz_fail(ctx, PHZ_E_HIP, "phz_phase_components", e));
    ctx->last_ms[PHZ_T_ROWS] = (float)sec.ms;
    return fin(PHZ_OK);
}
