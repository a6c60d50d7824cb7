// Device-side BAM path (SURVEY.md 8(f) next-1 on the GPU): from the BGZF members of a coordinate-sorted BAM in a file to the
// structure-of-arrays shards K_map reads, without the records ever visiting the host.  Replaces, like phz_bam.cpp, what phASER gets
// from `samtools view -h BAM 'chr': | samtools view -Sh [-F 0x400] [-f 2] -q MAPQ -` (phaser/phaser.py:1346, :505-513) plus the
// mapper's per-record text parsing (phaser/read_variant_map.py:27-64) -- same filters, same packed arrays (tests compare the two
// paths array by array).
//
//   host   member table + header + chromosome ranges (phz_bam_plan_file: a few probe members inflated with zlib)
//   H2D    the compressed members that hold wanted records
//   K_inflate (phz_inflate.hip)  one lane per BGZF member
//   k_seg_start   the stream is cut into ~256 KB segments; the first record boundary of each is GUESSED by a plausibility chain
//   k_hop<0>      one lane per segment hops its record chain: filters, kept-record count, sortedness; every guess is VERIFIED
//                 (segment k must end exactly where segment k+1 starts -- the same proof the host's parallel hop uses)
//   scan + k_hop<1>   kept-record list (offset, reference, op count, bases, name length)
//   scans + k_pack    one lane per kept record writes its slots of the shard arrays (normalised CIGAR, 2-bit bases, quals, AS)
// Anything the device path cannot prove (a member that is not valid DEFLATE, a boundary that does not verify, an unsorted file,
// 32-bit offsets exceeded) returns PHZ_E_UNSUPPORTED and the caller uses the host path.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "phz.h"
#include "phz_internal.h"
#include "phz_scan.h"

namespace {

__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ __forceinline__ int32_t ldi32(const uint8_t *p) { return (int32_t)ld32(p); }

struct Seg { uint64_t guess, end; uint32_t exact, piece; };     // end = end of the segment's piece

struct Filters {
    int min_mapq, flag_required, flag_forbidden;
    double isize_cutoff;
    const uint8_t *ref_mask;      // [n_ref] device, 1 = wanted
    int n_ref;
};

// can offset p of d[0, n) start a record?  -> offset of the next record, 0 when it cannot (same test as the host's plausible_at)
__device__ uint64_t plausible(const uint8_t *d, uint64_t n, uint64_t p, int n_ref, uint64_t *key) {
    if (p + 36 > n) return 0;
    const int32_t bs = ldi32(d + p);
    if (bs < 32 || bs > (1 << 24) || p + 4 + (uint64_t)bs > n) return 0;
    const uint8_t *r = d + p + 4;
    const int32_t ref = ldi32(r), pos0 = ldi32(r + 4), l_seq = ldi32(r + 16), nref = ldi32(r + 20);
    const uint32_t l_rn = r[8], n_cig = ld16(r + 12), flag = ld16(r + 14);
    // a QNAME has at least one character (SAM: [!-?A-~]{1,254}; an empty name would make every zero byte a candidate), FLAG has 12
    // defined bits, mate position >= -1
    if (ref < -1 || ref >= n_ref || nref < -1 || nref >= n_ref || pos0 < -1 || l_seq < 0 || l_rn < 2 || flag >= 4096 || ldi32(r + 24) < -1) return 0;
    const uint64_t need = 32 + (uint64_t)l_rn + 4 * (uint64_t)n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (need > (uint64_t)bs) return 0;
    if (r[32 + l_rn - 1] != 0) return 0;
    for (uint32_t k = 0; k + 1 < l_rn; k++) if (r[32 + k] < 33 || r[32 + k] > 126) return 0;
    *key = ((uint64_t)(ref < 0 ? 0x7fffffff : ref) << 32) | (uint32_t)(pos0 + 1);      // coordinate-sort key (unmapped last)
    return p + 4 + (uint64_t)bs;
}

__global__ __launch_bounds__(64) void k_seg_start(const uint8_t *d, const Seg *segs, int64_t nseg, int n_ref, uint64_t *start) {
    const int64_t k = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (k >= nseg) return;
    const Seg s = segs[k];
    if (s.exact) { start[k] = s.guess; return; }
    const uint64_t limit = s.guess + (32u << 20) < s.end ? s.guess + (32u << 20) : s.end;
    uint64_t found = s.end;
    for (uint64_t p = s.guess; p < limit; p++) {
        uint64_t q = p, prev_key = 0; int ok = 0;
        while (ok < 12) {
            uint64_t key;
            const uint64_t nx = plausible(d, s.end, q, n_ref, &key);
            if (!nx || key < prev_key) break;              // the records of a chain are in coordinate order, too
            prev_key = key;
            ok++; q = nx;
            if (q + 4 > s.end) { ok = 12; break; }
        }
        if (ok >= 12) { found = p; break; }
    }
    start[k] = found;
}

// number of ops of a record's normalised op list (see norm_ops in phz_bam.cpp / soa.pack_sam); out != nullptr: write them
__device__ int norm_ops_dev(const uint8_t *cig, int n_cig, int nb, uint32_t *out) {
    int n = 0;
    long long read_pos = 0;
    for (int i = 0; i < n_cig; i++) {
        const uint32_t c = ld32(cig + 4 * i);
        const uint32_t op = c & 15, len = c >> 4;
        if (op == 0 || op == 7 || op == 8) {
            const long long lo = read_pos < nb ? read_pos : nb, hi = read_pos + (long long)len < nb ? read_pos + (long long)len : nb;
            const uint32_t avail = (uint32_t)(hi > lo ? hi - lo : 0);
            if (avail == len) { if (out) out[n] = c; n++; }
            else {
                if (avail) { if (out) out[n] = (avail << 4) | op; n++; }
                if (out) out[n] = ((len - avail) << 4) | 9u;
                n++;
            }
            read_pos += len;
        } else if (op == 1) {
            const long long lo = read_pos < nb ? read_pos : nb, hi = read_pos + (long long)len < nb ? read_pos + (long long)len : nb;
            const uint32_t avail = (uint32_t)(hi > lo ? hi - lo : 0);
            if (out) out[n] = (avail << 4) | 1u;
            n++;
            read_pos += len;
        } else if (op == 4) {
            if (out) out[n] = c;
            n++;
            read_pos += len;
        } else if (op == 2 || op == 3) {
            if (out) out[n] = c;
            n++;
        }                                  // H, P and anything else leave no trace (as in the host packer)
    }
    return n;
}

struct SegOut {                 // per segment, written by the counting hop
    uint32_t kept;
    uint32_t flags;             // 1 corrupt chain, 2 boundary mismatch, 4 unsorted inside the segment
    int32_t first_ref, first_pos, last_ref, last_pos;      // of the kept records (first_ref = -2 when none)
    uint64_t end_pos;                                      // where the chain arrived (>= the segment's stop)
    uint64_t sq_sum, qn_sum, op_sum;                       // 64-bit sums of base groups, name bytes and 2 x CIGAR ops (upper bound of the
                                                           // normalised op count) over the kept records: the scans below are 32-bit
};

struct KeptOut {                // kept-record list (structure of arrays)
    uint64_t *off; int32_t *ref; uint32_t *nops, *sq, *nb, *lqn;
};

template <int WRITE>
__global__ __launch_bounds__(64) void k_hop(const uint8_t *d, const Seg *segs, int64_t nseg, const uint64_t *start, Filters F, SegOut *so,
                                            const uint32_t *kept_base, KeptOut K) {
    const int64_t k = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (k >= nseg) return;
    const Seg s = segs[k];
    const uint64_t n = s.end;
    uint64_t p = start[k];
    const uint64_t stop = (k + 1 < nseg && segs[k + 1].piece == s.piece) ? start[k + 1] : s.end;
    uint32_t kept = 0, flags = 0;
    uint64_t sq_sum = 0, qn_sum = 0, op_sum = 0;
    int32_t first_ref = -2, first_pos = 0, last_ref = -2, last_pos = 0;
    uint64_t w = WRITE ? kept_base[k] : 0;
    while (p < stop) {
        if (p + 36 > n) { flags |= 1; break; }
        const int32_t bs = ldi32(d + p);
        if (bs < 32 || (uint64_t)bs > n - p - 4) { flags |= 1; break; }
        const uint8_t *r = d + p + 4;
        const int32_t ref = ldi32(r);
        const uint32_t l_rn = r[8], mapq = r[9], n_cig = ld16(r + 12), flag = ld16(r + 14);
        const int32_t l_seq = ldi32(r + 16), tlen = ldi32(r + 28);
        if (l_seq < 0 || l_rn < 1 || 32 + (uint64_t)l_rn + 4 * (uint64_t)n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq > (uint64_t)bs) { flags |= 1; break; }
        bool keep = ref >= 0 && ref < F.n_ref && F.ref_mask[ref] && (int)mapq >= F.min_mapq && ((int)flag & F.flag_required) == F.flag_required &&
                    ((int)flag & F.flag_forbidden) == 0;
        if (keep && F.isize_cutoff != 0) { const double tl = tlen < 0 ? -(double)tlen : (double)tlen; keep = tl <= F.isize_cutoff; }
        if (keep) {
            const int32_t pos0 = ldi32(r + 4);
            if (last_ref != -2 && (ref < last_ref || (ref == last_ref && pos0 < last_pos))) flags |= 4;
            if (first_ref == -2) { first_ref = ref; first_pos = pos0; }
            last_ref = ref; last_pos = pos0;
            if (!WRITE) {
                const uint8_t *qual0 = r + 32 + l_rn + 4 * (uint64_t)n_cig + ((uint64_t)l_seq + 1) / 2;
                const uint32_t nb0 = l_seq <= 0 ? 1u : (qual0[0] == 0xFF ? 1u : (uint32_t)l_seq);
                sq_sum += (nb0 + 3) / 4; qn_sum += l_rn - 1; op_sum += 2 * (uint64_t)n_cig;
            }
            if (WRITE) {
                const uint8_t *cig = r + 32 + l_rn;
                const uint8_t *qual = cig + 4 * (uint64_t)n_cig + ((uint64_t)l_seq + 1) / 2;
                uint32_t nb;
                if (l_seq <= 0) nb = 1; else nb = qual[0] == 0xFF ? 1u : (uint32_t)l_seq;
                K.off[w] = p + 4; K.ref[w] = ref; K.nops[w] = (uint32_t)norm_ops_dev(cig, (int)n_cig, (int)nb, nullptr);
                K.sq[w] = (nb + 3) / 4; K.nb[w] = nb; K.lqn[w] = l_rn - 1;
                w++;
            }
            kept++;
        }
        p += 4 + (uint64_t)bs;
    }
    if (!(flags & 1) && p != stop) flags |= 2;
    if (!WRITE) { SegOut o; o.kept = kept; o.flags = flags; o.first_ref = first_ref; o.first_pos = first_pos; o.last_ref = last_ref; o.last_pos = last_pos;
                  o.sq_sum = sq_sum; o.qn_sum = qn_sum; o.op_sum = op_sum; o.end_pos = p; so[k] = o; }
}

__global__ void k_seg_kept(const SegOut *so, int64_t nseg, uint32_t *kept) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nseg) kept[k] = so[k].kept;
}

// ref_begin[r] = first kept record with reference >= r (the list is grouped by reference: the file is coordinate-sorted)
__global__ void k_ref_bounds(const int32_t *ref, int64_t n, int n_ref, int64_t *ref_begin) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_ref) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (ref[m] < r) lo = m + 1; else hi = m; }
    ref_begin[r] = lo;
}

struct DevShard {               // device pointers of one reference's output arrays (caller-allocated)
    int32_t *pos; uint32_t *cigar_off, *cigar, *seq_off; uint8_t *seq2, *qual; int32_t *aln; uint8_t *has_as; uint32_t *qname_off; char *qnames;
};

// value of the last AS tag among the aux fields [p, e); false when absent (same walk as aux_as in phz_bam.cpp)
__device__ bool aux_as_dev(const uint8_t *p, const uint8_t *e, int32_t *out) {
    bool found = false;
    while (p + 3 <= e) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        uint64_t sz = 0;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { const uint8_t *q = p; while (q < e && *q) q++; if (q >= e) return found; sz = (uint64_t)(q - p) + 1; break; }
            case 'B': {
                if (p + 5 > e) return found;
                const uint8_t sub = p[0];
                const uint32_t cnt = ld32(p + 1);
                uint64_t es = 0;
                switch (sub) { case 'c': case 'C': es = 1; break; case 's': case 'S': es = 2; break; case 'i': case 'I': case 'f': es = 4; break; default: return found; }
                sz = 5 + es * (uint64_t)cnt;
                break;
            }
            default: return found;
        }
        if (sz > (uint64_t)(e - p)) return found;
        if (t0 == 'A' && t1 == 'S' && ty != 'A' && ty != 'f' && ty != 'Z' && ty != 'H' && ty != 'B') {
            int32_t v = 0;
            switch (ty) {
                case 'c': v = (int8_t)p[0]; break;
                case 'C': v = p[0]; break;
                case 's': v = (int16_t)ld16(p); break;
                case 'S': v = (int32_t)ld16(p); break;
                case 'i': v = ldi32(p); break;
                case 'I': v = (int32_t)ld32(p); break;
            }
            *out = v; found = true;
        }
        p += sz;
    }
    return found;
}

// l_seq of the record at r (phase B needs it for the offset of the qualities)
__device__ __forceinline__ uint32_t nb_seq_len(const uint8_t *r) { const int32_t l = ldi32(r + 16); return l > 0 ? (uint32_t)l : 0u; }

__constant__ int8_t c_code_of[16] = {-2, 0, 1, -2, 2, -2, -2, -2, 3, -2, -2, -2, -2, -1, -2, -1};   // =ACMGRSVTWYHKDBN

// One wave per 64 kept records.  Phase A, lane = record: the scalar columns, the normalised CIGAR and the AS tag (a few dependent
// loads per lane).  Phase B, lane = byte: the wave walks its 64 records and moves each one's name, bases and qualities with
// coalesced accesses -- a lane reading its own record byte by byte pulls a whole cache line per byte through L2 (the first version
// of this kernel did, and was bound by exactly that: 0.21 s for 59M records).
__global__ __launch_bounds__(256) void k_pack(const uint8_t *d, KeptOut K, int64_t n_kept, const int64_t *ref_begin, const uint32_t *co,
                                              const uint32_t *so, const uint32_t *qo, const DevShard *shards) {
    __shared__ const uint8_t *s_src[256];
    __shared__ char *s_qn[256];
    __shared__ uint8_t *s_o2[256], *s_oq[256];
    __shared__ uint32_t s_meta[256];           // l_rn | n_cig << 8 | star << 24 (SEQ '*')
    __shared__ uint32_t s_nb[256];
    const int tid = threadIdx.x, lane = tid & 63, wbase = tid & ~63;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    const bool live = i < n_kept;
    if (live) {
        const int ref = K.ref[i];
        const int64_t b = ref_begin[ref], e = ref_begin[ref + 1];
        const int64_t k = i - b;
        const DevShard S = shards[ref];
        const uint32_t c0 = co[i] - co[b], s0 = so[i] - so[b], q0 = qo[i] - qo[b];
        const uint8_t *r = d + K.off[i];
        const uint32_t l_rn = r[8], n_cig = ld16(r + 12);
        const int32_t l_seq = ldi32(r + 16);
        const int32_t bs = ldi32(r - 4);
        const uint32_t nb = K.nb[i];
        S.pos[k] = ldi32(r + 4) + 1;
        S.cigar_off[k] = c0; S.seq_off[k] = s0; S.qname_off[k] = q0;
        if (i + 1 == e) { S.cigar_off[k + 1] = co[e] - co[b]; S.seq_off[k + 1] = so[e] - so[b]; S.qname_off[k + 1] = qo[e] - qo[b]; }
        const uint8_t *cig = r + 32 + l_rn;
        norm_ops_dev(cig, (int)n_cig, (int)nb, S.cigar + c0);
        const uint8_t *ql = cig + 4 * (uint64_t)n_cig + ((uint64_t)(l_seq > 0 ? l_seq : 0) + 1) / 2;
        int32_t as = 0;
        const bool has = aux_as_dev(ql + (l_seq > 0 ? l_seq : 0), r + bs, &as);
        S.aln[k] = has ? as : 0; S.has_as[k] = has ? 1 : 0;
        s_src[tid] = r; s_qn[tid] = S.qnames + q0; s_o2[tid] = S.seq2 + s0; s_oq[tid] = S.qual + (uint64_t)s0 * 4;
        s_meta[tid] = l_rn | (n_cig << 8) | (l_seq <= 0 ? 1u << 24 : 0u);
        s_nb[tid] = nb;
    } else {
        s_nb[tid] = 0; s_meta[tid] = 0;
    }
    __syncthreads();
    for (int j = 0; j < 64; j++) {
        const uint32_t nb = s_nb[wbase + j];
        if (nb == 0) continue;                 // record beyond the end of the list
        const uint32_t meta = s_meta[wbase + j];
        const uint32_t l_rn = meta & 0xFF, n_cig = (meta >> 8) & 0xFFFF;
        const uint8_t *r = s_src[wbase + j];
        char *qn = s_qn[wbase + j];
        for (uint32_t t = lane; t + 1 < l_rn; t += 64) qn[t] = (char)r[32 + t];
        uint8_t *o2 = s_o2[wbase + j], *oq = s_oq[wbase + j];
        if (meta >> 24) {                      // SEQ '*' QUAL '*': one IUPAC-other character with phred 9
            if (lane < 4) oq[lane] = lane == 0 ? (uint8_t)(9 | 0x80) : 0;
            if (lane == 0) o2[0] = 1;
            continue;
        }
        const uint8_t *sq = r + 32 + l_rn + 4 * (uint64_t)n_cig;
        const uint8_t *ql = sq + ((uint64_t)nb_seq_len(r) + 1) / 2;
        const bool noq = ql[0] == 0xFF;
        const uint32_t padded = ((nb + 3) / 4) * 4;
        for (uint32_t jb = lane; jb < ((padded + 63) & ~63u); jb += 64) {
            uint32_t code2 = 0; uint8_t q = 0;
            if (jb < nb) {
                const uint8_t by = sq[jb >> 1];
                const uint8_t nib = (jb & 1) ? (by & 15) : (by >> 4);
                const int c = c_code_of[nib];
                q = noq ? 9 : (ql[jb] > 127 ? 127 : ql[jb]);
                if (c >= 0) code2 = (uint32_t)c; else { code2 = c == -1 ? 0u : 1u; q |= 0x80; }
            }
            uint32_t v = code2 << (2 * (jb & 3));
            v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2);
            if (jb < padded) {
                oq[jb] = q;
                if ((jb & 3) == 0) o2[jb >> 2] = (uint8_t)v;
            }
        }
    }
}

// ---- QNAME interning on the device: ids in first-appearance order (what a sequential dictionary would hand out; mates and
// repeated templates share an id), continuing the numbering of the BAMs seen before.  The names of the ids handed out so far live
// in a device-side store (blob + offsets, id order).  Per call: an open-addressing table keyed by a 64-bit hash of the name, equality
// decided by comparing the name bytes; a slot belongs to one name for good.  Slot values: an OLD id (< 2^31), or 2^31 | record index
// of a name that is new in this call -- that value converges to the smallest record index of the name.
constexpr uint32_t NEWBIT = 0x80000000u, EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ uint64_t name_hash(const char *p, uint32_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; i < n; i++) { h ^= (uint8_t)p[i]; h *= 0x100000001b3ull; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h;
}
__device__ __forceinline__ bool bytes_eq(const char *pa, uint32_t la, const char *pb, uint32_t lb) {
    if (la != lb) return false;
    for (uint32_t i = 0; i < la; i++) if (pa[i] != pb[i]) return false;
    return true;
}
// the names already in the store claim their slots (they are distinct: no comparisons)
__global__ __launch_bounds__(256) void k_intern_old(const char *store, const uint32_t *store_off, int64_t n_old, uint32_t *table, uint32_t mask) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= n_old) return;
    uint32_t s = (uint32_t)name_hash(store + store_off[id], store_off[id + 1] - store_off[id]) & mask;
    while (atomicCAS(&table[s], EMPTY, (uint32_t)id) != EMPTY) s = (s + 1) & mask;
}
__global__ __launch_bounds__(256) void k_intern_insert(const char *blob, const uint32_t *off, int64_t n, const char *store, const uint32_t *store_off,
                                                       uint32_t *table, uint32_t mask, uint32_t *slot_of) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const char *nm = blob + off[i];
    const uint32_t ln = off[i + 1] - off[i];
    const uint32_t mine = NEWBIT | (uint32_t)i;
    uint32_t s = (uint32_t)name_hash(nm, ln) & mask;
    for (;;) {
        uint32_t cur = table[s];
        if (cur == EMPTY) {
            cur = atomicCAS(&table[s], EMPTY, mine);
            if (cur == EMPTY) break;                             // claimed for this name
        }
        if (cur & NEWBIT) {                                      // a name that is new in this call: compare with that record's name
            const uint32_t j = cur & ~NEWBIT;
            if (j == (uint32_t)i || bytes_eq(nm, ln, blob + off[j], off[j + 1] - off[j])) { atomicMin(&table[s], mine); break; }
        } else if (bytes_eq(nm, ln, store + store_off[cur], store_off[cur + 1] - store_off[cur])) break;      // a name of an earlier BAM
        s = (s + 1) & mask;
    }
    slot_of[i] = s;
}
__global__ __launch_bounds__(256) void k_intern_first(const uint32_t *table, const uint32_t *slot_of, int64_t n, uint32_t *first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) first[i] = table[slot_of[i]] == (NEWBIT | (uint32_t)i) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_intern_assign(const uint32_t *table, const uint32_t *slot_of, const uint32_t *first, const uint32_t *rank, int64_t n,
                                                       int32_t base, int32_t *qid, int32_t *first_idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = table[slot_of[i]];
    qid[i] = (v & NEWBIT) ? base + (int32_t)rank[v & ~NEWBIT] : (int32_t)v;
    if (first[i]) first_idx[rank[i]] = (int32_t)i;
}
// the names of the new ids appended to a store: name k = blob[off[first_idx[k]] ..), written at dst_off[k]
__global__ __launch_bounds__(256) void k_names_gather(const char *blob, const uint32_t *off, const int32_t *first_idx, int64_t m, const uint32_t *dst_off, char *dst) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const int32_t i = first_idx[k];
    const char *src = blob + off[i];
    const uint32_t ln = off[i + 1] - off[i];
    char *d = dst + dst_off[k];
    for (uint32_t t = 0; t < ln; t++) d[t] = src[t];
}
__global__ void k_add_u32(uint32_t *a, int64_t n, uint32_t x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += x;
}
__global__ __launch_bounds__(256) void k_names_len(const uint32_t *off, const int32_t *first_idx, int64_t m, uint32_t *len) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < m) { const int32_t i = first_idx[k]; len[k] = off[i + 1] - off[i]; }
}

}  // namespace

// The two big buffers of a device BAM read are kept in the ctx between calls (the larger one wins a slot): take / give back
static bool bam_keep_buffers() { const char *e = getenv("PHZ_BAM_KEEP_BUFFERS"); return !(e && atoi(e) == 0); }
static void *bam_take(DevBuf *slot, size_t bytes, size_t *cap) {
    if (slot && slot->p && slot->cap >= bytes) { void *p = slot->p; *cap = slot->cap; slot->p = nullptr; slot->cap = 0; return p; }
    // a fresh buffer gets ~3 % + 64 MB of head room: the BAMs of one sample differ by fractions of a percent, and a buffer cut to the byte sent every slightly larger
    // file to a new 15 GB hipMalloc (0.2-0.7 s each time on this runtime: profiles/r06/cli_4bam_full_before.txt, 'device buffers 684 ms' at the third of four BAMs)
    void *p = nullptr;
    const size_t roomy = bytes + bytes / 32 + ((size_t)64 << 20);
    if (bam_keep_buffers() && hipMalloc(&p, roomy) == hipSuccess) { *cap = roomy; return p; }
    (void)hipGetLastError();
    p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *cap = bytes;
    return p;
}
static void bam_give_back(DevBuf *slot, void *p, size_t cap) {
    if (!p) return;
    if (slot && bam_keep_buffers() && cap > slot->cap) { if (slot->p) (void)hipFree(slot->p); slot->p = p; slot->cap = cap; return; }
    (void)hipFree(p);
}

struct phz_bamdev {
    phz_ctx *ctx = nullptr;
    std::vector<std::pair<std::string, int32_t>> refs;
    void *d_stream = nullptr;           // inflated bytes of the needed members
    size_t d_stream_cap = 0;            // its size (the buffer goes back to the ctx's cache when the handle is closed)
    void *d_work = nullptr;             // kept-record list + offsets (one allocation)
    size_t d_work_cap = 0;
    KeptOut K{};
    int64_t n_kept = 0;
    uint32_t *co = nullptr, *so = nullptr, *qo = nullptr;      // exclusive prefix sums over the kept list ([n_kept + 1] each)
    int64_t *d_ref_begin = nullptr;
    std::vector<int64_t> ref_begin;     // host copy [n_ref + 1]
    std::vector<uint32_t> h_co, h_so, h_qo;     // values at the reference boundaries
    std::string err;
    ~phz_bamdev() {
        if (d_stream) bam_give_back(ctx ? &ctx->bam_stream : nullptr, d_stream, d_stream_cap);
        if (d_work) bam_give_back(ctx ? &ctx->bam_work : nullptr, d_work, d_work_cap);
    }
};

#define BD_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (call);                                                                           \
        if (_e != hipSuccess) { phz_fail(ctx, PHZ_E_HIP, #call, _e); delete h; phz_bam_plan_release(&plan); return PHZ_E_HIP; } \
    } while (0)

extern "C" {

int phz_bamdev_open(phz_ctx *ctx, const char *path, const char *const *ref_names, int n_names, const phz_bam_filters *f, phz_bamdev **out) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !path || !f || !out) return PHZ_E_ARG;
    *out = nullptr;
    const bool timing = getenv("PHZ_TIMING") != nullptr;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[phz timing]     bam device: %-40s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    PhzBamPlan plan;
    if (int st = phz_bam_plan_file(path, ref_names, n_names, &plan)) { phz_bam_plan_release(&plan); return st; }
    lap("plan (member table, header, chromosome ranges)");
    phz_bamdev *h = new phz_bamdev();
    h->ctx = ctx;
    h->refs = plan.refs;
    const int n_ref = (int)plan.refs.size();
    h->ref_begin.assign((size_t)n_ref + 1, 0);
    h->h_co.assign((size_t)n_ref + 1, 0); h->h_so.assign((size_t)n_ref + 1, 0); h->h_qo.assign((size_t)n_ref + 1, 0);
    if (plan.members.empty() || plan.pieces.empty()) { phz_bam_plan_release(&plan); *out = h; return PHZ_OK; }
    if (hipSetDevice(ctx->device) != hipSuccess) { delete h; phz_bam_plan_release(&plan); return PHZ_E_HIP; }
    hipStream_t sm = ctx->stream;
    hipEvent_t e0 = ctx->ev0, e1 = ctx->ev1;
    // ---- H2D of the compressed bytes: one contiguous file range per run of members; device member table with local offsets
    std::vector<phz_bgzf_member> mem(plan.members.size());
    uint64_t comp_bytes = 0, out_bytes = 0;
    std::vector<std::pair<uint64_t, uint64_t>> runs;          // file ranges [a, b)
    std::vector<uint64_t> run_dev;                            // their offsets in the device buffer
    {
        uint64_t run_a = 0, run_b = 0;
        for (size_t i = 0; i < plan.members.size(); i++) {
            const auto &m = plan.members[i];
            const uint64_t a = m.src, b = m.src + m.csize;
            if (i == 0 || a > run_b + 65536) {
                if (i) { runs.emplace_back(run_a, run_b); }
                run_a = a; run_b = b;
            } else run_b = b;
        }
        runs.emplace_back(run_a, run_b);
        for (auto &r : runs) { run_dev.push_back(comp_bytes); comp_bytes += (r.second - r.first + 15) & ~(uint64_t)15; }
    }
    // inflated layout: the needed members back to back in file order (mem_dev_dst[i] = where member i's output starts)
    std::vector<uint64_t> mem_dev_dst(plan.members.size());
    {
        size_t ri = 0;
        for (size_t i = 0; i < plan.members.size(); i++) {
            const auto &m = plan.members[i];
            while (ri + 1 < runs.size() && m.src >= runs[ri].second) ri++;
            mem[i].src = run_dev[ri] + (m.src - runs[ri].first);
            mem[i].csize = m.csize; mem[i].isize = m.isize; mem[i].dst = out_bytes;
            mem_dev_dst[i] = out_bytes;
            out_bytes += m.isize;
        }
    }
    auto dev_of = [&](uint64_t u) -> uint64_t {                // global inflated offset -> device stream offset
        size_t lo = 0, hi = plan.members.size();
        while (lo < hi) { const size_t m = (lo + hi) >> 1; if (plan.members[m].dst + plan.members[m].isize <= u) lo = m + 1; else hi = m; }
        if (lo >= plan.members.size()) return out_bytes;       // u == end of the last member
        return mem_dev_dst[lo] + (u - plan.members[lo].dst);
    };
    void *d_comp = nullptr, *d_mem = nullptr;
    // Every allocation failure of the device path is PHZ_E_NOMEM (the caller then decodes the file on the host; earlier BAMs' shards stay
    // resident, so HBM can legitimately be short here), and the runtime's sticky last-error is cleared so that the next kernel-launch check
    // of this ctx does not report a stale out-of-memory.  PHZ_BAMDEV_FORCE_NOMEM=1 (tests) takes this exit without exhausting a GPU.
    const bool force_nomem = getenv("PHZ_BAMDEV_FORCE_NOMEM") != nullptr;
    size_t d_comp_cap = 0;
    bool got = false;
    for (int attempt = 0; attempt < 2 && !force_nomem && !got; attempt++) {
        d_comp = bam_take(&ctx->bam_comp, comp_bytes + 64, &d_comp_cap);
        if (d_comp && hipMalloc(&d_mem, mem.size() * sizeof(phz_bgzf_member) + mem.size() * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); d_mem = nullptr; }      // member table + the trailers' CRC32s behind it
        if (d_comp && d_mem) h->d_stream = bam_take(&ctx->bam_stream, out_bytes + 64, &h->d_stream_cap);
        got = d_comp && d_mem && h->d_stream;
        if (!got) {             // short of memory: the kept buffers (too small for this file) go back to the runtime, then one more try
            if (d_comp) { (void)hipFree(d_comp); d_comp = nullptr; }
            if (d_mem) { (void)hipFree(d_mem); d_mem = nullptr; }
            if (h->d_stream) { (void)hipFree(h->d_stream); h->d_stream = nullptr; }
            if (ctx->bam_comp.p) { (void)hipFree(ctx->bam_comp.p); ctx->bam_comp = DevBuf(); }
            if (ctx->bam_stream.p) { (void)hipFree(ctx->bam_stream.p); ctx->bam_stream = DevBuf(); }
            if (ctx->bam_work.p) { (void)hipFree(ctx->bam_work.p); ctx->bam_work = DevBuf(); }
        }
    }
    if (!got) {
        (void)hipGetLastError();
        delete h; phz_bam_plan_release(&plan); return phz_fail(ctx, PHZ_E_NOMEM, "device BAM buffers");
    }
    lap("device buffers");
    // H2D and K_inflate overlapped: the members go over in chunks of ~1.3 GB of compressed bytes on a copy stream (the file is pageable
    // memory, so every copy keeps this thread busy staging it), and each chunk's members are inflated on the compute stream as soon as
    // its bytes have arrived -- the copy of chunk c+1 runs while chunk c inflates
    auto t_h2d0 = std::chrono::steady_clock::now();
    constexpr int NCOPY_MAX = 16;
    int NCOPY = 8;                      // host threads (and streams) that read the file and send it over; PHZ_BAM_NCOPY = 1..16
    { const char *e = getenv("PHZ_BAM_NCOPY"); if (e && atoi(e) >= 1 && atoi(e) <= NCOPY_MAX) NCOPY = atoi(e); }
    hipStream_t cs[NCOPY_MAX];
    for (int t = 0; t < NCOPY_MAX; t++) cs[t] = nullptr;
    for (int t = 0; t < NCOPY; t++) if (hipStreamCreateWithFlags(&cs[t], hipStreamNonBlocking) != hipSuccess) cs[t] = nullptr;
    if (phz_reserve(ctx, ctx->scalars, 64) != PHZ_OK || phz_reserve(ctx, ctx->scratch[11], mem.size() * (size_t)phz_inflate_scratch_bytes_per_member()) != PHZ_OK) {
        bam_give_back(&ctx->bam_comp, d_comp, d_comp_cap); (void)hipFree(d_mem); for (auto c : cs) if (c) (void)hipStreamDestroy(c);
        (void)hipGetLastError();
        delete h; phz_bam_plan_release(&plan); return PHZ_E_NOMEM;
    }
    int *d_status = (int *)ctx->scalars.p;
    (void)hipMemsetAsync(d_status, 0, 4, sm);
    (void)hipMemcpyAsync(d_mem, mem.data(), mem.size() * sizeof(phz_bgzf_member), hipMemcpyHostToDevice, sm);
    // every member's output is checked against the CRC32 of its trailer after it has been inflated (htslib does, so the reference's `samtools view` stops on a
    // damaged file that is still valid DEFLATE); PHZ_BAM_CRC=0 skips the check
    std::vector<uint32_t> crcs(plan.members.size());
    for (size_t i = 0; i < plan.members.size(); i++) crcs[i] = plan.members[i].crc;
    uint32_t *d_crc = (uint32_t *)((char *)d_mem + mem.size() * sizeof(phz_bgzf_member));
    bool crc_on = true;
    { const char *e = getenv("PHZ_BAM_CRC"); if (e && atoi(e) == 0) crc_on = false; }
    if (crc_on) (void)hipMemcpyAsync(d_crc, crcs.data(), crcs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, sm);
    (void)hipEventRecord(e0, sm);
    int st = PHZ_OK;
    std::vector<hipEvent_t> evs;
    // PHZ_BAM_REGISTER=1 (experiment, off by default): the mapped file's needed span registered with the runtime, DMA straight out of the page cache.  The
    // copies then run at PCIe speed without host threads (212 against 241 ms for copy + K_inflate of a 3.8 GB BAM), but faulting the mapping's 930,000 pages
    // into the page table costs 177 ms up front (profiles/r05/bam_device_sweep.txt): pread into page-locked staging never maps them and stays the default
    bool reg_ok = false; void *reg_p = nullptr;
    {
        const char *e = getenv("PHZ_BAM_REGISTER");
        if (e && atoi(e) == 1 && !runs.empty() && phz_bam_plan_map(&plan)) {
            const uint64_t r0 = runs.front().first & ~(uint64_t)4095;
            uint64_t r1 = (runs.back().second + 4095) & ~(uint64_t)4095;
            if (r1 > ((plan.file_size + 4095) & ~(uint64_t)4095)) r1 = (plan.file_size + 4095) & ~(uint64_t)4095;
            if (hipHostRegister((void *)(plan.file + r0), (size_t)(r1 - r0), hipHostRegisterDefault) == hipSuccess) { reg_ok = true; reg_p = (void *)(plan.file + r0); }
            else (void)hipGetLastError();
        }
    }
    lap("registration of the mapped file");
    // page-locked staging: NCOPY threads x 2 buffers, kept in the ctx for the next BAM of the sample
    constexpr uint64_t STAGE_BYTES = 8ull << 20;
    const int fd = ::open(path, O_RDONLY);
    bool stage_ok = !reg_ok && fd >= 0 && phz_reserve_host(ctx, ctx->h_bam_stage, (size_t)NCOPY * 2 * STAGE_BYTES) == PHZ_OK;
    if (!stage_ok && !reg_ok) {
        (void)hipGetLastError();
        if (!phz_bam_plan_map(&plan)) {                    // no page-locked staging and no mapping either
            if (fd >= 0) ::close(fd);
            bam_give_back(&ctx->bam_comp, d_comp, d_comp_cap); (void)hipFree(d_mem); for (auto c : cs) if (c) (void)hipStreamDestroy(c);
            delete h; phz_bam_plan_release(&plan); return PHZ_E_NOMEM;
        }
    }
    char *stage = (char *)ctx->h_bam_stage.p;
    hipEvent_t stage_ev[NCOPY_MAX * 2];
    for (auto &e : stage_ev) e = nullptr;
    for (int t = 0; t < NCOPY * 2; t++) if (stage_ok && hipEventCreateWithFlags(&stage_ev[t], hipEventDisableTiming) != hipSuccess) stage_ev[t] = nullptr;
    int n_is = 1, n_launch = 0;
    // K_inflate launches of consecutive chunks overlap only when their streams sit on different HARDWARE queues: the runtime spreads a process's streams over
    // GPU_MAX_HW_QUEUES of them (default 4) and this call alone has eight copy streams -- a copy stream that shares the queue of a running K_inflate waits
    // behind it, which serialised everything (profiles/r05/bam_device_sweep_streams.txt: 12 chunks on 4 streams 0.69 s with 4 queues, 0.30 s with 16).  With
    // >= 16 queues (the package asks for them before the runtime starts, phaser_amd/__init__.py): 3 streams x 1,280 MB chunks, file -> shards 0.24 s; a host
    // application that keeps the runtime's default gets ONE launch per 4 GB (0.27 s; 0.32 s with the old 1,280 MB chunks on one stream).
    bool many_queues = false;
    { const char *q = getenv("GPU_MAX_HW_QUEUES"); many_queues = q && atoi(q) >= 16 && getenv("PHZ_HW_QUEUES_LATE") == nullptr; }      // (LATE: the variable was set after the runtime had started)
    if (many_queues) n_is = 3;
    { const char *e = getenv("PHZ_BAM_INFLATE_STREAMS"); if (e && atoi(e) >= 1 && atoi(e) <= 8) n_is = atoi(e); }
    std::vector<hipStream_t> is((size_t)n_is, nullptr);
    if (n_is > 1) {
        (void)hipStreamSynchronize(sm);                  // the member table and the cleared status word are on the device before any other stream reads them
        bool all = true;
        for (size_t t = 0; t < is.size(); t++) if (hipStreamCreateWithFlags(&is[t], hipStreamNonBlocking) != hipSuccess) { is[t] = nullptr; all = false; }
        if (!all) {          // one stream could not be made: every stream that was goes away again, the chunks take the ctx stream
            for (size_t t = 0; t < is.size(); t++) if (is[t]) { (void)hipStreamDestroy(is[t]); is[t] = nullptr; }
            n_is = 1;
        }
    }
    {
        // a launch lasts as long as its slowest member (60-100 ms) however few it holds, and the chip holds 262,000 members at once: big chunks (PHZ_BAM_CHUNK_MB)
        const char *ch_env = getenv("PHZ_BAM_CHUNK_MB");
        const uint64_t CH = (ch_env && atoll(ch_env) > 0 ? (uint64_t)atoll(ch_env) : (many_queues ? 1280ull : 4096ull)) << 20;
        size_t ri = 0, i0 = 0;
        while (i0 < plan.members.size() && st == PHZ_OK) {
            while (ri + 1 < runs.size() && plan.members[i0].src >= runs[ri].second) ri++;
            size_t i1 = i0;
            const uint64_t a = plan.members[i0].src;
            uint64_t bnd = a;
            while (i1 < plan.members.size() && plan.members[i1].src < runs[ri].second && plan.members[i1].src + plan.members[i1].csize - a <= CH + (i1 == i0 ? CH : 0)) {
                bnd = plan.members[i1].src + plan.members[i1].csize; i1++;
            }
            if (i1 == i0) { bnd = plan.members[i0].src + plan.members[i0].csize; i1 = i0 + 1; }
            // The chunk goes over in NCOPY slices, each read by its own host thread with pread() into page-locked staging buffers of its own
            // (two per thread, in turn) and sent on its own stream.  Copying out of the mapped file instead -- pageable memory, staged by the
            // runtime at ~13 GB/s per thread -- also populated a page-table entry for every page of the file: 0.16 s of munmap afterwards for
            // a 3.8 GB BAM, on top of 0.34 s for the copy.
            {
                char *dst = (char *)d_comp + run_dev[ri] + (a - runs[ri].first);
                const uint64_t len = bnd - a;
                if (reg_ok) {
                    // the mapped file is registered with the runtime: the DMA engines read the page cache directly, four slices on four streams
                    // (tools/h2d_probe.py on this box: 56 GB/s out of a registered mapping; pread into page-locked staging + copy reached 13 GB/s
                    // here while K_inflate ran -- 52 GB/s alone --, and kept eight host threads busy)
                    const int nsl4 = len >= (64u << 20) ? 4 : 1;
                    bool okc = true;
                    for (int t = 0; t < nsl4; t++) {
                        const uint64_t lo = (len * (uint64_t)t / (uint64_t)nsl4) & ~(uint64_t)4095, hi = t + 1 == nsl4 ? len : ((len * (uint64_t)(t + 1) / (uint64_t)nsl4) & ~(uint64_t)4095);
                        if (hi > lo && hipMemcpyAsync(dst + lo, plan.file + a + lo, hi - lo, hipMemcpyHostToDevice, cs[t] ? cs[t] : sm) != hipSuccess) okc = false;
                    }
                    for (int t = 0; t < nsl4; t++) if (hipStreamSynchronize(cs[t] ? cs[t] : sm) != hipSuccess) okc = false;
                    if (!okc) { st = PHZ_E_HIP; break; }
                } else {
                const int nsl = (cs[0] && stage_ok && len >= (64u << 20)) ? NCOPY : 1;
                std::vector<std::thread> th;
                std::vector<int> thst((size_t)nsl, PHZ_OK);
                for (int t = 0; t < nsl; t++)
                    th.emplace_back([&, t] {
                        (void)hipSetDevice(ctx->device);
                        const uint64_t lo = (len * (uint64_t)t / (uint64_t)nsl) & ~(uint64_t)4095, hi = t + 1 == nsl ? len : ((len * (uint64_t)(t + 1) / (uint64_t)nsl) & ~(uint64_t)4095);
                        hipStream_t s2 = cs[t] ? cs[t] : sm;
                        if (!stage_ok) {                                   // no staging memory: the mapped file, as before
                            if (hipMemcpyAsync(dst + lo, plan.file + a + lo, hi - lo, hipMemcpyHostToDevice, s2) != hipSuccess) thst[(size_t)t] = PHZ_E_HIP;
                            (void)hipStreamSynchronize(s2);
                            return;
                        }
                        int which = 0;
                        for (uint64_t o = lo; o < hi; o += STAGE_BYTES, which ^= 1) {
                            const uint64_t m = hi - o < STAGE_BYTES ? hi - o : STAGE_BYTES;
                            char *sb = stage + ((size_t)t * 2 + (size_t)which) * STAGE_BYTES;
                            if (stage_ev[(size_t)t * 2 + (size_t)which] && hipEventSynchronize(stage_ev[(size_t)t * 2 + (size_t)which]) != hipSuccess) { thst[(size_t)t] = PHZ_E_HIP; return; }
                            uint64_t got = 0;
                            while (got < m) {
                                const ssize_t r = pread(fd, sb + got, (size_t)(m - got), (off_t)(a + o + got));
                                if (r <= 0) { thst[(size_t)t] = PHZ_E_ARG; return; }
                                got += (uint64_t)r;
                            }
                            if (hipMemcpyAsync(dst + o, sb, m, hipMemcpyHostToDevice, s2) != hipSuccess) { thst[(size_t)t] = PHZ_E_HIP; return; }
                            if (stage_ev[(size_t)t * 2 + (size_t)which]) (void)hipEventRecord(stage_ev[(size_t)t * 2 + (size_t)which], s2);
                            else (void)hipStreamSynchronize(s2);
                        }
                        (void)hipStreamSynchronize(s2);                   // the slice is on the device (and the staging buffers are free again)
                    });
                for (auto &x : th) x.join();
                for (int v : thst) if (v != PHZ_OK && st == PHZ_OK) st = v;
                if (st != PHZ_OK) break;
                }
            }
            // the K_inflate launches take turns on n_is streams: a chunk's members start while the previous launches are still running (see above)
            hipStream_t si = n_is > 1 ? is[(size_t)(n_launch % n_is)] : sm;
            st = phz_inflate_launch(ctx, (const uint8_t *)d_comp, (const phz_bgzf_member *)d_mem, (int64_t)i0, (int64_t)(i1 - i0), (uint8_t *)h->d_stream,
                                    (uint8_t *)ctx->scratch[11].p, d_status, si);
            if (st == PHZ_OK && crc_on) st = phz_crc_launch(ctx, (const phz_bgzf_member *)d_mem, (int64_t)i0, (int64_t)(i1 - i0), (const uint8_t *)h->d_stream, d_crc, d_status, si);
            n_launch++;
            i0 = i1;
        }
    }
    for (int t = 0; t < n_is; t++) if (is[(size_t)t]) { (void)hipStreamSynchronize(is[(size_t)t]); (void)hipStreamDestroy(is[(size_t)t]); }
    (void)hipEventRecord(e1, sm);
    int bad = 0;
    (void)hipMemcpyAsync(&bad, d_status, 4, hipMemcpyDeviceToHost, sm);
    (void)hipStreamSynchronize(sm);
    for (auto c : cs) if (c) { (void)hipStreamSynchronize(c); (void)hipStreamDestroy(c); }
    for (auto ev : evs) (void)hipEventDestroy(ev);
    for (auto e : stage_ev) if (e) (void)hipEventDestroy(e);
    if (reg_p) (void)hipHostUnregister(reg_p);
    if (fd >= 0) ::close(fd);
    const double h2d_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_h2d0).count();
    float inflate_ms = 0;
    (void)hipEventElapsedTime(&inflate_ms, e0, e1);
    ctx->last_ms[PHZ_T_INFLATE] = inflate_ms; ctx->total_ms[PHZ_T_INFLATE] += inflate_ms; ctx->launches[PHZ_T_INFLATE]++;
    bam_give_back(&ctx->bam_comp, d_comp, d_comp_cap); (void)hipFree(d_mem);
    if (st != PHZ_OK || bad) { delete h; phz_bam_plan_release(&plan); if (st == PHZ_OK) ctx->err = bad == 7 ? "a BGZF member does not give the CRC32 of its trailer" : "a BGZF member is not valid DEFLATE"; return st != PHZ_OK ? st : PHZ_E_UNSUPPORTED; }
    lap("H2D + K_inflate (+ free of the compressed copy)");
    phz_bam_plan_release(&plan);         // closes the file (nothing was mapped unless a fallback copied out of a mapping)
    lap("release of the plan");
    // ---- segments
    const uint64_t SEG = 256u << 10;
    std::vector<Seg> segs;
    for (size_t pi = 0; pi < plan.pieces.size(); pi++) {
        const uint64_t a = dev_of(plan.pieces[pi].first), b = a + (plan.pieces[pi].second - plan.pieces[pi].first);
        for (uint64_t g = a; g < b; g += SEG) segs.push_back({g, b, g == a ? 1u : 0u, (uint32_t)pi});
    }
    const int64_t nseg = (int64_t)segs.size();
    if (nseg == 0) { *out = h; return PHZ_OK; }
    void *d_seg = nullptr;
    const size_t seg_bytes = ((size_t)nseg * sizeof(Seg) + 255) & ~(size_t)255, start_bytes = ((size_t)(nseg + 1) * 8 + 255) & ~(size_t)255,
                 so_bytes = ((size_t)nseg * sizeof(SegOut) + 255) & ~(size_t)255, kept_bytes = ((size_t)(nseg + 2) * 4 + 255) & ~(size_t)255;
    const size_t mask_bytes = ((size_t)n_ref + 255) & ~(size_t)255;
    if (hipMalloc(&d_seg, seg_bytes + start_bytes + so_bytes + 2 * kept_bytes + mask_bytes) != hipSuccess) { (void)hipGetLastError(); delete h; return phz_fail(ctx, PHZ_E_NOMEM, "device BAM segments"); }
    Seg *dsegs = (Seg *)d_seg;
    uint64_t *dstart = (uint64_t *)((char *)d_seg + seg_bytes);
    SegOut *dso = (SegOut *)((char *)dstart + start_bytes);
    uint32_t *dkept = (uint32_t *)((char *)dso + so_bytes), *dkbase = (uint32_t *)((char *)dkept + kept_bytes);
    uint8_t *dmask = (uint8_t *)((char *)dkbase + kept_bytes);
    std::vector<uint8_t> mask((size_t)n_ref, ref_names ? 0 : 1);
    if (ref_names) for (int i = 0; i < n_ref; i++) for (int k = 0; k < n_names; k++) if (h->refs[(size_t)i].first == ref_names[k]) mask[(size_t)i] = 1;
    auto fail = [&](int code, const char *what) { (void)hipFree(d_seg); if (what) ctx->err = what; delete h; return code; };
    if (hipMemcpyAsync(dsegs, segs.data(), (size_t)nseg * sizeof(Seg), hipMemcpyHostToDevice, sm) != hipSuccess ||
        hipMemcpyAsync(dmask, mask.data(), (size_t)n_ref, hipMemcpyHostToDevice, sm) != hipSuccess) return fail(PHZ_E_HIP, "hipMemcpyAsync");
    Filters F; F.min_mapq = f->min_mapq; F.flag_required = f->flag_required; F.flag_forbidden = f->flag_forbidden; F.isize_cutoff = f->isize_cutoff;
    F.ref_mask = dmask; F.n_ref = n_ref;
    const uint8_t *d = (const uint8_t *)h->d_stream;
    (void)hipEventRecord(e0, sm);
    const unsigned gseg = (unsigned)((nseg + 63) / 64);
    hipLaunchKernelGGL(k_seg_start, dim3(gseg), dim3(64), 0, sm, d, (const Seg *)dsegs, nseg, n_ref, dstart);
    KeptOut none{};
    std::vector<SegOut> hso((size_t)nseg);
    uint32_t total_kept = 0;
    // The counting hop, repeated while a guessed boundary turns out to be a fake (a byte pattern inside a record that passes the
    // plausibility chain): segment k's chain is the truth when its own start is, so where it ARRIVES becomes the start of segment
    // k + 1, and the hop runs again.  Every boundary is verified in the end, or the file goes to the host path.
    for (int pass = 0;; pass++) {
        hipLaunchKernelGGL(k_hop<0>, dim3(gseg), dim3(64), 0, sm, d, (const Seg *)dsegs, nseg, (const uint64_t *)dstart, F, dso, (const uint32_t *)nullptr, none);
        if (hipMemcpyAsync(hso.data(), dso, (size_t)nseg * sizeof(SegOut), hipMemcpyDeviceToHost, sm) != hipSuccess || hipStreamSynchronize(sm) != hipSuccess)
            return fail(PHZ_E_HIP, "segment read-back");
        int repaired = 0;
        bool prev_clean = true;
        for (int64_t k = 0; k < nseg; k++) {
            const SegOut &o = hso[(size_t)k];
            const bool last_of_piece = !(k + 1 < nseg && segs[(size_t)k + 1].piece == segs[(size_t)k].piece);
            if ((o.flags & 1) && prev_clean) return fail(PHZ_E_ARG, "truncated or corrupt BAM record");
            if ((o.flags & 2) && prev_clean) {
                if (last_of_piece) return fail(PHZ_E_ARG, "truncated or corrupt BAM record");      // ran past the end of the piece
                if (hipMemcpyAsync(dstart + k + 1, &hso[(size_t)k].end_pos, 8, hipMemcpyHostToDevice, sm) != hipSuccess) return fail(PHZ_E_HIP, "boundary repair");
                repaired++;
            }
            prev_clean = (o.flags & 3) == 0;
        }
        if (!repaired) {
            bool any = false;
            for (auto &o : hso) if (o.flags & 3) any = true;
            if (!any) break;
        }
        if (timing) fprintf(stderr, "[phz timing]     bam device: %d guessed record boundaries were fakes, repaired from the preceding chain (pass %d)\n", repaired, pass);
        if (pass == 7) return fail(PHZ_E_UNSUPPORTED, "record boundaries did not converge");
        if (hipStreamSynchronize(sm) != hipSuccess) return fail(PHZ_E_HIP, "boundary repair");
    }
    hipLaunchKernelGGL(k_seg_kept, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, sm, (const SegOut *)dso, nseg, dkept);
    if (int s2 = scan_excl(ctx, dkept, dkbase, nseg, ctx->scratch[6])) return fail(s2, nullptr);
    if (hipMemcpyAsync(&total_kept, dkbase + nseg, 4, hipMemcpyDeviceToHost, sm) != hipSuccess || hipStreamSynchronize(sm) != hipSuccess)
        return fail(PHZ_E_HIP, "segment read-back");
    {   // records sorted (inside segments and across them); 64-bit totals within the 32-bit scans
        int32_t lr = -2, lp = 0;
        uint64_t sum = 0, sq = 0, qn = 0, ops = 0;
        for (int64_t k = 0; k < nseg; k++) {
            const SegOut &o = hso[(size_t)k];
            if (o.flags & 4) return fail(PHZ_E_UNSUPPORTED, "BAM is not coordinate-sorted");
            if (o.first_ref != -2) {
                if (lr != -2 && (o.first_ref < lr || (o.first_ref == lr && o.first_pos < lp))) return fail(PHZ_E_UNSUPPORTED, "BAM is not coordinate-sorted");
                lr = o.last_ref; lp = o.last_pos;
            }
            sum += o.kept; sq += o.sq_sum; qn += o.qn_sum; ops += o.op_sum;
        }
        uint64_t lim = (1ull << 32) - 16;
        { const char *e = getenv("PHZ_BAMDEV_LIMIT"); if (e) lim = (uint64_t)atoll(e); }      // tests lower it to exercise the caller's split
        if (sum >= (1ull << 31) || sq >= lim || qn >= lim || ops >= lim)
            return fail(PHZ_E_UNSUPPORTED, "call exceeds the 32-bit offsets of the device path");
    }
    lap("segments + counting hop");
    const int64_t nk = (int64_t)total_kept;
    h->n_kept = nk;
    // ---- kept-record list + prefix sums
    const size_t NK = (size_t)(nk ? nk : 1);
    const size_t a8 = (NK * 8 + 255) & ~(size_t)255, a4 = ((NK + 1) * 4 + 255) & ~(size_t)255, rb = ((size_t)(n_ref + 2) * 8 + 255) & ~(size_t)255;
    h->d_work = bam_take(&ctx->bam_work, a8 + 8 * a4 + rb, &h->d_work_cap);          // (kept between BAMs like the two stream buffers: this ~1 GB hipMalloc took 0.38 s now and then)
    if (!h->d_work) return fail(PHZ_E_NOMEM, "device BAM record list");
    char *w = (char *)h->d_work;
    h->K.off = (uint64_t *)w; w += a8;
    h->K.ref = (int32_t *)w; w += a4;
    h->K.nops = (uint32_t *)w; w += a4; h->K.sq = (uint32_t *)w; w += a4; h->K.nb = (uint32_t *)w; w += a4; h->K.lqn = (uint32_t *)w; w += a4;
    h->co = (uint32_t *)w; w += a4; h->so = (uint32_t *)w; w += a4; h->qo = (uint32_t *)w; w += a4;
    h->d_ref_begin = (int64_t *)w;
    if (nk > 0) {
        hipLaunchKernelGGL(k_hop<1>, dim3(gseg), dim3(64), 0, sm, d, (const Seg *)dsegs, nseg, (const uint64_t *)dstart, F, dso, (const uint32_t *)dkbase, h->K);
        if (int s2 = scan_excl(ctx, h->K.nops, h->co, nk, ctx->scratch[6])) return fail(s2, nullptr);
        if (int s2 = scan_excl(ctx, h->K.sq, h->so, nk, ctx->scratch[6])) return fail(s2, nullptr);
        if (int s2 = scan_excl(ctx, h->K.lqn, h->qo, nk, ctx->scratch[6])) return fail(s2, nullptr);
    } else {
        (void)hipMemsetAsync(h->co, 0, 4, sm); (void)hipMemsetAsync(h->so, 0, 4, sm); (void)hipMemsetAsync(h->qo, 0, 4, sm);
    }
    hipLaunchKernelGGL(k_ref_bounds, dim3((unsigned)((n_ref + 1 + 63) / 64)), dim3(64), 0, sm, (const int32_t *)h->K.ref, nk, n_ref, h->d_ref_begin);
    (void)hipEventRecord(e1, sm);
    if (hipMemcpyAsync(h->ref_begin.data(), h->d_ref_begin, (size_t)(n_ref + 1) * 8, hipMemcpyDeviceToHost, sm) != hipSuccess || hipStreamSynchronize(sm) != hipSuccess)
        return fail(PHZ_E_HIP, "reference bounds read-back");
    // totals must fit the 32-bit offsets of the scans: 64-bit sums of the per-segment kept counts are fine, the byte sums are
    // checked through the per-reference values (a wrapped scan makes a reference's span negative or absurd)
    for (int r = 0; r <= n_ref; r++) {
        const int64_t i = h->ref_begin[(size_t)r];
        (void)hipMemcpyAsync(&h->h_co[(size_t)r], h->co + i, 4, hipMemcpyDeviceToHost, sm);
        (void)hipMemcpyAsync(&h->h_so[(size_t)r], h->so + i, 4, hipMemcpyDeviceToHost, sm);
        (void)hipMemcpyAsync(&h->h_qo[(size_t)r], h->qo + i, 4, hipMemcpyDeviceToHost, sm);
    }
    if (hipStreamSynchronize(sm) != hipSuccess) return fail(PHZ_E_HIP, "offset read-back");
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ctx->last_ms[PHZ_T_BAMPACK] = ms; ctx->total_ms[PHZ_T_BAMPACK] += ms; ctx->launches[PHZ_T_BAMPACK]++;
    (void)hipFree(d_seg);
    lap("kept-record list + scans");
    if (timing)
        fprintf(stderr, "[phz timing]     bam device: H2D of %.1f MB overlapped with K_inflate (%.1f MB out): %.1f ms wall, first launch to last %.1f ms; boundaries + hop + scans %.1f ms, %lld records kept\n",
                comp_bytes / 1e6, out_bytes / 1e6, h2d_ms, inflate_ms, ms, (long long)nk);
    *out = h;
    return PHZ_OK;
}

int phz_bamdev_close(phz_bamdev *h) { delete h; return PHZ_OK; }
int phz_bamdev_n_ref(const phz_bamdev *h) { return (int)h->refs.size(); }
const char *phz_bamdev_ref_name(const phz_bamdev *h, int i) { return (i < 0 || (size_t)i >= h->refs.size()) ? nullptr : h->refs[(size_t)i].first.c_str(); }
int64_t phz_bamdev_ref_length(const phz_bamdev *h, int i) { return (i < 0 || (size_t)i >= h->refs.size()) ? -1 : (int64_t)h->refs[(size_t)i].second; }

int phz_bamdev_sizes_of(const phz_bamdev *h, int ref, phz_bamdev_sizes *out) {
    if (!h || !out || ref < 0 || (size_t)ref >= h->refs.size()) return PHZ_E_ARG;
    const size_t r = (size_t)ref;
    out->n_reads = h->ref_begin[r + 1] - h->ref_begin[r];
    out->n_ops = (int64_t)(uint32_t)(h->h_co[r + 1] - h->h_co[r]);
    out->n_seq_bytes = (int64_t)(uint32_t)(h->h_so[r + 1] - h->h_so[r]);
    out->n_qname_bytes = (int64_t)(uint32_t)(h->h_qo[r + 1] - h->h_qo[r]);
    return PHZ_OK;
}

// Fill the caller's (device) arrays of every reference: dst[r] is used for reference r when it holds records, ignored otherwise.
int phz_bamdev_pack(phz_bamdev *h, const phz_dev_shard *dst, int n_dst) {
    if (!h || !dst || n_dst != (int)h->refs.size()) return PHZ_E_ARG;
    phz_ctx *ctx = h->ctx;
    PhzEnter phz_guard_(ctx);
    if (h->n_kept == 0) return PHZ_OK;
    static_assert(sizeof(phz_dev_shard) == sizeof(DevShard), "shard pointer table layout");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t sm = ctx->stream;
    if (int s = phz_reserve(ctx, ctx->shard_tab, (size_t)n_dst * sizeof(DevShard))) return s;
    PHZ_HIP(ctx, hipMemcpyAsync(ctx->shard_tab.p, dst, (size_t)n_dst * sizeof(DevShard), hipMemcpyHostToDevice, sm));
    PHZ_HIP(ctx, hipEventRecord(ctx->ev0, sm));
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((h->n_kept + 255) / 256)), dim3(256), 0, sm, (const uint8_t *)h->d_stream, h->K, h->n_kept,
                       (const int64_t *)h->d_ref_begin, (const uint32_t *)h->co, (const uint32_t *)h->so, (const uint32_t *)h->qo,
                       (const DevShard *)ctx->shard_tab.p);
    PHZ_HIP(ctx, hipGetLastError());
    PHZ_HIP(ctx, hipEventRecord(ctx->ev1, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    float ms = 0;
    PHZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms[PHZ_T_BAMPACK] = ms; ctx->total_ms[PHZ_T_BAMPACK] += ms; ctx->launches[PHZ_T_BAMPACK]++;
    if (getenv("PHZ_TIMING")) fprintf(stderr, "[phz timing]     bam device: k_pack %.1f ms for %lld records\n", ms, (long long)h->n_kept);
    return PHZ_OK;
}

// QNAME ids of one shard on the device, continuing a numbering: the store holds the names of ids [0, n_old) in id order
// (store_off[n_old + 1]; NULL / 0 for an empty store).  qid[i] = the id of record i's name (an old id, or n_old + the number of new
// names that first appear before it does) -- exactly what phz_intern assigns.  first_idx[0, *n_new) = record of the first occurrence
// of every new name, in id order.
int phz_intern_device(phz_ctx *ctx, const char *qnames, const uint32_t *qname_off, int64_t n, const char *store, const uint32_t *store_off,
                      int64_t n_old, int32_t *qid, int32_t *first_idx, int64_t *n_new) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !n_new || n < 0 || n_old < 0 || (n > 0 && (!qnames || !qname_off || !qid || !first_idx)) || (n_old > 0 && (!store || !store_off)))
        return PHZ_E_ARG;
    *n_new = 0;
    if (n == 0) return PHZ_OK;
    if (n + n_old >= (1ll << 31) - 16) return PHZ_E_UNSUPPORTED;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t sm = ctx->stream;
    uint64_t cap = 1024;
    while (cap < 2 * (uint64_t)(n + n_old)) cap <<= 1;
    DevBuf *S = ctx->scratch;
    if (int s = phz_reserve(ctx, S[7], cap * 4)) return s;
    if (int s = phz_reserve(ctx, S[8], (size_t)n * 4)) return s;
    if (int s = phz_reserve(ctx, S[9], (size_t)n * 4)) return s;
    if (int s = phz_reserve(ctx, S[10], ((size_t)n + 1) * 4)) return s;
    uint32_t *table = (uint32_t *)S[7].p, *slot_of = (uint32_t *)S[8].p, *first = (uint32_t *)S[9].p, *rank = (uint32_t *)S[10].p;
    PHZ_HIP(ctx, hipMemsetAsync(table, 0xff, cap * 4, sm));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (n_old > 0) hipLaunchKernelGGL(k_intern_old, dim3((unsigned)((n_old + 255) / 256)), dim3(256), 0, sm, store, store_off, n_old, table, (uint32_t)(cap - 1));
    hipLaunchKernelGGL(k_intern_insert, dim3(grid), dim3(256), 0, sm, qnames, qname_off, n, store, store_off, table, (uint32_t)(cap - 1), slot_of);
    hipLaunchKernelGGL(k_intern_first, dim3(grid), dim3(256), 0, sm, (const uint32_t *)table, (const uint32_t *)slot_of, n, first);
    if (int s = scan_excl(ctx, first, rank, n, S[6])) return s;
    hipLaunchKernelGGL(k_intern_assign, dim3(grid), dim3(256), 0, sm, (const uint32_t *)table, (const uint32_t *)slot_of, (const uint32_t *)first,
                       (const uint32_t *)rank, n, (int32_t)n_old, qid, first_idx);
    PHZ_HIP(ctx, hipGetLastError());
    uint32_t total = 0;
    PHZ_HIP(ctx, hipMemcpyAsync(&total, rank + n, 4, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    *n_new = (int64_t)total;
    return PHZ_OK;
}

// Appends the names of new ids to a store: step 1 (dst == NULL): dst_off[0 .. m] = exclusive prefix sums of their lengths starting at
// `base_bytes`, *total_bytes = the store's size afterwards; step 2 (dst = the store blob, at least *total_bytes long): copies the bytes.
int phz_names_append_device(phz_ctx *ctx, const char *qnames, const uint32_t *qname_off, const int32_t *first_idx, int64_t m, uint32_t base_bytes,
                            uint32_t *dst_off, char *dst, int64_t *total_bytes) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || m < 0 || !total_bytes || (m > 0 && (!qnames || !qname_off || !first_idx || !dst_off))) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t sm = ctx->stream;
    if (m == 0) { *total_bytes = base_bytes; return PHZ_OK; }
    const unsigned grid = (unsigned)((m + 255) / 256);
    if (!dst) {
        DevBuf *S = ctx->scratch;
        if (int s = phz_reserve(ctx, S[8], (size_t)m * 4)) return s;
        if (int s = phz_reserve(ctx, S[9], ((size_t)m + 1) * 4)) return s;
        uint32_t *len = (uint32_t *)S[8].p, *pre = (uint32_t *)S[9].p;
        hipLaunchKernelGGL(k_names_len, dim3(grid), dim3(256), 0, sm, qname_off, first_idx, m, len);
        if (int s = scan_excl(ctx, len, pre, m, S[6])) return s;
        uint32_t total = 0;
        PHZ_HIP(ctx, hipMemcpyAsync(&total, pre + m, 4, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipStreamSynchronize(sm));
        if ((uint64_t)total + base_bytes >= (1ull << 32) - 16) return PHZ_E_UNSUPPORTED;
        // dst_off = pre + base_bytes (one more tiny pass keeps scan_excl generic)
        PHZ_HIP(ctx, hipMemcpyAsync(dst_off, pre, ((size_t)m + 1) * 4, hipMemcpyDeviceToDevice, sm));
        if (base_bytes) {
            hipLaunchKernelGGL(k_add_u32, dim3((unsigned)((m + 1 + 255) / 256)), dim3(256), 0, sm, dst_off, m + 1, base_bytes);
            PHZ_HIP(ctx, hipGetLastError());
        }
        PHZ_HIP(ctx, hipStreamSynchronize(sm));
        *total_bytes = (int64_t)total + base_bytes;
        return PHZ_OK;
    }
    hipLaunchKernelGGL(k_names_gather, dim3(grid), dim3(256), 0, sm, qnames, qname_off, first_idx, m, (const uint32_t *)dst_off, dst);
    PHZ_HIP(ctx, hipGetLastError());
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    return PHZ_OK;
}

}  // extern "C"
