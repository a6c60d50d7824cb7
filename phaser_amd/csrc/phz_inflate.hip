// K_inflate: raw-DEFLATE (RFC 1951) decode of BGZF members on the GPU -- the first stage of the device-side BAM path
// (what `samtools view` inflates for phASER at phaser/phaser.py:1346).
//
// A BGZF file is a sequence of independent gzip members of at most 64 KB of payload each (a whole-genome RNA-seq BAM: ~200,000 of
// them), so the parallelism is across members, not inside one: ONE LANE PER MEMBER, 64 members per wave, a few hundred waves in
// flight.  A lane is a slow decoder (one Huffman symbol per step, LZ77 copies through global memory), but the whole chip runs
// tens of thousands of them at once.  Per lane: the per-length code limits of the two Huffman codes live in registers (15
// independent comparisons find a code's length: no bit-serial walk, no divergence between lanes), the symbol tables in LDS
// ([symbol slot][lane] layout: lanes that are at the same slot hit different banks).  The kernel lives on memory latency (rocprofv3, a
// whole-genome BAM: waves wait 74 % of their cycles, the vector ALUs are busy 12 %), so the number of waves a CU can hold is its
// throughput: only the HOT first symbols of the literal/length table in code order -- the shortest codes, i.e. nearly all lookups -- are
// kept in LDS, the rest of the table lives in the member's global scratch (8.3 KB of LDS per wave instead of 26: 16 waves per CU --
// the register file's limit at 125 VGPRs -- instead of 6; HOT = 288 / 128 / 64 / 32: 167 / 142 / 140 / 101 ms for the 248,779 members of a
// whole-genome BAM, the last one because all 3,888 waves are resident at once).
//
// Every member is checked as it is decoded (code validity, distances, output size == ISIZE of the member trailer); a member that
// fails sets the call's status word and the host falls back to zlib for the whole file -- nothing is silently wrong.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "phz.h"
#include "phz_internal.h"

namespace {

constexpr int IL = 64;             // lanes (members) per workgroup
// PHZ_INF_ABL (tools/inflate_ab.sh, timing experiments only -- the output is wrong): 1 = no match copies, 2 = no literal stores.  Whole-genome BAM, 16 waves
// per CU: 109 ms as shipped, 51 ms without the copies, 105 ms without the literal stores, 42 ms without both: half of the kernel is the load -> store round
// trip of the LZ77 copies.  Storing a match's last chunks one step later (their loads under the next symbol's decode) changed nothing (100.7 against
// 100.2 ms): every trip of the loop drains the wave's memory counter anyway -- some lane of the 64 needs its next 16 input bytes, and on gfx9 loads and
// stores share one in-order counter.
#ifndef PHZ_INF_ABL
#define PHZ_INF_ABL 0
#endif
#ifndef PHZ_INF_HOT
#define PHZ_INF_HOT 32
#endif
constexpr int HOT = PHZ_INF_HOT;    // literal/length symbols (code order) kept in LDS; a multiple of 32, <= 288
constexpr int SCRATCH = 352 + 2 * 288;        // global scratch per member: 320 code lengths + 16 uint16 counters used while a table is built + the cold part of the symbol table (uint16 per slot)
constexpr int MAXL = 288, MAXD = 32, MAXLENS = 320;      // MAXL is a multiple of 32, MAXLENS even (bit plane / nibble packing)

struct BitIn {
    // Input arrives 16 bytes at a time and one load AHEAD of its use: `nxt` was requested when `cur` became current, so the
    // ~1 us of a global load is spent decoding the 128 bits before it.  (All 64 lanes of the wave stall when one of them waits.)
    const uint4 *p;                // next aligned 16-byte unit to request
    const uint4 *end;              // one past the last unit that may be read
    uint4 cur, nxt;
    int k;                         // dwords of cur already consumed (0..4)
    uint64_t bb; int bc;
    __device__ __forceinline__ uint4 fetch() { const uint4 z = make_uint4(0, 0, 0, 0); const uint4 v = p < end ? *p : z; p++; return v; }
    __device__ __forceinline__ void start(const uint8_t *at, const uint8_t *stop) {
        const uintptr_t a = (uintptr_t)at;
        p = (const uint4 *)(a & ~(uintptr_t)15);
        end = (const uint4 *)(((uintptr_t)stop + 15) & ~(uintptr_t)15);
        cur = fetch(); nxt = fetch();
        k = (int)((a & 15) >> 2);
        bb = 0; bc = 0;
        refill();
        const int skip = (int)(a & 3) * 8;
        bb >>= skip; bc -= skip;
        refill();
    }
    __device__ __forceinline__ void refill() {
        if (bc <= 32) {
            if (k == 4) { cur = nxt; nxt = fetch(); k = 0; }
            const uint32_t w = k == 0 ? cur.x : (k == 1 ? cur.y : (k == 2 ? cur.z : cur.w));
            k++;
            bb |= (uint64_t)w << bc; bc += 32;
        }
    }
    __device__ __forceinline__ uint32_t bits(int n) {          // n <= 16, caller keeps bc >= n
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        bb >>= n; bc -= n;
        return v;
    }
};

// Canonical Huffman decode without a bit-serial walk: the next 15 bits, reversed (DEFLATE packs codes most significant bit first),
// give the candidate code of every length l as rev >> (15 - l); a code of length l is a real one iff it is below limit[l] =
// first code of that length + number of codes of that length (prefixes of longer codes are >= limit[l]: canonical order).  The 15
// comparisons are independent (no dependency chain, no divergence between lanes), the shortest hit is the length, and the symbol
// sits at base[l] + code in the table that lists symbols in code order.  limit[] lives in registers, base[] in LDS.
template <class BaseAt, class SymAt>
__device__ __forceinline__ int decode_sym(BitIn &in, const uint16_t (&limit)[16], BaseAt base_at, SymAt sym_at) {
    const uint32_t rev = __builtin_bitreverse32((uint32_t)in.bb) >> 17;          // 15 bits, first stream bit on top
    uint32_t hits = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) hits |= ((rev >> (15 - l)) < (uint32_t)limit[l] ? 1u : 0u) << l;
    if (hits == 0) return -1;
    const int len = __builtin_ctz(hits);
    in.bb >>= len; in.bc -= len;
    return sym_at(base_at(len) + (int)(rev >> (15 - len)));
}

struct Member { uint64_t src; uint32_t csize, isize; uint64_t dst; };
typedef uint32_t __attribute__((aligned(1))) u32u;      // unaligned global accesses (gfx950 does them in hardware)
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t v4u_ __attribute__((ext_vector_type(4)));
typedef v4u_ __attribute__((aligned(1))) v4u;            // 16 unaligned bytes in ONE request

__constant__ uint8_t c_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// status codes written to the per-call status word (first failure wins)
enum { INF_OK = 0, INF_BAD_BLOCK = 1, INF_BAD_CODE = 2, INF_BAD_DIST = 3, INF_OVERRUN = 4, INF_SHORT = 5, INF_BAD_LENS = 6, INF_BAD_CRC = 7 };

// Per-lane tables in LDS, [slot][lane] (lanes on the same slot hit different banks), squeezed so that six waves fit a CU's 160 KB (26 KB each):
// literal/length symbols as 8 low bits + a bit plane for bit 8 (values < 288), distance symbols as bytes.
struct Tables {
    uint8_t syml_lo[HOT][IL];
    uint32_t syml_hi[HOT / 32][IL];
    uint8_t symd[MAXD][IL];
    int16_t basel[16][IL], based[16][IL];      // per code length: (index of its first symbol in the table) - (its first code)
};
struct Lane {
    Tables &t; const int lane;
    uint8_t *lens;                 // [MAXLENS] code lengths of the block being set up: global scratch of this member (touched only at
                                   // block headers; keeping them out of LDS is two more waves per CU)
    __device__ __forceinline__ int len_at(int i) const { return lens[i]; }
    __device__ __forceinline__ void set_len(int i, int v) { lens[i] = (uint8_t)v; }
    __device__ __forceinline__ uint16_t *cold() const { return (uint16_t *)(lens + 352); }
    __device__ __forceinline__ int syml(int k) const {
        if (k < HOT) return (int)t.syml_lo[k][lane] | (int)(((t.syml_hi[k >> 5][lane] >> (k & 31)) & 1u) << 8);
        return (int)cold()[k];
    }
    __device__ __forceinline__ void set_syml(int k, int v) {
        if (k < HOT) {
            t.syml_lo[k][lane] = (uint8_t)v;
            uint32_t &w = t.syml_hi[k >> 5][lane];
            w = (w & ~(1u << (k & 31))) | ((uint32_t)(v >> 8) << (k & 31));
        } else cold()[k] = (uint16_t)v;
    }
};

// canonical code of symbols [first, first + n) with lengths len_at(): counts -> cnt (returned in registers), symbols in code order
// through `put`; returns "left" of the Kraft sum (0 complete, > 0 incomplete, < 0 over-subscribed) and the number of coded symbols
template <class Put>
__device__ __forceinline__ int build_code(Lane &L, int first, int n, uint16_t (&limit)[16], int16_t (*base)[IL], int *coded, Put put) {
    const int lane = L.lane;
    uint16_t *tc = (uint16_t *)(L.lens + MAXLENS);       // counts, then running offsets while the table is filled (global scratch: rare)
    for (int l = 0; l < 16; l++) tc[l] = 0;
    for (int i = 0; i < n; i++) tc[L.len_at(first + i)]++;
    uint16_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; l++) cnt[l] = tc[l];
    int left = 1;
#pragma unroll
    for (int l = 1; l <= 15; l++) { left <<= 1; left -= cnt[l]; }
    *coded = n - cnt[0];
    if (left < 0) return left;
    uint32_t off = 0, code = 0;
    limit[0] = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) {
        tc[l] = (uint16_t)off;
        base[l][lane] = (int16_t)((int)off - (int)code);
        limit[l] = (uint16_t)(code + cnt[l]);            // <= 2^l <= 32768
        off += cnt[l];
        code = (code + cnt[l]) << 1;
    }
    for (int i = 0; i < n; i++) { const int l = L.len_at(first + i); if (l) put((int)tc[l]++, i); }
    return left;
}

__global__ __launch_bounds__(IL) void k_inflate(const uint8_t *comp, const Member *mem, int64_t n_members, uint8_t *out, uint8_t *lens_scratch,
                                                int *status) {
    __shared__ Tables T;
    const int lane = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * IL + lane;
    if (m >= n_members) return;
    Lane L{T, lane, lens_scratch + m * SCRATCH};
    const Member M = mem[m];
    uint8_t *o = out + M.dst;
    uint32_t op = 0;
    const uint32_t oend = M.isize;
    if (oend == 0) return;
    BitIn in;
    in.start(comp + M.src, comp + M.src + M.csize);
    int err = INF_OK;
    uint16_t cl[16], cd[16];
    bool last = false;
    // The kernel is bound by the number of memory REQUESTS it makes (a wave's byte store is 64 of them, one per member), not by bytes: literals are
    // collected eight to a register and leave as one 8-byte store -- when the register is full, or when something else needs the output to be
    // complete (a match, the end of a block).  A partly filled register is stored whole: the zero bytes past the run are inside the member's
    // own output and rewritten by what follows.
    uint64_t lit = 0; uint32_t nl = 0;             // pending literals: output bytes [op - nl, op)
    auto flush_literals = [&]() {
        if (nl) {
#if !(PHZ_INF_ABL & 2)
            uint8_t *q = o + (op - nl);
            if (op - nl + 8 <= oend) *(u64u *)q = lit;
            else for (uint32_t i = 0; i < nl; i++) q[i] = (uint8_t)(lit >> (8 * i));
#endif
            nl = 0; lit = 0;
        }
    };
    while (!last && !err) {
        in.refill();
        last = in.bits(1) != 0;
        const uint32_t type = in.bits(2);
        if (type == 0) {                                   // stored: skip to the byte boundary, LEN, ~LEN, bytes
            in.bits(in.bc & 7);
            in.refill();
            const uint32_t len = in.bits(16);
            in.refill();
            const uint32_t nlen = in.bits(16);
            if ((len ^ nlen) != 0xFFFFu) { err = INF_BAD_BLOCK; break; }
            if (op + len > oend) { err = INF_OVERRUN; break; }
            for (uint32_t i = 0; i < len; i++) { in.refill(); o[op++] = (uint8_t)in.bits(8); }
            continue;
        }
        if (type == 3) { err = INF_BAD_BLOCK; break; }
        int nlen = 288, ndist = 30;
        if (type == 1) {                                   // fixed code
            for (int i = 0; i < 144; i++) L.set_len(i, 8);
            for (int i = 144; i < 256; i++) L.set_len(i, 9);
            for (int i = 256; i < 280; i++) L.set_len(i, 7);
            for (int i = 280; i < 288; i++) L.set_len(i, 8);
            ndist = 32;                                    // the fixed distance code has 32 five-bit codes (30 and 31 never occur)
            for (int i = 0; i < 32; i++) L.set_len(288 + i, 5);
        } else {                                           // dynamic code: the code-length code first (its table in the distance array)
            nlen = (int)in.bits(5) + 257; ndist = (int)in.bits(5) + 1;
            const int ncode = (int)in.bits(4) + 4;
            if (nlen > 286 || ndist > 30) { err = INF_BAD_LENS; break; }
            for (int i = 0; i < 19; i++) L.set_len(i, 0);
            for (int i = 0; i < ncode; i++) { in.refill(); L.set_len(c_clorder[i], (int)in.bits(3)); }
            uint16_t cc[16];
            int coded;
            if (build_code(L, 0, 19, cc, T.based, &coded, [&](int k, int sym) { T.symd[k][lane] = (uint8_t)sym; }) < 0) { err = INF_BAD_LENS; break; }
            // literal/length and distance code lengths, run-length coded
            int idx = 0;
            while (idx < nlen + ndist && !err) {
                in.refill();
                const int sym = decode_sym(in, cc, [&](int l) { return (int)T.based[l][lane]; }, [&](int k) { return (int)T.symd[k & (MAXD - 1)][lane]; });
                if (sym < 0) { err = INF_BAD_CODE; break; }
                if (sym < 16) L.set_len(idx++, sym);
                else {
                    int rep, val = 0;
                    in.refill();
                    if (sym == 16) {
                        if (idx == 0) { err = INF_BAD_LENS; break; }
                        val = L.len_at(idx - 1); rep = 3 + (int)in.bits(2);
                    } else if (sym == 17) rep = 3 + (int)in.bits(3);
                    else rep = 11 + (int)in.bits(7);
                    if (idx + rep > nlen + ndist) { err = INF_BAD_LENS; break; }
                    while (rep--) L.set_len(idx++, val);
                }
            }
            if (err) break;
            if (L.len_at(256) == 0) { err = INF_BAD_LENS; break; }          // no end-of-block code
            // the distance lengths follow the literal/length lengths: move them to a fixed place
            for (int i = ndist - 1; i >= 0; i--) L.set_len(288 + i, L.len_at(nlen + i));
        }
        {
            int coded;
            const int left = build_code(L, 0, nlen, cl, T.basel, &coded, [&](int k, int sym) { L.set_syml(k, sym); });
            if (left < 0 || (left > 0 && coded != 1)) { err = INF_BAD_LENS; break; }
            // incomplete is fine for a single one-bit code, and for NO distance codes at all (a block of literals only)
            const int leftd = build_code(L, 288, ndist, cd, T.based, &coded, [&](int k, int sym) { T.symd[k][lane] = (uint8_t)sym; });
            if (leftd < 0 || (leftd > 0 && coded > 1)) { err = INF_BAD_LENS; break; }
        }
        // the block's symbols
        for (;;) {
            in.refill();
            int sym = decode_sym(in, cl, [&](int l) { return (int)T.basel[l][lane]; }, [&](int k) { return L.syml((unsigned)k < (unsigned)MAXL ? k : 0); });
            if (sym < 0) { err = INF_BAD_CODE; break; }
            if (sym < 256) {
                // (a literal run kept in an inner loop of its own, so that the match path below runs once for all lanes that have reached one, was
                // measured: 20 % slower -- the lanes at a match idle through the literal trips instead of advancing with them)
                if (op >= oend) { err = INF_OVERRUN; break; }
                lit |= (uint64_t)(uint32_t)sym << (8 * nl);
                nl++; op++;
#if PHZ_INF_ABL & 2          // timing experiment: literals dropped (wrong output)
                if (nl == 8) { nl = 0; lit = 0; }
#else
                if (nl == 8) { *(u64u *)(o + (op - 8)) = lit; nl = 0; lit = 0; }
#endif
                continue;
            }
            flush_literals();
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { err = INF_BAD_CODE; break; }
            in.refill();
            // base and extra bits of length code sym in closed form (RFC 1951 3.2.5: codes come in groups of four per extra bit): no table load
            const uint32_t lext = sym < 8 ? 0u : (sym == 28 ? 0u : (uint32_t)(sym - 4) >> 2);
            const uint32_t lbase = sym < 8 ? (uint32_t)sym + 3u : (sym == 28 ? 258u : ((4u + ((uint32_t)sym & 3u)) << lext) + 3u);
            const uint32_t len = lbase + in.bits((int)lext);
            in.refill();
            const int ds = decode_sym(in, cd, [&](int l) { return (int)T.based[l][lane]; }, [&](int k) { return (int)T.symd[k & (MAXD - 1)][lane]; });
            if (ds < 0 || ds >= 30) { err = INF_BAD_CODE; break; }
            in.refill();
            const uint32_t dext = ds < 4 ? 0u : (uint32_t)(ds - 2) >> 1;          // distance codes: groups of two per extra bit
            const uint32_t dbase = ds < 4 ? (uint32_t)ds + 1u : ((2u + ((uint32_t)ds & 1u)) << dext) + 1u;
            const uint32_t dist = dbase + in.bits((int)dext);
            if (dist > op) { err = INF_BAD_DIST; break; }
            if (op + len > oend) { err = INF_OVERRUN; break; }
            const uint8_t *src = o + op - dist;
            uint8_t *dst = o + op;
#if PHZ_INF_ABL & 1          // timing experiment: no match copies (wrong output)
            op += len; continue;
#endif
            // LZ77 copy.  Every chunk is a load -> store round trip through L2, so the chunk is as wide as the distance allows; wide
            // chunks may write a few bytes past the match (inside the member's own output, rewritten by what follows)
            if (dist >= 64 && op + ((len + 15u) & ~15u) <= oend) {
                // far enough back that 64 bytes of source cannot overlap what this round writes: up to four 16-byte chunks requested together,
                // then stored -- one round trip per 64 bytes (the wave waits for its LONGEST match every step)
                for (uint32_t i = 0; i < len; i += 64) {
                    const uint32_t r = len - i;
                    v4u a = *(const v4u *)(src + i), b = a, c = a, d = a;
                    if (r > 16) b = *(const v4u *)(src + i + 16);
                    if (r > 32) c = *(const v4u *)(src + i + 32);
                    if (r > 48) d = *(const v4u *)(src + i + 48);
                    *(v4u *)(dst + i) = a;
                    if (r > 16) *(v4u *)(dst + i + 16) = b;
                    if (r > 32) *(v4u *)(dst + i + 32) = c;
                    if (r > 48) *(v4u *)(dst + i + 48) = d;
                }
            } else if (dist >= 16 && op + ((len + 15u) & ~15u) <= oend) {
                for (uint32_t i = 0; i < len; i += 16) *(v4u *)(dst + i) = *(const v4u *)(src + i);
            } else if (dist >= 8 && op + ((len + 7u) & ~7u) <= oend) {
                for (uint32_t i = 0; i < len; i += 8) *(u64u *)(dst + i) = *(const u64u *)(src + i);
            } else if (dist >= 4) {
                uint32_t i = 0;
                for (; i + 4 <= len; i += 4) *(u32u *)(dst + i) = *(const u32u *)(src + i);
                for (; i < len; i++) dst[i] = src[i];
            } else {
                for (uint32_t i = 0; i < len; i++) dst[i] = src[i];
            }
            op += len;
        }
    }
    if (!err && op != oend) err = INF_SHORT;
    if (err) atomicCAS(status, 0, err);
}

// CRC-32 of every member's output against the value in its BGZF trailer (what htslib checks after inflating a block, so `samtools view` -- phaser.py:1346 --
// stops on a damaged file where DEFLATE alone would accept it: a flipped bit inside a literal's code gives another valid literal).  One lane per member like
// K_inflate, four bytes per step (slicing-by-4: four 256-entry tables in LDS, built by the workgroup), the output read in 16-byte requests.
constexpr int CRC_BLOCK = 256;
__global__ __launch_bounds__(CRC_BLOCK) void k_crc32(const uint8_t *out, const Member *mem, int64_t n_members, const uint32_t *want, int *status) {
    __shared__ uint32_t T[4][256];
    {
        uint32_t c = (uint32_t)threadIdx.x;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        T[0][threadIdx.x] = c;
    }
    __syncthreads();
    {
        uint32_t c = T[0][threadIdx.x];
        for (int t = 1; t < 4; t++) { c = T[0][c & 0xffu] ^ (c >> 8); T[t][threadIdx.x] = c; }
    }
    __syncthreads();
    const int64_t m = (int64_t)blockIdx.x * CRC_BLOCK + threadIdx.x;
    if (m >= n_members) return;
    const Member M = mem[m];
    const uint8_t *p = out + M.dst;
    const uint32_t n = M.isize;
    uint32_t crc = 0xFFFFFFFFu, i = 0;
    auto word = [&](uint32_t w) {
        crc ^= w;
        crc = T[3][crc & 0xffu] ^ T[2][(crc >> 8) & 0xffu] ^ T[1][(crc >> 16) & 0xffu] ^ T[0][crc >> 24];
    };
    for (; i + 16 <= n; i += 16) {
        const v4u v = *(const v4u *)(p + i);
        word(v[0]); word(v[1]); word(v[2]); word(v[3]);
    }
    for (; i < n; i++) crc = T[0][(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    if ((crc ^ 0xFFFFFFFFu) != want[m]) atomicCAS(status, 0, (int)INF_BAD_CRC);
}

}  // namespace

// internal: CRC check of members [first, first + count) on stream `s` (after their K_inflate on the same stream); want = the trailers' CRC32s (device, per member of the table)
int phz_crc_launch(phz_ctx *ctx, const phz_bgzf_member *members, int64_t first, int64_t count, const uint8_t *out, const uint32_t *want, int *d_status, hipStream_t s) {
    if (count <= 0) return PHZ_OK;
    hipLaunchKernelGGL(k_crc32, dim3((unsigned)((count + CRC_BLOCK - 1) / CRC_BLOCK)), dim3(CRC_BLOCK), 0, s, out, (const Member *)members + first, count, want + first, d_status);
    PHZ_HIP(ctx, hipGetLastError());
    return PHZ_OK;
}

// internal: members [first, first + count) of a device member table on stream `s`; d_status (device int, zeroed by the caller) and
// the code-length scratch ([n_members * 320] bytes) are shared by all launches of a call
int phz_inflate_launch(phz_ctx *ctx, const uint8_t *comp, const phz_bgzf_member *members, int64_t first, int64_t count, uint8_t *out,
                       uint8_t *lens_scratch, int *d_status, hipStream_t s) {
    if (count <= 0) return PHZ_OK;
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)((count + IL - 1) / IL)), dim3(IL), 0, s, comp, (const Member *)members + first, count, out,
                       lens_scratch + first * SCRATCH, d_status);
    PHZ_HIP(ctx, hipGetLastError());
    return PHZ_OK;
}
int phz_inflate_scratch_bytes_per_member() { return SCRATCH; }

// Inflate `n_members` BGZF members whose compressed bytes sit in device memory.  members[i] = {byte offset of the raw deflate
// stream in comp, its compressed size, ISIZE, offset of its output in out}; comp 16-byte aligned and readable for 16 bytes past the last member.
// *bad receives 0 or the code of the first member that failed (nothing else about the output can be trusted then).
extern "C" int phz_bgzf_inflate_device(phz_ctx *ctx, const uint8_t *comp, const phz_bgzf_member *members, int64_t n_members, uint8_t *out,
                                       int *bad) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !comp || !members || !out || !bad || n_members < 0) return PHZ_E_ARG;
    static_assert(sizeof(phz_bgzf_member) == sizeof(Member), "member record layout");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    *bad = 0;
    if (n_members == 0) return PHZ_OK;
    hipStream_t sm = ctx->stream;
    if (int s = phz_reserve(ctx, ctx->scalars, 64)) return s;
    if (int s = phz_reserve(ctx, ctx->scratch[11], (size_t)n_members * SCRATCH)) return s;
    int *d_status = (int *)ctx->scalars.p;
    PHZ_HIP(ctx, hipMemsetAsync(d_status, 0, 4, sm));
    PHZ_HIP(ctx, hipEventRecord(ctx->ev0, sm));
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)((n_members + IL - 1) / IL)), dim3(IL), 0, sm, comp, (const Member *)members, n_members, out, (uint8_t *)ctx->scratch[11].p, d_status);
    PHZ_HIP(ctx, hipGetLastError());
    PHZ_HIP(ctx, hipEventRecord(ctx->ev1, sm));
    PHZ_HIP(ctx, hipMemcpyAsync(bad, d_status, 4, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    float ms = 0;
    PHZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms[PHZ_T_INFLATE] = ms; ctx->total_ms[PHZ_T_INFLATE] += ms; ctx->launches[PHZ_T_INFLATE]++;
    return PHZ_OK;
}

// CRC-32 of the inflated members in `out` (device) against `crc` (device: the CRC32 field of every member's BGZF trailer, member order): *bad = 0, or 7 when
// some member's bytes do not give its checksum.
extern "C" int phz_bgzf_crc_device(phz_ctx *ctx, const uint8_t *out, const phz_bgzf_member *members, int64_t n_members, const uint32_t *crc, int *bad) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !out || !members || !crc || !bad || n_members < 0) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    *bad = 0;
    if (n_members == 0) return PHZ_OK;
    hipStream_t sm = ctx->stream;
    if (int s = phz_reserve(ctx, ctx->scalars, 64)) return s;
    int *d_status = (int *)ctx->scalars.p;
    PHZ_HIP(ctx, hipMemsetAsync(d_status, 0, 4, sm));
    if (int s = phz_crc_launch(ctx, members, 0, n_members, out, crc, d_status, sm)) return s;
    PHZ_HIP(ctx, hipMemcpyAsync(bad, d_status, 4, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    return PHZ_OK;
}
