// K_inflate: raw-DEFLATE (RFC 1951) decode of BGZF members on the GPU -- the first stage of the device-side BAM path
// (what `samtools view` inflates for phASER at phaser/phaser.py:1346).
//
// A BGZF file is a sequence of independent gzip members of at most 64 KB of payload each (a whole-genome RNA-seq BAM: ~200,000 of
// them), so the parallelism is across members, not inside one: ONE LANE PER MEMBER, 64 members per wave, a few hundred waves in
// flight.  A lane is a slow decoder (bit-serial canonical Huffman decode, byte-wise LZ77 copies through global memory), but the
// whole chip runs tens of thousands of them at once.  Per lane: the code-length counts of the two Huffman codes live in registers
// (the decode loop over code lengths 1..15 is unrolled, so every count is a fixed register), the symbol tables in LDS
// ([symbol slot][lane] layout: lanes that are at the same slot hit different banks).
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
constexpr int MAXL = 288, MAXD = 32, MAXLENS = 320;

struct BitIn {
    const uint32_t *p;             // next aligned dword
    const uint32_t *end;           // one past the last dword that may be read
    uint64_t bb; int bc;
    __device__ __forceinline__ void refill() {
        if (bc <= 32) {
            const uint32_t w = p < end ? *p : 0u;
            p++;
            bb |= (uint64_t)w << bc; bc += 32;
        }
    }
    __device__ __forceinline__ uint32_t bits(int n) {          // n <= 16, caller keeps bc >= n
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        bb >>= n; bc -= n;
        return v;
    }
};

// canonical Huffman decode with the per-length counts in registers: code lengths 1..15, symbols of equal length are consecutive
// in sym[] (ordered by symbol value) -- the classic counting decode, one bit per step, fully unrolled
template <class SymAt>
__device__ __forceinline__ int decode_sym(BitIn &in, const uint16_t (&cnt)[16], SymAt sym_at) {
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= 15; len++) {
        code |= (int)((uint32_t)in.bb & 1u);
        in.bb >>= 1; in.bc--;
        const int count = cnt[len];
        if (code - count < first) return sym_at(index + (code - first));
        index += count; first += count;
        first <<= 1; code <<= 1;
    }
    return -1;
}

struct Member { uint64_t src; uint32_t csize, isize; uint64_t dst; };

__constant__ uint16_t c_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// status codes written to the per-call status word (first failure wins)
enum { INF_OK = 0, INF_BAD_BLOCK = 1, INF_BAD_CODE = 2, INF_BAD_DIST = 3, INF_OVERRUN = 4, INF_SHORT = 5, INF_BAD_LENS = 6 };

__global__ __launch_bounds__(IL) void k_inflate(const uint8_t *comp, const Member *mem, int64_t n_members, uint8_t *out, int *status) {
    __shared__ uint16_t s_syml[MAXL][IL];
    __shared__ uint16_t s_symd[MAXD][IL];
    __shared__ uint8_t s_lens[MAXLENS][IL];
    __shared__ uint16_t s_cnt[16][IL];
    __shared__ uint16_t s_off[16][IL];
    const int lane = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * IL + lane;
    if (m >= n_members) return;
    const Member M = mem[m];
    uint8_t *o = out + M.dst;
    uint32_t op = 0;
    const uint32_t oend = M.isize;
    if (oend == 0) return;
    BitIn in;
    {
        const uintptr_t a = (uintptr_t)(comp + M.src);
        in.p = (const uint32_t *)(a & ~(uintptr_t)3);
        in.end = (const uint32_t *)(((uintptr_t)(comp + M.src + M.csize) + 3) & ~(uintptr_t)3);
        in.bb = 0; in.bc = 0;
        in.refill();
        const int skip = (int)(a & 3) * 8;
        in.bb >>= skip; in.bc -= skip;
        in.refill();
    }
    int err = INF_OK;
    uint16_t cl[16], cd[16];
    bool last = false;
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
            for (int i = 0; i < 144; i++) s_lens[i][lane] = 8;
            for (int i = 144; i < 256; i++) s_lens[i][lane] = 9;
            for (int i = 256; i < 280; i++) s_lens[i][lane] = 7;
            for (int i = 280; i < 288; i++) s_lens[i][lane] = 8;
            ndist = 32;                                    // the fixed distance code has 32 five-bit codes (30 and 31 never occur)
            for (int i = 0; i < 32; i++) s_lens[288 + i][lane] = 5;
        } else {                                           // dynamic code: the code-length code first (tables in the distance arrays)
            nlen = (int)in.bits(5) + 257; ndist = (int)in.bits(5) + 1;
            const int ncode = (int)in.bits(4) + 4;
            if (nlen > 286 || ndist > 30) { err = INF_BAD_LENS; break; }
            for (int i = 0; i < 19; i++) s_lens[i][lane] = 0;
            for (int i = 0; i < ncode; i++) { in.refill(); s_lens[c_clorder[i]][lane] = (uint8_t)in.bits(3); }
            for (int l = 0; l < 16; l++) s_cnt[l][lane] = 0;
            for (int i = 0; i < 19; i++) s_cnt[s_lens[i][lane]][lane]++;
            {
                int left = 1;
                for (int l = 1; l <= 7; l++) { left <<= 1; left -= s_cnt[l][lane]; }
                if (left < 0) { err = INF_BAD_LENS; break; }
            }
            s_off[1][lane] = 0;
            for (int l = 1; l < 15; l++) s_off[l + 1][lane] = s_off[l][lane] + s_cnt[l][lane];
            for (int i = 0; i < 19; i++) { const int l = s_lens[i][lane]; if (l) s_symd[s_off[l][lane]++][lane] = (uint16_t)i; }
            uint16_t cc[16];
#pragma unroll
            for (int l = 0; l < 16; l++) cc[l] = l <= 7 ? s_cnt[l][lane] : (uint16_t)0;
            // literal/length and distance code lengths, run-length coded
            int idx = 0;
            while (idx < nlen + ndist && !err) {
                in.refill();
                const int sym = decode_sym(in, cc, [&](int k) { return (int)s_symd[k & (MAXD - 1)][lane]; });
                if (sym < 0) { err = INF_BAD_CODE; break; }
                if (sym < 16) s_lens[idx++][lane] = (uint8_t)sym;
                else {
                    int rep, val = 0;
                    in.refill();
                    if (sym == 16) {
                        if (idx == 0) { err = INF_BAD_LENS; break; }
                        val = s_lens[idx - 1][lane]; rep = 3 + (int)in.bits(2);
                    } else if (sym == 17) rep = 3 + (int)in.bits(3);
                    else rep = 11 + (int)in.bits(7);
                    if (idx + rep > nlen + ndist) { err = INF_BAD_LENS; break; }
                    while (rep--) s_lens[idx++][lane] = (uint8_t)val;
                }
            }
            if (err) break;
            if (s_lens[256][lane] == 0) { err = INF_BAD_LENS; break; }      // no end-of-block code
            // the distance lengths follow the literal/length lengths: move them to a fixed place
            for (int i = ndist - 1; i >= 0; i--) s_lens[288 + i][lane] = s_lens[nlen + i][lane];
        }
        // literal/length table
        for (int l = 0; l < 16; l++) s_cnt[l][lane] = 0;
        for (int i = 0; i < nlen; i++) s_cnt[s_lens[i][lane]][lane]++;
        {
            int left = 1;
            for (int l = 1; l <= 15; l++) { left <<= 1; left -= s_cnt[l][lane]; }
            if (left < 0 || (left > 0 && nlen - s_cnt[0][lane] != 1)) { err = INF_BAD_LENS; break; }
        }
        s_off[1][lane] = 0;
        for (int l = 1; l < 15; l++) s_off[l + 1][lane] = s_off[l][lane] + s_cnt[l][lane];
        for (int i = 0; i < nlen; i++) { const int l = s_lens[i][lane]; if (l) s_syml[s_off[l][lane]++][lane] = (uint16_t)i; }
#pragma unroll
        for (int l = 0; l < 16; l++) cl[l] = s_cnt[l][lane];
        // distance table
        for (int l = 0; l < 16; l++) s_cnt[l][lane] = 0;
        for (int i = 0; i < ndist; i++) s_cnt[s_lens[288 + i][lane]][lane]++;
        {
            int left = 1;
            for (int l = 1; l <= 15; l++) { left <<= 1; left -= s_cnt[l][lane]; }
            // incomplete is fine for a single one-bit code, and for NO distance codes at all (a block of literals only)
            if (left < 0 || (left > 0 && ndist - s_cnt[0][lane] > 1)) { err = INF_BAD_LENS; break; }
        }
        s_off[1][lane] = 0;
        for (int l = 1; l < 15; l++) s_off[l + 1][lane] = s_off[l][lane] + s_cnt[l][lane];
        for (int i = 0; i < ndist; i++) { const int l = s_lens[288 + i][lane]; if (l) s_symd[s_off[l][lane]++][lane] = (uint16_t)i; }
#pragma unroll
        for (int l = 0; l < 16; l++) cd[l] = s_cnt[l][lane];
        // the block's symbols
        for (;;) {
            in.refill();
            int sym = decode_sym(in, cl, [&](int k) { return (int)s_syml[k < MAXL ? k : 0][lane]; });
            if (sym < 0) { err = INF_BAD_CODE; break; }
            if (sym < 256) {
                if (op >= oend) { err = INF_OVERRUN; break; }
                o[op++] = (uint8_t)sym;
                continue;
            }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { err = INF_BAD_CODE; break; }
            in.refill();
            const uint32_t len = c_lbase[sym] + in.bits(c_lext[sym]);
            in.refill();
            const int ds = decode_sym(in, cd, [&](int k) { return (int)s_symd[k & (MAXD - 1)][lane]; });
            if (ds < 0 || ds >= 30) { err = INF_BAD_CODE; break; }
            in.refill();
            const uint32_t dist = c_dbase[ds] + in.bits(c_dext[ds]);
            if (dist > op) { err = INF_BAD_DIST; break; }
            if (op + len > oend) { err = INF_OVERRUN; break; }
            const uint8_t *src = o + op - dist;
            uint8_t *dst = o + op;
            if (dist >= 4) {
                uint32_t i = 0;
                for (; i + 4 <= len; i += 4) {
                    const uint8_t b0 = src[i], b1 = src[i + 1], b2 = src[i + 2], b3 = src[i + 3];
                    dst[i] = b0; dst[i + 1] = b1; dst[i + 2] = b2; dst[i + 3] = b3;
                }
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

}  // namespace

// Inflate `n_members` BGZF members whose compressed bytes sit in device memory.  members[i] = {byte offset of the raw deflate
// stream in comp, its compressed size, ISIZE, offset of its output in out}; comp must be readable for 8 bytes past the last member.
// *bad receives 0 or the code of the first member that failed (nothing else about the output can be trusted then).
extern "C" int phz_bgzf_inflate_device(phz_ctx *ctx, const uint8_t *comp, const phz_bgzf_member *members, int64_t n_members, uint8_t *out,
                                       int *bad) {
    if (!ctx || !comp || !members || !out || !bad || n_members < 0) return PHZ_E_ARG;
    static_assert(sizeof(phz_bgzf_member) == sizeof(Member), "member record layout");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    *bad = 0;
    if (n_members == 0) return PHZ_OK;
    hipStream_t sm = ctx->stream;
    if (int s = phz_reserve(ctx, ctx->scalars, 64)) return s;
    int *d_status = (int *)ctx->scalars.p;
    PHZ_HIP(ctx, hipMemsetAsync(d_status, 0, 4, sm));
    PHZ_HIP(ctx, hipEventRecord(ctx->ev0, sm));
    hipLaunchKernelGGL(k_inflate, dim3((unsigned)((n_members + IL - 1) / IL)), dim3(IL), 0, sm, comp, (const Member *)members, n_members, out, d_status);
    PHZ_HIP(ctx, hipGetLastError());
    PHZ_HIP(ctx, hipEventRecord(ctx->ev1, sm));
    PHZ_HIP(ctx, hipMemcpyAsync(bad, d_status, 4, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    float ms = 0;
    PHZ_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->last_ms[PHZ_T_INFLATE] = ms; ctx->total_ms[PHZ_T_INFLATE] += ms; ctx->launches[PHZ_T_INFLATE]++;
    return PHZ_OK;
}
