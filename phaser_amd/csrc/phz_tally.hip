// K_tally family: what phaser/phaser.py does between the mapper's call files and the binomial test,
// for one chromosome over all of its BAM shards:
//   k_as_hist      :545-553   AS column histogram (host turns it into numpy.percentile's value)
//   k_line         :1287-1328 process_mapping_result: AS cutoff, allele class, per-variant line counters,
//                             first-appearance index; plus the "last BAM wins" owner of every QNAME's
//                             read_vars list (:576-581, stale-variable quirk)
//   k_keys/sort/k_unique      set construction :636-640 as sorted distinct (QNAME, variant, class) items
//   k_pairs        :1265-1285 + :1602-1632: every QNAME contributes one count to cell (class_a, class_b) of
//                             every variant pair it touches -- the nine set intersections of
//                             test_variant_connection, accumulated for all pairs at once
//   k_components   :1861-1882/:1985-1998 connected components (lock-free union-find)
// Integer work: sorting via rocPRIM's radix sort (AMD's device primitive library), everything else
// hand-written; pair cells are aggregated in an LDS hash table per workgroup and flushed with global atomics.
#include <cstring>
#include "phz_internal.h"
#include <rocprim/rocprim.hpp>

namespace {

constexpr uint64_t KEY_DROPPED = ~0ull;
constexpr int AS_LDS_BINS = 8192;       // AS in [-4096, 4096) is histogrammed in LDS

struct LinesDev {
    int64_t n;
    const int32_t *read_idx, *var_idx;
    const uint8_t *code;
    const int32_t *read_qid, *read_as;
    const uint8_t *read_has_as;
    double cutoff;
    int use_cutoff, bam;
};

__global__ __launch_bounds__(256) void k_as_hist(LinesDev L, unsigned long long *hist) {
    __shared__ unsigned int s_h[AS_LDS_BINS];
    for (int j = threadIdx.x; j < AS_LDS_BINS; j += 256) s_h[j] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L.n; i += (int64_t)gridDim.x * 256) {
        const int r = L.read_idx[i];
        if (L.read_has_as && !L.read_has_as[r]) continue;
        const int a = L.read_as[r];
        const int b = a + AS_LDS_BINS / 2;
        if ((unsigned)b < (unsigned)AS_LDS_BINS) atomicAdd(&s_h[b], 1u);
        else atomicAdd(&hist[(a + 32768) & 0xFFFF], 1ull);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < AS_LDS_BINS; j += 256)
        if (s_h[j]) atomicAdd(&hist[j - AS_LDS_BINS / 2 + 32768], (unsigned long long)s_h[j]);
}

// Call lines arrive in mapper order (record, then variant) and records are coordinate-sorted, so the lines of one
// workgroup touch a narrow run of variant indices, and deeply covered variants repeat hundreds of times in a row.
// Counters are therefore accumulated in an LDS window [vbase, vbase + TW) with LDS atomics and flushed once per
// workgroup; only lines outside the window (introns reaching far) use global atomics directly.
constexpr int TW = 1024;            // variants per LDS window
constexpr int LINES_PER_BLOCK = 2048;

__global__ __launch_bounds__(256) void k_line(LinesDev L, int64_t line_base, const uint8_t *a0, const uint8_t *a1,
                                              uint8_t *line_cls, int32_t *var_count, unsigned long long *var_first,
                                              int32_t *qid_owner, uint32_t *qid_first, int single_bam) {
    __shared__ int s_cnt[TW * 3];
    __shared__ unsigned long long s_first[TW];
    __shared__ int s_vbase;
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * LINES_PER_BLOCK;
    for (int j = tid; j < TW * 3; j += 256) s_cnt[j] = 0;
    for (int j = tid; j < TW; j += 256) s_first[j] = ~0ull;
    if (tid == 0) s_vbase = L.var_idx[i0];       // lines are (record, variant)-ordered: the first line holds ~the smallest index
    __syncthreads();
    const int vbase = s_vbase - 64 > 0 ? s_vbase - 64 : 0;       // a little room below (mate pairs / overlapping records)
    for (int64_t i = i0 + tid; i < i0 + LINES_PER_BLOCK && i < L.n; i += 256) {
        const int r = L.read_idx[i], v = L.var_idx[i];
        bool keep = true;
        if (L.use_cutoff) {
            if (L.read_has_as && !L.read_has_as[r]) keep = false;
            else keep = (double)L.read_as[r] >= L.cutoff;
        }
        if (!keep) { line_cls[line_base + i] = 255; continue; }
        const uint8_t c = L.code[i];
        // codes 5 / 6 come from the general (indel) mapper, which compared the text with the allele strings itself
        const int cls = c == 5 ? 0 : (c == 6 ? 1 : ((c < 4 && c == a0[v]) ? 0 : ((c < 4 && c == a1[v]) ? 1 : 2)));
        line_cls[line_base + i] = (uint8_t)cls;
        const unsigned d = (unsigned)(v - vbase);
        if (d < (unsigned)TW) {
            atomicAdd(&s_cnt[d * 3 + cls], 1);
            atomicMin(&s_first[d], (unsigned long long)(line_base + i));
        } else {
            atomicAdd(&var_count[(int64_t)v * 3 + cls], 1);
            atomicMin(&var_first[v], (unsigned long long)(line_base + i));
        }
        if (cls < 2) {
            const int q = L.read_qid[r];
            if (!single_bam) atomicMax(&qid_owner[q], L.bam);
            atomicMin(&qid_first[q], (uint32_t)(line_base + i));      // first ref/alt line of the QNAME over all BAMs
        }
    }
    __syncthreads();
    for (int j = tid; j < TW * 3; j += 256) {
        const int c = s_cnt[j];
        if (c) atomicAdd(&var_count[(int64_t)(vbase + j / 3) * 3 + (j % 3)], c);
    }
    for (int j = tid; j < TW; j += 256) {
        const unsigned long long f = s_first[j];
        if (f != ~0ull) atomicMin(&var_first[vbase + j], f);
    }
}

// key = qid:32 | var:28 | cls:2 | spare:1 | linked:1
__global__ __launch_bounds__(256) void k_keys(LinesDev L, int64_t line_base, const uint8_t *line_cls,
                                              const int32_t *qid_owner, uint64_t *keys, uint32_t *qid_vmin, int32_t *qid_vmax,
                                              int single_bam) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= L.n) return;
    const uint8_t cls = line_cls[line_base + i];
    uint64_t k = KEY_DROPPED;
    if (cls != 255) {
        const int r = L.read_idx[i];
        const uint32_t q = (uint32_t)L.read_qid[r];
        const uint32_t linked = (cls < 2 && (single_bam || qid_owner[q] == L.bam)) ? 1u : 0u;
        const uint32_t v = (uint32_t)L.var_idx[i];
        k = ((uint64_t)q << 32) | ((uint64_t)v << 4) | ((uint64_t)cls << 2) | linked;
        if (linked) { atomicMin(&qid_vmin[q], v); atomicMax(&qid_vmax[q], (int32_t)v); }     // span of the surviving read_vars list
    }
    keys[line_base + i] = k;
}

// Overlap-dictionary key order (SURVEY.md 8.1 rule 4, phaser.py:1271-1283): a variant's rank is the smallest
// (first ref/alt line of the QNAME, line) over the surviving read_vars entries of QNAMEs that hold >= 2 distinct
// variants; variants that never get a key keep the maximum value.  LDS window like k_line.
__global__ __launch_bounds__(256) void k_rank(LinesDev L, int64_t line_base, const uint8_t *line_cls, const int32_t *qid_owner,
                                              const uint32_t *qid_first, const uint32_t *qid_vmin, const int32_t *qid_vmax,
                                              unsigned long long *var_rank, int single_bam) {
    __shared__ unsigned long long s_rank[TW];
    __shared__ int s_vbase;
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * LINES_PER_BLOCK;
    for (int j = tid; j < TW; j += 256) s_rank[j] = ~0ull;
    if (tid == 0) s_vbase = L.var_idx[i0];
    __syncthreads();
    const int vbase = s_vbase - 64 > 0 ? s_vbase - 64 : 0;
    for (int64_t i = i0 + tid; i < i0 + LINES_PER_BLOCK && i < L.n; i += 256) {
        const uint8_t cls = line_cls[line_base + i];
        if (cls >= 2) continue;
        const uint32_t q = (uint32_t)L.read_qid[L.read_idx[i]];
        if (!(single_bam || qid_owner[q] == L.bam)) continue;
        if (qid_vmin[q] == (uint32_t)qid_vmax[q]) continue;
        const int v = L.var_idx[i];
        const unsigned long long key = ((unsigned long long)qid_first[q] << 32) | (unsigned long long)(line_base + i);
        const unsigned d = (unsigned)(v - vbase);
        if (d < (unsigned)TW) atomicMin(&s_rank[d], key);
        else atomicMin(&var_rank[v], key);
    }
    __syncthreads();
    for (int j = tid; j < TW; j += 256) {
        const unsigned long long f = s_rank[j];
        if (f != ~0ull) atomicMin(&var_rank[vbase + j], f);
    }
}

// after the sort: an element is the representative of its (qid, var, cls) run when it is the LAST of the
// run (it then carries linked = max over the run, because linked is the lowest key bit)
__global__ __launch_bounds__(256) void k_unique_flags(const uint64_t *keys, int64_t n, uint8_t *flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    bool f = false;
    if (k != KEY_DROPPED) f = (i == n - 1) || ((keys[i + 1] >> 2) != (k >> 2));
    flag[i] = f;
}

// items are sorted by (QNAME id, variant, class); QNAME ids follow first appearance in coordinate-sorted input, so one
// workgroup's items again fall in a narrow run of variants: same LDS window as k_line
__global__ __launch_bounds__(256) void k_distinct(const uint64_t *items, int64_t m, int32_t *var_distinct) {
    __shared__ int s_cnt[TW * 3];
    __shared__ int s_vbase;
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * LINES_PER_BLOCK;
    for (int j = tid; j < TW * 3; j += 256) s_cnt[j] = 0;
    if (tid == 0) s_vbase = (int)((uint32_t)(items[i0] >> 4) & 0x0FFFFFFFu);
    __syncthreads();
    const int vbase = s_vbase - TW / 2 > 0 ? s_vbase - TW / 2 : 0;
    for (int64_t i = i0 + tid; i < i0 + LINES_PER_BLOCK && i < m; i += 256) {
        const uint64_t k = items[i];
        const uint32_t v = (uint32_t)(k >> 4) & 0x0FFFFFFFu, cls = (uint32_t)(k >> 2) & 3u;
        const unsigned d = (unsigned)((int)v - vbase);
        if (d < (unsigned)TW) atomicAdd(&s_cnt[d * 3 + cls], 1);
        else atomicAdd(&var_distinct[(int64_t)v * 3 + cls], 1);
    }
    __syncthreads();
    for (int j = tid; j < TW * 3; j += 256) {
        const int c = s_cnt[j];
        if (c) atomicAdd(&var_distinct[(int64_t)(vbase + j / 3) * 3 + (j % 3)], c);
    }
}

__device__ __forceinline__ uint32_t hash64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (uint32_t)k;
}

// number of (i, j) item pairs inside one QNAME with different variants, counted from the smaller index; one atomic
// per workgroup of LINES_PER_BLOCK items (a per-wave atomic on the single total serialises: 1.9 ms for 10M items)
__global__ __launch_bounds__(256) void k_pair_count(const uint64_t *items, int64_t m, unsigned long long *total) {
    __shared__ unsigned int s_part[4];
    const int64_t i0 = (int64_t)blockIdx.x * LINES_PER_BLOCK;
    unsigned int c = 0;
    for (int64_t i = i0 + threadIdx.x; i < i0 + LINES_PER_BLOCK && i < m; i += 256) {
        const uint64_t k = items[i];
        const uint32_t q = (uint32_t)(k >> 32), v = (uint32_t)(k >> 4) & 0x0FFFFFFFu;
        for (int64_t j = i + 1; j < m; j++) {
            const uint64_t k2 = items[j];
            if ((uint32_t)(k2 >> 32) != q) break;
            if (((uint32_t)(k2 >> 4) & 0x0FFFFFFFu) != v) c++;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = (unsigned long long)s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (t) atomicAdd(total, t);
    }
}

constexpr int PH_SLOTS = 1024;     // LDS hash slots per workgroup
constexpr int PH_VALS = 10;        // 9 cells + linked flag
constexpr int PH_PROBES = 24;

__device__ __forceinline__ uint32_t global_slot(uint64_t *gkeys, uint32_t gmask, uint64_t key) {
    uint32_t s = hash64(key) & gmask;
    for (;;) {
        const unsigned long long prev = atomicCAS((unsigned long long *)&gkeys[s], (unsigned long long)KEY_DROPPED,
                                                  (unsigned long long)key);
        if (prev == KEY_DROPPED || prev == key) return s;
        s = (s + 1) & gmask;
    }
}
__device__ __forceinline__ void global_add(uint64_t *gkeys, int32_t *gvals, uint32_t gmask, uint64_t key, int cell, int val,
                                           int linked) {
    const uint32_t s = global_slot(gkeys, gmask, key);
    if (val) atomicAdd(&gvals[(int64_t)s * PH_VALS + cell], val);
    if (linked) atomicOr(&gvals[(int64_t)s * PH_VALS + 9], 1);
}

__global__ __launch_bounds__(256) void k_pairs(const uint64_t *items, int64_t m, uint64_t *gkeys, int32_t *gvals, uint32_t gmask) {
    __shared__ unsigned long long s_keys[PH_SLOTS];
    __shared__ int s_vals[PH_SLOTS * PH_VALS];
    for (int j = threadIdx.x; j < PH_SLOTS; j += 256) s_keys[j] = KEY_DROPPED;
    for (int j = threadIdx.x; j < PH_SLOTS * PH_VALS; j += 256) s_vals[j] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        const uint64_t k = items[i];
        const uint32_t q = (uint32_t)(k >> 32), v = (uint32_t)(k >> 4) & 0x0FFFFFFFu, cls = (uint32_t)(k >> 2) & 3u, ln = (uint32_t)k & 1u;
        for (int64_t j = i + 1; j < m; j++) {
            const uint64_t k2 = items[j];
            if ((uint32_t)(k2 >> 32) != q) break;
            const uint32_t v2 = (uint32_t)(k2 >> 4) & 0x0FFFFFFFu;
            if (v2 == v) continue;
            const uint32_t cls2 = (uint32_t)(k2 >> 2) & 3u, ln2 = (uint32_t)k2 & 1u;
            const uint64_t pk = ((uint64_t)v << 32) | v2;          // v < v2 because items are sorted
            const int cell = (int)(cls * 3 + cls2);
            const int linked = (int)(ln & ln2);
            uint32_t s = hash64(pk) & (PH_SLOTS - 1);
            bool done = false;
            for (int t = 0; t < PH_PROBES; t++) {
                const unsigned long long prev = atomicCAS(&s_keys[s], (unsigned long long)KEY_DROPPED, (unsigned long long)pk);
                if (prev == KEY_DROPPED || prev == pk) {
                    atomicAdd(&s_vals[s * PH_VALS + cell], 1);
                    if (linked) atomicOr(&s_vals[s * PH_VALS + 9], 1);
                    done = true;
                    break;
                }
                s = (s + 1) & (PH_SLOTS - 1);
            }
            if (!done) global_add(gkeys, gvals, gmask, pk, cell, 1, linked);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < PH_SLOTS; j += 256) {
        const uint64_t pk = s_keys[j];
        if (pk == KEY_DROPPED) continue;
        const uint32_t gs = global_slot(gkeys, gmask, pk);
        for (int c = 0; c < 9; c++) {
            const int val = s_vals[j * PH_VALS + c];
            if (val) atomicAdd(&gvals[(int64_t)gs * PH_VALS + c], val);
        }
        if (s_vals[j * PH_VALS + 9]) atomicOr(&gvals[(int64_t)gs * PH_VALS + 9], 1);
    }
}

__global__ __launch_bounds__(256) void k_edge_flags(const uint64_t *gkeys, int64_t cap, uint8_t *flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < cap) flag[i] = gkeys[i] != KEY_DROPPED;
}

__global__ __launch_bounds__(256) void k_edge_gather(const uint64_t *skeys, const uint32_t *sslot, int64_t ne, const int32_t *gvals,
                                                     int32_t *ea, int32_t *eb, int32_t *cells, uint8_t *linked, int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ne || i >= cap) return;
    const uint64_t k = skeys[i];
    ea[i] = (int32_t)(k >> 32); eb[i] = (int32_t)(uint32_t)k;
    const int64_t s = sslot[i];
#pragma unroll
    for (int c = 0; c < 9; c++) cells[i * 9 + c] = gvals[s * PH_VALS + c];
    linked[i] = (uint8_t)(gvals[s * PH_VALS + 9] & 1);
}

__global__ __launch_bounds__(256) void k_iota_u32(uint32_t *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

// ---- union-find
__device__ __forceinline__ int uf_find(int32_t *parent, int x) {
    int p = parent[x];
    while (p != x) {
        const int g = parent[p];
        if (g != p) parent[x] = g;      // path halving (benign race: only ever points closer to the root)
        x = p; p = g;
    }
    return x;
}
__global__ __launch_bounds__(256) void k_uf_init(int32_t *parent, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) parent[i] = (int32_t)i;
}
__global__ __launch_bounds__(256) void k_uf_hook(int32_t *parent, const int32_t *ea, const int32_t *eb, const uint8_t *keep, int64_t ne) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ne || (keep && !keep[i])) return;
    int a = uf_find(parent, ea[i]), b = uf_find(parent, eb[i]);
    while (a != b) {
        if (a < b) { const int t = a; a = b; b = t; }      // hook the larger root under the smaller
        const int old = atomicCAS(&parent[a], a, b);
        if (old == a) break;
        a = uf_find(parent, old); b = uf_find(parent, b);
    }
}
__global__ __launch_bounds__(256) void k_uf_flatten(int32_t *parent, int32_t *label, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) label[i] = uf_find(parent, (int)i);
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

struct Timer {
    phz_ctx *c; int slot;
    Timer(phz_ctx *ctx, int s) : c(ctx), slot(s) { (void)hipEventRecord(c->ev0, c->stream); }
    void stop() {
        (void)hipEventRecord(c->ev1, c->stream);
        (void)hipEventSynchronize(c->ev1);
        float ms = 0; (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
        c->last_ms[slot] = ms; c->total_ms[slot] += ms; c->launches[slot]++;
    }
};

int stage_lines(Staging &st, const phz_lines &h, int space, LinesDev *d) {
    d->n = h.n_calls; d->cutoff = h.as_cutoff; d->use_cutoff = h.use_cutoff; d->bam = h.bam_index;
    if (int s = st.in(h.read_idx, (size_t)h.n_calls, space, &d->read_idx)) return s;
    if (int s = st.in(h.var_idx, (size_t)h.n_calls, space, &d->var_idx)) return s;
    if (int s = st.in(h.code, (size_t)h.n_calls, space, &d->code)) return s;
    if (int s = st.in(h.read_qid, (size_t)h.n_reads, space, &d->read_qid)) return s;
    if (int s = st.in(h.read_as, (size_t)h.n_reads, space, &d->read_as)) return s;
    if (int s = st.in(h.read_has_as, (size_t)h.n_reads, space, &d->read_has_as)) return s;
    return PHZ_OK;
}

}  // namespace

extern "C" int phz_as_histogram(phz_ctx *ctx, const phz_lines *shard, int64_t *hist, int space) {
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
        unsigned grid = nblk(L.n); if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(k_as_hist, dim3(grid), dim3(256), 0, ctx->stream, L, dh);
    }
    PHZ_HIP(ctx, hipGetLastError());
    t.stop();
    if (space == PHZ_HOST) PHZ_HIP(ctx, hipMemcpyAsync(hist, dh, PHZ_AS_BINS * 8, hipMemcpyDeviceToHost, ctx->stream));
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PHZ_OK;
}

extern "C" int phz_tally(phz_ctx *ctx, const phz_lines *shards, int n_shards, int64_t nv, const uint8_t *a0, const uint8_t *a1,
                         int64_t n_qid, phz_tally_out *out, int64_t *n_edges, int space) {
    if (!ctx || (!shards && n_shards) || !out || !n_edges || nv < 0 || n_qid < 0) return PHZ_E_ARG;
    if (nv >= (1ll << 28)) return phz_fail(ctx, PHZ_E_ARG, "more than 2^28 variants in one chromosome");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    *n_edges = 0;
    Staging st(ctx);
    std::vector<LinesDev> L((size_t)n_shards);
    int64_t total = 0;
    for (int b = 0; b < n_shards; b++) { if (int s = stage_lines(st, shards[b], space, &L[b])) return s; total += L[b].n; }
    const uint8_t *d_a0, *d_a1;
    if (int s = st.in(a0, (size_t)nv, space, &d_a0)) return s;
    if (int s = st.in(a1, (size_t)nv, space, &d_a1)) return s;
    int32_t *d_cnt, *d_dist, *d_ea, *d_eb, *d_cells; int64_t *d_first; uint8_t *d_cls, *d_linked;
    if (int s = st.out(out->var_count, (size_t)nv * 3, space, &d_cnt)) return s;
    if (int s = st.out(out->var_first, (size_t)nv, space, &d_first)) return s;
    if (int s = st.out(out->var_distinct, (size_t)nv * 3, space, &d_dist)) return s;
    if (int s = st.out(out->line_cls, (size_t)total, space, &d_cls)) return s;
    if (int s = st.out(out->edge_a, (size_t)out->edge_cap, space, &d_ea)) return s;
    if (int s = st.out(out->edge_b, (size_t)out->edge_cap, space, &d_eb)) return s;
    if (int s = st.out(out->edge_cells, (size_t)out->edge_cap * 9, space, &d_cells)) return s;
    if (int s = st.out(out->edge_linked, (size_t)out->edge_cap, space, &d_linked)) return s;
    uint64_t *d_rank;
    if (int s = st.out(out->var_rank, (size_t)nv, space, &d_rank)) return s;
    if (total >= (1ll << 32)) return phz_fail(ctx, PHZ_E_ARG, "more than 2^32 call lines in one chromosome");

    hipStream_t sm = ctx->stream;
    Timer timer(ctx, PHZ_T_TALLY);
    PHZ_HIP(ctx, hipMemsetAsync(d_cnt, 0, (size_t)nv * 12, sm));
    PHZ_HIP(ctx, hipMemsetAsync(d_dist, 0, (size_t)nv * 12, sm));
    // scratch: 1 qid_owner, 2 keys, 3 keys sorted, 4 flags, 5 items, 6 rocprim temp, 7 counters, 8 gkeys, 9 gvals, 10.. edge sort
    DevBuf *S = ctx->scratch;
    if (int s = phz_reserve(ctx, S[1], (size_t)(n_qid ? n_qid : 1) * 4)) return s;
    if (int s = phz_reserve(ctx, S[2], (size_t)(total ? total : 1) * 8)) return s;
    if (int s = phz_reserve(ctx, S[3], (size_t)(total ? total : 1) * 8)) return s;
    if (int s = phz_reserve(ctx, S[4], (size_t)(total ? total : 1))) return s;
    if (int s = phz_reserve(ctx, S[5], (size_t)(total ? total : 1) * 8)) return s;
    if (int s = phz_reserve(ctx, S[7], 64)) return s;
    int32_t *qid_owner = (int32_t *)S[1].p;
    const size_t nq = (size_t)(n_qid ? n_qid : 1);
    if (int s = phz_reserve(ctx, S[16], nq * 12)) return s;          // per QNAME: first ref/alt line, min / max linked variant
    uint32_t *qid_first = (uint32_t *)S[16].p, *qid_vmin = qid_first + nq;
    int32_t *qid_vmax = (int32_t *)(qid_vmin + nq);
    PHZ_HIP(ctx, hipMemsetAsync(qid_first, 0xff, nq * 12, sm));      // first = vmin = UINT_MAX, vmax = -1
    PHZ_HIP(ctx, hipMemsetAsync(d_rank, 0xff, (size_t)nv * 8, sm));
    uint64_t *keys = (uint64_t *)S[2].p, *skeys = (uint64_t *)S[3].p, *items = (uint64_t *)S[5].p;
    uint8_t *flags = (uint8_t *)S[4].p;
    unsigned long long *counters = (unsigned long long *)S[7].p;
    PHZ_HIP(ctx, hipMemsetAsync(qid_owner, 0xff, (size_t)(n_qid ? n_qid : 1) * 4, sm));      // -1
    PHZ_HIP(ctx, hipMemsetAsync(counters, 0, 64, sm));
    PHZ_HIP(ctx, hipMemsetAsync(d_first, 0xff, (size_t)nv * 8, sm));       // unsigned max for atomicMin == -1 as int64 ("none")

    const int single_bam = n_shards <= 1 ? 1 : 0;     // one BAM: every QNAME's read_vars list is owned by that BAM
    int64_t base = 0;
    for (int b = 0; b < n_shards; b++) {
        if (L[b].n) hipLaunchKernelGGL(k_line, dim3((unsigned)((L[b].n + LINES_PER_BLOCK - 1) / LINES_PER_BLOCK)), dim3(256), 0, sm, L[b], base,
                                       d_a0, d_a1, d_cls, d_cnt, (unsigned long long *)d_first, qid_owner, qid_first, single_bam);
        base += L[b].n;
    }
    base = 0;
    for (int b = 0; b < n_shards; b++) {
        if (L[b].n) hipLaunchKernelGGL(k_keys, dim3(nblk(L[b].n)), dim3(256), 0, sm, L[b], base, d_cls, qid_owner, keys, qid_vmin, qid_vmax,
                                       single_bam);
        base += L[b].n;
    }
    base = 0;
    for (int b = 0; b < n_shards; b++) {
        if (L[b].n) hipLaunchKernelGGL(k_rank, dim3((unsigned)((L[b].n + LINES_PER_BLOCK - 1) / LINES_PER_BLOCK)), dim3(256), 0, sm, L[b], base,
                                       d_cls, qid_owner, qid_first, qid_vmin, qid_vmax, (unsigned long long *)d_rank, single_bam);
        base += L[b].n;
    }
    PHZ_HIP(ctx, hipGetLastError());
    int64_t m = 0;
    if (total > 0) {
        size_t tmp = 0;
        PHZ_HIP(ctx, rocprim::radix_sort_keys(nullptr, tmp, keys, skeys, (size_t)total, 0, 64, sm));
        if (int s = phz_reserve(ctx, S[6], tmp)) return s;
        PHZ_HIP(ctx, rocprim::radix_sort_keys(S[6].p, tmp, keys, skeys, (size_t)total, 0, 64, sm));
        hipLaunchKernelGGL(k_unique_flags, dim3(nblk(total)), dim3(256), 0, sm, skeys, total, flags);
        size_t tmp2 = 0;
        unsigned long long *d_m = counters + 1;
        PHZ_HIP(ctx, rocprim::select(nullptr, tmp2, skeys, flags, items, d_m, (size_t)total, sm));
        if (int s = phz_reserve(ctx, S[6], tmp2 > tmp ? tmp2 : tmp)) return s;
        PHZ_HIP(ctx, rocprim::select(S[6].p, tmp2, skeys, flags, items, d_m, (size_t)total, sm));
        unsigned long long hm = 0;
        PHZ_HIP(ctx, hipMemcpyAsync(&hm, d_m, 8, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipStreamSynchronize(sm));
        m = (int64_t)hm;
    }
    int64_t ne = 0;
    unsigned long long pair_events = 0;
    if (m > 0) {
        hipLaunchKernelGGL(k_distinct, dim3((unsigned)((m + LINES_PER_BLOCK - 1) / LINES_PER_BLOCK)), dim3(256), 0, sm, items, m, d_dist);
        hipLaunchKernelGGL(k_pair_count, dim3((unsigned)((m + LINES_PER_BLOCK - 1) / LINES_PER_BLOCK)), dim3(256), 0, sm, items, m, counters + 2);
        unsigned long long events = 0;
        PHZ_HIP(ctx, hipMemcpyAsync(&events, counters + 2, 8, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipStreamSynchronize(sm));
        pair_events = events;
        if (events > 0) {
            uint64_t cap = 1024;
            while (cap < 2 * events && cap < (1ull << 31)) cap <<= 1;
            if (int s = phz_reserve(ctx, S[8], cap * 8)) return s;
            if (int s = phz_reserve(ctx, S[9], cap * PH_VALS * 4)) return s;
            uint64_t *gkeys = (uint64_t *)S[8].p; int32_t *gvals = (int32_t *)S[9].p;
            PHZ_HIP(ctx, hipMemsetAsync(gkeys, 0xff, cap * 8, sm));
            PHZ_HIP(ctx, hipMemsetAsync(gvals, 0, cap * PH_VALS * 4, sm));
            hipLaunchKernelGGL(k_pairs, dim3(nblk(m)), dim3(256), 0, sm, items, m, gkeys, gvals, (uint32_t)(cap - 1));
            // compact used slots, sort by pair key for a deterministic edge order
            if (int s = phz_reserve(ctx, S[10], cap)) return s;              // flags
            if (int s = phz_reserve(ctx, S[11], cap * 4)) return s;          // iota
            if (int s = phz_reserve(ctx, S[12], cap * 4)) return s;          // slots (compacted)
            if (int s = phz_reserve(ctx, S[13], cap * 8)) return s;          // keys (compacted)
            if (int s = phz_reserve(ctx, S[14], cap * 8)) return s;          // keys sorted
            if (int s = phz_reserve(ctx, S[15], cap * 4)) return s;          // slots sorted
            uint8_t *ef = (uint8_t *)S[10].p; uint32_t *iota = (uint32_t *)S[11].p, *cslot = (uint32_t *)S[12].p, *sslot = (uint32_t *)S[15].p;
            uint64_t *ckeys = (uint64_t *)S[13].p, *sk = (uint64_t *)S[14].p;
            hipLaunchKernelGGL(k_edge_flags, dim3(nblk((int64_t)cap)), dim3(256), 0, sm, gkeys, (int64_t)cap, ef);
            hipLaunchKernelGGL(k_iota_u32, dim3(nblk((int64_t)cap)), dim3(256), 0, sm, iota, (int64_t)cap);
            size_t t1 = 0, t2 = 0, t3 = 0;
            unsigned long long *d_ne = counters + 3;
            PHZ_HIP(ctx, rocprim::select(nullptr, t1, gkeys, ef, ckeys, d_ne, (size_t)cap, sm));
            PHZ_HIP(ctx, rocprim::select(nullptr, t2, iota, ef, cslot, d_ne, (size_t)cap, sm));
            size_t tm = t1 > t2 ? t1 : t2;
            if (int s = phz_reserve(ctx, S[6], tm)) return s;
            PHZ_HIP(ctx, rocprim::select(S[6].p, t1, gkeys, ef, ckeys, d_ne, (size_t)cap, sm));
            PHZ_HIP(ctx, rocprim::select(S[6].p, t2, iota, ef, cslot, d_ne, (size_t)cap, sm));
            unsigned long long hne = 0;
            PHZ_HIP(ctx, hipMemcpyAsync(&hne, d_ne, 8, hipMemcpyDeviceToHost, sm));
            PHZ_HIP(ctx, hipStreamSynchronize(sm));
            ne = (int64_t)hne;
            if (ne > 0) {
                PHZ_HIP(ctx, rocprim::radix_sort_pairs(nullptr, t3, ckeys, sk, cslot, sslot, (size_t)ne, 0, 64, sm));
                if (int s = phz_reserve(ctx, S[6], t3)) return s;
                PHZ_HIP(ctx, rocprim::radix_sort_pairs(S[6].p, t3, ckeys, sk, cslot, sslot, (size_t)ne, 0, 64, sm));
                hipLaunchKernelGGL(k_edge_gather, dim3(nblk(ne)), dim3(256), 0, sm, sk, sslot, ne, gvals, d_ea, d_eb, d_cells, d_linked,
                                   out->edge_cap);
            }
        }
    }
    PHZ_HIP(ctx, hipGetLastError());
    timer.stop();
    ctx->counters[PHZ_C_LINES] += total; ctx->counters[PHZ_C_ITEMS] += m; ctx->counters[PHZ_C_PAIR_EVENTS] += (int64_t)pair_events;
    ctx->counters[PHZ_C_EDGES] += ne;
    *n_edges = ne;
    if (space == PHZ_HOST) {
        PHZ_HIP(ctx, hipMemcpyAsync(out->var_count, d_cnt, (size_t)nv * 12, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(out->var_first, d_first, (size_t)nv * 8, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(out->var_distinct, d_dist, (size_t)nv * 12, hipMemcpyDeviceToHost, sm));
        PHZ_HIP(ctx, hipMemcpyAsync(out->var_rank, d_rank, (size_t)nv * 8, hipMemcpyDeviceToHost, sm));
        if (total) PHZ_HIP(ctx, hipMemcpyAsync(out->line_cls, d_cls, (size_t)total, hipMemcpyDeviceToHost, sm));
        const size_t k = (size_t)(ne < out->edge_cap ? ne : out->edge_cap);
        if (k) {
            PHZ_HIP(ctx, hipMemcpyAsync(out->edge_a, d_ea, k * 4, hipMemcpyDeviceToHost, sm));
            PHZ_HIP(ctx, hipMemcpyAsync(out->edge_b, d_eb, k * 4, hipMemcpyDeviceToHost, sm));
            PHZ_HIP(ctx, hipMemcpyAsync(out->edge_cells, d_cells, k * 36, hipMemcpyDeviceToHost, sm));
            PHZ_HIP(ctx, hipMemcpyAsync(out->edge_linked, d_linked, k, hipMemcpyDeviceToHost, sm));
        }
    }
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    return ne > out->edge_cap ? PHZ_E_CAPACITY : PHZ_OK;
}

extern "C" int phz_components(phz_ctx *ctx, int64_t nv, int64_t n_edges, const int32_t *edge_a, const int32_t *edge_b,
                              const uint8_t *keep, int32_t *label, int space) {
    if (!ctx || nv < 0 || n_edges < 0 || (!label && nv)) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    Staging st(ctx);
    const int32_t *ea, *eb; const uint8_t *kp; int32_t *lab;
    if (int s = st.in(edge_a, (size_t)n_edges, space, &ea)) return s;
    if (int s = st.in(edge_b, (size_t)n_edges, space, &eb)) return s;
    if (int s = st.in(keep, (size_t)n_edges, space, &kp)) return s;
    if (int s = st.out(label, (size_t)nv, space, &lab)) return s;
    if (int s = phz_reserve(ctx, ctx->scratch[16], (size_t)(nv ? nv : 1) * 4)) return s;
    int32_t *parent = (int32_t *)ctx->scratch[16].p;
    Timer t(ctx, PHZ_T_COMPONENTS);
    if (nv) hipLaunchKernelGGL(k_uf_init, dim3(nblk(nv)), dim3(256), 0, ctx->stream, parent, nv);
    if (n_edges) hipLaunchKernelGGL(k_uf_hook, dim3(nblk(n_edges)), dim3(256), 0, ctx->stream, parent, ea, eb, kp, n_edges);
    if (nv) hipLaunchKernelGGL(k_uf_flatten, dim3(nblk(nv)), dim3(256), 0, ctx->stream, parent, lab, nv);
    PHZ_HIP(ctx, hipGetLastError());
    t.stop();
    if (space == PHZ_HOST && nv) PHZ_HIP(ctx, hipMemcpyAsync(label, lab, (size_t)nv * 4, hipMemcpyDeviceToHost, ctx->stream));
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PHZ_OK;
}
