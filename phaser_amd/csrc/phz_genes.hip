// K_genes: the data-parallel core of phaser_gene_ae -- for every (haplotypic_counts row,
// overlapping feature) pair, the number of distinct reads on each haplotype among the row's variants that fall inside the
// feature (variant_feature_reads, phaser_gene_ae/phaser_gene_ae.py:172-219: union of the per-variant read-label lists as a set).
//
// Layout: a row's labels of one haplotype are one contiguous run; lab_pos[p] = position of the variant label p belongs to,
// lab_prev[p] = run-relative index of the previous occurrence of the same label (-1 = none), both written by phz_hc_parse.
// A label occurrence counts iff its variant is inside the feature and no EARLIER occurrence of the same label is: the
// kernel walks the prev chain (a read covers a handful of variants, so chains are short).  Works for any subset of the
// row's variants, in any order.  HBM-bound: 8 B per label visited (+ the chain), no reuse -> plain coalesced streaming,
// one work item = (pair, haplotype, <= ITEM_LABELS labels) per wave so that a block with millions of labels spreads
// over the chip; per-item partial counts are added to the pair's counter.
#include <hip/hip_runtime.h>

#include "phz.h"
#include "phz_internal.h"

namespace {

__global__ __launch_bounds__(256) void k_gene_items(int64_t n_items, const int64_t *item_lo, const int32_t *item_n, const int64_t *item_run,
                                                     const int32_t *item_pair, const uint8_t *item_hap, const int32_t *pair_begin,
                                                     const int32_t *pair_end, const int32_t *pos_a, const int32_t *prev_a,
                                                     const int32_t *pos_b, const int32_t *prev_b, int32_t *counts) {
    // one wave per work item (most rows carry a few dozen labels; the big blocks arrive pre-split into ITEM-sized slices)
    const int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= n_items) return;
    const int lane = threadIdx.x & 63;
    const int hap = item_hap[it];
    const int32_t *lab_pos = hap ? pos_b : pos_a;
    const int32_t *lab_prev = hap ? prev_b : prev_a;
    const int64_t lo = item_lo[it], run = item_run[it];
    const int n = item_n[it];
    const int pair = item_pair[it];
    const int fb = pair_begin[pair], fe = pair_end[pair];
    int cnt = 0;
    for (int i = lane; i < n; i += 64) {
        const int64_t p = lo + i;
        const int x = lab_pos[p] - 1;
        if (x < fb || x > fe) continue;                       // end inclusive, as in the reference (:190)
        bool first = true;
        for (int q = lab_prev[p]; q >= 0; q = lab_prev[run + q]) {
            const int y = lab_pos[run + q] - 1;
            if (y >= fb && y <= fe) { first = false; break; }
        }
        cnt += first ? 1 : 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane == 0 && cnt) atomicAdd(&counts[2 * (int64_t)pair + hap], cnt);
}

}  // namespace

extern "C" int phz_gene_counts(phz_ctx *ctx, const phz_gene_work *w, int32_t *pair_counts, int space) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !w || (!pair_counts && w->n_pairs) || w->n_items < 0 || w->n_pairs < 0) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    Staging st(ctx);
    const int64_t *d_lo, *d_run; const int32_t *d_n, *d_pair, *d_pb, *d_pe, *d_posa, *d_preva, *d_posb, *d_prevb; const uint8_t *d_hap;
    if (int s = st.in(w->item_lo, (size_t)w->n_items, space, &d_lo)) return s;
    if (int s = st.in(w->item_n, (size_t)w->n_items, space, &d_n)) return s;
    if (int s = st.in(w->item_run, (size_t)w->n_items, space, &d_run)) return s;
    if (int s = st.in(w->item_pair, (size_t)w->n_items, space, &d_pair)) return s;
    if (int s = st.in(w->item_hap, (size_t)w->n_items, space, &d_hap)) return s;
    if (int s = st.in(w->pair_begin, (size_t)w->n_pairs, space, &d_pb)) return s;
    if (int s = st.in(w->pair_end, (size_t)w->n_pairs, space, &d_pe)) return s;
    if (int s = st.in(w->lab_pos_a, (size_t)w->n_lab_a, space, &d_posa)) return s;
    if (int s = st.in(w->lab_prev_a, (size_t)w->n_lab_a, space, &d_preva)) return s;
    if (int s = st.in(w->lab_pos_b, (size_t)w->n_lab_b, space, &d_posb)) return s;
    if (int s = st.in(w->lab_prev_b, (size_t)w->n_lab_b, space, &d_prevb)) return s;
    int32_t *d_counts;
    if (int s = st.out(pair_counts, (size_t)w->n_pairs * 2, space, &d_counts)) return s;
    hipStream_t sm = ctx->stream;
    (void)hipEventRecord(ctx->ev0, sm);
    PHZ_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)(w->n_pairs ? w->n_pairs : 1) * 8, sm));
    if (w->n_items > 0) {
        if (w->n_items >= (1ll << 31)) return phz_fail(ctx, PHZ_E_ARG, "too many gene work items");
        hipLaunchKernelGGL(k_gene_items, dim3((unsigned)((w->n_items + 3) / 4)), dim3(256), 0, sm, w->n_items, d_lo, d_n, d_run, d_pair, d_hap, d_pb, d_pe,
                           d_posa, d_preva, d_posb, d_prevb, d_counts);
        PHZ_HIP(ctx, hipGetLastError());
    }
    (void)hipEventRecord(ctx->ev1, sm);
    (void)hipEventSynchronize(ctx->ev1);
    float ms = 0; (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_ms[PHZ_T_GENES] = ms; ctx->total_ms[PHZ_T_GENES] += ms; ctx->launches[PHZ_T_GENES]++;
    if (space == PHZ_HOST && w->n_pairs) PHZ_HIP(ctx, hipMemcpyAsync(pair_counts, d_counts, (size_t)w->n_pairs * 8, hipMemcpyDeviceToHost, sm));
    PHZ_HIP(ctx, hipStreamSynchronize(sm));
    return PHZ_OK;
}
