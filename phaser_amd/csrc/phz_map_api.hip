// Host side of the mapper entry points of the C ABI (include/phz.h): argument checks and staging for PHZ_HOST callers around the
// launchers of phz_map.hip.
#include "phz_internal.h"

static int upload(phz_ctx *ctx, DevBuf &b, const void *src, size_t bytes) {
    if (int s = phz_reserve(ctx, b, bytes ? bytes : 1)) return s;
    if (bytes) PHZ_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PHZ_OK;
}

static int check_variants(phz_ctx *ctx, const phz_variants *v, int space) {
    // SNP-only fast path: reject indel mode up front (host-visible arrays only)
    if (space == PHZ_HOST && v->ref_len)
        for (int64_t i = 0; i < v->n; i++)
            if (v->ref_len[i] != 1) return phz_fail(ctx, PHZ_E_UNSUPPORTED, "variants with ref_len != 1 (indel mode) are not supported by K_map yet");
    return PHZ_OK;
}

extern "C" int phz_map_reads_batch(phz_ctx *ctx, int n_shards, const phz_reads *reads, const phz_variants *vars, int baseq,
                                   const phz_calls *out, int64_t *n_calls) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || n_shards < 0 || (n_shards && (!reads || !vars || !out || !n_calls))) return PHZ_E_ARG;
    for (int i = 0; i < n_shards; i++)
        if (reads[i].n_reads < 0 || vars[i].n < 0 || out[i].cap < 0) return phz_fail(ctx, PHZ_E_ARG, "negative size");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    return phz_launch_map_batch(ctx, n_shards, reads, vars, baseq, out, n_calls);
}

extern "C" int phz_map_reads(phz_ctx *ctx, const phz_reads *reads, const phz_variants *vars, int baseq,
                             phz_calls *out, int64_t *n_calls, int space) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !reads || !vars || !out || !n_calls) return PHZ_E_ARG;
    if (reads->n_reads < 0 || vars->n < 0 || out->cap < 0) return phz_fail(ctx, PHZ_E_ARG, "negative size");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    if (int s = check_variants(ctx, vars, space)) return s;
    if (space == PHZ_DEVICE) return phz_launch_map(ctx, *reads, *vars, baseq, *out, n_calls);
    if (space != PHZ_HOST) return phz_fail(ctx, PHZ_E_ARG, "bad memory space");

    const int64_t n = reads->n_reads;
    phz_reads dr = *reads;
    phz_variants dv = *vars;
    if (int s = upload(ctx, ctx->r_pos, reads->pos, (size_t)n * 4)) return s;
    if (int s = upload(ctx, ctx->r_coff, reads->cigar_off, (size_t)(n + 1) * 4)) return s;
    if (int s = upload(ctx, ctx->r_cig, reads->cigar, (size_t)reads->n_ops * 4)) return s;
    if (int s = upload(ctx, ctx->r_soff, reads->seq_off, (size_t)(n + 1) * 4)) return s;
    if (int s = upload(ctx, ctx->r_seq, reads->seq2, (size_t)reads->n_seq_bytes)) return s;
    if (int s = upload(ctx, ctx->r_qual, reads->qual, (size_t)reads->n_seq_bytes * 4)) return s;
    if (int s = upload(ctx, ctx->v_pos, vars->pos, (size_t)vars->n * 4)) return s;
    dr.pos = (const int32_t *)ctx->r_pos.p; dr.cigar_off = (const uint32_t *)ctx->r_coff.p;
    dr.cigar = (const uint32_t *)ctx->r_cig.p; dr.seq_off = (const uint32_t *)ctx->r_soff.p;
    dr.seq2 = (const uint8_t *)ctx->r_seq.p; dr.qual = (const uint8_t *)ctx->r_qual.p; dr.bq = nullptr;
    dv.pos = (const int32_t *)ctx->v_pos.p; dv.ref_len = nullptr;
    phz_calls dc;
    dc.cap = out->cap;
    const size_t cap = (size_t)(out->cap ? out->cap : 1);
    if (int s = phz_reserve(ctx, ctx->c_read, cap * 4)) return s;
    if (int s = phz_reserve(ctx, ctx->c_var, cap * 4)) return s;
    if (int s = phz_reserve(ctx, ctx->c_code, cap)) return s;
    if (int s = phz_reserve(ctx, ctx->c_aux0, cap * 4)) return s;
    if (int s = phz_reserve(ctx, ctx->c_aux1, cap * 4)) return s;
    dc.read_idx = (int32_t *)ctx->c_read.p; dc.var_idx = (int32_t *)ctx->c_var.p; dc.code = (uint8_t *)ctx->c_code.p;
    dc.aux0 = (uint32_t *)ctx->c_aux0.p; dc.aux1 = (uint32_t *)ctx->c_aux1.p;
    if (!out->aux0 || !out->aux1) { dc.aux0 = nullptr; dc.aux1 = nullptr; }
    int st = phz_launch_map(ctx, dr, dv, baseq, dc, n_calls);
    if (st != PHZ_OK && st != PHZ_E_CAPACITY) return st;
    const size_t m = (size_t)(*n_calls < out->cap ? *n_calls : out->cap);
    if (m) {
        PHZ_HIP(ctx, hipMemcpyAsync(out->read_idx, dc.read_idx, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        PHZ_HIP(ctx, hipMemcpyAsync(out->var_idx, dc.var_idx, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        PHZ_HIP(ctx, hipMemcpyAsync(out->code, dc.code, m, hipMemcpyDeviceToHost, ctx->stream));
        if (dc.aux0) {
            PHZ_HIP(ctx, hipMemcpyAsync(out->aux0, dc.aux0, m * 4, hipMemcpyDeviceToHost, ctx->stream));
            PHZ_HIP(ctx, hipMemcpyAsync(out->aux1, dc.aux1, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return st;
}
