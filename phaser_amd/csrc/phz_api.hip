// Host side of the C ABI (include/phz.h): context, staging for PHZ_HOST callers, timing.
#include "phz_internal.h"

#include <new>

extern "C" {

int phz_version(void) { return 100; }

const char *phz_strerror(int s) {
    switch (s) {
        case PHZ_OK: return "ok";
        case PHZ_E_ARG: return "invalid argument";
        case PHZ_E_HIP: return "HIP runtime error";
        case PHZ_E_CAPACITY: return "output capacity too small";
        case PHZ_E_UNSUPPORTED: return "unsupported input";
        case PHZ_E_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

const char *phz_last_error(const phz_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int phz_device_count(int *n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *n = e == hipSuccess ? c : 0;
    return e == hipSuccess ? PHZ_OK : PHZ_E_HIP;
}

int phz_ctx_create(int device, phz_ctx **out) {
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return PHZ_E_HIP;
    phz_ctx *c = new (std::nothrow) phz_ctx();
    if (!c) return PHZ_E_NOMEM;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return PHZ_E_HIP;
    }
    *out = c;
    return PHZ_OK;
}

static void free_buf(DevBuf &b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

int phz_ctx_destroy(phz_ctx *c) {
    if (!c) return PHZ_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf *all[] = {&c->desc, &c->tile_w0, &c->scalars, &c->r_pos, &c->r_coff, &c->r_cig, &c->r_soff, &c->r_seq,
                     &c->r_qual, &c->v_pos, &c->v_reflen, &c->c_read, &c->c_var, &c->c_code, &c->c_aux0, &c->c_aux1};
    for (DevBuf *b : all) free_buf(*b);
    for (DevBuf &b : c->scratch) free_buf(b);
    for (DevBuf &b : c->stage_pool) free_buf(b);
    for (DevBuf &b : c->tally_buf) free_buf(b);
    free_buf(c->tally_qcount);
    if (c->h_scalars.p) (void)hipHostFree(c->h_scalars.p);
    if (c->h_shard_tab.p) (void)hipHostFree(c->h_shard_tab.p);
    free_buf(c->shard_tab);
    for (hipEvent_t e : c->map_ev) if (e) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return PHZ_OK;
}

int phz_ctx_sync(phz_ctx *ctx) {
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PHZ_OK;
}

void *phz_ctx_stream(phz_ctx *ctx) { return (void *)ctx->stream; }

int phz_get_timing(phz_ctx *ctx, int slot, float *last_ms, double *total_ms, int64_t *launches) {
    if (slot < 0 || slot >= PHZ_T_COUNT) return PHZ_E_ARG;
    if (last_ms) *last_ms = ctx->last_ms[slot];
    if (total_ms) *total_ms = ctx->total_ms[slot];
    if (launches) *launches = ctx->launches[slot];
    return PHZ_OK;
}

int phz_reset_timing(phz_ctx *ctx) {
    for (int i = 0; i < PHZ_T_COUNT; i++) { ctx->last_ms[i] = 0; ctx->total_ms[i] = 0; ctx->launches[i] = 0; }
    for (int i = 0; i < PHZ_C_COUNT; i++) ctx->counters[i] = 0;
    return PHZ_OK;
}

int phz_get_counter(phz_ctx *ctx, int slot, int64_t *value) {
    if (!ctx || !value || slot < 0 || slot >= PHZ_C_COUNT) return PHZ_E_ARG;
    *value = ctx->counters[slot];
    return PHZ_OK;
}

}  // extern "C"

int phz_fail(phz_ctx *ctx, int status, const char *what, hipError_t e) {
    if (ctx) {
        ctx->err = what ? what : "";
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return status;
}

int phz_reserve(phz_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return PHZ_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return phz_fail(ctx, PHZ_E_NOMEM, "hipMalloc", e);
    b.cap = want;
    return PHZ_OK;
}

int phz_reserve_host(phz_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return PHZ_OK;
    if (b.p) (void)hipHostFree(b.p);
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
    if (e != hipSuccess) return phz_fail(ctx, PHZ_E_NOMEM, "hipHostMalloc", e);
    b.cap = want;
    return PHZ_OK;
}

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
    if (!ctx || n_shards < 0 || (n_shards && (!reads || !vars || !out || !n_calls))) return PHZ_E_ARG;
    for (int i = 0; i < n_shards; i++)
        if (reads[i].n_reads < 0 || vars[i].n < 0 || out[i].cap < 0) return phz_fail(ctx, PHZ_E_ARG, "negative size");
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    return phz_launch_map_batch(ctx, n_shards, reads, vars, baseq, out, n_calls);
}

extern "C" int phz_map_reads(phz_ctx *ctx, const phz_reads *reads, const phz_variants *vars, int baseq,
                             phz_calls *out, int64_t *n_calls, int space) {
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
    dr.seq2 = (const uint8_t *)ctx->r_seq.p; dr.qual = (const uint8_t *)ctx->r_qual.p;
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
    int st = phz_launch_map(ctx, dr, dv, baseq, dc, n_calls);
    if (st != PHZ_OK && st != PHZ_E_CAPACITY) return st;
    const size_t m = (size_t)(*n_calls < out->cap ? *n_calls : out->cap);
    if (m) {
        PHZ_HIP(ctx, hipMemcpyAsync(out->read_idx, dc.read_idx, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        PHZ_HIP(ctx, hipMemcpyAsync(out->var_idx, dc.var_idx, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        PHZ_HIP(ctx, hipMemcpyAsync(out->code, dc.code, m, hipMemcpyDeviceToHost, ctx->stream));
        PHZ_HIP(ctx, hipMemcpyAsync(out->aux0, dc.aux0, m * 4, hipMemcpyDeviceToHost, ctx->stream));
        PHZ_HIP(ctx, hipMemcpyAsync(out->aux1, dc.aux1, m * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return st;
}
