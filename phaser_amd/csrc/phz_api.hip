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
    for (DevBuf &b : c->import_buf) free_buf(b);
    for (DevBuf &b : c->resident_vars) free_buf(b);
    free_buf(c->tally_qcount); free_buf(c->scan_state);
    if (c->h_scalars.p) (void)hipHostFree(c->h_scalars.p);
    if (c->mail_host.p) (void)hipHostFree(c->mail_host.p);
    free_buf(c->mail_dev);
    if (c->h_shard_tab.p) (void)hipHostFree(c->h_shard_tab.p);
    if (c->h_bam_stage.p) (void)hipHostFree(c->h_bam_stage.p);
    free_buf(c->shard_tab); free_buf(c->map_tab); free_buf(c->bam_comp); free_buf(c->bam_stream); free_buf(c->bam_work);
    for (hipEvent_t e : c->map_ev) if (e) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    if (c->tab_ev) (void)hipEventDestroy(c->tab_ev);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return PHZ_OK;
}

// SURVEY.md 8(b) `phz_load_variants`: one chromosome's het-variant table (generate_mapping_table, phaser/phaser.py:1355-1413: the POS and len(REF) columns the
// mapper reads, read_variant_map.py:25-44) made RESIDENT in the ctx under `slot`; *resident receives device pointers that stay valid until the slot is
// loaded again or the ctx is destroyed, for phz_map_reads(..., PHZ_DEVICE) / phz_map_reads_batch over every BAM's shard of that chromosome.
int phz_load_variants(phz_ctx *ctx, int slot, const int32_t *pos, const uint8_t *ref_len, int64_t n, int space, phz_variants *resident) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !resident || slot < 0 || slot >= 65536 || n < 0 || (n && (!pos || !ref_len)) || (space != PHZ_HOST && space != PHZ_DEVICE)) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    for (int64_t i = 1; space == PHZ_HOST && i < n; i++)
        if (pos[i] < pos[i - 1]) return phz_fail(ctx, PHZ_E_ARG, "phz_load_variants: positions are not sorted");
    if (ctx->resident_vars.size() < (size_t)(2 * slot + 2)) ctx->resident_vars.resize((size_t)(2 * slot + 2));
    DevBuf &P = ctx->resident_vars[(size_t)(2 * slot)], &L = ctx->resident_vars[(size_t)(2 * slot + 1)];
    if (int s = phz_reserve(ctx, P, (size_t)(n ? n : 1) * 4)) return s;
    if (int s = phz_reserve(ctx, L, (size_t)(n ? n : 1))) return s;
    const hipMemcpyKind kind = space == PHZ_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (n) {
        PHZ_HIP(ctx, hipMemcpyAsync(P.p, pos, (size_t)n * 4, kind, ctx->stream));
        PHZ_HIP(ctx, hipMemcpyAsync(L.p, ref_len, (size_t)n, kind, ctx->stream));
        PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the caller's arrays may go away
    }
    resident->n = n; resident->pos = (const int32_t *)P.p; resident->ref_len = (const uint8_t *)L.p;
    return PHZ_OK;
}

int phz_ctx_sync(phz_ctx *ctx) {
    PhzEnter phz_guard_(ctx);
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PHZ_OK;
}

void *phz_ctx_stream(phz_ctx *ctx) { return (void *)ctx->stream; }

int phz_get_timing(phz_ctx *ctx, int slot, float *last_ms, double *total_ms, int64_t *launches) {
    PhzEnter phz_guard_(ctx);
    if (slot < 0 || slot >= PHZ_T_COUNT) return PHZ_E_ARG;
    if (last_ms) *last_ms = ctx->last_ms[slot];
    if (total_ms) *total_ms = ctx->total_ms[slot];
    if (launches) *launches = ctx->launches[slot];
    return PHZ_OK;
}

int phz_reset_timing(phz_ctx *ctx) {
    PhzEnter phz_guard_(ctx);
    for (int i = 0; i < PHZ_T_COUNT; i++) { ctx->last_ms[i] = 0; ctx->total_ms[i] = 0; ctx->launches[i] = 0; }
    for (int i = 0; i < PHZ_C_COUNT; i++) ctx->counters[i] = 0;
    return PHZ_OK;
}

int phz_get_counter(phz_ctx *ctx, int slot, int64_t *value) {
    PhzEnter phz_guard_(ctx);
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

// ---- PhzMail (phz_internal.h): gather kernel + one copy to page-locked memory
namespace {
struct MailArgs { const void *src[PhzMail::MAX]; uint32_t bytes[PhzMail::MAX], off[PhzMail::MAX]; };
__global__ __launch_bounds__(64) void k_mail(MailArgs a, char *dst) {
    const int e = blockIdx.x;
    const char *s = (const char *)a.src[e];
    for (uint32_t i = threadIdx.x; i < a.bytes[e]; i += 64) dst[a.off[e] + i] = s[i];
}
}  // namespace
int PhzMail::send() {
    if (!n) return PHZ_OK;
    if (int s = phz_reserve(ctx, ctx->mail_dev, total + 64)) return s;
    if (int s = phz_reserve_host(ctx, ctx->mail_host, total + 64)) return s;
    MailArgs a;
    for (int i = 0; i < MAX; i++) { a.src[i] = i < n ? src[i] : nullptr; a.bytes[i] = i < n ? bytes[i] : 0u; a.off[i] = i < n ? off[i] : 0u; }
    // The gather kernel stores straight into the page-locked host block (hipHostMalloc memory is mapped into the device's address space): a few dozen bytes over the
    // link, visible to the host when the kernel has completed.  The copy that used to follow the kernel -- device block -> host block on a copy engine -- cost every
    // host wait of a pass ~10 us of queue hand-over (PHZ_MAIL_COPY=1 keeps it, for the A/B).
    static const bool via_copy = getenv("PHZ_MAIL_COPY") != nullptr;
    hipLaunchKernelGGL(k_mail, dim3((unsigned)n), dim3(64), 0, ctx->stream, a, via_copy ? (char *)ctx->mail_dev.p : (char *)ctx->mail_host.p);
    PHZ_HIP(ctx, hipGetLastError());
    if (via_copy) PHZ_HIP(ctx, hipMemcpyAsync(ctx->mail_host.p, ctx->mail_dev.p, total, hipMemcpyDeviceToHost, ctx->stream));
    return PHZ_OK;
}

// Adopt results computed elsewhere as the resident tally of this ctx (see phz.h).  Arrays a caller leaves NULL stay unset; the device
// row stage needs var_count, var_first, var_distinct, var_rank, edge_a / edge_b, edge_linked, edge_stats, rl_start, rl_qid and rl_list.
extern "C" int phz_tally_import(phz_ctx *ctx, int64_t nv, int n_bams, const phz_tally_sizes *sz, const phz_tally_out *a, const uint32_t *rl_list, int space) {
    PhzEnter phz_guard_(ctx);
    if (!ctx || !sz || !a || nv < 0 || n_bams < 1) return PHZ_E_ARG;
    PHZ_HIP(ctx, hipSetDevice(ctx->device));
    ctx->tally_gen++;
    auto &T = ctx->tally;
    const size_t NV = (size_t)nv, NE = (size_t)sz->n_edges, NRL = NV * 2 * (size_t)n_bams, NR = (size_t)sz->n_read_list;
    if (ctx->import_buf.size() < 16) ctx->import_buf.resize(16);
    int slot = 0;
    auto take = [&](const void *src, size_t bytes, void **dst) -> int {
        *dst = nullptr;
        DevBuf &b = ctx->import_buf[(size_t)slot++];
        if (!src) return PHZ_OK;
        if (space == PHZ_DEVICE) { *dst = (void *)src; return PHZ_OK; }
        if (int s = phz_reserve(ctx, b, bytes ? bytes : 1)) return s;
        if (bytes) PHZ_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        *dst = b.p;
        return PHZ_OK;
    };
    void *p = nullptr;
    if (int s = take(a->var_count, NV * 12, &p)) return s; T.var_count = (int32_t *)p;
    if (int s = take(a->var_first, NV * 8, &p)) return s; T.var_first = (int64_t *)p;
    if (int s = take(a->var_distinct, NV * 12, &p)) return s; T.var_distinct = (int32_t *)p;
    if (int s = take(a->var_rank, NV * 8, &p)) return s; T.var_rank = (uint64_t *)p;
    if (int s = take(a->line_cls, (size_t)sz->n_lines, &p)) return s; T.line_cls = (uint8_t *)p;
    if (int s = take(a->edge_a, NE * 4, &p)) return s; T.ea = (int32_t *)p;
    if (int s = take(a->edge_b, NE * 4, &p)) return s; T.eb = (int32_t *)p;
    if (int s = take(a->edge_cells, NE * 36, &p)) return s; T.cells = (int32_t *)p;
    if (int s = take(a->edge_linked, NE, &p)) return s; T.linked = (uint8_t *)p;
    if (int s = take(a->edge_cto, NE * 12, &p)) return s; T.cto = (int32_t *)p;
    if (int s = take(a->rl_start, (NRL + 1) * 4, &p)) return s; T.rl_start = (uint32_t *)p;
    if (int s = take(a->rl_qid, NR * 4, &p)) return s; T.rl_qid = (int32_t *)p;
    if (int s = take(a->edge_stats, NE * 20, &p)) return s; T.stats = (int32_t *)p;
    if (int s = take(rl_list, NR * 4, &p)) return s; T.rl_list = (uint32_t *)p;
    PHZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    T.nv = nv; T.nb = n_bams; T.n_lines = sz->n_lines; T.n_kept = sz->n_kept; T.n_edges = sz->n_edges; T.n_rl = sz->n_read_list;
    return PHZ_OK;
}
