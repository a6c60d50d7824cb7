// Internal declarations shared by the HIP translation units of libphz.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>
#include <vector>

#include "phz.h"

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct phz_ctx {
    // Every public entry point that works on a ctx holds this lock for the whole call (PhzEnter): two threads that share a ctx are
    // serialised, never interleaved inside its stream / scratch / error string.  Recursive because entry points call each other.
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;      // device -> host copies that run beside the kernels of `stream` (phz_rowsdev_run's copy-as-written), created on first use
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    // timing
    float last_ms[PHZ_T_COUNT] = {0};
    double total_ms[PHZ_T_COUNT] = {0};
    int64_t launches[PHZ_T_COUNT] = {0};
    int64_t counters[PHZ_C_COUNT] = {0};     // work counters accumulated by phz_tally (phz_get_counter)
    // scratch
    DevBuf desc, tile_w0, scalars;
    DevBuf h_scalars;                  // pinned host mirror of `scalars` (hipHostMalloc)
    DevBuf h_bam_stage;                // page-locked staging of the device BAM path (phz_bamdev.hip)
    DevBuf bam_comp, bam_stream, bam_work;       // the device BAM path's big buffers (compressed members, inflated stream, kept-record list) kept between BAMs: a fresh hipMalloc of ~19 GB
                                       // took 1.1-1.5 s every few calls (profiles/r05/cli_4bam_full.txt); given back to the runtime when an allocation fails
                                       // and by phz_ctx_destroy; PHZ_BAM_KEEP_BUFFERS=0 turns the cache off
    DevBuf mail_dev, mail_host;        // PhzMail: gathered small read-backs (device block, page-locked host image)
    DevBuf shard_tab, h_shard_tab;     // shard table of a batched stage (device / pinned host image)
    hipEvent_t tab_ev = nullptr; bool tab_pending = false;      // the last upload of the pinned image: the next one waits for it before it overwrites the image (a stage that returns without a host wait -- phz_as_cutoff_enqueue -- leaves its copy queued)
    DevBuf map_tab;                    // K_map's own device copy of its shard table (shard_tab is shared with the tally / BAM stages)
    std::vector<char> map_tab_image; void *map_tab_dev = nullptr;      // the image last uploaded to map_tab (a repeated submission skips the copy)
    std::vector<hipEvent_t> map_ev;    // event pairs around every k_map launch of a batch
    // staging for PHZ_HOST callers
    DevBuf r_pos, r_coff, r_cig, r_soff, r_seq, r_qual, v_pos, v_reflen;
    DevBuf c_read, c_var, c_code, c_aux0, c_aux1;
    // generic per-call scratch slots (tally / components), grown on demand and reused across calls
    DevBuf scratch[24];
    // device copies of PHZ_HOST callers' arrays (Staging): slot k of a call reuses stage_pool[k], grown on demand, so the
    // steady state allocates nothing
    std::vector<DevBuf> stage_pool;
    std::vector<DevBuf> tally_buf;     // result + read-list buffers of phz_tally
    std::vector<DevBuf> import_buf;    // arrays adopted by phz_tally_import
    std::vector<DevBuf> resident_vars; // phz_load_variants: slot s = {pos, ref_len} of one chromosome's het-variant table at [2 s], [2 s + 1]
    bool tally_dirty = false, tally_table_dirty = true;     // per-QNAME counters / variant-pair table not known to be clean
    // single-pass scan (phz_sort.h): ticket counter + one status word per tile, valid for the current epoch only (never cleared between scans)
    DevBuf scan_state; uint32_t scan_epoch = 0, scan_ticket_base = 0;
    uint64_t tally_gen = 0;            // bumped by every phz_tally / phz_tally_import: stamps what later stages prepared for "the resident tally"
    uint64_t tally_table_cap = 0;      // slots of the variant-pair table that the last phz_tally needed
    DevBuf tally_qcount;               // lines per QNAME: all zero between phz_tally calls (never shared with other stages)
    // results of the last phz_tally, resident in HBM until the next one (phz_tally_fetch / phz_components read them)
    struct {
        int64_t nv = 0, n_lines = 0, n_kept = 0, n_edges = 0, n_rl = 0;
        int nb = 0;
        int32_t *var_count = nullptr, *var_distinct = nullptr, *ea = nullptr, *eb = nullptr, *cells = nullptr, *cto = nullptr, *stats = nullptr, *rl_qid = nullptr;
        int64_t *var_first = nullptr;
        uint64_t *var_rank = nullptr;
        uint32_t *rl_start = nullptr, *rl_list = nullptr;
        uint8_t *linked = nullptr, *line_cls = nullptr;
    } tally;
    int map_tile_reads = 0;
    int map_slot_cap = 0;      // calls per tile slot of K_map's staging area
    int64_t map_ovf_cap = 0;   // calls the overflow area behind the slots holds (grown to what the densest submission needed)
    long long map_ovf_image[5] = {0, 0, 0, 0, 0};      // the overflow-area record last uploaded (+ where): a repeated submission skips the copy
};

// Small values a stage reads back before it can go on (counts, sizes, flags: a few words each, scattered over device buffers) travel as ONE block:
// a tiny kernel gathers them into ctx->mail_dev and one asynchronous copy brings the block to PAGE-LOCKED host memory.  (A hipMemcpyAsync to a pageable
// destination -- a stack variable, a std::vector -- is not asynchronous at all: the call waits for the stream and stages the bytes, 20-30 us each; a stage
// that read ten values back paid that ten times over, measured as 150-250 us of idle GPU per host wait in the phasing pass.)
struct PhzMail {
    static constexpr int MAX = 16;
    phz_ctx *ctx;
    const void *src[MAX]; uint32_t bytes[MAX], off[MAX];
    int n = 0; uint32_t total = 0;
    explicit PhzMail(phz_ctx *c) : ctx(c) {}
    int add(const void *dev, size_t nbytes) {          // -> slot; the value is at<T>(slot) after send() + a host wait on the ctx stream
        if (n >= MAX) return -1;
        src[n] = dev; bytes[n] = (uint32_t)nbytes; off[n] = total;
        total += ((uint32_t)nbytes + 7u) & ~7u;
        return n++;
    }
    int send();
    template <class T> const T *at(int slot) const { return (const T *)((const char *)ctx->mail_host.p + off[slot]); }
};

struct PhzEnter {
    std::unique_lock<std::recursive_mutex> lk;
    explicit PhzEnter(phz_ctx *ctx) { if (ctx) lk = std::unique_lock<std::recursive_mutex>(ctx->mu); }
};

int phz_fail(phz_ctx *ctx, int status, const char *what, hipError_t e = hipSuccess);
int phz_reserve(phz_ctx *ctx, DevBuf &b, size_t bytes);
int phz_reserve_host(phz_ctx *ctx, DevBuf &b, size_t bytes);      // pinned host memory

#define PHZ_HIP(ctx, call)                                                      \
    do {                                                                        \
        hipError_t _e = (call);                                                 \
        if (_e != hipSuccess) return phz_fail((ctx), PHZ_E_HIP, #call, _e);     \
    } while (0)

// launchers implemented in the kernel translation units (device pointers only)
int phz_launch_map(phz_ctx *ctx, const phz_reads &r, const phz_variants &v, int baseq, const phz_calls &out,
                   int64_t *n_calls);
int phz_launch_map_batch(phz_ctx *ctx, int n, const phz_reads *r, const phz_variants *v, int baseq, const phz_calls *out,
                         int64_t *n_calls);

// Scoped staging of host arrays for PHZ_HOST callers: the k-th array of a call lives in ctx->stage_pool[k] (kept across
// calls, grown on demand -- no allocation in the steady state).
struct Staging {
    phz_ctx *ctx;
    size_t next = 0;
    explicit Staging(phz_ctx *c) : ctx(c) {}
    int slot(size_t bytes, void **d) {
        if (next >= ctx->stage_pool.size()) ctx->stage_pool.resize(next + 1);
        DevBuf &b = ctx->stage_pool[next++];
        if (int s = phz_reserve(ctx, b, bytes ? bytes : 1)) return s;
        *d = b.p;
        return PHZ_OK;
    }
    // returns a device pointer holding `count` items copied from host pointer src (or src itself in device space)
    template <class T> int in(const T *src, size_t count, int space, const T **dst) {
        if (space == PHZ_DEVICE || src == nullptr) { *dst = src; return PHZ_OK; }
        void *d = nullptr;
        const size_t bytes = count * sizeof(T);
        if (int s = slot(bytes, &d)) return s;
        if (bytes) { hipError_t e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream); if (e != hipSuccess) return phz_fail(ctx, PHZ_E_HIP, "H2D", e); }
        *dst = (const T *)d;
        return PHZ_OK;
    }
    // device buffer for an output of `count` elements; host pointer remembered by the caller for the copy back
    template <class T> int out(T *host, size_t count, int space, T **dev) {
        if (space == PHZ_DEVICE) { *dev = host; return PHZ_OK; }
        void *d = nullptr;
        if (int s = slot(count * sizeof(T), &d)) return s;
        *dev = (T *)d;
        return PHZ_OK;
    }
};

// ---- plan of a (chromosome-restricted) BAM read, shared by the host decode and the device decode (phz_bam.cpp)
struct PhzBamPlan {
    std::vector<std::pair<std::string, int32_t>> refs;
    std::vector<uint8_t> head;                             // inflated bytes from the file start; the header ends at first_record
    size_t first_record = 0;
    std::vector<std::pair<uint64_t, uint64_t>> pieces;     // [u0, u1) of the inflated stream (global offsets), cut at record boundaries
    struct Mem { uint64_t src; uint32_t csize, isize; uint64_t dst; uint32_t crc; };          // crc: the CRC32 field of the member's trailer
    std::vector<Mem> members;                              // members that overlap a piece, file order; dst = global inflated offset
    const uint8_t *file = nullptr; size_t file_size = 0;   // the mapped file: NULL until phz_bam_plan_map (valid until phz_bam_plan_release)
    void *owner = nullptr;
};
int phz_bam_plan_file(const char *path, const char *const *ref_names, int n_names, PhzBamPlan *out);
const uint8_t *phz_bam_plan_map(PhzBamPlan *p);          // maps the file on first use (the plan itself reads it with pread)
void phz_bam_plan_release(PhzBamPlan *p);

// K_inflate launcher for callers that pipeline it with their own copies (phz_inflate.hip)
int phz_inflate_launch(phz_ctx *ctx, const uint8_t *comp, const phz_bgzf_member *members, int64_t first, int64_t count, uint8_t *out,
                       uint8_t *lens_scratch, int *d_status, hipStream_t s);
int phz_inflate_scratch_bytes_per_member();
// CRC-32 check of the same members' output against the trailers' values (want: device array, one per member of the table), status code 7 into d_status
int phz_crc_launch(phz_ctx *ctx, const phz_bgzf_member *members, int64_t first, int64_t count, const uint8_t *out, const uint32_t *want, int *d_status, hipStream_t s);

