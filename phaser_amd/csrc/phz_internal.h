// Internal declarations shared by the HIP translation units of libphz.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "phz.h"

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct phz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
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
    std::vector<hipEvent_t> map_ev;    // event pairs around every k_map launch of a batch
    // staging for PHZ_HOST callers
    DevBuf r_pos, r_coff, r_cig, r_soff, r_seq, r_qual, v_pos, v_reflen;
    DevBuf c_read, c_var, c_code, c_aux0, c_aux1;
    // generic per-call scratch slots (tally / components), grown on demand and reused across calls
    DevBuf scratch[24];
    int map_tile_reads = 0;
    int map_slot_cap = 0;      // calls per tile slot of K_map's staging area (grown on demand)
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

// Scoped staging of host arrays for PHZ_HOST callers: device copies live until the object dies.
struct Staging {
    phz_ctx *ctx;
    std::vector<void *> owned;
    explicit Staging(phz_ctx *c) : ctx(c) {}
    ~Staging() { for (void *p : owned) (void)hipFree(p); }
    // returns a device pointer holding `bytes` bytes copied from host pointer src (or src itself in device space)
    template <class T> int in(const T *src, size_t count, int space, const T **dst) {
        if (space == PHZ_DEVICE || src == nullptr) { *dst = src; return PHZ_OK; }
        void *d = nullptr;
        size_t bytes = count * sizeof(T);
        hipError_t e = hipMalloc(&d, bytes ? bytes : 1);
        if (e != hipSuccess) return phz_fail(ctx, PHZ_E_NOMEM, "hipMalloc(staging)", e);
        owned.push_back(d);
        if (bytes) { e = hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream); if (e != hipSuccess) return phz_fail(ctx, PHZ_E_HIP, "H2D", e); }
        *dst = (const T *)d;
        return PHZ_OK;
    }
    // device buffer for an output of `count` elements; host pointer remembered by the caller for the copy back
    template <class T> int out(T *host, size_t count, int space, T **dev) {
        if (space == PHZ_DEVICE) { *dev = host; return PHZ_OK; }
        void *d = nullptr;
        size_t bytes = count * sizeof(T);
        hipError_t e = hipMalloc(&d, bytes ? bytes : 1);
        if (e != hipSuccess) return phz_fail(ctx, PHZ_E_NOMEM, "hipMalloc(staging)", e);
        owned.push_back(d);
        *dev = (T *)d;
        return PHZ_OK;
    }
};
