// Internal declarations shared by the HIP translation units of libphz.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

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
    // scratch
    DevBuf desc, tile_w0, scalars;
    // staging for PHZ_HOST callers
    DevBuf r_pos, r_coff, r_cig, r_soff, r_seq, r_qual, v_pos, v_reflen;
    DevBuf c_read, c_var, c_code, c_aux0, c_aux1;
};

int phz_fail(phz_ctx *ctx, int status, const char *what, hipError_t e = hipSuccess);
int phz_reserve(phz_ctx *ctx, DevBuf &b, size_t bytes);

#define PHZ_HIP(ctx, call)                                                      \
    do {                                                                        \
        hipError_t _e = (call);                                                 \
        if (_e != hipSuccess) return phz_fail((ctx), PHZ_E_HIP, #call, _e);     \
    } while (0)

// launchers implemented in the kernel translation units (device pointers only)
int phz_launch_map(phz_ctx *ctx, const phz_reads &r, const phz_variants &v, int baseq, const phz_calls &out,
                   int64_t *n_calls);
