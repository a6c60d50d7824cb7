// Lock-free union-find on a device edge list (connected components, phaser/phaser.py:1861-1882 / :1985-1998): shared by
// phz_components (phz_tally.hip) and the device row stage (phz_rowsdev.hip).  label[v] = smallest member of v's component.
#pragma once
#include "phz_internal.h"

namespace {

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

}  // namespace
