// hipemu.h -- TEST INFRASTRUCTURE ONLY.  A minimal lock-step emulation of the HIP programming model on the host, so that the
// *logic* of the kernels under phaser_amd/csrc can be exercised by the CPU test suite (-m "not gpu") in a container without a GPU.
// Nothing under phaser_amd/ includes, links or loads this: the product path is libphz.so built by hipcc for gfx950, and it fails
// loudly without a GPU (tests/test_abi.py).  Timing, memory spaces and occupancy are NOT modelled; parity on the real device is
// established by the -m gpu tests.
//
// Model: one block at a time; every thread of the block is a ucontext fiber.  A fiber runs until it reaches __syncthreads() or a
// wave-level operation (__shfl*, __ballot, ...), where it yields until all live threads of the block / wave have arrived.  Wave
// operations require every live lane of the wave to execute the same sequence of operations (i.e. they must sit in wave-uniform
// control flow, or after other lanes have returned): anything else deadlocks here and is reported -- on the hardware it would be
// a data race on the exec mask.  Waves are 64 lanes wide (gfx950).
#ifndef HIPEMU_H
#define HIPEMU_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <ucontext.h>

#include <algorithm>
#include <functional>
#include <tuple>
#include <utility>

struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---------------------------------------------------------------------------------------------- runtime API subset
typedef enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorUnknown = 999 } hipError_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
struct hipemu_stream;
typedef hipemu_stream *hipStream_t;
struct hipemu_event { double t; };
typedef hipemu_event *hipEvent_t;
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2 };

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : (e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipError"); }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event{0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    e->t = (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
template <class T> inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) { *p = (T *)malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }

// ---------------------------------------------------------------------------------------------- execution model
namespace hipemu {

struct Fiber {
    ucontext_t ctx;
    dim3 tid;
    int lane = 0, wave = 0;
    bool done = false;
    uint64_t bar_count = 0;     // barriers this thread has passed
    uint64_t op_count = 0;      // wave operations this lane has arrived at
};
struct WaveBox { alignas(16) unsigned char buf[2][64][16]; };

struct Block {
    dim3 bid, bdim, gdim;
    int nthreads = 0;
    Fiber *fib = nullptr;
    WaveBox *waves = nullptr;
    ucontext_t sched;
    bool progress = false;
};

extern Block *g_blk;
extern Fiber *g_cur;

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body);
void yield();
void syncthreads();
// every live lane of the calling lane's wave contributes `sz` (<= 16) bytes; returns all contributions and the mask of contributors
void wave_xchg(const void *val, size_t sz, unsigned char (*all)[16], uint64_t *mask);

template <class T> struct Gather {
    unsigned char all[64][16];
    uint64_t mask;
    explicit Gather(const T &v) { static_assert(sizeof(T) <= 16, "wave operand too wide"); wave_xchg(&v, sizeof(T), all, &mask); }
    T at(int lane) const { T r; memset(&r, 0, sizeof(T)); if (lane >= 0 && lane < 64 && ((mask >> lane) & 1)) memcpy(&r, all[lane], sizeof(T)); return r; }
};

}  // namespace hipemu

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __constant__ static const
#define ext_vector_type(N) vector_size(4 * (N))          // clang's vector attribute spelled for g++ (the emulated units use 4-byte elements only)
inline unsigned __builtin_bitreverse32(unsigned v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(v);
}
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define threadIdx (hipemu::g_cur->tid)
#define blockIdx (hipemu::g_blk->bid)
#define blockDim (hipemu::g_blk->bdim)
#define gridDim (hipemu::g_blk->gdim)
static const int warpSize = 64;

inline void __syncthreads() { hipemu::syncthreads(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

template <class T> inline T __shfl(T v, int src, int width = 64) {
    hipemu::Gather<T> g(v);
    const int lane = hipemu::g_cur->lane;
    const int base = lane & ~(width - 1);
    return g.at(base + (src & (width - 1)));
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    hipemu::Gather<T> g(v);
    const int lane = hipemu::g_cur->lane;
    const int base = lane & ~(width - 1);
    return (lane - base) >= (int)d ? g.at(lane - (int)d) : v;
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    hipemu::Gather<T> g(v);
    const int lane = hipemu::g_cur->lane;
    const int base = lane & ~(width - 1);
    return (lane - base) + (int)d < width ? g.at(lane + (int)d) : v;
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) {
    hipemu::Gather<T> g(v);
    const int lane = hipemu::g_cur->lane;
    (void)width;
    return g.at(lane ^ m);
}
#define __HIP_MEMORY_SCOPE_AGENT 4
template <class T> inline T __hip_atomic_load(const T *p, int, int) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
template <class T, class V> inline void __hip_atomic_store(T *p, V v, int, int) { __atomic_store_n(p, (T)v, __ATOMIC_SEQ_CST); }
inline void __builtin_amdgcn_s_sleep(int) {}
// on the hardware a compiler-level ordering point between the LDS accesses of one wave's lanes (they execute in lock step); here every lane is a
// fiber of its own, so the ordering point has to be a rendezvous of the wave
inline void __builtin_amdgcn_wave_barrier();
inline unsigned long long wall_clock64() { return 0ull; }
inline unsigned long long __ballot(int pred) {
    hipemu::Gather<int> g(pred ? 1 : 0);
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) if (g.at(l)) m |= 1ull << l;
    return m;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline void __builtin_amdgcn_wave_barrier() { (void)__ballot(1); }
inline int __all(int pred) {
    hipemu::Gather<int> g(pred ? 1 : 0);
    for (int l = 0; l < 64; l++) if (((g.mask >> l) & 1) && !g.at(l)) return 0;
    return 1;
}
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }

// atomics: fibers are cooperative and blocks run one after the other, so plain read-modify-write is atomic
template <class T> inline T atomicAdd(T *p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> inline T atomicSub(T *p, T v) { T o = *p; *p = (T)(o - v); return o; }
template <class T> inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicOr(T *p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <class T> inline T atomicAnd(T *p, T v) { T o = *p; *p = (T)(o & v); return o; }
template <class T> inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
// mixed literal / variable operand types, as HIP's overload set accepts them
inline unsigned atomicAdd(unsigned *p, int v) { return atomicAdd<unsigned>(p, (unsigned)v); }
inline unsigned long long atomicAdd(unsigned long long *p, int v) { return atomicAdd<unsigned long long>(p, (unsigned long long)v); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned v) { return atomicAdd<unsigned long long>(p, (unsigned long long)v); }
inline unsigned atomicSub(unsigned *p, int v) { return atomicSub<unsigned>(p, (unsigned)v); }
inline int atomicOr(int *p, unsigned v) { return atomicOr<int>(p, (int)v); }
inline unsigned atomicOr(unsigned *p, int v) { return atomicOr<unsigned>(p, (unsigned)v); }

using std::max;
using std::min;

template <class... KArgs, class... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t, hipStream_t, Args &&...args) {
    std::tuple<typename std::decay<KArgs>::type...> t(std::forward<Args>(args)...);
    hipemu::run_grid(grid, block, [&]() { std::apply(kernel, t); });
}

#endif
