#!/usr/bin/env python3
"""Runs ON THE GPU BOX: where does the file -> HBM path of the device BAM decoder spend its time?  (round-4 verdict, weak #6: 3.8 GB of BGZF go over at
13 GB/s against ~55 GB/s for PCIe Gen5.)  A file of random bytes in /tmp (page cache) is moved with: pread() into pageable / page-locked buffers on T threads,
page-locked -> device copies alone, the full pread -> copy pipeline on T threads and streams, and hipHostRegister of the mapped file + one copy out of it.
usage: tools/h2d_probe.py [GB=2]"""
import ctypes as C
import mmap
import os
import sys
import threading
import time

import numpy as np
import torch

GB = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
N = int(GB * (1 << 30)) & ~((8 << 20) - 1)
path = "/tmp/h2d_probe.bin"
CH = 8 << 20
t0 = time.perf_counter()
with open(path, "wb") as f:
    blk = np.random.default_rng(1).integers(0, 256, 64 << 20, dtype=np.uint8).tobytes()
    for o in range(0, N, len(blk)):
        f.write(blk[:min(len(blk), N - o)])
print("wrote %.2f GB in %.2f s; cpu quota: %s; cpus visible %d" % (N / 1e9, time.perf_counter() - t0, open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?", os.cpu_count()), flush=True)
fd = os.open(path, os.O_RDONLY)
dev = torch.empty(N, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def run_threads(T, fn):
    th = [threading.Thread(target=fn, args=(t, T)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for pinned in (False, True):
    for T in (1, 2, 4, 8, 16, 32):
        bufs = [torch.empty(2 * CH, dtype=torch.uint8, pin_memory=pinned) for _ in range(T)]
        views = [memoryview(b.numpy()) for b in bufs]
        def rd(t, T_):
            lo = N * t // T_ // CH * CH; hi = N * (t + 1) // T_ // CH * CH if t + 1 < T_ else N
            w = 0
            for o in range(lo, hi, CH):
                os.preadv(fd, [views[t][w * CH:(w + 1) * CH]], o); w ^= 1
        dt = run_threads(T, rd)
        print("pread -> %s buffers, %2d threads: %6.1f GB/s" % ("page-locked" if pinned else "pageable   ", T, N / dt / 1e9), flush=True)
        del bufs, views

# page-locked -> device alone
src = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True); src.fill_(7)
for S in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(S)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = max(1, N // (1 << 30))
    for r in range(reps):
        for i, s in enumerate(streams):
            lo = (1 << 30) * i // S; hi = (1 << 30) * (i + 1) // S
            with torch.cuda.stream(s):
                dev[r * (1 << 30) + lo:r * (1 << 30) + hi].copy_(src[lo:hi], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("page-locked -> device, %d streams, 1 GiB pieces: %6.1f GB/s" % (S, reps * (1 << 30) / dt / 1e9), flush=True)
for piece in (1 << 20, 8 << 20, 64 << 20):
    s = torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for o in range(0, 1 << 30, piece):
            dev[o:o + piece].copy_(src[o:o + piece], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("page-locked -> device, 1 stream, %3d MiB pieces: %6.1f GB/s" % (piece >> 20, (1 << 30) / dt / 1e9), flush=True)
del src

# the pipeline of phz_bamdev.hip: T threads, two page-locked buffers each, pread then async copy on the thread's stream
for CHK in (8 << 20, 32 << 20):
    for T in (4, 8, 16, 32):
        bufs = [torch.empty(2 * CHK, dtype=torch.uint8, pin_memory=True) for _ in range(T)]
        views = [memoryview(b.numpy()) for b in bufs]
        streams = [torch.cuda.Stream() for _ in range(T)]
        def pipe(t, T_):
            torch.cuda.set_device(0)
            lo = N * t // T_ // CHK * CHK; hi = N * (t + 1) // T_ // CHK * CHK if t + 1 < T_ else N
            ev = [None, None]; w = 0
            with torch.cuda.stream(streams[t]):
                for o in range(lo, hi, CHK):
                    m = min(CHK, hi - o)
                    if ev[w] is not None: ev[w].synchronize()
                    os.preadv(fd, [views[t][w * CHK:w * CHK + m]], o)
                    dev[o:o + m].copy_(bufs[t][w * CHK:w * CHK + m], non_blocking=True)
                    ev[w] = torch.cuda.Event(); ev[w].record(streams[t]); w ^= 1
        dt = run_threads(T, pipe)
        print("pipeline pread -> page-locked -> device, %2d MiB buffers, %2d threads: %6.1f GB/s" % (CHK >> 20, T, N / dt / 1e9), flush=True)
        del bufs, views, streams

# hipHostRegister of the mapped file, then ONE copy out of the registered mapping
hip = C.CDLL("libamdhip64.so")
mm = mmap.mmap(fd, N, prot=mmap.PROT_READ)
arr = np.frombuffer(mm, dtype=np.uint8)
ptr = arr.ctypes.data
for flags, name in ((0, "default"), (8, "hipHostRegisterReadOnly")):
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(C.c_void_p(ptr), C.c_size_t(N), C.c_uint(flags))
    t_reg = time.perf_counter() - t0
    if rc != 0:
        print("hipHostRegister(mapped file, %s) failed rc=%d after %.3f s" % (name, rc, t_reg), flush=True)
        continue
    t0 = time.perf_counter()
    rc2 = hip.hipMemcpy(C.c_void_p(dev.data_ptr()), C.c_void_p(ptr), C.c_size_t(N), C.c_int(1))
    torch.cuda.synchronize(); t_cp = time.perf_counter() - t0
    t0 = time.perf_counter(); hip.hipHostUnregister(C.c_void_p(ptr)); t_un = time.perf_counter() - t0
    print("hipHostRegister(mapped file, %s): register %.3f s (%.1f GB/s), copy rc=%d %.3f s (%.1f GB/s), unregister %.3f s" %
          (name, t_reg, N / t_reg / 1e9, rc2, t_cp, N / t_cp / 1e9, t_un), flush=True)
    break
del arr
os.close(fd); os.remove(path)
