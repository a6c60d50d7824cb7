#!/usr/bin/env python3
"""GPU-box probe: CPU read/compute speed on page-locked (hipHostMalloc) arrays vs pageable ones, and D2H rates into both."""
import time, numpy as np, torch
n = 20_000_000
dev = torch.arange(n, dtype=torch.int32, device="cuda")
for name, host in (("pinned", torch.empty(n, dtype=torch.int32, pin_memory=True)), ("pageable", torch.empty(n, dtype=torch.int32))):
    torch.cuda.synchronize(); t = time.perf_counter(); host.copy_(dev); torch.cuda.synchronize(); d2h = time.perf_counter() - t
    a = host.numpy()
    t = time.perf_counter(); s = int(a.sum()); t1 = time.perf_counter() - t
    t = time.perf_counter(); b = np.maximum(a, 5); t2 = time.perf_counter() - t
    t = time.perf_counter(); c = a.copy(); t3 = time.perf_counter() - t
    print("%-9s D2H %.1f ms (%.1f GB/s)  sum %.1f ms  maximum %.1f ms  copy %.1f ms" % (name, d2h * 1e3, n * 4 / d2h / 1e9, t1 * 1e3, t2 * 1e3, t3 * 1e3))
