#!/usr/bin/env python3
"""GPU-box measurement of the PCIe-inclusive mapper rate: phz_map_reads on a shard that lives in HOST memory (the library stages it
to the GPU itself), pageable and pinned, next to the resident-shard rate.  Never the bench's `value`."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import dataclasses
import torch
from phaser_amd import workloads
from phaser_amd.mapper import Mapper
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
v, shard, _ = workloads.make_shard("chr1", workloads.CHR1_LEN, 40_000, n, 20240807, "cuda:0")
m = Mapper(0)
calls = m.map(shard, v.pos.to("cuda:0"), 10); cap = calls.n + 16
def rate(sh, vpos, reps=3):
    m.map(sh, vpos, 10, cap=cap)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        m.map(sh, vpos, 10, cap=cap)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
t_dev = rate(shard, v.pos.to("cuda:0"), 10)
host = shard.to("cpu")
t_page = rate(host, v.pos)
pinned = dataclasses.replace(host, **{f.name: (getattr(host, f.name).pin_memory() if isinstance(getattr(host, f.name), torch.Tensor) else getattr(host, f.name))
                                      for f in dataclasses.fields(host)})
t_pin = rate(pinned, v.pos)
gb = shard.nbytes_map_inputs() / 1e9
print("records %d (%.2f GB of shard): resident %.2f ms (%.1f G rec/s) | host pageable %.1f ms (%.2f G rec/s, %.1f GB/s) | host pinned %.1f ms (%.2f G rec/s, %.1f GB/s)"
      % (shard.n, gb, t_dev * 1e3, shard.n / t_dev / 1e9, t_page * 1e3, shard.n / t_page / 1e9, gb / t_page, t_pin * 1e3, shard.n / t_pin / 1e9, gb / t_pin))
