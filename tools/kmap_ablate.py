#!/usr/bin/env python3
"""GPU-box profiling helper: times K_map variants on the bench shard (env switches PHZ_MAP_BLOCK / _RPT / _DBG).  Not part of the product path."""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import workloads, _lib
from phaser_amd.mapper import Mapper
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
v, shard, _ = workloads.make_shard("chr1", workloads.CHR1_LEN, 40_000, n, 20240807, "cuda:0")
m = Mapper(0); vpos = v.pos.to("cuda:0")
calls = m.map(shard, vpos, 10); cap = calls.n + 16
ref = calls.n
for blk, rpt in [("128", "2"), ("256", "2"), ("128", "2")]:      # the other shapes are no longer built (phz_map.hip)
    os.environ["PHZ_MAP_BLOCK"] = blk; os.environ["PHZ_MAP_RPT"] = rpt
    c2 = m.map(shard, vpos, 10, cap=cap)
    m.ctx.reset_timing()
    t0 = time.perf_counter()
    for _ in range(8):
        m.map(shard, vpos, 10, cap=cap)
    wall = (time.perf_counter() - t0) / 8 * 1e3
    print("blk=%s rpt=%s  k_map avg %.3f ms   wall/step %.3f ms  calls %d (%s)" %
          (blk, rpt, m.ctx.timing()[1] / 8, wall, c2.n, "same" if c2.n == ref else "DIFF"), flush=True)
