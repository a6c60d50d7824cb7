#!/usr/bin/env python3
"""GPU-box profiling helper: times K_map on the bench shard with ablation switches (PHZ_MAP_DBG bits:
1 = no seq/qual gather, 2 = no emit pass, 4 = no look-back, 8 = no CIGAR walk).  Not part of the product path."""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import workloads, _lib
from phaser_amd.mapper import Mapper
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
v, shard, _ = workloads.make_shard("chr1", workloads.CHR1_LEN, 40_000, n, 20240807, "cuda:0")
m = Mapper(0); vpos = v.pos.to("cuda:0")
calls = m.map(shard, vpos, 10); cap = calls.n + 16
for blk, rpt in [("256","2"),("128","2"),("64","2"),("64","4"),("128","4"),("64","2")]:
  os.environ["PHZ_MAP_BLOCK"] = blk
  if True:
    os.environ["PHZ_MAP_RPT"] = rpt
    for dbg in [0, 8]:
        os.environ["PHZ_MAP_DBG"] = str(dbg)
        m.map(shard, vpos, 10, cap=cap)
        m.ctx.reset_timing()
        t0 = time.perf_counter()
        for _ in range(8):
            m.map(shard, vpos, 10, cap=cap)
        wall = (time.perf_counter() - t0) / 8 * 1e3
        print("blk=%s rpt=%s dbg=%2d  k_map avg %.3f ms   wall/step %.3f ms" % (blk, rpt, dbg, m.ctx.timing()[1] / 8, wall), flush=True)
