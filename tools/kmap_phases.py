#!/usr/bin/env python3
"""GPU box: K_map kernel time on the bench genome with phases switched off through PHZ_MAP_DBG (bit 1: no base / quality reads, 2: no flush into the
staging slots, 8: no walk of the multi-op records, 16: ..., 32: lean walker off).  The outputs of the switched-off runs are wrong on purpose: timing only."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import workloads, _lib
from phaser_amd.mapper import Mapper
plan = workloads.genome_plan(80_000_000, 1_500_000)
dev = "cuda:0"
vs = []; sh = []
for chrom, ln, n_snps, n_rec, seed in plan:
    v, shard, _ = workloads.make_shard(chrom, ln, n_snps, n_rec, seed, dev)
    vs.append(v.pos); sh.append(shard)
m = Mapper(0)
first = m.map_batch(sh, vs, 10)
call, bufs, N = m.prepare_batch(sh, vs, 10, [c.n + 16 for c in first])
DBGS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 2, 8, 10, 11, 32, 1]
WARM, REPS = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 10)
for dbg in DBGS:
    os.environ["PHZ_MAP_DBG"] = str(dbg)
    for _ in range(WARM):
        call()
    m.ctx.reset_timing()
    torch.cuda.synchronize()
    for _ in range(REPS):
        call()
    torch.cuda.synchronize()
    _, tot, n = m.ctx.timing(_lib.PHZ_T_MAP)
    print("PHZ_MAP_DBG=%-3d k_map %.3f ms" % (dbg, tot / n))
