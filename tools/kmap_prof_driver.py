#!/usr/bin/env python3
"""Minimal driver for rocprofv3 runs: builds the bench shard and launches K_map a few times."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import workloads
from phaser_amd.mapper import Mapper
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
v, shard, _ = workloads.make_shard("chr1", workloads.CHR1_LEN, 40_000, n, 20240807, "cuda:0")
m = Mapper(0); vpos = v.pos.to("cuda:0")
calls = m.map(shard, vpos, 10); cap = calls.n + 16
for _ in range(reps):
    m.map(shard, vpos, 10, cap=cap)
print("k_map ms", m.ctx.timing()[0], "calls", calls.n, "input bytes", shard.nbytes_map_inputs())
