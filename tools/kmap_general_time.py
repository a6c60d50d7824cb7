#!/usr/bin/env python3
"""GPU-box timing of K_map_general (indel mode) on the bench shard: the same het SNPs handed over as allele strings, so the
call list must match K_map's (codes 5/6 instead of base codes) and the time shows the price of the general path."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import numpy as np
import torch
from phaser_amd import workloads
from phaser_amd.mapper import Mapper
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
v, shard, _ = workloads.make_shard("chr1", workloads.CHR1_LEN, 40_000, n, 20240807, "cuda:0")
m = Mapper(0); vpos = v.pos.to("cuda:0")
base = m.map(shard, vpos, 10)
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
nv = len(v)
ab = np.empty(2 * nv, dtype=np.uint8); ab[0::2] = letters[v.ref.numpy()]; ab[1::2] = letters[v.alt.numpy()]
aoff = torch.arange(2 * nv + 1, dtype=torch.int32)
ref_len = torch.ones(nv, dtype=torch.uint8)
calls, _ = m.map_general(shard, vpos, ref_len, aoff, torch.from_numpy(np.concatenate([ab, np.zeros(1, np.uint8)])), 10)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    calls, _ = m.map_general(shard, vpos, ref_len, aoff, torch.from_numpy(np.concatenate([ab, np.zeros(1, np.uint8)])), 10)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
same = calls.n == base.n and bool(torch.equal(calls.read_idx, base.read_idx)) and bool(torch.equal(calls.var_idx, base.var_idx))
print("K_map_general: %d records, %d calls (K_map %d, same (read, variant) list: %s) | %.2f ms per pass incl. allocation -> %.2f G records/s"
      % (shard.n, calls.n, base.n, same, dt * 1e3, shard.n / dt / 1e9))
