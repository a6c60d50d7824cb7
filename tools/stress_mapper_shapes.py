#!/usr/bin/env python3
"""GPU-box stress of K_map on the extreme record shapes of tests/test_gpu_mapper.py (records of up to ~100 CIGAR operations, het SNPs
every few bases) with seeds the tests do not use: every call list against the C oracle.  usage: tools/stress_mapper_shapes.py [n_seeds=20] [first_seed=100]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
import random
import test_gpu_mapper as T
from phaser_amd.mapper import Mapper
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
m = Mapper(0)
ora = os.path.join(REPO, "oracle")
many = T.test_many_op_records_vs_oracle.__wrapped__ if hasattr(T.test_many_op_records_vs_oracle, "__wrapped__") else T.test_many_op_records_vs_oracle
for k in range(n):
    rng = random.Random(s0 + k)
    gaps = rng.choice([2, 8, 25, 60, 100]); every = rng.choice([4, 6, 12, 25, 60])
    many(m, ora, s0 + k, gaps, every)
    print("seed %d: up to %d gaps per record, a het SNP every %d bases -> identical" % (s0 + k, gaps, every), flush=True)
print("all %d identical" % n)
