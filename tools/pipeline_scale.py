#!/usr/bin/env python3
"""GPU-box timing of the phasing stages (T1-O2) at scale; prints a stage breakdown."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import workloads, synth, vcf
from phaser_amd.engine import Engine, Config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
snps = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000
t0 = time.perf_counter()
v, shard, _ = workloads.make_shard("chr1", workloads.CHR1_LEN, snps, n, 20240807, "cuda:0")
vs = vcf.load_variants("\n".join(synth.vcf_lines([v])))
torch.cuda.synchronize(); t1 = time.perf_counter()
eng = Engine(vs, ["bench"], Config())
eng.add_shard(0, "chr1", shard, int(shard.qid.max()) + 1)
torch.cuda.synchronize(); t2 = time.perf_counter()
eng.close_bam(0); t3 = time.perf_counter()
m = eng.tally_all(); t4 = time.perf_counter()
noise = eng.noise_from_counts(*m)
frag = eng._fragments(noise)["chr1"]; t5 = time.perf_counter()
print("records %d snps %d | gen %.2fs map %.3fs as_cutoff %.3fs tally %.3fs (kernel %.2f ms) fragment(host) %.2fs %s | phased %d lines %d blocks %d"
      % (n, snps, t1 - t0, t2 - t1, t3 - t2, t4 - t3, eng.ctx.timing(2)[0], t5 - t4, {k: round(v, 3) for k, v in eng.stats.items()},
         frag["phased"], frag["lines"], frag["n_blocks"]))
if len(sys.argv) > 3:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    eng._fragments(noise)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
