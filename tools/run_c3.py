#!/usr/bin/env python3
"""GPU-box end-to-end run at BASELINE.json configs[2] shape: autosomes, ~1.5M het SNPs, ~80M records, one sample, one GPU.
Shards are generated directly in HBM (no BAM: there is no samtools to write one at this size); everything after that is the
product path: K_map per chromosome, AS cutoff, K_tally, pair tests, components, block phasing, the five files written to disk."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import workloads, synth, vcf
from phaser_amd.engine import Engine, Config
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
        114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
total_len = sum(HG38)
t0 = time.perf_counter()
vsets = []; shards = {}
for i, ln in enumerate(HG38):
    chrom = "chr%d" % (i + 1)
    n_snps = int(1_500_000 * scale * ln / total_len); n_rec = int(80_000_000 * scale * ln / total_len)
    v, shard, _ = workloads.make_shard(chrom, ln, n_snps, n_rec, 777 + i, "cuda:0")
    vsets.append(v); shards[chrom] = shard
torch.cuda.synchronize(); t1 = time.perf_counter()
vs = vcf.load_variants("\n".join(synth.vcf_lines(vsets)))
t2 = time.perf_counter()
eng = Engine(vs, ["gtex_like"], Config(host_threads=int(sys.argv[2]) if len(sys.argv) > 2 else 1, want_vcf=False))
for chrom, shard in shards.items():
    eng.add_shard(0, chrom, shard, int(shard.qid.max()) + 1)
torch.cuda.synchronize(); t3 = time.perf_counter()
eng.close_bam(0); t4 = time.perf_counter()
files = eng.finish(chunks=True); t5 = time.perf_counter()
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
for name, body in files.items():
    open("/tmp/c3." + name + ".txt", "wb").writelines(body)
t6 = time.perf_counter()
nrec = sum(s.n for s in shards.values()); ncalls = sum(sh.calls.n for c in eng.shards for sh in eng.shards[c] if sh is not None)
print("C3 x%.2f: %d records, %d het SNPs, %d calls | generate %.1fs | vcf parse %.1fs | K_map all chroms %.3fs | AS cutoff %.3fs | "
      "tally+phasing+rows %.1fs | write %.1fs | phased %d | hot path total %.1fs -> %.0f calls/s, %.0f phased variants/s"
      % (scale, nrec, vs.het_count, ncalls, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, eng.phased, t6 - t2,
         ncalls / (t6 - t2), eng.phased / (t6 - t3)))
print({k: sum(len(x) for x in v) for k, v in files.items()}, {k: round(v, 2) for k, v in eng.stats.items()})
