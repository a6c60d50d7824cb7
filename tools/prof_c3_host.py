import os, sys, time, cProfile, pstats
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, REPO)
import torch
from phaser_amd import workloads, synth, vcf
from phaser_amd.engine import Engine, Config
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
        114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468]
total_len = sum(HG38)
vsets = []; shards = {}
for i, ln in enumerate(HG38):
    chrom = "chr%d" % (i + 1)
    n_snps = int(1_500_000 * ln / total_len); n_rec = int(80_000_000 * ln / total_len)
    v, shard, _ = workloads.make_shard(chrom, ln, n_snps, n_rec, 777 + i, "cuda:0")
    vsets.append(v); shards[chrom] = shard
vs = vcf.load_variants("\n".join(synth.vcf_lines(vsets)))
eng = Engine(vs, ["gtex_like"], Config(host_threads=32, want_vcf=False))
for chrom, shard in shards.items():
    eng.add_shard(0, chrom, shard, int(shard.qid.max()) + 1)
torch.cuda.synchronize()
eng.close_bam(0)
pr = cProfile.Profile(); pr.enable()
files = eng.finish(chunks=True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
print({k: round(v, 3) for k, v in eng.stats.items()})
