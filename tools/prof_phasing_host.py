#!/usr/bin/env python3
"""GPU-box profile of one phasing pass (stages T1-O2) over the configs[2] genome, host side: cProfile of Engine.close_bam +
Engine.finish on resident shards and call lists (the same objects bench.py times), sorted by cumulative and by own time.
usage: tools/prof_phasing_host.py [scale=1.0]"""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import dist as pdist, synth, vcf as pvcf, workloads
from phaser_amd.engine import Config, Engine
from phaser_amd.mapper import Mapper
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
plan = workloads.genome_plan(scale=scale)
vsets = {}; shards = {}
for chrom, ln, n_snps, n_rec, seed in plan:
    v, sh, _ = workloads.make_shard(chrom, ln, n_snps, n_rec, seed, "cuda:0")
    vsets[chrom] = v; shards[chrom] = sh
mapper = Mapper(0)
chroms = [p[0] for p in plan]
calls = mapper.map_batch([shards[c] for c in chroms], [vsets[c].pos for c in chroms], 10)
vs = pvcf.load_variants("\n".join(synth.vcf_lines([vsets[c] for c in chroms])))
threads = max(1, min(64, 4 * pdist.effective_cpus()))


def one_pass():
    eng = Engine(vs, ["bench"], Config(baseq=10, host_threads=threads, want_vcf=False), mapper=mapper)
    eng.set_owned(chroms)
    for i, c in enumerate(chroms):
        eng.add_mapped(0, c, shards[c], calls[i], int(shards[c].qid.max()) + 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.close_bam(0)
    files = eng.finish(chunks=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, eng


for _ in range(3):
    dt, eng = one_pass(); print("pass %.4f s, %d phased" % (dt, eng.phased), {k: round(v, 4) for k, v in eng.stats.items()})
    del eng; pdist.cleanup_spool()
pr = cProfile.Profile(); pr.enable(); dt, eng = one_pass(); pr.disable()
print("profiled pass %.4f s" % dt)
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28); st.sort_stats("tottime").print_stats(18)
# own time per function in microseconds (pstats prints milliseconds with three decimals: too coarse for a 9 ms pass)
rows = sorted(((v[2], v[3], v[0], k) for k, v in st.stats.items()), reverse=True)[:45]
print("own us   cum us   calls  function")
for tt, ct, cc, (fn, ln, name) in rows:
    print("%7.0f %8.0f %6d  %s:%d(%s)" % (tt * 1e6, ct * 1e6, cc, os.path.basename(fn), ln, name))
