#!/usr/bin/env python3
"""GPU-box tool: the phasing pass (stages T1-O2, text resident in HBM) against the size of the data.
  python tools/pass_sweep.py [--shares 1,0.5,0.25,0.125] [--c2] [--passes 7] [--only-c2]
For every share s of configs[2] (22 shards, 80 M x s records, 1.5 M x s het SNPs) and, with --c2, for the configs[1] shard (chr1, 50 M records,
40,000 het SNPs): K_map once, then `passes` phasing passes; prints median / min wall time of a pass, the sum of the library's HIP-event stage
timers, phased variants and kept call lines.  The intercept of pass time over share is the FIXED part of a pass (launches, host waits, glue)."""
import argparse
import gc
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402


def passes_on(mapper, vs, chroms, shards, calls, n_pass, fetch_text=False):
    from phaser_amd import _lib
    from phaser_amd.engine import Engine, Config
    out = []
    for rep in range(n_pass + 1):
        eng = Engine(vs, ["bench"], Config(baseq=10, host_threads=16, want_vcf=False, fetch_text=fetch_text), mapper=mapper)
        eng.set_owned(chroms)
        for i, c in enumerate(chroms):
            eng.add_mapped(0, c, shards[c], calls[i], int(shards[c].qid.max()) + 1)
        mapper.ctx.reset_timing()
        gc.collect(); gc.disable()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.close_bam(0)
        files = eng.finish(chunks=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        gpu_ms = sum(mapper.ctx.timing(sl)[1] for sl in (_lib.PHZ_T_ASHIST, _lib.PHZ_T_TALLY, _lib.PHZ_T_COMPONENTS, _lib.PHZ_T_ROWS))
        if rep:
            out.append((dt * 1e3, gpu_ms, eng.phased, eng.total_lines, mapper.ctx.timing(_lib.PHZ_T_TALLY)[1],
                        {k: round(v * 1e3, 3) for k, v in eng.stats.items() if k.endswith("_s")}))
        del eng, files
    return out


def report(name, res):
    ms = sorted(r[0] for r in res)
    med = ms[len(ms) // 2]
    r = res[0]
    print("%-28s pass median %7.3f ms  min %7.3f  gpu-event sum %6.3f (tally %5.3f)  phased %8d  kept lines %9d  -> %6.1f M phased variants/s   host stages ms %s"
          % (name, med, ms[0], sorted(x[1] for x in res)[len(res) // 2], sorted(x[4] for x in res)[len(res) // 2], r[2], r[3], r[2] / med / 1e3, r[5]), flush=True)
    return med


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shares", default="1,0.5,0.25,0.125")
    ap.add_argument("--passes", type=int, default=7)
    ap.add_argument("--c2", action="store_true")
    ap.add_argument("--only-c2", action="store_true")
    ap.add_argument("--d2h", action="store_true", help="also a series with the text copied to page-locked host memory")
    a = ap.parse_args()
    from phaser_amd import workloads, synth, vcf as pvcf
    from phaser_amd.mapper import Mapper
    dev = "cuda:0"
    mapper = Mapper(0)
    pts = []
    if not a.only_c2:
        for s in [float(x) for x in a.shares.split(",")]:
            plan = workloads.genome_plan(int(80_000_000 * s), int(1_500_000 * s))
            vsets = {}; shards = {}
            for chrom, ln, n_snps, n_rec, seed in plan:
                v, shard, _ = workloads.make_shard(chrom, ln, n_snps, n_rec, seed, dev)
                vsets[chrom] = v; shards[chrom] = shard
            chroms = [p[0] for p in plan]
            calls = mapper.map_batch([shards[c] for c in chroms], [vsets[c].pos for c in chroms], 10)
            vs = pvcf.load_variants("\n".join(synth.vcf_lines([vsets[c] for c in chroms])))
            med = report("configs[2] x %.3f" % s, passes_on(mapper, vs, chroms, shards, calls, a.passes))
            if a.d2h:
                report("configs[2] x %.3f +D2H" % s, passes_on(mapper, vs, chroms, shards, calls, a.passes, fetch_text=True))
            pts.append((s, med))
            del vsets, shards, calls, vs
            torch.cuda.empty_cache()
        if len(pts) >= 2:
            n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
            sxx = sum(p[0] * p[0] for p in pts); sxy = sum(p[0] * p[1] for p in pts)
            slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
            print("least squares over the shares: pass = %.3f ms fixed + %.3f ms x share" % (icpt, slope), flush=True)
    if a.c2 or a.only_c2:
        v, shard, _ = workloads.make_shard("chr1", workloads.CHR1_LEN, 40_000, 50_000_000, 20240807, dev)
        calls = mapper.map_batch([shard], [v.pos], 10)
        vs = pvcf.load_variants("\n".join(synth.vcf_lines([v])))
        report("configs[1] chr1 50M/40k", passes_on(mapper, vs, ["chr1"], {"chr1": shard}, calls, a.passes))
        if a.d2h:
            report("configs[1] +D2H", passes_on(mapper, vs, ["chr1"], {"chr1": shard}, calls, a.passes, fetch_text=True))


if __name__ == "__main__":
    main()
