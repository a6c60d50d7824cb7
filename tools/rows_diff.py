#!/usr/bin/env python3
"""Runs ON THE GPU BOX (debugging aid): one chromosome of the configs[2] plan through the product with the device row stage, with the host
row stage, and through oracle/phasing_oracle.py; prints, per file, whether the three agree (canonical form) and the first differing rows."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import torch
from helpers import OUTPUTS, call_text, canonical
from phaser_amd import synth, vcf as pvcf, workloads
from phaser_amd.engine import Config, Engine
from phaser_amd.mapper import Mapper


def main():
    import phasing_oracle as po
    which = sys.argv[1] if len(sys.argv) > 1 else "chr22"
    with_oracle = os.environ.get("PHZ_DIFF_ORACLE", "1") == "1"
    plan = [p for p in workloads.genome_plan() if p[0] == which]
    chrom, ln, n_snps, nr, seed = plan[0]
    mapper = Mapper(0)
    v, sh, _ = workloads.make_shard(chrom, ln, n_snps, nr, seed, "cuda:0")
    calls = mapper.map(sh, v.pos, 10)
    vs = pvcf.load_variants("\n".join(synth.vcf_lines([v])))
    outs = {}
    for mode in ("device", "host"):
        eng = Engine(vs, ["bam0"], Config(want_vcf=False, device_rows=(mode == "device"), host_threads=8), mapper=mapper)
        eng.add_mapped(0, chrom, sh, calls, int(sh.qid.max()) + 1)
        eng.close_bam(0)
        outs[mode] = eng.finish()
        print(mode, "rows on the", eng.rows_path, "phased", eng.phased, getattr(eng, "rows_fallback", ""), flush=True)
    if with_oracle:
        ph = po.Phaser(["bam0"], baseq=10)
        ph.add_bam([call_text(v, sh, calls)])
        outs["oracle"] = ph.finish()
        print("oracle phased", ph.phased, flush=True)
    names = list(outs)
    for name in OUTPUTS:
        can = {k: canonical(name, outs[k][name]).split("\n") for k in names}
        for a in names[1:]:
            x, y = can[names[0]], can[a]
            same = x == y
            print("%-20s %s vs %s: %s (%d / %d rows)" % (name, names[0], a, "identical" if same else "DIFFERENT", len(x), len(y)))
            if not same:
                sx, sy = set(x), set(y)
                for r in sorted(sx - sy)[:4]:
                    print("   only %s: %s" % (names[0], r[:400]))
                for r in sorted(sy - sx)[:4]:
                    print("   only %s: %s" % (a, r[:400]))
        raw_same = outs["device"][name] == outs["host"][name]
        print("%-20s device vs host raw bytes: %s" % (name, "identical" if raw_same else "DIFFERENT"))


if __name__ == "__main__":
    main()
