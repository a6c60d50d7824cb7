#!/usr/bin/env python3
"""GPU-box check of --output_read_ids 1 at scale: the five files of the device row stage against the host twin, byte for byte, on a share of configs[2] (22 chromosomes) and on
the configs[1] shard (chr1, 50 M records over 40,000 het SNPs: read sets of tens of thousands of QNAMEs, the workgroup / global-pool paths of the read-set kernels), with the
QNAME columns on.  QNAMEs are synthetic ("q<id>" per chromosome: ids are per chromosome, the pool is what phz_rowsdev_opts.qname_* carries).
usage: tools/read_ids_scale.py [share=0.1] [--c2]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402


def run(mapper, vs, chroms, shards, calls, device_rows):
    from phaser_amd.engine import Engine, Config
    eng = Engine(vs, ["scale"], Config(baseq=10, host_threads=32, want_vcf=False, output_read_ids=1, device_rows=device_rows), mapper=mapper)
    eng.set_owned(chroms)
    for i, c in enumerate(chroms):
        n_qid = int(shards[c].qid.max()) + 1
        eng.add_mapped(0, c, shards[c], calls[i], n_qid, qnames=["%s.q%d" % (c, k) for k in range(n_qid)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.close_bam(0)
    out = eng.finish(binary=True)
    dt = time.perf_counter() - t0
    return out, eng, dt


def compare(name, mapper, vs, chroms, shards, calls):
    dev, e1, t1 = run(mapper, vs, chroms, shards, calls, True)
    host, e2, t2 = run(mapper, vs, chroms, shards, calls, False)
    assert e1.rows_path == "device" and e2.rows_path == "host", (e1.rows_path, e2.rows_path, getattr(e1, "rows_fallback", ""))
    ok = True
    for k in dev:
        same = dev[k] == host[k]
        ok = ok and same
        print("  %-22s %12d bytes  sha256/16 %s  %s" % (k, len(dev[k]), hashlib.sha256(dev[k]).hexdigest()[:16], "= host twin" if same else "DIFFERS from the host twin"))
    rows = dev["haplotypic_counts"].split(b"\n")
    widest = max(rows, key=len)
    print("%s: device %.3f s, host twin %.3f s, phased %d, haplotypic_counts rows %d, widest row %d bytes (%d QNAMEs in its first list) -> %s" % (
        name, t1, t2, e1.phased, len(rows) - 2, len(widest), widest.split(b"\t")[14].count(b",") + 1 if widest.count(b"\t") >= 19 else 0, "IDENTICAL" if ok else "DIFFERENT"), flush=True)
    return ok


def main():
    share = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 0.1
    from phaser_amd import workloads, synth, vcf as pvcf
    from phaser_amd.mapper import Mapper
    dev = "cuda:0"
    mapper = Mapper(0)
    ok = True
    plan = workloads.genome_plan(int(80_000_000 * share), int(1_500_000 * share))
    vsets = {}; shards = {}
    for chrom, ln, n_snps, n_rec, seed in plan:
        v, shard, _ = workloads.make_shard(chrom, ln, n_snps, n_rec, seed, dev)
        vsets[chrom] = v; shards[chrom] = shard
    chroms = [p[0] for p in plan]
    calls = mapper.map_batch([shards[c] for c in chroms], [vsets[c].pos for c in chroms], 10)
    vs = pvcf.load_variants("\n".join(synth.vcf_lines([vsets[c] for c in chroms])))
    ok = compare("configs[2] x %.2f" % share, mapper, vs, chroms, shards, calls) and ok
    del vsets, shards, calls, vs
    torch.cuda.empty_cache()
    if "--c2" in sys.argv:
        v, shard, _ = workloads.make_shard("chr1", workloads.CHR1_LEN, 40_000, 12_000_000, 20240807, dev)          # a quarter of configs[1]'s depth: 3 M QNAME strings in Python
        calls = mapper.map_batch([shard], [v.pos], 10)
        vs = pvcf.load_variants("\n".join(synth.vcf_lines([v])))
        ok = compare("configs[1] shape (chr1, 12 M records, 40,000 het SNPs)", mapper, vs, ["chr1"], {"chr1": shard}, calls) and ok
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
