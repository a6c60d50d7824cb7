#!/usr/bin/env python3
"""GPU-box stress: product (GPU path) vs the pinned oracle on freshly seeded inputs nobody has seen -- random numbers of BAMs and
chromosomes, error rates, block-size limits, option flags.  Exits non-zero on the first canonical difference in any of the five
files.  usage: tools/stress_parity.py [iterations=40] [first_seed=100]"""
import os, random, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import phasing_oracle as po
from helpers import OUTPUTS, canonical
from phaser_amd import samio, synth, vcf
from phaser_amd.engine import Config, Engine
from phaser_amd.mapper import Mapper
subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mapper = Mapper(0)
CONTIGS = [("chr3", 198295559), ("chr11", 135086622), ("chr19", 58617616)] + [("ctg%02d" % k, 3_000_000 + 1000 * k) for k in range(40)]
for it in range(iters):
    rng = random.Random(seed0 + it)
    nchrom = rng.choice([1, 1, 2, 3, 3, 30]); nbam = rng.choice([1, 1, 2, 3, 3, 11])       # now and then: scaffolds by the dozen, a sample of many BAMs
    many = nchrom * nbam > 9
    contigs = CONTIGS[:nchrom]
    err = rng.choice([0.001, 0.002, 0.02, 0.05, 0.08]); mbs = rng.choice([3, 5, 8, 15])
    cfg = {"max_block_size": mbs}
    if rng.random() < 0.3: cfg["unphased_vars"] = 0
    if rng.random() < 0.3: cfg["unique_ids"] = 1
    if rng.random() < 0.3: cfg["cc_threshold"] = rng.choice([0.001, 0.05, 0.2])
    if rng.random() < 0.3: cfg["as_q_cutoff"] = rng.choice([0.0, 0.2, 0.5])
    if nbam > 1 and rng.random() < 0.3: cfg["haplo_count_bam_exclude"] = [rng.randrange(nbam)]
    if rng.random() < 0.2 or os.environ.get("PHZ_STRESS_READ_IDS") == "1": cfg["output_read_ids"] = 1          # (PHZ_STRESS_READ_IDS=1: every iteration writes the QNAME columns)
    if rng.random() < 0.25: cfg["gw_phase_method"] = 1          # MAF-weighted genome-wide phase (the synthetic VCF carries AF=...): on the device since round 5
    sparse = rng.random() < 0.25                # some BAMs have no read at all on some chromosomes (block order then follows the first BAM that has one)
    vs_ = []; bams = {"x%d.bam" % b: {} for b in range(nbam)}
    for ci, (chrom, ln) in enumerate(contigs):
        dense = rng.random() < 0.35 and not many  # het SNPs every 5-40 bp: tens of calls per read, components of hundreds of variants
        v, gs, ge, w = synth.make_variants(chrom, 1, rng.choice([600_000, 1_500_000]), rng.choice([600, 1500]) if dense else rng.choice([60, 150, 260]),
                                           seed0 * 7 + 13 * it + ci, n_genes=rng.choice([2, 4]) if dense else rng.choice([4, 10]))
        L = rng.choice([76, 76, 150, 600]) if dense else rng.choice([76, 76, 76, 150])
        vs_.append(v)
        for bi, bam in enumerate(bams):
            rb = synth.make_reads(v, gs, ge, w, rng.choice([300, 800]) if ((dense and L > 150) or many) else rng.choice([1500, 4000, 7000]), seed0 * 11 + 17 * it + 10 * ci + bi, L=L,
                                  qname_prefix="q" if rng.random() < 0.7 else "q%d." % bi, err_rate=err)
            rf = rb.select(synth.samtools_keep(rb, 255))
            bams[bam][chrom] = "" if (sparse and rng.random() < 0.4 and not (ci == len(contigs) - 1 and bi == nbam - 1)) else "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf_text = "\n".join(synth.vcf_lines(vs_)) + "\n"
    # product
    vset = vcf.load_variants(vcf_text, gw_phase_method=cfg.get("gw_phase_method", 0))
    eng = Engine(vset, po.bam_display_names(list(bams.keys())), Config(host_threads=rng.choice([1, 4]), **cfg), mapper=mapper)
    interners = {}
    for bi, (bam, per_chrom) in enumerate(bams.items()):
        for chrom in vset.chroms:
            for c2, sh in samio.shards_from_sam(per_chrom[chrom], interners, 0.0).items():
                eng.add_shard(bi, c2, sh.to("cuda"), len(interners[c2]), interners[c2].names)
        for c2 in interners:
            eng.n_qid[c2] = len(interners[c2])
        eng.close_bam(bi)
    got = eng.finish()
    # oracle
    pool, _, _ = po.load_vcf(vcf_text)
    ph = po.Phaser(po.bam_display_names(list(bams.keys())), **cfg)
    with tempfile.TemporaryDirectory() as tmp:
        for bam, per_chrom in bams.items():
            texts = []
            for c in pool:
                tp = os.path.join(tmp, "t.tsv"); open(tp, "w").write("".join("\t".join(r) + "\n" for r in po.variant_table_rows(pool[c], gw_phase_method=cfg.get("gw_phase_method", 0))[0]))
                op = os.path.join(tmp, "c.tsv")
                subprocess.run([os.path.join(REPO, "oracle", "rvm_oracle"), "--variant_table", tp, "--baseq", "10", "--o", op], input=per_chrom[c].encode(), check=True)
                texts.append(open(op).read())
            ph.add_bam(texts)
    want = ph.finish()
    bad = [n for n in OUTPUTS if canonical(n, got[n]) != canonical(n, want[n])]
    print("iter %d seed %d: chroms %d bams %d%s err %.3f cfg %s -> phased %d %s" % (it, seed0 + it, nchrom, nbam, " sparse" if sparse else "", err, cfg, eng.phased, "OK" if not bad else "DIFF " + str(bad)), flush=True)
    if bad or eng.phased != ph.phased:
        # which path disagrees, and where: the same inputs through the other row stage / without the QNAME columns, first differing rows of every file
        def product(**over):
            c2 = dict(cfg); c2.update(over)
            e2 = Engine(vset, po.bam_display_names(list(bams.keys())), Config(host_threads=1, **c2), mapper=mapper)
            it2 = {}
            for bi, (bam, per_chrom) in enumerate(bams.items()):
                for chrom in vset.chroms:
                    for c3, sh in samio.shards_from_sam(per_chrom[chrom], it2, 0.0).items():
                        e2.add_shard(bi, c3, sh.to("cuda"), len(it2[c3]), it2[c3].names)
                for c3 in it2:
                    e2.n_qid[c3] = len(it2[c3])
                e2.close_bam(bi)
            return e2.finish(), e2
        for label, over in (("rows path as run", {}), ("device_rows=False", {"device_rows": False}), ("output_read_ids=0 (device rows)", {"output_read_ids": 0})):
            g2, e2 = product(**over)
            w2 = want
            if over.get("output_read_ids") == 0 and cfg.get("output_read_ids"):
                ph2 = po.Phaser(po.bam_display_names(list(bams.keys())), **{k: v for k, v in cfg.items() if k != "output_read_ids"})
                with tempfile.TemporaryDirectory() as tmp:
                    for bam, per_chrom in bams.items():
                        texts = []
                        for c in pool:
                            tp = os.path.join(tmp, "t.tsv"); open(tp, "w").write("".join("\t".join(r) + "\n" for r in po.variant_table_rows(pool[c], gw_phase_method=cfg.get("gw_phase_method", 0))[0]))
                            op = os.path.join(tmp, "c.tsv")
                            subprocess.run([os.path.join(REPO, "oracle", "rvm_oracle"), "--variant_table", tp, "--baseq", "10", "--o", op], input=per_chrom[c].encode(), check=True)
                            texts.append(open(op).read())
                        ph2.add_bam(texts)
                w2 = ph2.finish()
            b2 = [n for n in OUTPUTS if canonical(n, g2[n]) != canonical(n, w2[n])]
            print("   %-34s rows on the %-6s -> %s" % (label, e2.rows_path, "identical" if not b2 else "DIFF " + str(b2)), flush=True)
            for n in b2:
                a_ = canonical(n, g2[n]).split("\n"); b_ = canonical(n, w2[n]).split("\n")
                k = next((i for i in range(min(len(a_), len(b_))) if a_[i] != b_[i]), min(len(a_), len(b_)))
                print("      %s: %d rows (product) vs %d (oracle), first difference at row %d" % (n, len(a_), len(b_), k))
                for i in range(max(0, k - 1), min(k + 3, max(len(a_), len(b_)))):
                    print("        product: %s" % (a_[i][:200] if i < len(a_) else "<none>")); print("        oracle : %s" % (b_[i][:200] if i < len(b_) else "<none>"))
                same_set = sorted(a_) == sorted(b_)
                print("      same multiset of rows: %s" % same_set)
        sys.exit(1)
print("all %d iterations identical" % iters)
