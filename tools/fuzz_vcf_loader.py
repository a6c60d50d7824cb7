#!/usr/bin/env python3
"""Build-container fuzz (needs /root/reference): the native het-variant loader (phz_vcf_parse) vs the variant tables the reference's
own process_vcf / generate_mapping_table writes, on synthetic VCFs whose lines are randomly mutated (FILTER lists, multi-allelic ALT,
phased / unphased / homozygous / missing genotypes, extra FORMAT fields, INFO AF lists, indels) under random --pass_only,
--include_indels, --gw_phase_method.  usage: PYTHONHASHSEED=0 tools/fuzz_vcf_loader.py [iterations=40] [seed=1]"""
import os, random, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, REPO)
import make_golden as mg
from phaser_amd import synth, vcf
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
phaser, rvm = mg.build_reference()
bad = 0
for it in range(iters):
    rng = random.Random(seed0 * 1000 + it)
    contigs = [("chr3", 198295559), ("chr11", 135086622)][:rng.choice([1, 2])]
    vs_ = []; sams = {"x.bam": {}}
    for ci, (chrom, ln) in enumerate(contigs):
        v, gs, ge, w = synth.make_variants(chrom, 1, 800_000, rng.choice([80, 200]), seed0 * 77 + 5 * it + ci, n_genes=6)
        rb = synth.make_reads(v, gs, ge, w, 1500, seed0 * 99 + 7 * it + ci)
        rf = rb.select(synth.samtools_keep(rb, 255))
        sams["x.bam"][chrom] = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
        vs_.append(v)
    lines = []
    for l in synth.vcf_lines(vs_):
        if l.startswith("#"):
            lines.append(l); continue
        c = l.split("\t")
        r = rng.random()
        if r < 0.06: c[6] = rng.choice(["q10", "q10;PASS", "LowQual", "."])
        r = rng.random()
        if r < 0.05: c[9] = rng.choice(["1|1", "0|0", ".|1", "./.", "0/1", "1/0"])
        elif r < 0.10:
            c[4] = c[4] + "," + rng.choice([x for x in "ACGT" if x not in (c[3], c[4])]); c[9] = rng.choice(["1|2", "2|1", "0|2", "1/2"])
        elif r < 0.14:
            c[3] = c[3] + rng.choice(["A", "CG"]) if rng.random() < 0.5 else c[3]; c[4] = c[4] + ("TT" if rng.random() < 0.5 else "")
        if rng.random() < 0.5:
            nalt = c[4].count(",") + 1
            c[7] = rng.choice(["", "DP=10;"]) + "AF=" + ",".join("%.3g" % rng.choice([0.01, 0.12, 0.5, 0.77, 1e-05]) for _ in range(nalt if rng.random() < 0.9 else nalt + 1))
        if rng.random() < 0.2:
            c[8] = "GT:DP"; c[9] = c[9] + ":7"
        elif rng.random() < 0.03:
            c[8] = "DP"; c[9] = "7"
        lines.append("\t".join(c))
    vcf_text = "\n".join(lines) + "\n"
    kw = {"pass_only": rng.choice([0, 1]), "include_indels": rng.choice([0, 1]), "gw_phase_method": rng.choice([0, 1])}
    with tempfile.TemporaryDirectory() as tmp:
        try:
            res, calls = mg.run_pipeline(phaser, rvm, vcf_text, sams, tmp, **kw)
        except BaseException as e:
            print("iter %d: reference raised %s: %s" % (it, type(e).__name__, str(e)[:80])); continue
    vset = vcf.load_variants(vcf_text, threads=3, **kw)
    ok = True
    for chrom, cv in vset.chroms.items():
        want = calls.get(("table", chrom))
        got = "".join("\t".join(r) + "\n" for r in cv.table_rows())
        if want is None:
            ok = ok and len(cv) == 0
        elif got != want:
            ok = False
            w = want.split("\n"); g = got.split("\n")
            k = next((i for i in range(min(len(w), len(g))) if w[i] != g[i]), min(len(w), len(g)))
            print("  %s line %d: want %r got %r" % (chrom, k, w[k:k + 1], g[k:k + 1]))
    bad += not ok
    print("iter %d %s: %d het sites kept -> %s" % (it, kw, vset.het_count, "OK" if ok else "DIFF"), flush=True)
print("%d iterations, %d with differences" % (iters, bad))
