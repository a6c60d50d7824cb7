#!/usr/bin/env python3
"""Build-container fuzz (needs /root/reference): the Python restatement of the phasing core (oracle/phasing_oracle.py, fed by the C
mapper oracle) vs the reference's own process_vcf on freshly seeded inputs -- 1-3 chromosomes, 1-3 BAMs with shared QNAMEs, error
rates up to 8 %, random max_block_size / cc_threshold / as_q_cutoff / unphased_vars / unique_ids / output_read_ids / gw_phase_method / BAM exclusion,
BAMs without reads on some chromosomes.
Canonical comparison of all five files.  usage: PYTHONHASHSEED=0 tools/fuzz_oracle_phasing.py [iterations=60] [seed=500]"""
import os, random, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import make_golden as mg
import phasing_oracle as po
from helpers import OUTPUTS, canonical
from phaser_amd import synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
phaser, rvm = mg.build_reference()
subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
CONTIGS = [("chr3", 198295559), ("chr11", 135086622), ("chr19", 58617616)]
bad_total = 0
for it in range(iters):
    rng = random.Random(seed0 + it)
    nchrom = rng.choice([1, 1, 2, 3]); nbam = rng.choice([1, 1, 2, 3])
    contigs = CONTIGS[:nchrom]
    err = rng.choice([0.001, 0.002, 0.02, 0.05, 0.08])
    okw = {"max_block_size": rng.choice([3, 5, 8, 15])}
    if rng.random() < 0.3: okw["unphased_vars"] = 0
    if rng.random() < 0.3: okw["unique_ids"] = 1
    if rng.random() < 0.3: okw["cc_threshold"] = rng.choice([0.001, 0.05, 0.2])
    if rng.random() < 0.3: okw["as_q_cutoff"] = rng.choice([0.2, 0.5])
    if rng.random() < 0.2: okw["output_read_ids"] = 1
    if rng.random() < 0.25: okw["gw_phase_method"] = 1          # MAF-weighted genome-wide phase (phaser.py:982-1025; the synthetic VCF carries AF=)
    sparse = rng.random() < 0.3                                    # some BAMs have no read at all on some chromosomes (read_vars keys follow the first BAM with a kept line)
    excl = [rng.randrange(nbam)] if nbam > 1 and rng.random() < 0.3 else []
    vs_ = []; names = ["x%d.bam" % b for b in range(nbam)]
    sams = {b: {} for b in names}
    for ci, (chrom, ln) in enumerate(contigs):
        dense = rng.random() < float(os.environ.get("PHZ_FUZZ_DENSE", "0"))      # off by default: the reference needs minutes to hours on dense shapes (tools/pin_dense_vs_reference.py)
        v, gs, ge, w = synth.make_variants(chrom, 1, rng.choice([600_000, 1_500_000]), rng.choice([600, 1500]) if dense else rng.choice([60, 150, 260]),
                                           seed0 * 7 + 13 * it + ci, n_genes=rng.choice([2, 4]) if dense else rng.choice([4, 10]))
        vs_.append(v)
        L = rng.choice([76, 150, 600]) if dense else 76
        for bi, bam in enumerate(names):
            rb = synth.make_reads(v, gs, ge, w, rng.choice([200, 500]) if (dense and L > 150) else rng.choice([1500, 4000]), seed0 * 11 + 17 * it + 10 * ci + bi, L=L,
                                  qname_prefix="q" if rng.random() < 0.7 else "q%d." % bi, err_rate=err)
            rf = rb.select(synth.samtools_keep(rb, 255))
            sams[bam][chrom] = "" if (sparse and rng.random() < 0.4 and not (ci == len(contigs) - 1 and bi == nbam - 1)) else "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf_text = "\n".join(synth.vcf_lines(vs_)) + "\n"
    refkw = dict(okw)
    if excl:
        refkw["haplo_count_bam_exclude"] = ",".join(str(x + 1) for x in excl)
    with tempfile.TemporaryDirectory() as tmp:
        want, calls = mg.run_pipeline(phaser, rvm, vcf_text, sams, tmp, **refkw)
    ph = po.Phaser(po.bam_display_names(names), haplo_count_bam_exclude=excl, **okw)
    pool, _, _ = po.load_vcf(vcf_text)
    for bam in names:
        ph.add_bam([calls[(bam, c)] for c in pool])      # the reference's own call files: this fuzz isolates the phasing core
    got = ph.finish()
    bad = [n for n in OUTPUTS if canonical(n, got[n]) != canonical(n, want[n])]
    bad_total += bool(bad)
    print("iter %d: chroms %d bams %d err %.3f %s excl %s%s -> %s" % (it, nchrom, nbam, err, okw, excl, " sparse" if sparse else "", "OK" if not bad else "DIFF " + str(bad)), flush=True)
print("%d iterations, %d with differences" % (iters, bad_total))
