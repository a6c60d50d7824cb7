#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING the reference (phASER v1.2.0) in this container.

Only works where /root/reference exists (the build container).  The reference's
Python is imported from a scratch copy under /tmp (its mapper compiled with the
reference's own setup.py, exactly as phaser/README.md:20-25 prescribes); nothing
of it is copied into the repository -- only inputs we generate ourselves and the
outputs the reference computes for them land in tests/golden/.

Recipe validated in SURVEY.md Appendix A:
  * stub `pysam` (imported at phaser.py:15, never used)
  * set phaser.args / devnull / haplo_count_bam_exclude / sample_column
  * replace call_mapping_script (phaser.py:1330-1353, the samtools pipeline) with a
    function that feeds pre-filtered SAM text to do_read_variant_map
  * call process_vcf (phaser.py:378) with write_vcf=0

Run:  PYTHONHASHSEED=0 python tools/make_golden.py [--only NAME]
"""
import argparse
import gzip
import hashlib
import io
import random
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/phaser"
GOLD = os.path.join(REPO, "tests", "golden")
BUILD = "/tmp/phz_ref_build"

if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

sys.path.insert(0, REPO)


def build_reference():
    if not os.path.isdir(REF):
        sys.exit("reference not present; golden fixtures can only be regenerated in the build container")
    os.makedirs(BUILD, exist_ok=True)
    for f in os.listdir(REF):
        if f.endswith(".py"):
            shutil.copy(os.path.join(REF, f), BUILD)
    so = [f for f in os.listdir(BUILD) if f.startswith("read_variant_map") and f.endswith(".so")]
    if not so:
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=BUILD,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, BUILD)
    sys.modules["pysam"] = types.ModuleType("pysam")
    import phaser  # noqa
    import read_variant_map  # noqa
    return phaser, read_variant_map


def default_args(**kw):
    ns = argparse.Namespace(
        bam="a.bam", vcf="in.vcf", sample="S1", mapq="255", baseq=10, paired_end="1", o="out",
        python_string="python3", haplo_count_bam_exclude="", haplo_count_blacklist="", cc_threshold=0.01,
        isize="0", as_q_cutoff=0.05, blacklist="", write_vcf=0, include_indels=0, output_read_ids=0,
        remove_dups=1, pass_only=1, unphased_vars=1, chr_prefix="", gw_phase_method=0, gw_af_field="AF",
        gw_phase_vcf=0, gw_phase_vcf_min_confidence=0.9, threads=1, max_block_size=15, temp_dir="",
        max_items_per_thread=100000, show_warning=0, debug=0, chr="", unique_ids=0, id_separator="_",
        output_network="", process_slow=0)
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def run_mapper(rvm, sam_text, table_text, baseq=10, isize=0.0):
    """Drive do_read_variant_map (read_variant_map.py:3) on SAM text; returns the call TSV text."""
    d = tempfile.mkdtemp()
    tp = os.path.join(d, "table.tsv"); op = os.path.join(d, "out.tsv")
    open(tp, "w").write(table_text)
    old_in, old_out = sys.stdin, sys.stdout
    sys.stdin = io.StringIO(sam_text); sys.stdout = io.StringIO()
    try:
        rvm.do_read_variant_map(tp, baseq, op, 1, isize)
    finally:
        sys.stdin, sys.stdout = old_in, old_out
    out = open(op).read()
    shutil.rmtree(d)
    return out


def run_pipeline(phaser, rvm, vcf_text, sams, outdir, capture_calls=True, haplo_blacklist=None, **argkw):
    """Drive process_vcf (phaser.py:378) end to end. `sams` = ordered {bam_path: sam_text}."""
    ns = default_args(bam=",".join(sams.keys()), **argkw)
    phaser.args = ns
    phaser.devnull = open(os.devnull, "w")
    if ns.haplo_count_bam_exclude != "":
        phaser.haplo_count_bam_exclude = [x - 1 for x in map(int, ns.haplo_count_bam_exclude.split(","))]
    else:
        phaser.haplo_count_bam_exclude = []
    phaser.sample_column = 9
    work = tempfile.mkdtemp()
    calls = {}

    def fake_call_mapping_script(inp):
        chrom, bed, table, samtools_arg, bam, mapq, isize = inp
        out = phaser.new_temp_file()
        old_in, old_out = sys.stdin, sys.stdout
        sys.stdin = io.StringIO(sams[bam][chrom]); sys.stdout = io.StringIO()
        try:
            rvm.do_read_variant_map(table, ns.baseq, out, 1, isize)
        finally:
            sys.stdin, sys.stdout = old_in, old_out
        if capture_calls:
            calls[(bam, chrom)] = open(out).read()
            calls[("table", chrom)] = open(table).read()
        return out

    phaser.call_mapping_script = fake_call_mapping_script
    vp = os.path.join(work, "in.vcf")
    open(vp, "w").write(vcf_text)
    if ns.write_vcf == 1:
        # write_vcf (phaser.py:1661) re-reads the ORIGINAL gzipped VCF through `gunzip -c | cut`; its last step shells out
        # to bgzip/tabix, which are not installed here: the plain out.vcf it wrote before that is what we keep.
        with gzip.open(vp + ".gz", "wt") as f:
            f.write(vcf_text)
        ns.vcf = vp + ".gz"
        phaser.csi_index = 0
    vcf_tmp = tempfile.NamedTemporaryFile(delete=False); vcf_tmp.close()
    prefix = os.path.join(work, "out")
    old_out = sys.stdout
    log = io.StringIO()
    sys.stdout = log
    try:
        phaser.process_vcf(open(vp), "", [ns.id_separator, ":"], haplo_blacklist or set(), time.time(), vcf_tmp, prefix, True, 0)
    except subprocess.CalledProcessError as e:
        if not (ns.write_vcf == 1 and "bgzip" in str(e.cmd)):
            raise
    finally:
        sys.stdout = old_out
    os.makedirs(outdir, exist_ok=True)
    res = {}
    for suf in ["allelic_counts", "variant_connections", "haplotypes", "haplotypic_counts", "allele_config"]:
        res[suf] = open(prefix + "." + suf + ".txt").read()
    res["log"] = log.getvalue()
    if ns.write_vcf == 1:
        res["vcf"] = open(prefix + ".vcf").read()
    shutil.rmtree(work)
    return res, calls


def wgz(path, text):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(text.encode())


def sha(text):
    return hashlib.sha256(text.encode()).hexdigest()


# --------------------------------------------------------------------------------------------- fixtures

def fx_kat(phaser, rvm):
    """Micro known-answer tests on split_read / identify_allele (read_variant_map.py:165-258)."""
    I = "I"
    cases = [
        ("plain", 100, "ACGTACGTAC", I * 10, "10M", [(p, "A,C", 1) for p in (99, 100, 109, 110)]),
        ("low_baseq", 100, "ACGTACGTAC", "II#IIIIIII", "10M", [(p, "A,C", 1) for p in (101, 102, 103)]),
        ("deletion", 100, "ACGTACGTAC", I * 10, "4M2D6M", [(p, "A,C", 1) for p in (103, 104, 105, 106)]),
        ("ins_after_snp", 100, "ACGTTTACGT", I * 10, "4M2I4M", [(p, "A,C", 1) for p in (102, 103, 104)]),
        ("clip_splice", 100, "NNACGTACGTAC", I * 12, "2S4M10N6M", [(p, "A,C", 1) for p in list(range(99, 122))]),
        ("ins_seg2_dropped", 100, "ACGTACGTTTT", I * 11, "4M10N3M1I3M", [(p, "A,C", 1) for p in range(113, 121)]),
        ("ins_seg2_misplaced", 100, "ACGTCCCCCCCCC", I * 13, "1M1N3M1I8M", [(p, "A,C", 1) for p in range(100, 112)]),
        ("hard_eq_x", 100, "ACGTAC", I * 6, "3H2=2X2M5H", [(p, "A,C", 1) for p in range(99, 107)]),
        ("indel_vars", 100, "ACGTACGTAC", I * 10, "10M", [(103, "TA,T", 2), (108, "AC,A", 2), (109, "CG,C", 2)]),
        ("seq_star", 100, "*", "*", "10M", [(p, "A,C", 1) for p in range(99, 111)]),
        ("del_then_ins", 100, "ACGTGACGT", I * 9, "4M2D1I4M", [(p, "A,C", 1) for p in range(99, 112)]),
        ("ins_lowq", 100, "ACGTTTACGT", "IIII##IIII", "4M2I4M", [(p, "A,C", 1) for p in (102, 103, 104)]),
        ("ins_one_lowq", 100, "ACGTTTACGT", "IIII#IIIII", "4M2I4M", [(p, "A,C", 1) for p in (102, 103, 104)]),
        ("snp_lowq_ins", 100, "ACGTTTACGT", "III#IIIIII", "4M2I4M", [(p, "A,C", 1) for p in (102, 103, 104)]),
        ("ins_at_start", 100, "TTACGTACGT", I * 10, "2I8M", [(p, "A,C", 1) for p in range(99, 109)]),
        ("ins_after_clip", 100, "GGTTACGTAC", I * 10, "2S2I6M", [(p, "A,C", 1) for p in range(99, 107)]),
        ("two_ins_same_key", 100, "ACGTTTGGAC", I * 10, "4M2I2I2M", [(p, "A,C", 1) for p in range(100, 107)]),
        ("ins_right_after_N", 100, "ACGTTTACGT", I * 10, "4M5N2I4M", [(p, "A,C", 1) for p in range(100, 114)]),
        ("ins_seg2_key_in_range", 100, "ACGTAAAAAAAAAATTCCCC", I * 20, "2M2N12M2I4M",
         [(p, "A,C", 1) for p in range(100, 124)]),
        ("pad_op", 100, "ACGTACGTAC", I * 10, "4M2P6M", [(p, "A,C", 1) for p in range(99, 111)]),
        ("short_qual", 100, "ACGTACGTAC", "IIII", "10M", [(p, "A,C", 1) for p in range(99, 111)]),
        ("short_seq_two_M", 100, "ACGTAC", I * 6, "4M3N4M", [(p, "A,C", 1) for p in range(99, 112)]),
        ("iupac", 100, "ACRTDC=TAC", I * 10, "10M", [(p, "A,C", 1) for p in range(99, 111)]),
        ("dup_pos_vars", 100, "ACGTACGTAC", I * 10, "10M", [(104, "A,C", 1), (104, "A,G", 1), (105, "C,T", 1)]),
        ("baseq0", 100, "ACGNACGTAC", "!!!!!!!!!!", "10M", [(p, "A,C", 1) for p in range(100, 110)]),
        ("all_deleted_var", 100, "ACGTAC", I * 6, "2M3D4M", [(101, "CGT,C", 3), (102, "GT,G", 2), (103, "T,A", 1)]),
        ("ins_in_multibase_var", 100, "ACGTTTACGT", I * 10, "4M2I4M", [(102, "GT,G", 2), (103, "TA,T", 2), (101, "CGTA,C", 4)]),
    ]
    out = []
    for name, pos, seq, qual, cigar, vars_ in cases:
        for baseq in ([10] if name != "baseq0" else [0, 10]):
            rvm.args = {"baseq": baseq, "splice": 1}
            segs = rvm.split_read(pos, seq, qual, cigar, "r")
            res = []
            for (vp, alleles, reflen) in vars_:
                v = rvm.variant(["1", str(vp), "1_%d_x" % vp, ".", alleles, str(reflen), "0|1", "None"])
                per_seg = [rvm.identify_allele(s, pos, v) for s in segs]
                res.append({"pos": vp, "alleles": alleles, "ref_len": reflen, "per_segment": per_seg})
            out.append({"name": name, "pos": pos, "seq": seq, "qual": qual, "cigar": cigar, "baseq": baseq,
                        "segments": [[s.read_start, s.read_stop, s.pseudo_read,
                                      {str(k): v for k, v in s.insertions.items()}] for s in segs],
                        "variants": res})
    json.dump(out, open(os.path.join(GOLD, "kat_micro.json"), "w"), indent=1)
    print("kat_micro: %d cases" % len(out))


def table_text(v, include_maf="None"):
    from phaser_amd import synth
    rows = []
    pos = v.pos.tolist(); ref = v.ref.tolist(); alt = v.alt.tolist()
    for i in range(len(v)):
        r, a = synth.BASES[ref[i]], synth.BASES[alt[i]]
        uid = "%s_%d_%s_%s" % (v.chrom, pos[i], r, a)
        rows.append("\t".join([v.chrom, str(pos[i]), uid, v.rsid[i], r + "," + a, "1", v.gt[i], include_maf]))
    return "\n".join(rows) + "\n"


def dataset(name, chrom, start, end, n_snps, n_pairs, seed, n_genes=None, prefix="s0.b0.r", L=76, vseed=None):
    from phaser_amd import synth
    v, gs, ge, w = synth.make_variants(chrom, start, end, n_snps, seed if vseed is None else vseed, n_genes=n_genes)
    rb = synth.make_reads(v, gs, ge, w, n_pairs, seed + 1, L=L, qname_prefix=prefix)
    rf = rb.select(synth.samtools_keep(rb, 255))
    return v, rf, (v, gs, ge, w)


def fx_mapper_small(phaser, rvm):
    from phaser_amd import synth
    d = os.path.join(GOLD, "mapper_small"); os.makedirs(d, exist_ok=True)
    v, rf, _ = dataset("mapper_small", "chr22", 1, 2_000_000, 150, 4000, 101, n_genes=12)
    sam = "\n".join(synth.sam_lines(rf, [("chr22", 50818468)])) + "\n"
    tab = table_text(v)
    meta = {"records": len(rf), "variants": len(v), "runs": []}
    wgz(os.path.join(d, "in.sam.gz"), sam); open(os.path.join(d, "table.tsv"), "w").write(tab)
    for baseq, isize in [(10, 0.0), (30, 0.0), (10, 260.0), (0, 0.0)]:
        out = run_mapper(rvm, sam, tab, baseq, isize)
        fn = "calls_bq%d_is%d.tsv.gz" % (baseq, int(isize))
        wgz(os.path.join(d, fn), out)
        meta["runs"].append({"baseq": baseq, "isize": isize, "file": fn, "lines": out.count("\n")})
    json.dump(meta, open(os.path.join(d, "meta.json"), "w"), indent=1)
    print("mapper_small:", meta)


def fx_mapper_unsorted(phaser, rvm):
    """A stream whose records are OUT of coordinate order: the reference's variant buffer is forward-only (read_variant_map.py:37-50 prunes
    what lies behind the current record -- for every record, also one the isize filter drops --, :88-93 and :106-112 never rewind the
    variant stream), so a record that steps backwards misses the variants already dropped.  Same records and table as mapper_small."""
    d = os.path.join(GOLD, "mapper_unsorted"); os.makedirs(d, exist_ok=True)
    src = os.path.join(GOLD, "mapper_small")
    sam = gzip.open(os.path.join(src, "in.sam.gz"), "rt").read()
    tab = open(os.path.join(src, "table.tsv")).read()
    head = [l for l in sam.split("\n") if l.startswith("@")]
    recs = [l for l in sam.split("\n") if l and not l.startswith("@")]
    rng = random.Random(5)
    shuffled = list(recs)
    for _ in range(len(recs) // 3):                            # local disorder ...
        i = rng.randrange(len(shuffled) - 1)
        shuffled[i], shuffled[i + 1] = shuffled[i + 1], shuffled[i]
    for _ in range(5):                                         # ... and a few long-range moves to the end of the stream
        shuffled.append(shuffled.pop(rng.randrange(len(shuffled) // 2)))
    far = list(shuffled)
    for _ in range(6):                                         # records moved far FORWARD: what follows them steps backwards, and the
        i = rng.randrange(len(far) // 8, len(far))             # variants they make the mapper skip are gone for good
        far.insert(max(0, i - rng.randrange(200, 700)), far.pop(i))
    meta = {"records": len(shuffled), "runs": []}
    for name, stream in (("local", shuffled), ("far", far)):
        text = "\n".join(head + stream) + "\n"
        wgz(os.path.join(d, "in_%s.sam.gz" % name), text)
        for baseq, isize in [(10, 0.0), (10, 260.0)]:
            out = run_mapper(rvm, text, tab, baseq, isize)
            fn = "calls_%s_bq%d_is%d.tsv.gz" % (name, baseq, int(isize))
            wgz(os.path.join(d, fn), out)
            meta["runs"].append({"stream": name, "baseq": baseq, "isize": isize, "file": fn, "lines": out.count("\n")})
    json.dump(meta, open(os.path.join(d, "meta.json"), "w"), indent=1)
    print("mapper_unsorted:", meta)


def split_by_chrom(sam_lists):
    return sam_lists


def fx_pipeline(phaser, rvm):
    from phaser_amd import synth
    contigs = [("chr21", 46709983), ("chr22", 50818468)]
    # --- case A: one chromosome, one BAM
    d = os.path.join(GOLD, "pipe_one"); os.makedirs(d, exist_ok=True)
    v, rf, _ = dataset("pipe_one", "chr22", 1, 3_000_000, 300, 9000, 201, n_genes=20)
    sam = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf = "\n".join(synth.vcf_lines([v])) + "\n"
    res, calls = run_pipeline(phaser, rvm, vcf, {"a.bam": {"chr22": sam}}, d)
    open(os.path.join(d, "in.vcf"), "w").write(vcf)
    wgz(os.path.join(d, "a.chr22.sam.gz"), sam)
    for k, t in res.items():
        wgz(os.path.join(d, "out." + k + ".txt.gz"), t)
    wgz(os.path.join(d, "calls.a.chr22.tsv.gz"), calls[("a.bam", "chr22")])
    print("pipe_one: records=%d" % len(rf), [l for l in res["log"].splitlines() if "PHASED" in l or "noise" in l or "cutoff" in l])

    # --- case B: two chromosomes, two BAMs sharing QNAMEs (T3 overwrite quirk), BAM 2 has other seeds
    d = os.path.join(GOLD, "pipe_two"); os.makedirs(d, exist_ok=True)
    vs = []; sams = {"t1.bam": {}, "t2.bam": {}}
    for ci, (chrom, ln) in enumerate(contigs):
        v, gs, ge, w = synth.make_variants(chrom, 1, 2_000_000, 160, 300 + ci, n_genes=12)
        vs.append(v)
        for bi, bam in enumerate(sams):
            # same qname prefix in both BAMs => QNAME collisions across BAMs on purpose
            rb = synth.make_reads(v, gs, ge, w, 3000, 310 + 10 * ci + bi, qname_prefix="s0.r")
            rf = rb.select(synth.samtools_keep(rb, 255))
            sams[bam][chrom] = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf = "\n".join(synth.vcf_lines(vs)) + "\n"
    res, calls = run_pipeline(phaser, rvm, vcf, sams, d)
    open(os.path.join(d, "in.vcf"), "w").write(vcf)
    for bam in sams:
        for chrom in sams[bam]:
            wgz(os.path.join(d, "%s.%s.sam.gz" % (bam.replace(".bam", ""), chrom)), sams[bam][chrom])
    for k, t in res.items():
        wgz(os.path.join(d, "out." + k + ".txt.gz"), t)
    print("pipe_two:", [l for l in res["log"].splitlines() if "PHASED" in l or "noise" in l or "cutoff" in l])

    # --- case C: noisy data (high error) to force conflicting blocks through split / brute-force / stitch paths
    for tag, err, mbs, seed in [("a", 0.03, 6, 401), ("b", 0.05, 4, 501), ("c", 0.08, 15, 601)]:
        d = os.path.join(GOLD, "pipe_noisy_" + tag); os.makedirs(d, exist_ok=True)
        v, gs, ge, w = synth.make_variants("chr22", 1, 400_000, 260, seed, n_genes=6)
        rb = synth.make_reads(v, gs, ge, w, 9000, seed + 1, err_rate=err)
        rf = rb.select(synth.samtools_keep(rb, 255))
        sam = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
        vcf = "\n".join(synth.vcf_lines([v])) + "\n"
        res, calls = run_pipeline(phaser, rvm, vcf, {"n.bam": {"chr22": sam}}, d, max_block_size=mbs)
        open(os.path.join(d, "in.vcf"), "w").write(vcf)
        wgz(os.path.join(d, "n.chr22.sam.gz"), sam)
        json.dump({"max_block_size": mbs, "err_rate": err}, open(os.path.join(d, "meta.json"), "w"))
        for k, t in res.items():
            wgz(os.path.join(d, "out." + k + ".txt.gz"), t)
        print("pipe_noisy_" + tag, [l for l in res["log"].splitlines() if "PHASED" in l or "noise" in l or "dropped" in l])


def fx_sparse(phaser, rvm):
    """Round 5 (found by tools/stress_parity.py): the chromosome order of the block files when the FIRST BAM has no kept line on a chromosome.  read_vars is keyed
    by the chromosome process_mapping_result returns -- "" for a call file without kept lines (phaser.py:1299) -- and later BAMs append their new keys (:573-574),
    so the blocks of such a chromosome come AFTER those of the chromosomes the first BAM covers, whatever the VCF order.  Three chromosomes (VCF order chr3,
    chr11, chr19), three BAMs: BAM 1 has reads on chr11 only, BAM 2 on chr19 and chr3, BAM 3 on all; --write_vcf numbers the blocks (PI) in that order too."""
    from phaser_amd import synth
    contigs = [("chr3", 198295559), ("chr11", 135086622), ("chr19", 58617616)]
    d = os.path.join(GOLD, "pipe_sparse"); os.makedirs(d, exist_ok=True)
    vs = []; sams = {"s1.bam": {}, "s2.bam": {}, "s3.bam": {}}
    covers = {"s1.bam": {"chr11"}, "s2.bam": {"chr19", "chr3"}, "s3.bam": {"chr3", "chr11", "chr19"}}
    for ci, (chrom, ln) in enumerate(contigs):
        v, gs, ge, w = synth.make_variants(chrom, 1, 1_000_000, 120, 700 + ci, n_genes=8)
        vs.append(v)
        for bi, bam in enumerate(sams):
            rb = synth.make_reads(v, gs, ge, w, 2500, 710 + 10 * ci + bi, qname_prefix="s0.r")
            rf = rb.select(synth.samtools_keep(rb, 255))
            lines = list(synth.sam_lines(rf, contigs))
            if chrom not in covers[bam]:
                lines = [l for l in lines if l.startswith("@")]          # what `samtools view -h BAM chrom:` prints for a chromosome without reads: the header
            sams[bam][chrom] = "\n".join(lines) + "\n"
    vcf = "\n".join(synth.vcf_lines(vs)) + "\n"
    res, calls = run_pipeline(phaser, rvm, vcf, sams, d, write_vcf=1)
    open(os.path.join(d, "in.vcf"), "w").write(vcf)
    for bam in sams:
        for chrom in sams[bam]:
            wgz(os.path.join(d, "%s.%s.sam.gz" % (bam.replace(".bam", ""), chrom)), sams[bam][chrom])
    for k, t in res.items():
        wgz(os.path.join(d, "out." + k + ".txt.gz"), t)
    first = [l.split("\t")[0].split("_")[0] for l in res["allele_config"].split("\n")[1:] if l]
    order = [c for i, c in enumerate(first) if i == 0 or first[i - 1] != c]
    print("pipe_sparse: chromosome order of allele_config:", order, [l for l in res["log"].splitlines() if "PHASED" in l])
    assert order == ["chr11", "chr3", "chr19"], order          # chr11 with BAM 1; chr3 and chr19 with BAM 2, in VCF order


def fx_c1(phaser, rvm):
    """BASELINE.json configs[0]: chr22:1-20Mb, 1k het SNPs, 100k records.  Inputs are regenerated from the
    seed by tests (too large to commit); expected outputs are committed gz + sha256 of the inputs."""
    from phaser_amd import synth
    d = os.path.join(GOLD, "c1"); os.makedirs(d, exist_ok=True)
    contigs = [("chr22", 50818468)]
    v, gs, ge, w = synth.make_variants("chr22", 1, 20_000_000, 1000, 20240807, n_genes=100)
    rb = synth.make_reads(v, gs, ge, w, 68000, 20240808)
    rf = rb.select(synth.samtools_keep(rb, 255))
    sam = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf = "\n".join(synth.vcf_lines([v])) + "\n"
    t0 = time.time()
    calls_txt = run_mapper(rvm, sam, table_text(v), 10, 0.0)
    t_map = time.time() - t0
    t0 = time.time()
    res, calls = run_pipeline(phaser, rvm, vcf, {"c1.bam": {"chr22": sam}}, d, capture_calls=False)
    t_all = time.time() - t0
    for k, t in res.items():
        wgz(os.path.join(d, "out." + k + ".txt.gz"), t)
    wgz(os.path.join(d, "calls.tsv.gz"), calls_txt)
    meta = {"records": len(rf), "variants": len(v), "sam_sha256": sha(sam), "vcf_sha256": sha(vcf),
            "call_lines": calls_txt.count("\n"), "ref_mapper_seconds": round(t_map, 3),
            "ref_pipeline_seconds": round(t_all, 3),
            "gen": {"region": ["chr22", 1, 20000000], "n_snps": 1000, "n_genes": 100, "vseed": 20240807,
                    "n_pairs": 68000, "rseed": 20240808, "mapq": 255}}
    json.dump(meta, open(os.path.join(d, "meta.json"), "w"), indent=1)
    print("c1:", meta, [l for l in res["log"].splitlines() if "PHASED" in l])


def fx_indels(phaser, rvm):
    """--include_indels 1: deletion / insertion variants next to SNPs, reads carrying I / D / N ops (general ref_len path,
    read_variant_map.py:236-258 with ref_len > 1 and multi-base alleles)."""
    from phaser_amd import synth
    contigs = [("chr21", 46709983), ("chr22", 50818468)]
    d = os.path.join(GOLD, "pipe_indel"); os.makedirs(d, exist_ok=True)
    v, gs, ge, w = synth.make_variants("chr22", 1, 1_500_000, 320, 701, n_genes=10, indel_frac=0.3)
    rb = synth.make_reads(v, gs, ge, w, 9000, 702, err_rate=0.004)
    rf = rb.select(synth.samtools_keep(rb, 255))
    sam = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf = "\n".join(synth.vcf_lines([v])) + "\n"
    res, calls = run_pipeline(phaser, rvm, vcf, {"i.bam": {"chr22": sam}}, d, include_indels=1)
    open(os.path.join(d, "in.vcf"), "w").write(vcf)
    wgz(os.path.join(d, "i.chr22.sam.gz"), sam)
    wgz(os.path.join(d, "calls.i.chr22.tsv.gz"), calls[("i.bam", "chr22")])
    open(os.path.join(d, "table.chr22.tsv"), "w").write(calls[("table", "chr22")])
    for k, t in res.items():
        wgz(os.path.join(d, "out." + k + ".txt.gz"), t)
    txt = calls[("i.bam", "chr22")]
    multi = sum(1 for l in txt.splitlines() if len(l.split("\t")[3]) > 1)
    print("pipe_indel: records=%d call lines=%d (multi-base texts %d)" % (len(rf), txt.count("\n"), multi),
          [l for l in res["log"].splitlines() if "PHASED" in l or "heterozygous" in l])


OPT_CASES = {
    "gw_maf": dict(gw_phase_method=1),
    "no_unphased": dict(unphased_vars=0),
    "read_ids": dict(output_read_ids=1),
    "bam_exclude": dict(haplo_count_bam_exclude="2"),
    "unique_ids": dict(unique_ids=1),
    "thresholds": dict(cc_threshold=0.5, as_q_cutoff=0, baseq=30),
    "as_q20": dict(as_q_cutoff=0.2),
    "isize": dict(isize="260"),
    "separator": dict(id_separator="~"),
    "blacklist": dict(),          # haplotypic-count blacklist passed as a set (see below)
}


def opts_inputs():
    from phaser_amd import synth
    contigs = [("chr21", 46709983), ("chr22", 50818468)]
    vs = []; sams = {"o1.bam": {}, "o2.bam": {}}
    for ci, (chrom, ln) in enumerate(contigs):
        v, gs, ge, w = synth.make_variants(chrom, 1, 800_000, 90, 880 + ci, n_genes=5)
        vs.append(v)
        for bi, bam in enumerate(sams):
            rb = synth.make_reads(v, gs, ge, w, 1400, 890 + 10 * ci + bi, qname_prefix="s0.r", err_rate=0.01)
            rf = rb.select(synth.samtools_keep(rb, 255))
            sams[bam][chrom] = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    return vs, sams, "\n".join(synth.vcf_lines(vs)) + "\n"


def fx_options(phaser, rvm):
    """The flags that reach the hot path (phaser.py:30-78), one reference run each on a small 2-chromosome x 2-BAM sample."""
    d0 = os.path.join(GOLD, "pipe_opts"); os.makedirs(d0, exist_ok=True)
    vs, sams, vcf = opts_inputs()
    open(os.path.join(d0, "in.vcf"), "w").write(vcf)
    for bam in sams:
        for chrom in sams[bam]:
            wgz(os.path.join(d0, "%s.%s.sam.gz" % (bam.replace(".bam", ""), chrom)), sams[bam][chrom])
    bl = set("%s_%d" % (v.chrom, int(p)) for v in vs for p in v.pos.tolist()[::7])
    json.dump({"cases": {k: {kk: vv for kk, vv in kw.items()} for k, kw in OPT_CASES.items()}, "blacklist": sorted(bl)},
              open(os.path.join(d0, "cases.json"), "w"), indent=1)
    for name, kw in OPT_CASES.items():
        d = os.path.join(d0, name); os.makedirs(d, exist_ok=True)
        kw = dict(kw)
        if name == "isize":
            # the insert-size filter lives in the mapper call (phaser.py:1346 --isize_cutoff); run_pipeline passes it through
            pass
        res, _ = run_pipeline(phaser, rvm, vcf, sams, d, capture_calls=False, haplo_blacklist=(bl if name == "blacklist" else set()), **kw)
        for k, t in res.items():
            wgz(os.path.join(d, "out." + k + ".txt.gz"), t)
        print("pipe_opts/%s" % name, [l.strip() for l in res["log"].splitlines() if "PHASED" in l])


def bed_overlaps(iv, chrom, pos1, ref_len):
    """bedtools semantics, brute force: a VCF record is the 0-based interval [POS-1, POS-1+len(REF)); hit = >= 1 shared base."""
    s, e = pos1 - 1, pos1 - 1 + max(1, ref_len)
    return any(c == chrom and a < e and s < b for c, a, b in iv)


def fx_blacklist_bed(phaser, rvm):
    """--blacklist / --haplo_count_blacklist (phaser.py:220-243).  bedtools is not installed, so the reference cannot run its own
    BED step here; what it does with the result is pinned instead: process_vcf is run on the VCF with the lines `bedtools intersect -v`
    would remove already removed (brute-force overlap above), and with the haplotype-count blacklist set `bedtools intersect`
    would yield (same brute force over the remaining lines).  The product is then fed the FULL VCF plus the two BED files."""
    d0 = os.path.join(GOLD, "pipe_bed"); os.makedirs(d0, exist_ok=True)
    vs, sams, vcf = opts_inputs()
    rng = random.Random(4242)
    drop = []; mark = []
    for v in vs:
        pos = v.pos.tolist()
        for k in range(6):                      # blocks of neighbouring variants, HLA-style
            i = rng.randrange(0, len(pos) - 4)
            drop.append((v.chrom, pos[i] - 1 - rng.randrange(0, 200), pos[i + rng.randrange(0, 3)] + rng.randrange(0, 200)))
        j = rng.randrange(0, len(pos))
        drop.append((v.chrom, pos[j] - 1, pos[j]))                      # exactly one base: the variant itself
        drop.append((v.chrom, pos[(j + 5) % len(pos)], pos[(j + 5) % len(pos)] + 40))      # starts right AFTER a variant: no hit
        drop.append((v.chrom, max(0, pos[(j + 9) % len(pos)] - 41), pos[(j + 9) % len(pos)] - 1))  # ends right BEFORE a variant: no hit
        for k in range(10):
            i = rng.randrange(0, len(pos) - 2)
            mark.append((v.chrom, pos[i] - 1 - rng.randrange(0, 50), pos[i + rng.randrange(0, 2)] + rng.randrange(0, 50)))
        mark.append((v.chrom, pos[3], pos[3] + 1))                      # the base after a variant: no hit
    drop.append(("chrUn_other", 0, 10 ** 9)); mark.append(("chr_absent", 5, 50))
    rng.shuffle(drop); rng.shuffle(mark)
    open(os.path.join(d0, "blacklist.bed"), "w").write("track name=test\n" + "".join("%s\t%d\t%d\n" % x for x in drop))
    open(os.path.join(d0, "haplo_blacklist.bed"), "w").write("".join("%s\t%d\t%d\tx\n" % x for x in mark))
    kept = []; hb = set(); n_drop = 0
    for line in vcf.split("\n"):
        if not line or line[0] == "#":
            kept.append(line); continue
        c = line.split("\t")
        if bed_overlaps(drop, c[0], int(c[1]), len(c[3])):
            n_drop += 1
            continue
        kept.append(line)
        if bed_overlaps(mark, c[0], int(c[1]), len(c[3])):
            hb.add(c[0] + "_" + str(int(c[1])))
    res, _ = run_pipeline(phaser, rvm, "\n".join(kept), sams, d0, capture_calls=False, haplo_blacklist=hb)
    for k, t in res.items():
        wgz(os.path.join(d0, "out." + k + ".txt.gz"), t)
    json.dump({"dropped_lines": n_drop, "haplo_blacklist": sorted(hb)}, open(os.path.join(d0, "meta.json"), "w"), indent=1)
    print("pipe_bed: %d VCF lines dropped, %d variants on the haplotype-count blacklist" % (n_drop, len(hb)),
          [l.strip() for l in res["log"].splitlines() if "PHASED" in l])


def fx_write_vcf(phaser, rvm):
    """Phased VCF text (write_vcf, phaser.py:1661-1855) for pipe_one / pipe_noisy_c inputs under the three --gw_phase_vcf modes."""
    for src, mbs in [("pipe_one", 15), ("pipe_noisy_c", 15), ("pipe_two", 15)]:
        d = os.path.join(GOLD, src)
        vcf = open(os.path.join(d, "in.vcf")).read()
        if src == "pipe_two":
            sams = {b + ".bam": {c: gzip.open(os.path.join(d, "%s.%s.sam.gz" % (b, c)), "rt").read() for c in ("chr21", "chr22")} for b in ("t1", "t2")}
        else:
            bam = "a" if src == "pipe_one" else "n"
            sams = {bam + ".bam": {"chr22": gzip.open(os.path.join(d, bam + ".chr22.sam.gz"), "rt").read()}}
        for mode in (0, 1, 2):
            res, _ = run_pipeline(phaser, rvm, vcf, sams, d, capture_calls=False, write_vcf=1, gw_phase_vcf=mode, max_block_size=mbs,
                                  gw_phase_vcf_min_confidence=0.9)
            wgz(os.path.join(d, "out.vcf_gw%d.txt.gz" % mode), res["vcf"])
        print("write_vcf", src, len(res["vcf"].splitlines()), "lines")


def fx_write_vcf_more(phaser, rvm):
    """More phased-VCF goldens: noisy inputs (low-confidence blocks), the MAF-weighted genome-wide phase (--gw_phase_method 1),
    an odd id separator and unique ids, with --gw_phase_vcf 1 / 2 at lower --gw_phase_vcf_min_confidence thresholds."""
    jobs = []
    for tag in ("a", "b"):
        d = os.path.join(GOLD, "pipe_noisy_" + tag)
        mbs = json.load(open(os.path.join(d, "meta.json")))["max_block_size"]
        sams = {"n.bam": {"chr22": gzip.open(os.path.join(d, "n.chr22.sam.gz"), "rt").read()}}
        jobs.append((d, open(os.path.join(d, "in.vcf")).read(), sams, {"max_block_size": mbs}))
    d0 = os.path.join(GOLD, "pipe_opts")
    sams = {b + ".bam": {c: gzip.open(os.path.join(d0, "%s.%s.sam.gz" % (b, c)), "rt").read() for c in ("chr21", "chr22")} for b in ("o1", "o2")}
    for name in ("gw_maf", "separator", "unique_ids"):
        jobs.append((os.path.join(d0, name), open(os.path.join(d0, "in.vcf")).read(), sams, dict(OPT_CASES[name])))
    for d, vcf, sams, kw in jobs:
        for mode, conf in ((1, 0.6), (2, 0.6), (2, 0.95)):
            res, _ = run_pipeline(phaser, rvm, vcf, sams, d, capture_calls=False, write_vcf=1, gw_phase_vcf=mode, gw_phase_vcf_min_confidence=conf, **kw)
            wgz(os.path.join(d, "out.vcf_gw%d_c%d.txt.gz" % (mode, int(conf * 100))), res["vcf"])
        print("write_vcf_more", os.path.relpath(d, GOLD), len(res["vcf"].splitlines()), "lines")


def gene_ae_features(hc_text, seed):
    """Synthetic BED features for a haplotypic_counts file: clusters over runs of variants, nested features, features that
    END exactly at (variant position - 1) inside multi-variant blocks (the inclusive-end quirk of variant_feature_reads,
    phaser_gene_ae.py:190), empty regions and a contig the counts never mention."""
    import random
    rng = random.Random(seed)
    rows = [l.split("\t") for l in hc_text.split("\n")[1:] if l]
    per = {}
    for r in rows:
        for u in r[3].split(","):
            f = u.split("_")
            per.setdefault(r[0], set()).add(int(f[1]))
    feats = []
    for chrom in sorted(per):
        pos = sorted(per[chrom])
        i = 0
        while i < len(pos):
            k = rng.randint(1, 9)
            a = pos[i]; b = pos[min(len(pos) - 1, i + k - 1)]
            feats.append((chrom, max(0, a - 1 - rng.randint(0, 300)), b + rng.randint(0, 300), "g%d" % len(feats)))
            if rng.random() < 0.3:              # nested / overlapping neighbour
                feats.append((chrom, max(0, a - 1), b, "g%d_inner" % len(feats)))
            if rng.random() < 0.15:
                feats.append((chrom, b + 400, b + 500, "g%d_empty" % len(feats)))
            if rng.random() < 0.25:             # exactly one variant
                feats.append((chrom, a - 1, a, "g%d_one" % len(feats)))
            i += k if rng.random() < 0.8 else max(1, k - 2)
        for r in rows:
            vs = r[3].split(",")
            if r[0] == chrom and len(vs) >= 3 and rng.random() < 0.35:
                p = [int(u.split("_")[1]) for u in vs]
                j = rng.randint(1, len(p) - 1)
                feats.append((chrom, max(0, p[0] - 1 - rng.randint(0, 50)), p[j] - 1, "g%d_edge" % len(feats)))
    feats.append(("chrUn_absent", 100, 5000, "g_absent"))
    return "".join("%s\t%d\t%d\t%s\n" % f for f in feats if f[2] > f[1])


def gene_ae_boundary_features(hc_text):
    """Features that sit exactly on the edges the two overlap rules of phaser_gene_ae.py disagree about: the IntervalTree query is
    half-open (`tree[start-1:stop]`, :107) while the per-variant test is closed at the feature end (`(pos-1) - feature.end <= 0`,
    :191).  For every block with >= 3 variants p0 < p1 < p2 ... (1-based): a feature ending right at p1-1 (closed rule pulls p1 in),
    one starting right after p1, a single-base feature on p1, a twin with identical coordinates, features touching the block from
    the left / right by zero and by one base."""
    rows = [l.split("\t") for l in hc_text.split("\n")[1:] if l]
    feats = []
    seen = set()
    for r in rows:
        vs = r[3].split(",")
        if len(vs) < 3 or (r[0], r[1]) in seen:
            continue
        seen.add((r[0], r[1]))
        p = [int(u.split("_")[1]) for u in vs]
        s0, e0 = int(r[1]) - 1, int(r[2])                 # the row's query interval [start-1, stop)
        k = len(feats)
        feats += [(r[0], p[0] - 1, p[1] - 1, "b%d_end_on_next" % k), (r[0], p[1], p[2], "b%d_starts_after" % k),
                  (r[0], p[1] - 1, p[1], "b%d_single" % k), (r[0], p[0] - 1, p[1] - 1, "b%d_twin" % k),
                  (r[0], e0, e0 + 50, "b%d_right_touch" % k), (r[0], max(0, s0 - 50), s0, "b%d_left_touch" % k),
                  (r[0], max(0, s0 - 50), s0 + 1, "b%d_left_one" % k), (r[0], e0 - 1, e0 + 50, "b%d_right_one" % k)]
        if len(feats) > 160:
            break
    return "".join("%s\t%d\t%d\t%s\n" % f for f in feats if f[2] > f[1])


def fx_gene_ae(phaser, rvm):
    """phaser_gene_ae: run the reference's script on haplotypic_counts files the reference's phASER wrote
    (other fixtures) and on synthetic BED features, with the REAL `intervaltree` package behind it: intervaltree 3.1.0 is not installed for
    this interpreter but its pure-Python sources sit in the image's conda tree (/opt/conda/lib/python3.9/site-packages/intervaltree; its one
    dependency, sortedcontainers, is installed here), so the package is loaded from there by path.  Every case is run a second time with an
    independently written brute-force stand-in (a list scan with the documented overlap rule) and both outputs must be identical -- the
    stand-in is what rounds 1-4 generated these fixtures with; the fixtures did not change when the real package took over."""
    import collections
    import importlib.util
    import runpy
    real_dir = "/opt/conda/lib/python3.9/site-packages/intervaltree"
    spec = importlib.util.spec_from_file_location("intervaltree", os.path.join(real_dir, "__init__.py"), submodule_search_locations=[real_dir])
    real_mod = importlib.util.module_from_spec(spec); sys.modules["intervaltree"] = real_mod; spec.loader.exec_module(real_mod)
    assert real_mod.IntervalTree.__module__.startswith("intervaltree") and os.path.dirname(real_mod.__file__) == real_dir
    Interval = collections.namedtuple("Interval", ["begin", "end", "data"])

    class BruteTree:
        def __init__(self):
            self.ivs = []

        def __setitem__(self, sl, data):
            if sl.start >= sl.stop:
                raise ValueError("IntervalTree: Null Interval objects not allowed in IntervalTree")
            self.ivs.append(Interval(sl.start, sl.stop, data))

        def __getitem__(self, sl):
            if sl.start >= sl.stop:
                return set()
            return set(iv for iv in self.ivs if iv.begin < sl.stop and iv.end > sl.start)
    brute_mod = types.ModuleType("intervaltree"); brute_mod.IntervalTree = BruteTree; brute_mod.Interval = Interval
    script = "/root/reference/phaser_gene_ae/phaser_gene_ae.py"
    cases = [("pipe_one", "pipe_one", 11, []), ("pipe_two", "pipe_two", 12, []), ("pipe_two_mincov", "pipe_two", 12, ["--min_cov", "5"]),
             ("pipe_two_gw06", "pipe_two", 12, ["--gw_cutoff", "0.6"]), ("pipe_noisy_c", "pipe_noisy_c", 13, []), ("c1", "c1", 14, []),
             ("opts_gw_maf", os.path.join("pipe_opts", "gw_maf"), 15, ["--min_haplo_maf", "0.1"]),
             ("opts_bam_exclude", os.path.join("pipe_opts", "bam_exclude"), 16, []),
             ("opts_blacklist", os.path.join("pipe_opts", "blacklist"), 17, ["--gw_cutoff", "0.75", "--min_cov", "2"]),
             ("pipe_two_strict", "pipe_two", 18, ["--gw_cutoff", "1.01"]), ("pipe_noisy_b", "pipe_noisy_b", 19, []),
             ("opts_gw_maf_strict", os.path.join("pipe_opts", "gw_maf"), 20, ["--min_haplo_maf", "0.35", "--min_cov", "1"])]
    cases.append(("pipe_two_bounds", "pipe_two", 0, []))
    cases.append(("noisy_c_bounds", "pipe_noisy_c", 0, ["--min_cov", "1"]))
    for name, src, seed, extra in cases:
        hc = gzip.open(os.path.join(GOLD, src, "out.haplotypic_counts.txt.gz"), "rt").read()
        d = os.path.join(GOLD, "gene_ae", name); os.makedirs(d, exist_ok=True)
        bed = gene_ae_boundary_features(hc) if name.endswith("_bounds") else gene_ae_features(hc, seed)
        with tempfile.TemporaryDirectory() as tmp:
            hp = os.path.join(tmp, "hc.txt"); bp = os.path.join(tmp, "f.bed"); op = os.path.join(tmp, "o.txt")
            open(hp, "w").write(hc); open(bp, "w").write(bed)
            outs = []
            for mod in (real_mod, brute_mod):
                sys.modules["intervaltree"] = mod
                argv = sys.argv
                sys.argv = [script, "--haplotypic_counts", hp, "--features", bp, "--o", op] + extra
                buf = io.StringIO(); old = sys.stdout; sys.stdout = buf
                try:
                    runpy.run_path(script, run_name="__main__")
                finally:
                    sys.stdout = old; sys.argv = argv
                outs.append(open(op).read()); os.remove(op)
            assert outs[0] == outs[1], "intervaltree 3.1.0 and the brute-force overlap disagree on " + name
            out = outs[0]
        open(os.path.join(d, "features.bed"), "w").write(bed)
        wgz(os.path.join(d, "out.gene_ae.txt.gz"), out)
        json.dump({"haplotypic_counts": src.replace(os.sep, "/") + "/out.haplotypic_counts.txt.gz", "args": extra}, open(os.path.join(d, "case.json"), "w"))
        print("gene_ae", name, len(bed.splitlines()), "features", len(out.splitlines()) - 1, "rows")


def fx_expr_matrix(phaser, rvm):
    """phaser_pop/phaser_expr_matrix.py: run the reference's script on a directory of gene_ae files (the
    reference's own gene_ae outputs of other fixtures, one sample per file, sample names made distinct) and keep the two
    matrices it writes.  bgzip / tabix are not installed: the script's shell calls fail and leave the plain .bed files, which is
    what we keep."""
    import runpy
    script = "/root/reference/phaser_pop/phaser_expr_matrix.py"
    src = os.path.join(GOLD, "gene_ae")
    bed = open(os.path.join(src, "pipe_two", "features.bed")).read()
    assert bed == open(os.path.join(src, "pipe_two_gw06", "features.bed")).read() == open(os.path.join(src, "pipe_two_mincov", "features.bed")).read()
    d = os.path.join(GOLD, "expr_matrix"); os.makedirs(os.path.join(d, "gene_ae"), exist_ok=True)
    files = {}
    for case, tag in (("pipe_two", "A"), ("pipe_two_gw06", "B"), ("pipe_two_mincov", "C")):
        text = gzip.open(os.path.join(src, case, "out.gene_ae.txt.gz"), "rt").read()
        lines = text.split("\n")
        head, rows = lines[0], [l for l in lines[1:] if l]
        for bam in ("t1", "t2"):
            name = "%s_%s" % (tag, bam)
            body = [l.rsplit("\t", 1)[0] + "\t" + name for l in rows if l.rsplit("\t", 1)[1] == bam]
            files[name + ".gene_ae.txt"] = head + "\n" + "\n".join(body) + "\n"
    outs = {}
    for order in ("sorted", "reversed"):          # os.listdir order is arbitrary: pin the script's behaviour under both
        with tempfile.TemporaryDirectory() as tmp:
            gdir = os.path.join(tmp, "in"); os.makedirs(gdir)
            for fn, body in files.items():
                open(os.path.join(gdir, fn), "w").write(body)
            bp = os.path.join(tmp, "f.bed"); open(bp, "w").write(bed)
            cwd = os.getcwd(); os.chdir(tmp)
            argv = sys.argv
            sys.argv = [script, "--gene_ae_dir", gdir, "--features", bp, "--o", os.path.join(tmp, "out")]
            buf = io.StringIO(); old = sys.stdout; sys.stdout = buf
            real_listdir = os.listdir
            os.listdir = lambda p=".": sorted(real_listdir(p), reverse=(order == "reversed"))
            try:
                runpy.run_path(script, run_name="__main__")
            finally:
                os.listdir = real_listdir
                sys.stdout = old; sys.argv = argv; os.chdir(cwd)
            outs[order] = (open(os.path.join(tmp, "out.bed")).read(), open(os.path.join(tmp, "out.gw_phased.bed")).read(), buf.getvalue())
    for fn, body in files.items():
        wgz(os.path.join(d, "gene_ae", fn + ".gz"), body)
    open(os.path.join(d, "features.bed"), "w").write(bed)
    for order, (out_all, out_gw, log) in outs.items():
        wgz(os.path.join(d, "out.%s.bed.gz" % order), out_all); wgz(os.path.join(d, "out.%s.gw_phased.bed.gz" % order), out_gw)
        wgz(os.path.join(d, "out.%s.log.txt.gz" % order), log)
        print("expr_matrix", order, len(files), "files ->", out_all.split("\n")[0].count("\t") - 3, "sample columns,", len(out_all.splitlines()) - 1, "rows")


FIXTURES = {"sparse": fx_sparse, "kat": fx_kat, "mapper_small": fx_mapper_small, "mapper_unsorted": fx_mapper_unsorted, "pipeline": fx_pipeline, "c1": fx_c1, "write_vcf": fx_write_vcf, "write_vcf_more": fx_write_vcf_more, "indels": fx_indels, "options": fx_options, "gene_ae": fx_gene_ae, "expr_matrix": fx_expr_matrix, "blacklist_bed": fx_blacklist_bed}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    phaser, rvm = build_reference()
    os.makedirs(GOLD, exist_ok=True)
    for name, fn in FIXTURES.items():
        if a.only in ("", name):
            fn(phaser, rvm)
