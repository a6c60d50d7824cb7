"""GPU end-to-end parity: K_map + K_tally + components + host assembly vs the reference's five output files
(tests/golden/pipe_*, c1/) and vs the pinned oracle on fresh seeded inputs.  Canonical forms per SURVEY.md 8(a)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLD, REPO, gz_text
from helpers import OUTPUTS, canonical

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mapper():
    from phaser_amd.mapper import Mapper
    return Mapper(0)


def run_product(mapper, vcf_text, bams, device="cpu", include_indels=0, load_kw=None, isize=0.0, **cfgkw):
    """bams: ordered {bam_path: {chrom: sam_text}}"""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from phasing_oracle import bam_display_names          # naming helper only (test side)
    from phaser_amd import samio, vcf
    from phaser_amd.engine import Config, Engine
    vs = vcf.load_variants(vcf_text, include_indels=include_indels, **(load_kw or {}))
    eng = Engine(vs, bam_display_names(list(bams.keys())), Config(**cfgkw), mapper=mapper)
    interners = {}
    for bi, (bam, per_chrom) in enumerate(bams.items()):
        for chrom in vs.chroms:
            if chrom not in per_chrom:
                continue
            shards = samio.shards_from_sam(per_chrom[chrom], interners, isize)
            for c2, sh in shards.items():
                eng.add_shard(bi, c2, sh.to(device), len(interners[c2]), interners[c2].names)
        for c2 in interners:
            eng.n_qid[c2] = len(interners[c2])
        eng.close_bam(bi)
    return eng.finish(), eng


def compare(out, gold_dir):
    for name in OUTPUTS:
        want = gz_text(os.path.join(gold_dir, "out.%s.txt.gz" % name))
        assert canonical(name, out[name]) == canonical(name, want), name


@pytest.mark.parametrize("device", ["cpu", "cuda"])
def test_pipe_one(mapper, device):
    d = os.path.join(GOLD, "pipe_one")
    out, eng = run_product(mapper, open(os.path.join(d, "in.vcf")).read(),
                           {"a.bam": {"chr22": gz_text(os.path.join(d, "a.chr22.sam.gz"))}}, device)
    compare(out, d)
    log = gz_text(os.path.join(d, "out.log.txt.gz"))
    for line in eng.log:
        assert line in log, line


@pytest.mark.parametrize("device_rows", [True, False])
def test_pipe_sparse_first_bam_misses_chromosomes(mapper, device_rows):
    """tests/golden/pipe_sparse, written by the reference: three chromosomes, three BAMs, the first with reads on chr11 only -- the blocks of chr11 come first, then
    chr3 and chr19 (the chromosomes BAM 2 brings in, VCF order), in the five files and in the block numbers (PI) of the phased VCF.  Found by tools/stress_parity.py
    in round 5 (the product listed the chromosomes in VCF order); both row stages."""
    from phaser_amd import vcfout
    d = os.path.join(GOLD, "pipe_sparse")
    vcf_text = open(os.path.join(d, "in.vcf")).read()
    bams = {b + ".bam": {c: gz_text(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr3", "chr11", "chr19")} for b in ("s1", "s2", "s3")}
    out, eng = run_product(mapper, vcf_text, bams, "cuda", want_vcf=True, device_rows=device_rows)
    assert eng.rows_path == ("device" if device_rows else "host")
    compare(out, d)
    first = [l.split("\t")[0].split("_")[0] for l in out["allele_config"].split("\n")[1:] if l]
    assert [c for i, c in enumerate(first) if i == 0 or first[i - 1] != c] == ["chr11", "chr3", "chr19"]
    text, up, pc = vcfout.phased_vcf_text(vcf_text, 9, eng)
    assert text == gz_text(os.path.join(d, "out.vcf.txt.gz"))


@pytest.mark.parametrize("host_threads", [1, 3])
def test_pipe_two_bams_two_chroms(mapper, host_threads):
    """host_threads > 1: the native block phasing / row writer runs multi-threaded (the reference's --threads)."""
    d = os.path.join(GOLD, "pipe_two")
    bams = {}
    for b in ("t1", "t2"):
        bams[b + ".bam"] = {c: gz_text(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")}
    out, eng = run_product(mapper, open(os.path.join(d, "in.vcf")).read(), bams, "cuda", host_threads=host_threads)
    compare(out, d)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_pipe_noisy(mapper, tag):
    d = os.path.join(GOLD, "pipe_noisy_" + tag)
    meta = json.load(open(os.path.join(d, "meta.json")))
    out, eng = run_product(mapper, open(os.path.join(d, "in.vcf")).read(),
                           {"n.bam": {"chr22": gz_text(os.path.join(d, "n.chr22.sam.gz"))}}, "cuda",
                           max_block_size=meta["max_block_size"])
    compare(out, d)


def test_c1(mapper, c1_inputs):
    out, eng = run_product(mapper, c1_inputs["vcf"], {"c1.bam": {"chr22": c1_inputs["sam"]}}, "cuda")
    compare(out, os.path.join(GOLD, "c1"))


def test_cli_from_bam_matches_reference(tmp_path):
    """The drop-in CLI on an UNFILTERED BAM + gzipped VCF (own BGZF reader, own duplicate / proper-pair / MAPQ
    filters) reproduces what the reference wrote for the same sample (fixture pipe_one)."""
    import gzip
    from phaser_amd import bamio, phaser, synth
    v, gs, ge, w = synth.make_variants("chr22", 1, 3_000_000, 300, 201, n_genes=20)
    rb = synth.make_reads(v, gs, ge, w, 9000, 202)
    bam = str(tmp_path / "a.bam")
    bamio.readbatch_to_bam(bam, [rb], [("chr21", 46709983), ("chr22", 50818468)])
    d = os.path.join(GOLD, "pipe_one")
    vcfgz = str(tmp_path / "in.vcf.gz")
    with gzip.open(vcfgz, "wt") as f:
        f.write(open(os.path.join(d, "in.vcf")).read())
    prefix = str(tmp_path / "out")
    rc = phaser.main(["--vcf", vcfgz, "--bam", bam, "--sample", "S1", "--mapq", "255", "--baseq", "10", "--paired_end", "1",
                      "--o", prefix, "--write_vcf", "1", "--threads", "3"])
    assert rc == 0
    out = {name: open(prefix + "." + name + ".txt").read() for name in OUTPUTS}
    compare(out, d)
    # --write_vcf 1 (the default): phased VCF, bgzipped + tabix-indexed
    assert gzip.open(prefix + ".vcf.gz", "rt").read() == gz_text(os.path.join(d, "out.vcf_gw0.txt.gz"))
    assert gzip.open(prefix + ".vcf.gz.tbi", "rb").read()[:4] == b"TBI\x01"


@pytest.mark.parametrize("index", ["tbi", "stale"])
def test_cli_bam_prefetch_from_the_tabix_index(tmp_path, index, capfd):
    """A tabix-indexed VCF (what the reference asks for, phaser.py:31): the first BAM's decode starts from the index's contig names before the VCF is gunzipped.
    "stale": an index left over from another file names other contigs -- the prefetch is discarded after the VCF is parsed and the BAM is read again;
    same five files either way (fixture pipe_one)."""
    import gzip
    from phaser_amd import bamio, phaser, synth, vcfout
    v, gs, ge, w = synth.make_variants("chr22", 1, 3_000_000, 300, 201, n_genes=20)
    rb = synth.make_reads(v, gs, ge, w, 9000, 202)
    bam = str(tmp_path / "a.bam")
    bamio.readbatch_to_bam(bam, [rb], [("chr21", 46709983), ("chr22", 50818468)])
    d = os.path.join(GOLD, "pipe_one")
    vcfgz = str(tmp_path / "in.vcf.gz")
    assert vcfout.write_bgzf(vcfgz, open(os.path.join(d, "in.vcf")).read(), 2, index="vcf")
    if index == "stale":
        other = str(tmp_path / "other.vcf.gz")
        text = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\nchr21\t100\tr1\tA\tC\t.\tPASS\t.\tGT\t0|1\n"
        assert vcfout.write_bgzf(other, text, 1, index="vcf")
        os.replace(other + ".tbi", vcfgz + ".tbi")
    prefix = str(tmp_path / "out")
    os.environ["PHZ_TIMING"] = "1"
    try:
        rc = phaser.main(["--vcf", vcfgz, "--bam", bam, "--sample", "S1", "--mapq", "255", "--baseq", "10", "--paired_end", "1", "--o", prefix, "--write_vcf", "0", "--threads", "3"])
    finally:
        del os.environ["PHZ_TIMING"]
    assert rc == 0
    compare({name: open(prefix + "." + name + ".txt").read() for name in OUTPUTS}, d)
    err = capfd.readouterr().err
    assert ("bam prefetch during the VCF parse: used" in err) if index == "tbi" else ("bam prefetch during the VCF parse: discarded" in err), err[-1500:]


@pytest.mark.parametrize("hashseed,twin", [("0", "0"), ("4242", "0"), ("0", "1")])
def test_cli_py_hash_order_gives_the_reference_bytes(tmp_path, hashseed, twin):
    """--py_hash_order 1: the drop-in CLI (GPU path from an unfiltered BAM) writes the reference's five files BYTE FOR BYTE (fixture pipe_one was written by
    the reference under PYTHONHASHSEED=0, CPython 3.10): the raw tier of SURVEY.md 8(a).  The native tier restates that interpreter's str hash and set, so
    the bytes do not depend on the seed of the interpreter the CLI runs under; twin 1 = the pure-Python replay with real sets (needs seed 0)."""
    import gzip
    import subprocess
    from phaser_amd import bamio, synth
    v, gs, ge, w = synth.make_variants("chr22", 1, 3_000_000, 300, 201, n_genes=20)
    rb = synth.make_reads(v, gs, ge, w, 9000, 202)
    bam = str(tmp_path / "a.bam")
    bamio.readbatch_to_bam(bam, [rb], [("chr21", 46709983), ("chr22", 50818468)])
    d = os.path.join(GOLD, "pipe_one")
    vcfgz = str(tmp_path / "in.vcf.gz")
    with gzip.open(vcfgz, "wt") as f:
        f.write(open(os.path.join(d, "in.vcf")).read())
    prefix = str(tmp_path / "out")
    env = dict(os.environ, PYTHONHASHSEED=hashseed, PYTHONPATH=REPO, PHZ_PYORDER_PYTHON=twin)
    r = subprocess.run([sys.executable, "-m", "phaser_amd.phaser", "--vcf", vcfgz, "--bam", bam, "--sample", "S1", "--mapq", "255", "--baseq", "10", "--paired_end", "1",
                        "--o", prefix, "--write_vcf", "0", "--threads", "3", "--py_hash_order", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for name in OUTPUTS:
        assert open(prefix + "." + name + ".txt").read() == gz_text(os.path.join(d, "out.%s.txt.gz" % name)), name


@pytest.mark.parametrize("seed,err", [(9001, 0.002), (9002, 0.04)])
def test_fresh_seed_vs_oracle(mapper, oracle_build, tmp_path, seed, err):
    """Inputs nobody has seen before (2 chromosomes x 2 BAMs with shared QNAMEs): product (GPU) vs the pinned oracle."""
    import subprocess
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import phasing_oracle as po
    from phaser_amd import synth
    contigs = [("chr5", 181538259), ("chr9", 138394717)]
    vs = []; bams = {"x1.bam": {}, "x2.bam": {}}
    for ci, (chrom, ln) in enumerate(contigs):
        v, gs, ge, w = synth.make_variants(chrom, 1, 1_500_000, 220, seed + ci, n_genes=10)
        vs.append(v)
        for bi, bam in enumerate(bams):
            rb = synth.make_reads(v, gs, ge, w, 5000, seed + 10 * ci + bi + 100, qname_prefix="q", err_rate=err)
            rf = rb.select(synth.samtools_keep(rb, 255))
            bams[bam][chrom] = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf_text = "\n".join(synth.vcf_lines(vs)) + "\n"
    got, eng = run_product(mapper, vcf_text, bams, "cuda", max_block_size=8)
    # oracle side
    pool, _, _ = po.load_vcf(vcf_text)
    ph = po.Phaser(po.bam_display_names(list(bams.keys())), max_block_size=8)
    for bam, per_chrom in bams.items():
        texts = []
        for c in pool:
            tp = tmp_path / "t.tsv"; tp.write_text("".join("\t".join(r) + "\n" for r in po.variant_table_rows(pool[c])[0]))
            op = tmp_path / "c.tsv"
            subprocess.run([os.path.join(oracle_build, "rvm_oracle"), "--variant_table", str(tp), "--baseq", "10", "--o", str(op)],
                           input=per_chrom[c].encode(), check=True)
            texts.append(op.read_text())
        ph.add_bam(texts)
    want = ph.finish()
    for name in OUTPUTS:
        assert canonical(name, got[name]) == canonical(name, want[name]), name
    assert eng.phased == ph.phased and eng.phased > 50


@pytest.mark.parametrize("read_ids", [0, 1])
def test_chromosome_without_reads_in_any_bam(mapper, oracle_build, tmp_path, read_ids):
    """Three chromosomes in the VCF, two BAMs: the middle chromosome has het SNPs but no read in either BAM (no shard, no QNAME table), the first one has
    reads only in the second BAM.  Device row stage without and with the QNAME columns (--output_read_ids 1) vs the pinned oracle."""
    import subprocess
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import phasing_oracle as po
    from phaser_amd import synth
    contigs = [("chr3", 198295559), ("chr11", 135086622), ("chr19", 58617616)]
    vs = []; bams = {"x1.bam": {}, "x2.bam": {}}
    for ci, (chrom, ln) in enumerate(contigs):
        v, gs, ge, w = synth.make_variants(chrom, 1, 1_000_000, 150, 9700 + ci, n_genes=8)
        vs.append(v)
        for bi, bam in enumerate(bams):
            rb = synth.make_reads(v, gs, ge, w, 3000, 9710 + 10 * ci + bi, qname_prefix="q", err_rate=0.01)
            rf = rb.select(synth.samtools_keep(rb, 255))
            empty = ci == 1 or (ci == 0 and bi == 0)
            bams[bam][chrom] = "" if empty else "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf_text = "\n".join(synth.vcf_lines(vs)) + "\n"
    got, eng = run_product(mapper, vcf_text, bams, "cuda", max_block_size=8, output_read_ids=read_ids)
    assert eng.rows_path == "device"
    pool, _, _ = po.load_vcf(vcf_text)
    ph = po.Phaser(po.bam_display_names(list(bams.keys())), max_block_size=8, output_read_ids=read_ids)
    for bam, per_chrom in bams.items():
        texts = []
        for c in pool:
            tp = tmp_path / "t.tsv"; tp.write_text("".join("\t".join(r) + "\n" for r in po.variant_table_rows(pool[c])[0]))
            op = tmp_path / "c.tsv"
            subprocess.run([os.path.join(oracle_build, "rvm_oracle"), "--variant_table", str(tp), "--baseq", "10", "--o", str(op)], input=per_chrom[c].encode(), check=True)
            texts.append(op.read_text())
        ph.add_bam(texts)
    want = ph.finish()
    for name in OUTPUTS:
        assert canonical(name, got[name]) == canonical(name, want[name]), name
    assert eng.phased == ph.phased and eng.phased > 30


@pytest.mark.parametrize("seed,err,pairs,mbs,read_ids", [(9101, 0.004, 30000, 15, 0), (9102, 0.06, 14000, 6, 0), (9103, 0.01, 9000, 10, 1)])
def test_deep_coverage_vs_oracle(mapper, oracle_build, tmp_path, seed, err, pairs, mbs, read_ids):
    """Few genes, thousands of reads over every het SNP, two BAMs with shared QNAMEs: read sets of thousands of QNAMEs per haplotype (the
    workgroup / global-table paths of the read-set kernels), rows with tens of kilobytes of labels (the direct write path), long blocks and,
    with 6 % base errors, conflicting components (weak-point split, brute force, stitching on the GPU).  Product vs the pinned oracle, and the
    device row stage vs the host row stage byte for byte.  The third case writes the QNAME columns of --output_read_ids 1 on the device: read sets of
    thousands of QNAMEs listed by the wave / workgroup read-set kernels' first-appearance flags."""
    import subprocess
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import phasing_oracle as po
    from phaser_amd import synth
    contigs = [("chr7", 159345973)]
    chrom = "chr7"
    v, gs, ge, w = synth.make_variants(chrom, 1, 400_000, 90, seed, n_genes=3)
    bams = {"d1.bam": {}, "d2.bam": {}}
    for bi, bam in enumerate(bams):
        rb = synth.make_reads(v, gs, ge, w, pairs, seed + bi + 100, qname_prefix="q", err_rate=err)
        rf = rb.select(synth.samtools_keep(rb, 255))
        bams[bam][chrom] = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf_text = "\n".join(synth.vcf_lines([v])) + "\n"
    got, eng = run_product(mapper, vcf_text, bams, "cuda", max_block_size=mbs, output_read_ids=read_ids)
    assert eng.rows_path == "device", getattr(eng, "rows_fallback", "")
    host, heng = run_product(mapper, vcf_text, bams, "cuda", max_block_size=mbs, device_rows=False, output_read_ids=read_ids)
    assert heng.rows_path == "host"
    for name in OUTPUTS:
        assert got[name] == host[name], name
    pool, _, _ = po.load_vcf(vcf_text)
    ph = po.Phaser(po.bam_display_names(list(bams.keys())), max_block_size=mbs, output_read_ids=read_ids)
    for bam, per_chrom in bams.items():
        texts = []
        for c in pool:
            tp = tmp_path / "t.tsv"; tp.write_text("".join("\t".join(r) + "\n" for r in po.variant_table_rows(pool[c])[0]))
            op = tmp_path / "c.tsv"
            subprocess.run([os.path.join(oracle_build, "rvm_oracle"), "--variant_table", str(tp), "--baseq", "10", "--o", str(op)],
                           input=per_chrom[c].encode(), check=True)
            texts.append(op.read_text())
        ph.add_bam(texts)
    want = ph.finish()
    for name in OUTPUTS:
        assert canonical(name, got[name]) == canonical(name, want[name]), name
    assert eng.phased == ph.phased and eng.phased > 20
    assert eng.stats["rowsdev_n_big_segments"] > 0
    # phz_hap_counts: distinct reads per (variant, allele, BAM) list of the tally still resident from the host-twin pass -- lists of thousands
    # of entries here (wave / workgroup read-set kernels) -- against numpy on the fetched lists; device and host destinations
    import ctypes as C
    from phaser_amd import _lib
    heng._fetch_tally()
    G = heng.G
    nseg = G["nv"] * 2 * G["nb"]
    rs = G["rl_start"].astype(np.int64); rq = G["rl_qid"]
    want_n = np.array([len(np.unique(rq[rs[e]:rs[e + 1]])) for e in range(nseg)], dtype=np.int32)
    got_h = np.full(nseg, -1, dtype=np.int32)
    heng.ctx.check(heng.lib.phz_hap_counts(heng.ctx.h, C.c_void_p(got_h.ctypes.data), nseg, _lib.PHZ_HOST))
    got_d = torch.full((nseg,), -1, dtype=torch.int32, device="cuda")
    heng.ctx.check(heng.lib.phz_hap_counts(heng.ctx.h, C.c_void_p(got_d.data_ptr()), nseg, _lib.PHZ_DEVICE))
    assert np.array_equal(got_h, want_n) and np.array_equal(got_d.cpu().numpy(), want_n) and int(want_n.max()) > 64          # (lists beyond one thread's 32 entries: the wave kernel; the first case has lists of thousands)


@pytest.mark.parametrize("seed,n_snps,err,pairs,mbs,L", [(9201, 400, 0.003, 3000, 15, 76), (9202, 1500, 0.03, 2500, 10, 76), (9301, 1500, 0.01, 200, 15, 1000)])
def test_dense_variants_vs_oracle(mapper, oracle_build, tmp_path, seed, n_snps, err, pairs, mbs, L):
    """Het SNPs every 5-40 bp inside a few genes: a read covers up to ~20 of them (the mapper's candidate buffers past the 8-call fast
    path), a QNAME contributes hundreds of variant pairs (long groups in the tally tiles), components run to hundreds of variants and,
    with 3 % base errors, have to be split and stitched.  The third case has 1,000-base reads: ~60 calls per read (past the 32 ordinals
    of the candidate buffer: in-lane fallback of the mapper), ~7,000 variant pairs per QNAME.  Product vs the pinned oracle; device row
    stage vs host row stage."""
    import subprocess
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import phasing_oracle as po
    from phaser_amd import synth
    contigs = [("chr7", 159345973)]
    chrom = "chr7"
    v, gs, ge, w = synth.make_variants(chrom, 1, 600_000, n_snps, seed, n_genes=4)
    bams = {"d1.bam": {}, "d2.bam": {}}
    for bi, bam in enumerate(bams):
        rb = synth.make_reads(v, gs, ge, w, pairs, seed + bi + 100, L=L, qname_prefix="q", err_rate=err)
        rf = rb.select(synth.samtools_keep(rb, 255))
        bams[bam][chrom] = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
    vcf_text = "\n".join(synth.vcf_lines([v])) + "\n"
    got, eng = run_product(mapper, vcf_text, bams, "cuda", max_block_size=mbs)
    host, heng = run_product(mapper, vcf_text, bams, "cuda", max_block_size=mbs, device_rows=False)
    assert heng.rows_path == "host"
    for name in OUTPUTS:
        assert got[name] == host[name], (name, eng.rows_path)
    pool, _, _ = po.load_vcf(vcf_text)
    ph = po.Phaser(po.bam_display_names(list(bams.keys())), max_block_size=mbs)
    for bam, per_chrom in bams.items():
        texts = []
        for c in pool:
            tp = tmp_path / "t.tsv"; tp.write_text("".join("\t".join(r) + "\n" for r in po.variant_table_rows(pool[c])[0]))
            op = tmp_path / "c.tsv"
            subprocess.run([os.path.join(oracle_build, "rvm_oracle"), "--variant_table", str(tp), "--baseq", "10", "--o", str(op)],
                           input=per_chrom[c].encode(), check=True)
            texts.append(op.read_text())
        ph.add_bam(texts)
    want = ph.finish()
    for name in OUTPUTS:
        assert canonical(name, got[name]) == canonical(name, want[name]), name
    assert eng.phased == ph.phased and eng.phased > 100


@pytest.mark.parametrize("seed,n_snps,err,pairs,mbs", [(9401, 1300, 0.0, 5000, 15), (9404, 800, 0.006, 3000, 12)])
def test_block_and_component_beyond_the_device_tables(mapper, oracle_build, tmp_path, seed, n_snps, err, pairs, mbs, monkeypatch):
    """One gene with 1,300 het SNPs every few bases under deep coverage: ONE connected component of ~1,300 variants.  Without base errors it is a clean
    component and stays ONE haplotype block whatever --max_block_size says (phaser.py:2116-2136) -- a block of more than 512 variants, beyond the device
    stage's gwStat table and LDS piece arrays; with errors it is a component beyond the phasing kernel's 256 variants that the host routine splits
    (phaser.py:2125-2157) into dozens of blocks.  Up to round 4 the first case sent the WHOLE pass to the host stage (verdict missing #4 / next #6); now
    only that component costs: the pass stays on the device, the exception path (phz_phase_block inside the call) really runs on the GPU box, and the
    pair-key table starts from 16 slots so that its growth path runs too.  Product = host row stage byte for byte = the pinned oracle."""
    import subprocess
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import phasing_oracle as po
    from phaser_amd import synth
    contigs = [("chr7", 159345973)]
    chrom = "chr7"
    v, gs, ge, w = synth.make_variants(chrom, 1, 600_000, n_snps, seed, n_genes=1)
    rb = synth.make_reads(v, gs, ge, w, pairs, seed + 100, L=76, qname_prefix="q", err_rate=err)
    rf = rb.select(synth.samtools_keep(rb, 255))
    bams = {"big.bam": {chrom: "\n".join(synth.sam_lines(rf, contigs)) + "\n"}}
    vcf_text = "\n".join(synth.vcf_lines([v])) + "\n"
    monkeypatch.setenv("PHZ_ROWS_PAIR_SLOTS", "16")
    got, eng = run_product(mapper, vcf_text, bams, "cuda", max_block_size=mbs)
    assert eng.rows_path == "device", getattr(eng, "rows_fallback", "")
    assert eng.stats.get("rowsdev_n_exceptions", 0) >= 1, eng.stats                 # the component went through the host phase_v3 inside the device stage
    assert err == 0.0 or eng.stats.get("rowsdev_n_pair_table_growths", 0) >= 1       # (without base errors no pair has conflicting reads: nothing to test, no key)
    sizes = [int(l.split("\t")[4]) for l in got["haplotypes"].split("\n")[1:] if l]
    if err == 0.0:
        assert max(sizes) > 512, max(sizes)                                             # one block beyond the tables
    else:
        assert len([x for x in sizes if x > 1]) > 20 and max(sizes) <= 512
    monkeypatch.delenv("PHZ_ROWS_PAIR_SLOTS")
    host, heng = run_product(mapper, vcf_text, bams, "cuda", max_block_size=mbs, device_rows=False)
    assert heng.rows_path == "host"
    for name in OUTPUTS:
        assert got[name] == host[name], name
    pool, _, _ = po.load_vcf(vcf_text)
    ph = po.Phaser(po.bam_display_names(list(bams.keys())), max_block_size=mbs)
    tp = tmp_path / "t.tsv"; tp.write_text("".join("\t".join(r) + "\n" for r in po.variant_table_rows(pool[chrom])[0]))
    op = tmp_path / "c.tsv"
    subprocess.run([os.path.join(oracle_build, "rvm_oracle"), "--variant_table", str(tp), "--baseq", "10", "--o", str(op)], input=bams["big.bam"][chrom].encode(), check=True)
    ph.add_bam([op.read_text()])
    want = ph.finish()
    for name in OUTPUTS:
        assert canonical(name, got[name]) == canonical(name, want[name]), name
    assert eng.phased == ph.phased and eng.phased > 500


@pytest.mark.parametrize("src,mode", [("pipe_one", 0), ("pipe_one", 1), ("pipe_one", 2), ("pipe_noisy_c", 2), ("pipe_two", 1)])
def test_phased_vcf_matches_reference(mapper, src, mode):
    """write_vcf (phaser.py:1661-1855): the phased VCF text equals what the reference wrote, byte for byte."""
    from phaser_amd import vcfout
    d = os.path.join(GOLD, src)
    vcf_text = open(os.path.join(d, "in.vcf")).read()
    if src == "pipe_two":
        bams = {b + ".bam": {c: gz_text(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")} for b in ("t1", "t2")}
    else:
        b = "a" if src == "pipe_one" else "n"
        bams = {b + ".bam": {"chr22": gz_text(os.path.join(d, b + ".chr22.sam.gz"))}}
    out, eng = run_product(mapper, vcf_text, bams, "cuda")
    got, up, pc = vcfout.phased_vcf_text(vcf_text, 9, eng, gw_phase_vcf=mode)
    assert got == gz_text(os.path.join(d, "out.vcf_gw%d.txt.gz" % mode))


def test_include_indels_pipeline(mapper):
    """--include_indels 1: deletions / insertions in the variant table, general mapper (K_map_general) + the same tally."""
    d = os.path.join(GOLD, "pipe_indel")
    out, eng = run_product(mapper, open(os.path.join(d, "in.vcf")).read(),
                           {"i.bam": {"chr22": gz_text(os.path.join(d, "i.chr22.sam.gz"))}}, "cuda", include_indels=1)
    compare(out, d)
    assert eng.vs.chroms["chr22"].is_general


def _opt_cases():
    return list(json.load(open(os.path.join(GOLD, "pipe_opts", "cases.json")))["cases"].keys())


@pytest.mark.parametrize("name", _opt_cases())
def test_option_flags(mapper, name):
    """Every flag that reaches the hot path, product (GPU) vs what the reference wrote (tests/golden/pipe_opts)."""
    from helpers import option_case_kwargs
    d0 = os.path.join(GOLD, "pipe_opts")
    meta = json.load(open(os.path.join(d0, "cases.json")))
    load, cfg, baseq, isize = option_case_kwargs(name, meta["cases"][name], meta["blacklist"])
    bams = {b + ".bam": {c: gz_text(os.path.join(d0, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")} for b in ("o1", "o2")}
    out, eng = run_product(mapper, open(os.path.join(d0, "in.vcf")).read(), bams, "cuda", load_kw=load, isize=isize, **cfg)
    compare(out, os.path.join(d0, name))


# (test_full_size_pipeline_invariants moved to tests/test_gpu_scale.py::test_configs1_chr1_50m_records: configs[1] is checked there on ALL 50M records and against the phasing oracle)

def test_population_flow_three_samples(mapper, tmp_path):
    """phaser -> phaser_gene_ae -> phaser_expr_matrix for three samples (the phaser_pop flow of BASELINE configs[4] in miniature):
    every stage is the GPU product path; the matrix cells must be the gene-level counts, which must come from the haplotypic counts."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import gene_ae_oracle as go
    from phaser_amd import expr_matrix, gene_ae, synth
    contigs = [("chr7", 159345973)]
    gdir = tmp_path / "gene_ae"; gdir.mkdir()
    bed = None
    per_sample = {}
    for s in range(3):
        v, gs, ge, w = synth.make_variants("chr7", 1, 2_000_000, 260, 500, n_genes=12)          # same variant sites for every sample
        if bed is None:
            bed = "".join("chr7\t%d\t%d\tgene%d\n" % (int(a) - 1, int(b), i) for i, (a, b) in enumerate(zip(gs.tolist(), ge.tolist())))
            (tmp_path / "genes.bed").write_text(bed)
        rb = synth.make_reads(v, gs, ge, w, 6000, 600 + s, qname_prefix="p%d." % s)
        rf = rb.select(synth.samtools_keep(rb, 255))
        sam = "\n".join(synth.sam_lines(rf, contigs)) + "\n"
        out, eng = run_product(mapper, "\n".join(synth.vcf_lines([v])) + "\n", {"sample%d.bam" % s: {"chr7": sam}}, "cuda")
        hc = out["haplotypic_counts"]
        table = gene_ae.gene_ae(hc.encode(), bed)
        assert go.canonical(table) == go.canonical(go.gene_ae(hc, bed))                          # K_genes path == pinned oracle
        (gdir / ("sample%d.gene_ae.txt" % s)).write_text(table)
        per_sample["sample%d" % s] = [r.split("\t") for r in table.split("\n")[1:] if r]
    a, g, log = expr_matrix.expr_matrix(str(gdir), str(tmp_path / "genes.bed"))
    assert not log
    rows = [r.split("\t") for r in a.split("\n") if r]
    assert rows[0][:4] == ["#contig", "start", "stop", "name"] and rows[0][4:] == ["sample0", "sample1", "sample2"]
    assert len(rows) - 1 == len(bed.splitlines())
    for k, r in enumerate(rows[1:]):
        for si, s in enumerate(rows[0][4:]):
            assert r[4 + si] == per_sample[s][k][4] + "|" + per_sample[s][k][5]
    assert any(c != "0|0" for r in rows[1:] for c in r[4:])


def test_population_flow_matches_reference(mapper, tmp_path):
    """next-4 on the GPU box: SAM inputs of fixture pipe_two -> phaser (K_map / K_tally) -> phaser_gene_ae (K_genes) under the three
    option sets of tests/golden/expr_matrix -> phaser_expr_matrix, compared with the matrices the REFERENCE's phaser_expr_matrix.py
    wrote from the reference's own gene_ae files of the reference's own haplotypic counts (tools/make_golden.py fx_expr_matrix)."""
    from phaser_amd import expr_matrix, gene_ae
    d = os.path.join(GOLD, "pipe_two")
    bams = {b + ".bam": {c: gz_text(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")} for b in ("t1", "t2")}
    out, eng = run_product(mapper, open(os.path.join(d, "in.vcf")).read(), bams, "cuda")
    hc = out["haplotypic_counts"].encode()
    e = os.path.join(GOLD, "expr_matrix")
    bed = open(os.path.join(e, "features.bed")).read()
    gdir = tmp_path / "in"; gdir.mkdir()
    for tag, kw in (("A", {}), ("B", {"gw_cutoff": 0.6}), ("C", {"min_cov": 5})):
        text = gene_ae.gene_ae(hc, bed, **kw)
        lines = text.split("\n")
        head, rows = lines[0], [l for l in lines[1:] if l]
        for bam in ("t1", "t2"):
            name = "%s_%s" % (tag, bam)
            body = [l.rsplit("\t", 1)[0] + "\t" + name for l in rows if l.rsplit("\t", 1)[1] == bam]
            (gdir / (name + ".gene_ae.txt")).write_text(head + "\n" + "\n".join(body) + "\n")
    for order in ("sorted", "reversed"):
        a, g, log = expr_matrix.expr_matrix(str(gdir), os.path.join(e, "features.bed"), order)
        assert a == gz_text(os.path.join(e, "out.%s.bed.gz" % order)), order
        assert g == gz_text(os.path.join(e, "out.%s.gw_phased.bed.gz" % order)), order


def test_cli_blacklist_beds_match_reference(tmp_path):
    """--blacklist and --haplo_count_blacklist through the drop-in CLI (phaser.py:220-243): the FULL VCF plus two BED files in,
    the five files the reference wrote for the equivalently pre-filtered VCF / blacklist set out (tests/golden/pipe_bed)."""
    from phaser_amd import phaser
    d0 = os.path.join(GOLD, "pipe_opts"); d = os.path.join(GOLD, "pipe_bed")
    bams = []
    for b in ("o1", "o2"):
        p = tmp_path / (b + ".sam")
        p.write_text("".join(gz_text(os.path.join(d0, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")))
        bams.append(str(p))
    prefix = str(tmp_path / "out")
    rc = phaser.main(["--vcf", os.path.join(d0, "in.vcf"), "--bam", ",".join(bams), "--sample", "S1", "--mapq", "255", "--baseq", "10",
                      "--paired_end", "1", "--o", prefix, "--write_vcf", "0", "--threads", "2",
                      "--blacklist", os.path.join(d, "blacklist.bed"), "--haplo_count_blacklist", os.path.join(d, "haplo_blacklist.bed")])
    assert rc == 0
    # the text inputs are named *.sam; the reference strips only ".bam" from the display name (phaser.py:473), the golden says o1 / o2
    out = {name: open(prefix + "." + name + ".txt").read().replace("\to1.sam\t", "\to1\t").replace("\to2.sam\t", "\to2\t") for name in OUTPUTS}
    compare(out, d)
    ase = [l.split("\t") for l in out["haplotypic_counts"].split("\n")[1:] if l]
    assert any(int(r[6]) > 0 for r in ase)           # some block really lost a variant to the haplotype-count blacklist


def test_cli_bam_prefetch_equals_the_plain_order(tmp_path, monkeypatch, capfd):
    """The first BAM is decoded on the GPU while the VCF is parsed, for the chromosomes vcf.contig_names_guess names (phaser.main).  The
    files must be the ones the plain order (PHZ_BAM_PREFETCH=0: VCF first, then the BAM for the parsed table's chromosomes) writes --
    with a BAM that holds a reference the VCF lacks, a VCF contig without a het site, two BAMs, and with a VCF whose contigs do NOT
    come in runs (the guess misses one: the prefetch has to be thrown away)."""
    import gzip
    from phaser_amd import bamio, phaser, synth, vcf
    refs = [("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrM", 16569)]
    vs, rbs1, rbs2 = [], [], []
    for i, c in enumerate(("chr20", "chr21", "chr22", "chrM")):
        v, gs, ge, w = synth.make_variants(c, 1, 2_000_000 if c != "chrM" else 16000, 260 if c != "chrM" else 12, 301 + i, n_genes=14 if c != "chrM" else 1)
        if c != "chrM":
            vs.append(v)                                  # chrM: reads in the BAMs, no variants in the first VCF
        else:
            v_m = v
        rbs1.append(synth.make_reads(v, gs, ge, w, 5000 if c != "chrM" else 300, 401 + i))
        rbs2.append(synth.make_reads(v, gs, ge, w, 3000 if c != "chrM" else 200, 501 + i))
    b1, b2 = str(tmp_path / "t1.bam"), str(tmp_path / "t2.bam")
    bamio.readbatch_to_bam(b1, rbs1, refs); bamio.readbatch_to_bam(b2, rbs2, refs)
    lines = synth.vcf_lines(vs)
    head = [l for l in lines if l.startswith("#")]; body = [l for l in lines if not l.startswith("#")]
    homs = ["chr19\t%d\t.\tA\tG\t50\tPASS\t.\tGT\t1|1" % (100 * k) for k in range(1, 40)]            # a contig with no het site, first in the file
    runs = "\n".join(head + homs + body) + "\n"
    # two chrM het sites in the middle of the chr20 run and nowhere else: the bisection does not see them
    k20 = max(i for i, l in enumerate(body) if l.startswith("chr20\t")) // 2
    scattered = "\n".join(head + homs + body[:k20] + [l for l in synth.vcf_lines([v_m]) if not l.startswith("#")][:2] + body[k20:]) + "\n"
    assert "chrM" not in vcf.contig_names_guess(scattered.encode())
    assert sorted(vcf.contig_names_guess(runs.encode())) == ["chr19", "chr20", "chr21", "chr22"]

    def run(text, tag, prefetch):
        vp = str(tmp_path / (tag + ".vcf.gz"))
        with gzip.open(vp, "wt") as f:
            f.write(text)
        monkeypatch.setenv("PHZ_BAM_PREFETCH", prefetch)
        monkeypatch.setenv("PHZ_TIMING", "1")
        prefix = str(tmp_path / (tag + "_" + prefetch))
        assert phaser.main(["--vcf", vp, "--bam", b1 + "," + b2, "--sample", "S1", "--mapq", "255", "--baseq", "10", "--paired_end", "1", "--o", prefix,
                            "--write_vcf", "1", "--threads", "3"]) == 0
        out = {name: open(prefix + "." + name + ".txt").read() for name in OUTPUTS}
        out["vcf"] = gzip.open(prefix + ".vcf.gz", "rt").read()
        return out

    plain = run(runs, "runs", "0")
    assert len(plain["haplotypes"].split("\n")) > 20 and "\nchr22\t" in plain["haplotypes"] and "\tt2\t" in plain["haplotypic_counts"]
    assert "bam prefetch" not in capfd.readouterr().err
    assert run(runs, "runs", "1") == plain
    assert "bam prefetch during the VCF parse: used" in capfd.readouterr().err
    sc_plain = run(scattered, "scat", "0")
    assert "\nchrM\t" in sc_plain["allelic_counts"]
    capfd.readouterr()
    assert run(scattered, "scat", "1") == sc_plain
    assert "bam prefetch during the VCF parse: discarded (chromosomes ['chrM'] not in the guess)" in capfd.readouterr().err


def test_four_ranks_from_bams_equal_one_rank(tmp_path):
    """The CLI from BAM files as four ranks sharing the GPU (gloo) over a sample with three chromosomes: every rank decodes only its own
    chromosomes on the device (the fourth rank none at all), the files are byte for byte the one-rank run's."""
    import gzip, subprocess
    from phaser_amd import bamio, phaser, synth
    refs = [("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468)]
    vs, rbs1, rbs2 = [], [], []
    for i, (c, _) in enumerate(refs):
        v, gs, ge, w = synth.make_variants(c, 1, 2_000_000, 200 + 40 * i, 601 + i, n_genes=10)
        vs.append(v)
        rbs1.append(synth.make_reads(v, gs, ge, w, 4000 + 1500 * i, 701 + i)); rbs2.append(synth.make_reads(v, gs, ge, w, 2500, 801 + i))
    b1, b2 = str(tmp_path / "t1.bam"), str(tmp_path / "t2.bam")
    bamio.readbatch_to_bam(b1, rbs1, refs); bamio.readbatch_to_bam(b2, rbs2, refs)
    vp = str(tmp_path / "in.vcf.gz")
    with gzip.open(vp, "wt") as f:
        f.write("\n".join(synth.vcf_lines(vs)) + "\n")
    common = ["--vcf", vp, "--bam", b1 + "," + b2, "--sample", "S1", "--mapq", "255", "--baseq", "10", "--paired_end", "1", "--write_vcf", "1", "--threads", "2"]
    one = str(tmp_path / "one")
    assert phaser.main(common + ["--o", one]) == 0
    four = str(tmp_path / "four")
    port = 29700 + (os.getpid() % 1000)
    procs = []
    for rank in range(4):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="4", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PHZ_DIST_BACKEND="gloo", PYTHONPATH=REPO)
        procs.append(subprocess.Popen([sys.executable, "-m", "phaser_amd.phaser"] + common + ["--o", four], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    assert "using 4 GPU(s)" in logs[0]
    for name in OUTPUTS:
        assert open(four + "." + name + ".txt").read() == open(one + "." + name + ".txt").read(), name
    assert gzip.open(four + ".vcf.gz", "rb").read() == gzip.open(one + ".vcf.gz", "rb").read()
    assert len(open(one + ".haplotypes.txt").read().split("\n")) > 50


def test_cli_fatal_error_waits_for_the_prefetch(tmp_path):
    """A fatal_error raised while the BAM prefetch is running (here: the sample is not in the VCF, found right after the prefetch was started)
    leaves main() only when that thread is done -- no GPU work of ours is in flight when the interpreter goes down."""
    import gzip, threading
    from phaser_amd import bamio, phaser, synth
    v, gs, ge, w = synth.make_variants("chr22", 1, 3_000_000, 300, 201, n_genes=20)
    rb = synth.make_reads(v, gs, ge, w, 60000, 203)
    bam = str(tmp_path / "a.bam")
    bamio.readbatch_to_bam(bam, [rb], [("chr22", 50818468)])
    vcfgz = str(tmp_path / "in.vcf.gz")
    with gzip.open(vcfgz, "wt") as f:
        f.write("\n".join(synth.vcf_lines([v])) + "\n")
    with pytest.raises(SystemExit):
        phaser.main(["--vcf", vcfgz, "--bam", bam, "--sample", "NOT_THERE", "--mapq", "255", "--baseq", "10", "--paired_end", "1", "--o", str(tmp_path / "o"),
                     "--threads", "2"])
    assert not any(t.name == "phz-bam-prefetch" and t.is_alive() for t in threading.enumerate())


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 2), ("gloo", 3)])
def test_two_ranks_one_gpu_real_kernels(tmp_path, backend, world):
    """The multi-rank path with REAL kernels on both ranks: chromosomes LPT-assigned, per-BAM AS histograms all-reduced, noise counters
    all-reduced, the fragment tables all-gathered as int64 tensors, row text spooled to files and spliced by rank 0.  The assembled files must be
    what the reference wrote (fixture pipe_two: two BAMs with shared QNAMEs, two chromosomes -> one chromosome per rank).
    backend gloo: the two ranks share the one GPU of the box; backend nccl (= RCCL over xGMI): one rank per GPU, runs where two GPUs are
    visible and is skipped on a one-GPU box.  world 3: the third rank owns no chromosome -- it still has to take part in every collective."""
    import subprocess
    import torch
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("the RCCL path needs two GPUs")
    d = os.path.join(GOLD, "pipe_two")
    bams = []
    for b in ("t1", "t2"):
        p = tmp_path / (b + ".sam")
        p.write_text("".join(gz_text(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")))
        bams.append(str(p))
    prefix = str(tmp_path / "out")
    port = 29600 + (os.getpid() % 1000)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PHZ_DIST_BACKEND=backend, PYTHONPATH=REPO)
        cmd = [sys.executable, "-m", "phaser_amd.phaser", "--vcf", os.path.join(d, "in.vcf"), "--bam", ",".join(bams), "--sample", "S1",
               "--mapq", "255", "--baseq", "10", "--paired_end", "1", "--o", prefix, "--write_vcf", "0", "--threads", "2"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    assert "using %d GPU(s)" % world in logs[0] and "phASER" not in logs[1]          # rank 0 speaks, rank 1 is silent
    out = {name: open(prefix + "." + name + ".txt").read().replace("\tt1.sam\t", "\tt1\t").replace("\tt2.sam\t", "\tt2\t") for name in OUTPUTS}
    compare(out, d)
    assert not [f for f in os.listdir(tmp_path) if f.startswith("phz_spool_")]          # spool files removed


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_one_rank_forced_through_every_collective(tmp_path, backend):
    """The RCCL branch on the ONE GPU a test box has (round-5 verdict: `librccl` had never been loaded by this code): PHZ_DIST_FORCE_COLLECTIVES=1
    makes a single rank take the whole multi-rank path -- process group over backend "nccl" (= RCCL) with world size 1, the dense AS histogram
    all-reduced on the device, the noise counters all-reduced, the broadcasts and int64 all-gathers of the fragment tables, row text through the
    spool file and spliced by byte ranges, the closing barrier -- through the CLI on the two-BAM / two-chromosome fixture.  The five files must be
    the reference's, and with nccl the process must really have RCCL mapped."""
    import subprocess
    d = os.path.join(GOLD, "pipe_two")
    bams = []
    for b in ("t1", "t2"):
        p = tmp_path / (b + ".sam")
        p.write_text("".join(gz_text(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")))
        bams.append(str(p))
    prefix = str(tmp_path / "out")
    code = ("import sys\nfrom phaser_amd import phaser, dist as pdist\nrc = phaser.main(sys.argv[1:])\nimport torch.distributed as td\n"
            "print('PG', td.is_initialized() and td.get_backend(), 'WORLD', td.get_world_size(), 'LIVE', pdist.collectives_live())\n"
            "print('RCCL_MAPPED', 'librccl' in open('/proc/self/maps').read())\nsys.exit(rc)\n")
    env = dict(os.environ, PHZ_DIST_FORCE_COLLECTIVES="1", PHZ_DIST_BACKEND=backend, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + (os.getpid() % 1000)),
               PYTHONPATH=REPO)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code, "--vcf", os.path.join(d, "in.vcf"), "--bam", ",".join(bams), "--sample", "S1", "--mapq", "255", "--baseq", "10",
                        "--paired_end", "1", "--o", prefix, "--write_vcf", "0", "--threads", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert "PG %s WORLD 1 LIVE True" % backend in r.stdout, r.stdout
    if backend == "nccl":
        assert "RCCL_MAPPED True" in r.stdout, r.stdout
    out = {name: open(prefix + "." + name + ".txt").read().replace("\tt1.sam\t", "\tt1\t").replace("\tt2.sam\t", "\tt2\t") for name in OUTPUTS}
    compare(out, d)
    assert not [f for f in os.listdir(tmp_path) if f.startswith("phz_spool_")]          # spool file removed


def test_as_histogram_sparse_equals_dense(mapper):
    """phz_as_histogram_sparse (one rank: the histogram stays on the device, its occupied bins come back) = the dense 64 Ki-bin histogram of
    phz_as_histogram_batch, on a shard whose alignment scores are spread over thousands of values; more occupied bins than the caller's room
    is reported, not truncated."""
    import ctypes as C
    import numpy as np
    from phaser_amd import _lib
    g = torch.Generator().manual_seed(99)
    n_reads, n_lines = 50_000, 400_000
    aln = torch.randint(-1500, 1500, (n_reads,), generator=g, dtype=torch.int32)
    aln[:100] = torch.arange(-32768, -32768 + 100, dtype=torch.int32)
    has = (torch.rand(n_reads, generator=g) > 0.1).to(torch.uint8)
    read_idx = torch.sort(torch.randint(0, n_reads, (n_lines,), generator=g, dtype=torch.int32)).values
    dev = torch.device("cuda:0")
    t = {k: v.to(dev) for k, v in dict(read_idx=read_idx, var_idx=torch.zeros(n_lines, dtype=torch.int32), code=torch.zeros(n_lines, dtype=torch.uint8),
                                       qid=torch.zeros(n_reads, dtype=torch.int32), aln=aln, has=has).items()}
    p = lambda x: C.c_void_p(x.data_ptr())
    ln = _lib.phz_lines(n_lines, p(t["read_idx"]), p(t["var_idx"]), p(t["code"]), n_reads, p(t["qid"]), p(t["aln"]), p(t["has"]), 0.0, 0, 0, 0, 0)
    arr = (_lib.phz_lines * 2)(ln, ln)
    ctx = mapper.ctx
    dense = torch.zeros(_lib.PHZ_AS_BINS, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.phz_as_histogram_batch(ctx.h, arr, 2, p(dense)))
    h = dense.cpu().numpy()
    want = np.bincount(aln.numpy()[read_idx.numpy()][has.numpy()[read_idx.numpy()] != 0].astype(np.int64) + 32768, minlength=65536) * 2
    assert np.array_equal(h, want)
    bins = np.empty(4096, np.int32); counts = np.empty(4096, np.int64); k = C.c_int32(0)
    st = ctx.lib.phz_as_histogram_sparse(ctx.h, arr, 2, 4096, C.c_void_p(bins.ctypes.data), C.c_void_p(counts.ctypes.data), C.byref(k))
    nz = np.flatnonzero(h)
    if len(nz) <= 4096:
        assert st == 0 and k.value == len(nz)
        assert np.array_equal(bins[:k.value], nz) and np.array_equal(counts[:k.value], h[nz])
    else:
        assert st == _lib.PHZ_E_CAPACITY and k.value == len(nz)
    st = ctx.lib.phz_as_histogram_sparse(ctx.h, arr, 2, 64, C.c_void_p(bins.ctypes.data), C.c_void_p(counts.ctypes.data), C.byref(k))
    assert st == _lib.PHZ_E_CAPACITY and k.value == len(nz)


@pytest.mark.parametrize("dtype,ranges,n", [("uint32", [(0, 21)], 1_500_000), ("uint64", [(0, 21), (32, 54)], 1_460_000), ("uint32", [(0, 32)], 3_000_001), ("uint64", [(0, 40)], 70_000),
                                             ("uint32", [(0, 18)], 5)])
def test_device_sort_matches_a_stable_host_sort(mapper, dtype, ranges, n):
    """The radix sort behind the row stage's ordering rules (phz_sort.h: one launch per pass, tiles of 4,096 keys in ticket order, decoupled look-back over the
    tiles' digit counts) at the sizes of a genome's passes (1.5 M variants, 1.46 M pairs), on the GPU: equal to numpy's stable sort and to the three-launch
    passes it replaced.  Hundreds of tiles really run concurrently here, which the emulation (tests/test_emu_sort.py) cannot show."""
    import ctypes as C
    ctx = mapper.ctx
    rng = np.random.default_rng(23)
    mask = 0
    for lo_, hi_ in ranges:
        mask |= ((1 << (hi_ - lo_)) - 1) << lo_
    keys = (rng.integers(0, 1 << 62, size=n, dtype=np.uint64) & np.uint64(mask)).astype(dtype)
    keys[rng.integers(0, n, n // 4)] = keys[0]
    vals = np.arange(n, dtype=np.uint32)
    order = np.arange(n)
    for lo_, hi_ in ranges:
        d = (keys[order].astype(np.uint64) >> np.uint64(lo_)) & np.uint64((1 << (hi_ - lo_)) - 1)
        order = order[np.argsort(d, kind="stable")]
    rg = np.asarray(ranges, dtype=np.int32).reshape(-1)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    for three in (0, 1):
        for rep in range(3 if three == 0 else 1):
            ko = np.empty_like(keys); vo = np.empty_like(vals)
            ctx.check(ctx.lib.phz_selftest_sort(ctx.h, keys.dtype.itemsize, vp(keys), vp(vals), n, vp(rg), len(ranges), three, vp(ko), vp(vo)))
            assert np.array_equal(vo, vals[order]) and np.array_equal(ko, keys[order]), (three, rep)


@pytest.mark.parametrize("strip", ["none", "some", "second_bam"])
def test_as_cutoff_on_the_device_equals_the_host_percentile(mapper, monkeypatch, strip):
    """close_bam leaves the AS percentile on the device (phz_as_cutoff_enqueue: histogram -> order statistics -> numpy's interpolation in one workgroup, no host
    wait; the tally kernels read the block) -- against the same pass with the percentile taken on the host (PHZ_AS_CUTOFF_HOST=1: phz_as_cutoff), which is pinned
    against numpy.percentile.  Fixture pipe_two as written by the reference; with the AS tag stripped from a third of the records; with the second BAM
    carrying no AS tag at all ('no alignment score value found in reads, cannot use cutoff', phaser.py:553: every line of that BAM is kept).  Same five files,
    same log lines; as_q_cutoff 0.37 puts the percentile between two order statistics (gamma != 0)."""
    import re
    d = os.path.join(GOLD, "pipe_two")
    bams = {}
    for bi, b in enumerate(("t1", "t2")):
        bams[b + ".bam"] = {}
        for c in ("chr21", "chr22"):
            text = gz_text(os.path.join(d, "%s.%s.sam.gz" % (b, c)))
            if strip == "some":
                lines = text.split("\n")
                text = "\n".join(re.sub(r"\tAS:i:-?\d+", "", l) if (i % 3 == 0 and not l.startswith("@")) else l for i, l in enumerate(lines))
            elif strip == "second_bam" and bi == 1:
                text = re.sub(r"\tAS:i:-?\d+", "", text)
            bams[b + ".bam"][c] = text
    vcf_text = open(os.path.join(d, "in.vcf")).read()
    runs = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("PHZ_AS_CUTOFF_HOST", "1")
        else:
            monkeypatch.delenv("PHZ_AS_CUTOFF_HOST", raising=False)
        for q in (0.05, 0.37):
            out, eng = run_product(mapper, vcf_text, bams, "cuda", as_q_cutoff=q)
            assert all(isinstance(l, str) for l in eng.log)
            runs[(mode, q)] = (out, [l for l in eng.log if "alignment score" in l])
    for q in (0.05, 0.37):
        assert runs[("device", q)][1] == runs[("host", q)][1] and len(runs[("device", q)][1]) == 2
        for name in OUTPUTS:
            assert runs[("device", q)][0][name] == runs[("host", q)][0][name], (name, q)
    if strip == "none":
        compare(runs[("device", 0.05)][0], d)
    if strip == "second_bam":
        assert "cannot use cutoff" in runs[("device", 0.05)][1][1] and "using alignment score cutoff" in runs[("device", 0.05)][1][0]


def test_as_value_outside_int16_is_refused_by_the_pass(mapper, monkeypatch):
    """An AS tag beyond int16 (the histogram's band): the device-side percentile flags it in its block and phz_tally refuses the input at its first host wait
    (PHZ_E_UNSUPPORTED), exactly as the host percentile refuses it in close_bam (PHZ_AS_CUTOFF_HOST=1)."""
    import re
    from phaser_amd import _lib
    d = os.path.join(GOLD, "pipe_one")
    sam = gz_text(os.path.join(d, "a.chr22.sam.gz"))
    lines = sam.split("\n")
    for k, l in enumerate(lines):          # (the histogram is over the records of the call lines: every 20th record, so that some of them carry a call)
        if k % 20 == 0 and not l.startswith("@") and "AS:i:" in l:
            lines[k] = re.sub(r"AS:i:-?\d+", "AS:i:70000", l)
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("PHZ_AS_CUTOFF_HOST", "1")
        with pytest.raises(_lib.PhzError) as e:
            run_product(mapper, open(os.path.join(d, "in.vcf")).read(), {"a.bam": {"chr22": "\n".join(lines)}}, "cuda")
        assert e.value.status == _lib.PHZ_E_UNSUPPORTED and "int16" in str(e.value), mode
    monkeypatch.delenv("PHZ_AS_CUTOFF_HOST", raising=False)
    out, eng = run_product(mapper, open(os.path.join(d, "in.vcf")).read(), {"a.bam": {"chr22": sam}}, "cuda")      # the ctx stays usable
    compare(out, d)


def test_copy_as_written_on_later_passes(mapper, monkeypatch):
    """Passes after the first over one variant set let phz_rowsdev_run copy the finished text into a page-locked region beside its last writer kernels (events per file,
    second stream, allele_config first): the bytes of pass 2 and 3 equal pass 1's (texts fetched after the run) and the reference's (fixture pipe_two); switched off: the same."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from phasing_oracle import bam_display_names
    from phaser_amd import rowsdev, samio, vcf
    from phaser_amd.engine import Config, Engine
    d = os.path.join(GOLD, "pipe_two")
    vs = vcf.load_variants(open(os.path.join(d, "in.vcf")).read())
    bams = {b + ".bam": {c: gz_text(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")} for b in ("t1", "t2")}

    def one_pass():
        eng = Engine(vs, bam_display_names(list(bams.keys())), Config(), mapper=mapper)
        interners = {}
        for bi, (bam, per_chrom) in enumerate(bams.items()):
            for chrom in vs.chroms:
                for c2, sh in samio.shards_from_sam(per_chrom[chrom], interners, 0.0).items():
                    eng.add_shard(bi, c2, sh.to("cuda"), len(interners[c2]), interners[c2].names)
            for c2 in interners:
                eng.n_qid[c2] = len(interners[c2])
            eng.close_bam(bi)
        out = eng.finish()
        assert eng.rows_path == "device"
        return out, eng
    first, e1 = one_pass()
    assert rowsdev.tables_for(e1).__dict__.get("_text_total", 0) > 0
    second, _ = one_pass()
    third, _ = one_pass()
    monkeypatch.setenv("PHZ_ROWS_COPY_AS_WRITTEN", "0")
    fourth, _ = one_pass()
    for name in OUTPUTS:
        assert first[name] == second[name] == third[name] == fourth[name], name
    compare(first, d)
