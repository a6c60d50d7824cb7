"""GPU parity at the STATED sizes of BASELINE.json's configs (round-4 verdict item 1), all on one MI355X:
  * configs[1]  chr1, 40k het SNPs, one 50M-record BAM: every record against the C mapper oracle, the properties of the call list, the
                invariants that tie the five files together, and chr1's five files against the pinned phasing oracle;
  * configs[2]  whole genome, 22 chromosome shards, ~80M records, ~1.5M het SNPs, one BAM;
  * configs[3]  the same sample with FOUR 80M-record BAMs whose QNAMEs collide (320M records, 37 GB of shards): every record of every
                BAM against the mapper oracle (one BAM's host copy at a time), invariants over the genome, the five files of two
                chromosomes against the phasing oracle;
  * configs[4]  one GPU's loop of the 128-sample batch: full-size samples (80M records each, different variant / read sets) streamed
                through ONE device context, per-sample mapper parity on every record and five-file parity on a sampled chromosome
                (16 samples, one GPU's share: tools/run_c5.py, log under profiles/), plus the small varying-size stream with
                phaser_gene_ae / phaser_expr_matrix behind it.
Each test combines a bit-exact check of K_map's call list for EVERY record against the C mapper oracle run on all host cores, the
size-independent relations between the five output files, and full equality with oracle/phasing_oracle.py (canonical form) on whole
chromosomes at full size.  The phasing oracle is slow (minutes per chromosome at these depths), so those comparisons run as worker
processes (tools/oracle_chrom_worker.py) started as soon as their call files exist and are joined by the LAST test of the module.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO
from helpers import OUTPUTS, call_text, canonical, oracle_map_readbatch

pytestmark = pytest.mark.gpu

JOBS = []          # phasing-oracle workers in flight: (label, Popen, phased variants of the product, sha256 of the product's five files in canonical form, start time)


def start_oracle_job(label, tmp_path, call_texts, got, phased, names):
    """Start oracle/phasing_oracle.py on the call files of one run (one text per BAM) in a worker process; the product's five files `got`
    are reduced to the hash the worker prints.  Joined by test_zz_phasing_oracle_jobs."""
    import hashlib
    import subprocess
    import time
    d = tmp_path / ("oracle_" + label); d.mkdir()
    paths = []
    for b, t in enumerate(call_texts):
        f = d / ("calls%d.tsv" % b); f.write_text(t); paths.append(str(f))
    h = hashlib.sha256()
    for name in OUTPUTS:
        h.update(canonical(name, got[name]).encode())
    pr = subprocess.Popen([sys.executable, os.path.join(REPO, "tools", "oracle_chrom_worker.py"), ",".join(paths), "10", "-", "-", "-", ",".join(names)],
                          stdout=subprocess.PIPE, text=True)
    JOBS.append((label, pr, phased, h.hexdigest(), time.time(), sum(t.count("\n") for t in call_texts)))


@pytest.fixture(scope="module")
def mapper():
    from phaser_amd.mapper import Mapper
    return Mapper(0)


def build(plan, n_bams, keep):
    """-> variants per chromosome, shards[bam][chrom], samples[bam][chrom] (host prefix of `keep` records)."""
    from phaser_amd import workloads
    vsets = {}; shards = [dict() for _ in range(n_bams)]; samples = [dict() for _ in range(n_bams)]
    for chrom, ln, n_snps, n_rec, seed in plan:
        for b in range(n_bams):
            v, sh, smp = workloads.make_shard(chrom, ln, n_snps, n_rec, seed, "cuda:0", keep_sample=keep, read_seed=seed + 1 + 7919 * b)
            vsets[chrom] = v; shards[b][chrom] = sh; samples[b][chrom] = smp
    return vsets, shards, samples


def run_engine(mapper, vsets, shards, plan, names=None, **cfg):
    from phaser_amd import synth, vcf
    from phaser_amd.engine import Config, Engine
    vs = vcf.load_variants("\n".join(synth.vcf_lines([vsets[p[0]] for p in plan])))
    eng = Engine(vs, names or ["bam%d" % b for b in range(len(shards))], Config(want_vcf=False, **cfg), mapper=mapper)
    for b, per_chrom in enumerate(shards):
        # QNAME ids are shared by the BAMs of a chromosome: id k of every BAM is the same template name "q<k>"
        eng.add_shards(b, [(c, sh, int(sh.qid.max()) + 1) for c, sh in per_chrom.items()])
        eng.close_bam(b)
    return eng, eng.finish()


def check_oracle_prefix(oracle_build, eng, vsets, samples, plan):
    for b, per_chrom in enumerate(samples):
        for chrom, *_ in plan:
            smp = per_chrom[chrom]
            o_r, o_v, o_c, _ = oracle_map_readbatch(oracle_build, smp, vsets[chrom].pos.numpy(), 10, with_text=False)
            calls = eng.shards[chrom][b].calls
            m = len(o_r)
            assert m > 0
            got_r = calls.read_idx[:m + 1].cpu().numpy()
            assert np.array_equal(got_r[:m], o_r), (chrom, b)
            assert np.array_equal(calls.var_idx[:m].cpu().numpy(), o_v), (chrom, b)
            assert np.array_equal(calls.code[:m].cpu().numpy(), o_c), (chrom, b)
            assert calls.n == m or int(got_r[m]) >= len(smp)          # the next call belongs to a record past the prefix


def check_oracle_all_records(oracle_build, eng, vsets, samples, plan):
    """Every record of every shard (host copies kept by build()): the C oracle maps them on all host cores, the call list must equal K_map's."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from full_parity_c3 import oracle_all_records
    from phaser_amd import dist as pdist
    cores = max(1, pdist.effective_cpus())
    total = 0
    for b, per_chrom in enumerate(samples):
        for chrom, *_ in plan:
            smp = per_chrom[chrom]
            o_r, o_v, o_c = oracle_all_records(oracle_build, smp, vsets[chrom].pos.numpy(), 10, cores)
            calls = eng.shards[chrom][b].calls
            assert calls.n == len(o_r), (chrom, b)
            assert np.array_equal(calls.read_idx.cpu().numpy(), o_r) and np.array_equal(calls.var_idx.cpu().numpy(), o_v) and \
                np.array_equal(calls.code.cpu().numpy(), o_c), (chrom, b)
            total += len(smp)
    return total


def check_invariants(eng, out, plan, n_bams):
    rows = lambda name: [l.split("\t") for l in out[name].split("\n")[1:] if l]
    kept = 0
    for chrom, *_ in plan:
        R = eng.chrom_view(chrom)
        assert (R["var_distinct"] <= R["var_count"]).all()
        kept += R["kept"]
    assert kept == eng.total_lines == eng.G["n_kept"]
    order = {p[0]: i for i, p in enumerate(plan)}
    al = rows("allelic_counts")
    assert all(int(r[5]) + int(r[6]) == int(r[7]) for r in al)
    # allelic_counts: per first BAM the chromosomes in VCF order (rule 2); inside one chromosome of one BAM first-appearance order
    seq = [order[r[0]] for r in al]
    drops = sum(1 for x, y in zip(seq, seq[1:]) if y < x)
    assert drops <= n_bams - 1
    for chrom in (plan[0][0], plan[-1][0]):
        R = eng.chrom_view(chrom)
        idx = {u: i for i, u in enumerate(eng.vs.chroms[chrom].uid)}
        for r in [x for x in al if x[0] == chrom][::211]:
            i = idx[r[2]]
            assert (int(r[5]), int(r[6])) == (int(R["var_distinct"][i][0]), int(R["var_distinct"][i][1]))
    hap = rows("haplotypes")
    blocks = [r for r in hap if int(r[4]) > 1]
    # blocks: chromosomes in VCF order (rule 4 + :863-867), every phased variant in exactly one block
    bseq = [order[r[0]] for r in blocks]
    assert bseq == sorted(bseq)
    phased_ids = [(r[0], x) for r in blocks for x in r[5].split(",")]          # names are rsids: unique per chromosome only
    assert len(phased_ids) == len(set(phased_ids)) == eng.phased
    assert sum(int(r[4]) * (int(r[4]) - 1) for r in blocks) == len(rows("allele_config"))
    assert all(int(r[7]) + int(r[8]) == int(r[9]) for r in hap)
    ase = rows("haplotypic_counts")
    assert all(int(r[9]) + int(r[10]) == int(r[11]) for r in ase)
    for r in [x for x in ase if int(x[4]) > 1][::307]:
        la = set(t for g in r[16].split(";") for t in g.split(",") if t); lb = set(t for g in r[17].split(";") for t in g.split(",") if t)
        assert len(la) == int(r[9]) and len(lb) == int(r[10])
    conn = rows("variant_connections")
    assert all(int(r[2]) <= int(r[3]) for r in conn)
    assert len(conn) == int(eng.G["linked"].sum())


def check_replica_vs_oracle(mapper, plan_small, n_bams, min_phased=1000):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import phasing_oracle as po
    from phaser_amd import synth
    vsets, shards, _ = build(plan_small, n_bams, 0)
    eng, got = run_engine(mapper, vsets, shards, plan_small, host_threads=4)
    ph = po.Phaser(["bam%d" % b for b in range(n_bams)])
    for b in range(n_bams):
        ph.add_bam([call_text(vsets[p[0]], shards[b][p[0]], eng.shards[p[0]][b].calls) for p in plan_small])
    want = ph.finish()
    for name in OUTPUTS:
        assert canonical(name, got[name]) == canonical(name, want[name]), name
    assert eng.phased == ph.phased and eng.phased > min_phased


def oracle_calls_of(oracle_build, smp, vpos, cores):
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from full_parity_c3 import oracle_all_records
    return oracle_all_records(oracle_build, smp, vpos, 10, cores)


def same_calls(calls, o):
    return calls.n == len(o[0]) and np.array_equal(calls.read_idx.cpu().numpy(), o[0]) and np.array_equal(calls.var_idx.cpu().numpy(), o[1]) and \
        np.array_equal(calls.code.cpu().numpy(), o[2])


def test_configs1_chr1_50m_records(mapper, oracle_build, tmp_path):
    """configs[1]: chr1 full, 40k het SNPs, ONE 50M-record BAM.  All 50M records against the C oracle (bit-exact call list), the properties of
    the list (mapper order, idempotence, split invariance), the cross-file invariants, and chr1's five files (~6M call lines over 40k variants:
    read sets of hundreds of QNAMEs per variant) against the phasing oracle (worker process, joined at the end of the module)."""
    from phaser_amd import dist as pdist, workloads
    v, shard, sample = workloads.make_shard("chr1", workloads.CHR1_LEN, 40_000, 50_000_000, 20240807, "cuda:0", keep_sample=1 << 40)
    assert len(sample) == shard.n >= 49_000_000
    vpos = v.pos.to("cuda:0")
    a = mapper.map(shard, vpos, 10)
    o = oracle_calls_of(oracle_build, sample, v.pos.numpy(), max(1, pdist.effective_cpus()))
    assert same_calls(a, o), "K_map != C oracle on the 50M records of configs[1]"
    del sample, o
    b = mapper.map(shard, vpos, 10)
    assert a.n == b.n > 5_000_000
    for f in ("read_idx", "var_idx", "code", "aux0", "aux1"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f                        # idempotence
    key = a.read_idx.to(torch.int64) * (len(v) + 1) + a.var_idx.to(torch.int64)
    assert bool((key[1:] > key[:-1]).all())                                        # (record, variant) order, no duplicates
    mid = shard.n // 2 + 12345
    lo = mapper.map(shard.slice(0, mid), vpos, 10); hi = mapper.map(shard.slice(mid, shard.n), vpos, 10)
    assert lo.n + hi.n == a.n
    assert torch.equal(torch.cat([lo.read_idx, hi.read_idx + mid]), a.read_idx)
    assert torch.equal(torch.cat([lo.var_idx, hi.var_idx]), a.var_idx) and torch.equal(torch.cat([lo.code, hi.code]), a.code)
    del b, lo, hi, key
    plan1 = [("chr1", workloads.CHR1_LEN, 40_000, shard.n, 20240807)]
    eng, out = run_engine(mapper, {"chr1": v}, [{"chr1": shard}], plan1, names=["big"], host_threads=16)
    assert eng.rows_path == "device", getattr(eng, "rows_fallback", "")
    assert torch.equal(eng.shards["chr1"][0].calls.read_idx, a.read_idx) and torch.equal(eng.shards["chr1"][0].calls.code, a.code)
    check_invariants(eng, out, plan1, 1)
    assert eng.total_lines > 5_000_000 and eng.phased > 10_000
    start_oracle_job("configs1_chr1", tmp_path, [call_text(v, shard, eng.shards["chr1"][0].calls)], out, eng.phased, ["big"])


def test_whole_genome_one_bam(mapper, oracle_build, tmp_path):
    """configs[2]: 22 chromosome shards, 80M records, 1.5M het SNPs."""
    from phaser_amd import workloads
    plan = workloads.genome_plan()
    vsets, shards, samples = build(plan, 1, 1 << 40)            # host copies of ALL records (13 GB) for the mapper oracle
    eng, out = run_engine(mapper, vsets, shards, plan, host_threads=16)
    assert eng.rows_path == "device", getattr(eng, "rows_fallback", "")
    assert sum(sh.n for sh in shards[0].values()) > 79_000_000 and eng.vs.het_count > 1_400_000
    # chr1 at full size through the pinned phasing oracle (a worker process) while the mapper check below runs
    check_invariants(eng, out, plan, 1)              # reads the ctx's resident tally: before any other pass on this mapper
    big = plan[0][0]
    one, got1 = run_engine(mapper, {big: vsets[big]}, [{big: shards[0][big]}], plan[:1], names=["bench"])
    assert one.phased > 50_000
    start_oracle_job("configs2_chr1", tmp_path, [call_text(vsets[big], shards[0][big], eng.shards[big][0].calls)], got1, one.phased, ["bench"])
    assert check_oracle_all_records(oracle_build, eng, vsets, samples, plan) > 79_000_000
    del eng, out, shards, samples, one, got1
    torch.cuda.empty_cache()
    check_replica_vs_oracle(mapper, workloads.genome_plan(scale=0.02), 1)


def test_whole_genome_four_bams_full_size(mapper, oracle_build, tmp_path):
    """configs[3] at its stated size on one GPU: the sample of configs[2] with FOUR BAMs of 80M records each (320M records, 37 GB of shards) whose
    QNAME ids collide, so the cross-BAM merge (last BAM owns a QNAME's read_vars list, phaser.py:558-581) runs on every chromosome.  Every record
    of every BAM against the C mapper oracle -- one BAM's host copy (13 GB) at a time --, the invariants over the genome, and the five files of
    chr21 and of chr22 with all four BAMs against the phasing oracle (workers)."""
    from phaser_amd import dist as pdist, workloads
    plan = workloads.genome_plan()
    cores = max(1, pdist.effective_cpus())
    n_bams = 4
    vsets = {}; shards = [dict() for _ in range(n_bams)]; direct = [dict() for _ in range(n_bams)]
    total = 0
    for b in range(n_bams):
        for chrom, ln, n_snps, n_rec, seed in plan:
            v, sh, smp = workloads.make_shard(chrom, ln, n_snps, n_rec, seed, "cuda:0", keep_sample=1 << 40, read_seed=seed + 1 + 7919 * b)
            vsets[chrom] = v; shards[b][chrom] = sh
            c = mapper.map(sh, v.pos, 10)
            assert same_calls(c, oracle_calls_of(oracle_build, smp, v.pos.numpy(), cores)), (chrom, b)
            direct[b][chrom] = (c.n, c.read_idx, c.var_idx, c.code)
            total += len(smp)
            del smp
    assert total > 319_000_000
    eng, out = run_engine(mapper, vsets, shards, plan, host_threads=16)
    assert eng.rows_path == "device", getattr(eng, "rows_fallback", "")
    for b in range(n_bams):                         # the batched submission of the pipeline gives the lists that were checked shard by shard
        for chrom, *_ in plan:
            calls = eng.shards[chrom][b].calls; n, r_, v_, c_ = direct[b][chrom]
            assert calls.n == n and torch.equal(calls.read_idx, r_) and torch.equal(calls.var_idx, v_) and torch.equal(calls.code, c_), (chrom, b)
    del direct
    check_invariants(eng, out, plan, n_bams)
    names = ["bam%d" % b for b in range(n_bams)]
    for chrom in ("chr21", "chr22"):
        sub_plan = [p for p in plan if p[0] == chrom]
        one, got1 = run_engine(mapper, {chrom: vsets[chrom]}, [{chrom: shards[b][chrom]} for b in range(n_bams)], sub_plan, names=names)
        assert one.rows_path == "device" and one.phased > 5_000
        start_oracle_job("configs3_" + chrom, tmp_path, [call_text(vsets[chrom], shards[b][chrom], one.shards[chrom][b].calls) for b in range(n_bams)],
                         got1, one.phased, names)
        del one, got1
    del eng, out, shards
    torch.cuda.empty_cache()
    check_replica_vs_oracle(mapper, workloads.genome_plan(total_records=20_000_000, scale=0.02), 4, min_phased=200)


def test_sample_stream_full_size(mapper, oracle_build, tmp_path):
    """configs[4], one rank's loop at the stated size: four whole-genome samples of 80M records each (different variant sets and reads) streamed one
    after the other through ONE device context.  Per sample: every record of every chromosome against the C mapper oracle, the invariants of the
    five files over the genome, and the five files of one chromosome (a different one per sample) against the phasing oracle (worker).  The
    16-sample run of one GPU's share is tools/run_c5.py (profiles/r05/run_c5_16samples.txt)."""
    from phaser_amd import workloads
    sampled = ["chr22", "chr19", "chr21", "chr20"]
    for s in range(4):
        plan = workloads.genome_plan(seed=4000 + 100 * s)
        vsets, shards, samples = build(plan, 1, 1 << 40)
        name = "sample%03d" % s
        eng, out = run_engine(mapper, vsets, shards, plan, names=[name], host_threads=16)
        assert eng.rows_path == "device", getattr(eng, "rows_fallback", "")
        assert check_oracle_all_records(oracle_build, eng, vsets, samples, plan) > 79_000_000
        del samples
        check_invariants(eng, out, plan, 1)
        c = sampled[s]
        one, got1 = run_engine(mapper, {c: vsets[c]}, [{c: shards[0][c]}], [p for p in plan if p[0] == c], names=[name])
        assert one.phased > 5_000
        start_oracle_job("configs4_%s_%s" % (name, c), tmp_path, [call_text(vsets[c], shards[0][c], one.shards[c][0].calls)], got1, one.phased, [name])
        del eng, out, shards, one, got1
        torch.cuda.empty_cache()


def test_sample_stream_configs4_shape(mapper, tmp_path):
    """configs[4] shape on one rank: whole-genome samples (22 chromosomes each, different variant sets, read sets and SIZES) streamed
    one after the other through ONE device context -- the per-rank loop of the 128-sample batch (tools/run_c5.py) -- each through
    the hot path, phaser_gene_ae and, at the end, phaser_expr_matrix.  Every sample's five files must equal the pinned phasing
    oracle's (no state of sample k may leak into sample k+1: the scratch buffers, staging slots and shard tables are reused at other
    sizes), every gene table the pinned gene_ae oracle's, and the matrices must carry exactly those tables' cells."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import phasing_oracle as po
    import gene_ae_oracle as go
    from phaser_amd import expr_matrix, gene_ae, workloads
    scales = [0.008, 0.02, 0.004, 0.012]              # growing and shrinking: buffer reuse both ways
    gdir = tmp_path / "gene_ae"; gdir.mkdir()
    bed = None; tables = {}
    for s, scale in enumerate(scales):
        plan = workloads.genome_plan(scale=scale, seed=4000 + 100 * s)
        vsets, shards, _ = build(plan, 1, 0)
        eng, got = run_engine(mapper, vsets, shards, plan, names=["sample%03d" % s], host_threads=4)
        ph = po.Phaser(["sample%03d" % s])
        ph.add_bam([call_text(vsets[p[0]], shards[0][p[0]], eng.shards[p[0]][0].calls) for p in plan])
        want = ph.finish()
        for name in OUTPUTS:
            assert canonical(name, got[name]) == canonical(name, want[name]), (s, name)
        assert eng.phased == ph.phased and eng.phased > 300
        hc = got["haplotypic_counts"]
        if bed is None:                               # one gene model for the batch: merged spans of the first sample's rows
            spans = {}
            for line in hc.split("\n")[1:]:
                if line:
                    c = line.split("\t", 3); spans.setdefault(c[0], []).append((int(c[1]) - 1, int(c[2])))
            feats = []
            for chrom, sp in spans.items():
                sp.sort(); a0, b0 = sp[0]
                for a, b in sp[1:]:
                    if a - b0 < 5000: b0 = max(b0, b)
                    else: feats.append((chrom, a0, b0)); a0, b0 = a, b
                feats.append((chrom, a0, b0))
            bed = "".join("%s\t%d\t%d\tg%d\n" % (c, max(0, a - 50), b + 50, k) for k, (c, a, b) in enumerate(feats))
            (tmp_path / "genes.bed").write_text(bed)
        table = gene_ae.gene_ae(hc.encode(), bed, ctx=mapper.ctx)
        assert go.canonical(table) == go.canonical(go.gene_ae(hc, bed)), s
        (gdir / ("sample%03d.gene_ae.txt" % s)).write_text(table)
        tables["sample%03d" % s] = {r.split("\t")[3]: r.split("\t") for r in table.split("\n")[1:] if r}
        del eng, got, shards
        torch.cuda.empty_cache()
    a, g, log = expr_matrix.expr_matrix(str(gdir), str(tmp_path / "genes.bed"))
    assert not log
    rows = [r.split("\t") for r in a.split("\n") if r]
    assert rows[0][4:] == sorted(tables) and len(rows) - 1 == len(bed.splitlines())
    hdr = open(str(gdir / "sample000.gene_ae.txt")).readline().rstrip("\n").split("\t")
    ia, ib = hdr.index("aCount"), hdr.index("bCount")
    for r in rows[1:]:
        for k, smp in enumerate(sorted(tables)):
            t = tables[smp].get(r[3])
            assert t is not None and r[4 + k] == "%s|%s" % (t[ia], t[ib]), (r[3], smp)


def test_zz_phasing_oracle_jobs():
    """Joins the phasing-oracle workers the tests above started: the five files of every whole chromosome they were given must equal the product's
    (canonical form; same phased-variant count)."""
    import time
    assert JOBS, "no oracle job was started"
    bad = []
    for label, pr, phased, sha, t0, n_lines in JOBS:
        res = pr.communicate()[0].split()
        print("oracle job %-28s %8d call lines  %6d phased variants  oracle %6.1f CPU-s, done %4.0f s after its start: %s"
              % (label, n_lines, phased, float(res[1]) if len(res) > 1 else -1, time.time() - t0,
                 "identical" if pr.returncode == 0 and len(res) == 3 and int(res[0]) == phased and res[2] == sha else "DIFFERENT"))
        if not (pr.returncode == 0 and len(res) == 3 and int(res[0]) == phased and res[2] == sha):
            bad.append(label)
    del JOBS[:]
    assert not bad, "the five files differ from the phasing oracle: %s" % bad
