#!/usr/bin/env python3
"""Runs ON THE GPU BOX.  BASELINE.json configs[4] -- "128 synthetic samples x whole genome streamed through 8 x MI355X" -- as ONE GPU's share: S whole-genome
samples (default 16 = 128 / 8, full size: ~80M records and ~1.5M het SNPs each, different variant sets and reads per sample) streamed one after the other
through ONE device context, each through the hot path (resident shards -> the five files), phaser_gene_ae and, at the end, phaser_expr_matrix over all samples.
The samples of a rank are independent (no collective): the 8-GPU job is eight of these loops.

Parity, per sample (the reference's per-sample loop is phaser.py main() once per sample, phaser_pop's batch scripts):
  * K_map's call list of EVERY record of EVERY chromosome against the C mapper oracle (oracle/rvm_oracle.c on all host cores);
  * the five files of one whole chromosome (a different one per sample) against oracle/phasing_oracle.py (worker processes, joined at the end),
    from a product run on that chromosome alone (its own AS cutoff / noise level, like the oracle's);
  * cross-file invariants of the genome-wide five files (row arithmetic, block membership, allele_config cardinality).
Synthetic shards are generated in HBM (not timed); the timed part of a sample is VCF table -> Engine -> K_map -> phasing pass -> text in host memory.
usage: tools/run_c5.py [samples=16] [scale=1.0] [threads=32] [check=1]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np
import torch
from phaser_amd import dist as pdist, expr_matrix, gene_ae, synth, vcf, workloads
from phaser_amd.engine import Engine, Config
from phaser_amd.mapper import Mapper

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 32
check = (int(sys.argv[4]) if len(sys.argv) > 4 else 1) != 0


def invariants(eng, out, plan):
    rows = lambda name: [l.split("\t") for l in out[name].split("\n")[1:] if l]
    al = rows("allelic_counts")
    assert all(int(r[5]) + int(r[6]) == int(r[7]) for r in al)
    hap = rows("haplotypes")
    blocks = [r for r in hap if int(r[4]) > 1]
    order = {p[0]: i for i, p in enumerate(plan)}
    bseq = [order[r[0]] for r in blocks]
    assert bseq == sorted(bseq)
    ids = [(r[0], x) for r in blocks for x in r[5].split(",")]
    assert len(ids) == len(set(ids)) == eng.phased
    assert sum(int(r[4]) * (int(r[4]) - 1) for r in blocks) == len(rows("allele_config"))
    ase = rows("haplotypic_counts")
    assert all(int(r[9]) + int(r[10]) == int(r[11]) for r in ase)
    conn = rows("variant_connections")
    assert all(int(r[2]) <= int(r[3]) for r in conn)
    return len(al), len(blocks), len(ase), len(conn)


def main():
    from helpers import OUTPUTS, call_text, canonical
    from full_parity_c3 import oracle_all_records
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
    mapper = Mapper(0)
    cores = max(1, pdist.effective_cpus())
    tmp = tempfile.mkdtemp(prefix="phz_c5_")
    out_dir = os.path.join(tmp, "out"); os.makedirs(out_dir + "/gene_ae")
    bed_path = out_dir + "/genes.bed"
    sampled = ["chr22", "chr21", "chr20", "chr19", "chr18", "chr17", "chr16", "chr15"]
    jobs = []
    t_path = t_gene = 0.0; n_rec = n_phased = n_calls_total = 0
    t_all = time.perf_counter()
    print("configs[4], one GPU's share: %d samples at scale %.2f through one device context; oracle on %d host CPUs; parity checks %s" % (S, scale, cores, "on" if check else "OFF"), flush=True)
    for s in range(S):
        plan = workloads.genome_plan(seed=9000 + 100 * s, scale=scale)
        vsets = {}; shards = {}; samples = {}
        t0 = time.perf_counter()
        for chrom, ln, n_snps, n, seed in plan:
            v, shard, smp = workloads.make_shard(chrom, ln, n_snps, n, seed, "cuda:0", keep_sample=(1 << 40) if check else 0)
            vsets[chrom] = v; shards[chrom] = shard; samples[chrom] = smp
        vtext = "\n".join(synth.vcf_lines([vsets[p[0]] for p in plan]))
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        # ---- timed: the hot path of one sample (VCF table -> K_map over the 22 resident shards -> phasing pass -> the five files' text in host memory)
        name = "sample%03d" % s
        t0 = time.perf_counter()
        vs = vcf.load_variants(vtext, threads=threads)
        eng = Engine(vs, [name], Config(host_threads=threads, want_vcf=False), mapper=mapper)
        eng.add_shards(0, [(c, sh, int(sh.qid.max()) + 1) for c, sh in shards.items()])
        eng.close_bam(0)
        files = eng.finish(chunks=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for fname, body in files.items():
            with open("%s/%s.%s.txt" % (out_dir, name, fname), "wb") as f:
                f.writelines(body)
        hc = b"".join(bytes(x) for x in files["haplotypic_counts"])
        out = {k: b"".join(bytes(x) for x in body).decode() for k, body in files.items()}
        assert eng.rows_path == "device", getattr(eng, "rows_fallback", "")
        if s == 0:       # one gene model for the batch: merged spans of the first sample's rows
            spans = {}
            for line in hc.split(b"\n")[1:]:
                if line:
                    c = line.split(b"\t", 3); spans.setdefault(c[0].decode(), []).append((int(c[1]) - 1, int(c[2])))
            feats = []
            for chrom, sp in spans.items():
                sp.sort(); a0, b0 = sp[0]
                for a, b in sp[1:]:
                    if a - b0 < 5000: b0 = max(b0, b)
                    else: feats.append((chrom, a0, b0)); a0, b0 = a, b
                feats.append((chrom, a0, b0))
            open(bed_path, "w").write("".join("%s\t%d\t%d\tg%d\n" % (c, max(0, a - 50), b + 50, k) for k, (c, a, b) in enumerate(feats)))
        bed = open(bed_path).read()
        t2 = time.perf_counter()
        table = gene_ae.gene_ae(hc, bed, ctx=mapper.ctx, threads=threads)
        open("%s/gene_ae/%s.gene_ae.txt" % (out_dir, name), "w").write(table)
        t3 = time.perf_counter()
        recs = sum(sh.n for sh in shards.values()); calls = sum(eng.shards[c][0].calls.n for c in shards)
        t_path += t1 - t0; t_gene += t3 - t2; n_rec += recs; n_phased += eng.phased; n_calls_total += calls
        line = "sample %2d: %d records, %d het SNPs, %d allele calls, %d phased variants | hot path %.3f s (generation %.1f s, not timed), gene_ae %.2f s" % (
            s, recs, vs.het_count, calls, eng.phased, t1 - t0, t_gen, t3 - t2)
        if check:
            tc = time.perf_counter()
            for chrom, *_ in plan:
                o_r, o_v, o_c = oracle_all_records(os.path.join(REPO, "oracle"), samples[chrom], vsets[chrom].pos.numpy(), 10, cores)
                c = eng.shards[chrom][0].calls
                assert c.n == len(o_r) and np.array_equal(c.read_idx.cpu().numpy(), o_r) and np.array_equal(c.var_idx.cpu().numpy(), o_v) and \
                    np.array_equal(c.code.cpu().numpy(), o_c), (s, chrom)
                samples[chrom] = None
            inv = invariants(eng, out, plan)
            ch = sampled[s % len(sampled)]
            from phaser_amd import vcf as pvcf
            one = Engine(pvcf.load_variants("\n".join(synth.vcf_lines([vsets[ch]]))), [name], Config(want_vcf=False), mapper=mapper)
            one.add_shards(0, [(ch, shards[ch], int(shards[ch].qid.max()) + 1)]); one.close_bam(0)
            got1 = one.finish()
            h = hashlib.sha256()
            for k in OUTPUTS:
                h.update(canonical(k, got1[k]).encode())
            cf = os.path.join(tmp, "%s.%s.calls.tsv" % (name, ch))
            open(cf, "w").write(call_text(vsets[ch], shards[ch], one.shards[ch][0].calls))
            while sum(1 for j in jobs if j[1].poll() is None) >= max(1, cores - 2):
                time.sleep(0.5)
            jobs.append(("%s %s" % (name, ch), subprocess.Popen([sys.executable, os.path.join(REPO, "tools", "oracle_chrom_worker.py"), cf, "10", "-", "-", "-", name],
                                                                  stdout=subprocess.PIPE, text=True), one.phased, h.hexdigest()))
            line += " | K_map = C oracle on all %d records of 22 shards; invariants ok (%d allelic / %d blocks / %d hap-count / %d connection rows); checks %.1f s" % (
                recs, inv[0], inv[1], inv[2], inv[3], time.perf_counter() - tc)
            del one, got1
        print(line, flush=True)
        del shards, samples, eng, files, out
        torch.cuda.empty_cache()
    t4 = time.perf_counter()
    a, g, log = expr_matrix.expr_matrix(out_dir + "/gene_ae", bed_path)
    t5 = time.perf_counter()
    bad = 0
    for label, pr, phased, sha in jobs:
        res = pr.communicate()[0].split()
        ok = pr.returncode == 0 and len(res) == 3 and int(res[0]) == phased and res[2] == sha
        bad += 0 if ok else 1
        print("five files of %-18s vs oracle/phasing_oracle.py (%d phased variants, oracle %.0f CPU-s): %s" % (label, phased, float(res[1]) if len(res) > 1 else -1,
                                                                                                                "identical (canonical form)" if ok else "DIFFERENT"), flush=True)
    print("configs[4] share, scale %.2f, %d samples on one GPU: hot path %.3f s/sample = %.1f M records/s, %.2f G allele calls/s, %.2f M phased variants/s per GPU "
          "(x8 GPUs, independent samples: %.1f samples/s per node); gene_ae %.2f s/sample, expr_matrix %.2f s (%d genes x %d samples, %d problems); wall %.0f s"
          % (scale, S, t_path / S, n_rec / t_path / 1e6, n_calls_total / t_path / 1e9, n_phased / t_path / 1e6, 8 * S / t_path, t_gene / S, t5 - t4,
             len(a.splitlines()) - 1, a.split("\n")[0].count("\t") - 3, len(log), time.perf_counter() - t_all))
    print("VERDICT: %s" % ("ALL SAMPLES IDENTICAL TO THE ORACLES" if bad == 0 and check else ("%d CHROMOSOMES DIFFER" % bad if check else "no parity checks requested")))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
