#!/usr/bin/env python3
"""GPU-box run of BASELINE.json configs[4] in miniature: S whole-genome samples (different SNP sets / seeds) streamed through one GPU,
each through the hot path (shards -> five files), phaser_gene_ae and, at the end, phaser_expr_matrix over all samples.  The driver's
8-GPU runs give every rank its own samples; this is one rank's loop.  Synthetic shards are generated in HBM (not timed).
usage: tools/run_c5.py [samples=3] [scale=0.25] [threads=32]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import _lib, expr_matrix, gene_ae, synth, vcf, workloads
from phaser_amd.engine import Engine, Config
from phaser_amd.mapper import Mapper
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
        114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 32
total_len = sum(HG38)
mapper = Mapper(0)
out_dir = "/tmp/c5"; os.makedirs(out_dir + "/gene_ae", exist_ok=True)
# one gene model for all samples: genes laid out from the first sample's variant clusters
bed_path = out_dir + "/genes.bed"
t_path = t_gene = 0.0; n_rec = n_phased = 0
for s in range(S):
    vsets = []; shards = {}
    for i, ln in enumerate(HG38):
        chrom = "chr%d" % (i + 1)
        n_snps = int(1_500_000 * scale * ln / total_len); n = int(80_000_000 * scale * ln / total_len)
        v, shard, _ = workloads.make_shard(chrom, ln, n_snps, n, 9000 + 100 * s + i, "cuda:0")
        vsets.append(v); shards[chrom] = shard
    vtext = "\n".join(synth.vcf_lines(vsets))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vs = vcf.load_variants(vtext, threads=threads)
    eng = Engine(vs, ["sample%03d" % s], Config(host_threads=threads, want_vcf=False), mapper=mapper)
    for chrom, shard in shards.items():
        eng.add_shard(0, chrom, shard, int(shard.qid.max()) + 1)
    eng.close_bam(0)
    files = eng.finish(chunks=True)
    for name, body in files.items():
        with open("%s/sample%03d.%s.txt" % (out_dir, s, name), "wb") as f:
            f.writelines(body)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    hc = b"".join(bytes(x) for x in files["haplotypic_counts"])
    if s == 0:       # genes = merged spans of the first sample's rows, like tools/gene_ae_scale.py
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
    open("%s/gene_ae/sample%03d.gene_ae.txt" % (out_dir, s), "w").write(table)
    t3 = time.perf_counter()
    t_path += t1 - t0; t_gene += t3 - t2; n_rec += sum(sh.n for sh in shards.values()); n_phased += eng.phased
    print("sample %d: %d records, hot path %.2fs, gene_ae %.2fs" % (s, sum(sh.n for sh in shards.values()), t1 - t0, t3 - t2), flush=True)
    del shards, eng, files
    torch.cuda.empty_cache()
t4 = time.perf_counter()
a, g, log = expr_matrix.expr_matrix(out_dir + "/gene_ae", bed_path)
t5 = time.perf_counter()
print("C5 x%.2f, %d samples on one GPU: hot path %.2f s/sample (%.2f M records/s, %.0f phased variants/s), gene_ae %.2f s/sample, "
      "expr_matrix %.2fs (%d genes x %d samples, %d problems)" % (scale, S, t_path / S, n_rec / t_path / 1e6, n_phased / t_path, t_gene / S, t5 - t4,
                                                                  len(a.splitlines()) - 1, a.split("\n")[0].count("\t") - 3, len(log)))
