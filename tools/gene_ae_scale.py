#!/usr/bin/env python3
"""GPU-box run of the phaser_gene_ae drop-in at scale: input = the haplotypic_counts.txt tools/run_c3.py leaves in /tmp
(configs[2] shape), features = synthetic genes built from the rows' spans.  Prints stage timings, the K_genes roofline
figures and a parity check against the pinned oracle on one chromosome (the oracle is far too slow for the whole file)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np
from phaser_amd import _lib, gene_ae
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/c3.haplotypic_counts.txt"
text = open(path, "rb").read()
t0 = time.perf_counter()
# genes: merge row spans that lie within 5 kb of each other, then add an overlapping "transcript" for every third gene
spans = {}
for line in text.split(b"\n")[1:]:
    if line:
        c = line.split(b"\t", 3)
        spans.setdefault(c[0].decode(), []).append((int(c[1]) - 1, int(c[2])))
feats = []
for chrom, sp in spans.items():
    sp.sort()
    cur_a, cur_b = sp[0]
    genes = []
    for a, b in sp[1:]:
        if a - cur_b < 5000:
            cur_b = max(cur_b, b)
        else:
            genes.append((cur_a, cur_b)); cur_a, cur_b = a, b
    genes.append((cur_a, cur_b))
    for i, (a, b) in enumerate(genes):
        feats.append("%s\t%d\t%d\t%s_g%d" % (chrom, max(0, a - 50), b + 50, chrom, i))
        if i % 3 == 0 and b - a > 10:
            feats.append("%s\t%d\t%d\t%s_g%d_t2" % (chrom, a, a + (b - a) // 2, chrom, i))
bed = "\n".join(feats) + "\n"
t1 = time.perf_counter()
ctx = _lib.Context(0)
stats = {}
gene_ae.gene_ae(text[:200000].rsplit(b"\n", 1)[0] + b"\n", bed, ctx=ctx, threads=32)        # warm-up (library load, first launch)
t2 = time.perf_counter()
out = gene_ae.gene_ae(text, bed, ctx=ctx, threads=32, stats=stats)
t3 = time.perf_counter()
print("gene_ae: %.1f MB counts, %d features | features built %.1fs | whole call %.2fs | %s" % (len(text) / 1e6, len(feats), t1 - t0, t3 - t2, stats))
if stats.get("k_genes_ms"):
    gb = stats["labels_visited"] * 8 / 1e9
    print("K_genes: %.3f ms, %d labels visited in %d items -> %.1f GB/s algorithmic (8 B per label)" %
          (stats["k_genes_ms"], stats["labels_visited"], stats["items"], gb / (stats["k_genes_ms"] / 1e3)))
# parity on one chromosome vs the pinned oracle, timed as the CPU baseline
import gene_ae_oracle as go
chrom = "chr21"
sub = b"\n".join([text.split(b"\n", 1)[0]] + [l for l in text.split(b"\n")[1:] if l.startswith(chrom.encode() + b"\t")]) + b"\n"
bed_sub = "".join(l + "\n" for l in feats if l.startswith(chrom + "\t"))
t4 = time.perf_counter()
want = go.gene_ae(sub.decode(), bed_sub)
t5 = time.perf_counter()
got = gene_ae.gene_ae(sub, bed_sub, ctx=ctx, threads=32)
t6 = time.perf_counter()
nrows = sub.count(b"\n") - 1
print("parity on %s (%d rows, %d features): %s | oracle %.2fs (%.0f rows/s, 1 core) vs product %.3fs" %
      (chrom, nrows, bed_sub.count("\n"), "IDENTICAL" if go.canonical(got) == go.canonical(want) else "DIFFERENT", t5 - t4, nrows / (t5 - t4), t6 - t5))
