#!/usr/bin/env python3
"""GPU-box fuzz: the phaser_gene_ae drop-in (native parser + K_genes + native rows) vs oracle/gene_ae_oracle.py (itself fuzzed against
the reference's script by tools/fuzz_oracle_gene_ae.py) on random feature sets and argument combinations over the committed
haplotypic_counts fixtures.  usage: tools/fuzz_product_gene_ae.py [iterations=100] [seed=1]"""
import collections, gzip, io, os, random, runpy, sys, tempfile, types
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, os.path.join(REPO, "oracle"))
os.environ.setdefault("PYTHONHASHSEED", "0")
import make_golden as mg
import gene_ae_oracle as go
sys.path.insert(0, REPO)
from phaser_amd import _lib, gene_ae
ctx = _lib.Context(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SRC = ["pipe_one", "pipe_two", "pipe_noisy_a", "pipe_noisy_b", "pipe_noisy_c", "c1", "pipe_opts/gw_maf", "pipe_opts/bam_exclude", "pipe_opts/blacklist", "pipe_opts/no_unphased"]
bad = 0
for it in range(iters):
    src = rng.choice(SRC)
    hc = gzip.open(os.path.join(mg.GOLD, src, "out.haplotypic_counts.txt.gz"), "rt").read()
    bed = mg.gene_ae_features(hc, rng.randrange(10 ** 6))
    # extra random features: tiny, huge, duplicated names, shuffled order
    lines = [l for l in bed.split("\n") if l]
    chroms = sorted(set(l.split("\t")[0] for l in lines))
    for _ in range(rng.randint(0, 15)):
        c = rng.choice(chroms); a = rng.randint(0, 3_000_000); lines.append("%s\t%d\t%d\tx%d" % (c, a, a + rng.choice([1, 2, 50, 5000, 2_000_000]), rng.randint(0, 5)))
    rng.shuffle(lines)
    bed = "\n".join(lines) + "\n"
    args = []; kw = {}
    if rng.random() < 0.6:
        kw["gw_cutoff"] = rng.choice([0.5, 0.6, 0.75, 0.9, 1.0, 1.01]); args += ["--gw_cutoff", str(kw["gw_cutoff"])]
    if rng.random() < 0.4:
        kw["min_cov"] = rng.choice([1, 2, 5, 20]); args += ["--min_cov", str(kw["min_cov"])]
    if rng.random() < 0.4:
        kw["min_haplo_maf"] = rng.choice([0.05, 0.1, 0.35, 0.5]); args += ["--min_haplo_maf", str(kw["min_haplo_maf"])]
    want = go.gene_ae(hc, bed, **kw)
    got = gene_ae.gene_ae(hc.encode(), bed, ctx=ctx, threads=rng.choice([1, 4]), **kw)
    ok = go.canonical(got) == go.canonical(want)
    bad += not ok
    if not ok or it % 20 == 0:
        print("iter %d %s %s -> %s (%d rows)" % (it, src, args, "OK" if ok else "DIFF", len(want.splitlines()) - 1), flush=True)
print("%d iterations, %d with differences" % (iters, bad))
