#!/usr/bin/env python3
"""Build-container fuzz (needs /root/reference): oracle/gene_ae_oracle.py vs the reference's phaser_gene_ae.py (run with the real
intervaltree 3.1.0 package, loaded by path from the image's conda tree like tools/make_golden.py does) on random feature sets and argument combinations over the committed
haplotypic_counts fixtures.  usage: tools/fuzz_oracle_gene_ae.py [iterations=100] [seed=1]"""
import collections, gzip, io, os, random, runpy, sys, tempfile, types
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools")); sys.path.insert(0, os.path.join(REPO, "oracle"))
os.environ.setdefault("PYTHONHASHSEED", "0")
import make_golden as mg
import gene_ae_oracle as go
# the REAL intervaltree 3.1.0: pure-Python sources in the image's conda tree, loaded by path (its dependency sortedcontainers is installed here)
import importlib.util
_real = "/opt/conda/lib/python3.9/site-packages/intervaltree"
_spec = importlib.util.spec_from_file_location("intervaltree", os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
mod = importlib.util.module_from_spec(_spec); sys.modules["intervaltree"] = mod; _spec.loader.exec_module(mod)
script = "/root/reference/phaser_gene_ae/phaser_gene_ae.py"
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
    with tempfile.TemporaryDirectory() as tmp:
        hp = os.path.join(tmp, "hc.txt"); bp = os.path.join(tmp, "f.bed"); op = os.path.join(tmp, "o.txt")
        open(hp, "w").write(hc); open(bp, "w").write(bed)
        argv = sys.argv; sys.argv = [script, "--haplotypic_counts", hp, "--features", bp, "--o", op] + args
        old = sys.stdout; sys.stdout = io.StringIO()
        try:
            runpy.run_path(script, run_name="__main__")
        finally:
            sys.stdout = old; sys.argv = argv
        want = open(op).read()
    got = go.gene_ae(hc, bed, **kw)
    ok = go.canonical(got) == go.canonical(want)
    bad += not ok
    if not ok or it % 20 == 0:
        print("iter %d %s %s -> %s (%d rows)" % (it, src, args, "OK" if ok else "DIFF", len(want.splitlines()) - 1), flush=True)
print("%d iterations, %d with differences" % (iters, bad))
