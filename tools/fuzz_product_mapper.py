#!/usr/bin/env python3
"""GPU-box fuzz: the product mapper (K_map / K_map_general behind phaser_amd.read_variant_map.do_read_variant_map) vs the C oracle
(oracle/rvm_oracle, itself fuzzed against the reference's compiled mapper by tools/fuzz_oracle_mapper.py) on random, deliberately odd SAM records -- every CIGAR operator incl. H / P, zero-length ops, leading / trailing I and D,
several introns, SEQ shorter or longer than the CIGAR implies, SEQ '*', QUAL '*', IUPAC bases, dense variant tables with
multi-base REF -- byte-for-byte TSV comparison.  usage: tools/fuzz_product_mapper.py [rounds=200] [seed=1]"""
import os, random, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import io
from phaser_amd import read_variant_map as prvm
from phaser_amd.mapper import Mapper
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mapper = Mapper(0)
subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
rng = random.Random(seed)
OPS = "MMMMMIDNSHP=X"
total = diffs = lines = 0
skipped = {}
for rd in range(rounds):
    indel_mode = rng.random() < 0.3
    # variant table: dense, sorted, unique positions
    positions = sorted(rng.sample(range(100, 1400), rng.randint(5, 120)))
    rows = []
    for p in positions:
        ref = "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 1, 1, 2, 3]) if indel_mode else 1))
        alt = "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 1, 2]) if indel_mode else 1))
        rows.append("\t".join(["chrF", str(p), "chrF_%d_%s_%s" % (p, ref, alt), "rs%d" % p, ref + "," + alt, str(len(ref)), rng.choice(["0|1", "1|0", "0/1"]), "None"]))
    table = "\n".join(rows) + "\n"
    recs = []
    pos = 90
    for r in range(rng.randint(20, 60)):
        pos += rng.randint(0, 40)
        if rng.random() < 0.12:          # a long, indel-rich record: hundreds of operations, more than 64 insertions in one segment
            nops = rng.randint(80, 400)
            parts = [(rng.choice([1, 1, 2, 3, 5]), rng.choice("MMMIIDD=X" if rng.random() < 0.97 else "N")) for _ in range(nops)]
            cig = "".join("%d%s" % p_ for p_ in parts)
            qlen = sum(k for k, o in parts if o in "MI=XS") + rng.choice([0, 0, -3, 4])
            qlen = max(0, qlen)
        else:
            nops = rng.randint(1, 7)
            cig = "".join("%d%s" % (rng.choice([0, 1, 2, 3, 5, 8, 13, 30, 76, 200]), rng.choice(OPS)) for _ in range(nops)) or "*"
            qlen = rng.choice([0, 1, 5, 20, 76, 100, 150])
        seq = "".join(rng.choice("ACGTACGTACGTNRYK") for _ in range(qlen)) or "*"
        if rng.random() < 0.1:
            qual = "*"
        else:
            ql = qlen if rng.random() < 0.8 else max(0, qlen + rng.randint(-5, 5))
            qual = "".join(chr(33 + rng.choice([2, 9, 10, 11, 25, 37, 41])) for _ in range(ql)) or "*"
        tags = ["NH:i:1"] + (["AS:i:%d" % rng.randint(0, 152)] if rng.random() < 0.9 else []) + (["AS:i:%d" % rng.randint(0, 152)] if rng.random() < 0.1 else [])
        recs.append("\t".join(["q%d" % rng.randint(0, 30), "99", "chrF", str(pos), "255", cig, "=", str(pos + 100), str(rng.choice([0, 150, -150, 400])), seq, qual] + tags))
    sam = "@SQ\tSN:chrF\tLN:5000\n" + "\n".join(recs) + "\n"
    baseq = rng.choice([0, 10, 10, 11, 30]); isize = rng.choice([0.0, 0.0, 200.0])
    with tempfile.TemporaryDirectory() as tmp:
        tp = os.path.join(tmp, "t.tsv"); op = os.path.join(tmp, "o.tsv"); pp = os.path.join(tmp, "p.tsv"); open(tp, "w").write(table)
        p = subprocess.run([os.path.join(REPO, "oracle", "rvm_oracle"), "--variant_table", tp, "--baseq", str(baseq), "--isize_cutoff", str(isize), "--o", op],
                           input=sam.encode(), capture_output=True)
        want = open(op).read() if p.returncode == 0 and os.path.exists(op) else "<rc %d>" % p.returncode
        old_in, old_out = sys.stdin, sys.stdout
        sys.stdin = io.StringIO(sam); sys.stdout = io.StringIO()
        try:
            prvm.do_read_variant_map(tp, baseq, pp, 1, isize, _mapper=mapper)
            got = open(pp).read()
        except BaseException as e:
            got = "<%s: %s>" % (type(e).__name__, e)
        finally:
            sys.stdin, sys.stdout = old_in, old_out
    total += 1; lines += want.count("\n")
    if got != want:
        diffs += 1
        if diffs <= 3:
            open("/tmp/fuzz_fail_%d.sam" % diffs, "w").write(sam); open("/tmp/fuzz_fail_%d.tsv" % diffs, "w").write(table)
            w = want.split("\n"); g = got.split("\n")
            k = next((i for i in range(min(len(w), len(g))) if w[i] != g[i]), min(len(w), len(g)))
            print("round %d DIFF (baseq %d isize %s indel %s) at line %d: want %r got %r  [%d vs %d lines]" % (rd, baseq, isize, indel_mode, k, w[k:k + 1], g[k:k + 1], len(w), len(g)))
print("%d rounds compared (%d call lines), %d differ; skipped (reference raised): %s" % (total, lines, diffs, skipped))
