#!/usr/bin/env python3
"""GPU-box timing of the mapper drop-in (Seam 1) on BASELINE configs[0]: SAM text of 101,119 records (24 MB) + the 1,000-row variant
table through phaser_amd.read_variant_map.do_read_variant_map -- native SAM parse / pack, K_map, native TSV -- against the
reference's compiled mapper on the same input (0.97 s).  The output bytes are checked against the reference's own call
file (tests/golden/c1)."""
import gzip, io, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from phaser_amd import synth, read_variant_map as prvm
from phaser_amd.mapper import Mapper
from helpers import variant_table_text
GOLD = os.path.join(REPO, "tests", "golden", "c1")
g = json.load(open(os.path.join(GOLD, "meta.json")))["gen"]
v, gs, ge, w = synth.make_variants(g["region"][0], g["region"][1], g["region"][2], g["n_snps"], g["vseed"], n_genes=g["n_genes"])
rb = synth.make_reads(v, gs, ge, w, g["n_pairs"], g["rseed"]); rf = rb.select(synth.samtools_keep(rb, g["mapq"]))
sam = "\n".join(synth.sam_lines(rf, [("chr22", 50818468)])) + "\n"
open("/tmp/c1.table.tsv", "w").write(variant_table_text(v))
m = Mapper(0)
want = [f for f in os.listdir(GOLD) if "calls" in f]
for rep in range(4):
    class _In:                                   # what sys.stdin looks like to the drop-in: a text stream with a .buffer
        buffer = io.BytesIO(sam.encode())
    old = sys.stdin; sys.stdin = _In()
    t0 = time.perf_counter()
    prvm.do_read_variant_map("/tmp/c1.table.tsv", 10, "/tmp/c1.calls.tsv", 1, 0, _mapper=m, threads=16)
    dt = time.perf_counter() - t0
    sys.stdin = old
    print("pass %d: do_read_variant_map %.1f ms for %d records (%.1f MB of SAM text) -> %.1fx the reference's 0.97 s" % (rep, dt * 1e3, len(rf), len(sam) / 1e6, 0.97 / dt))
got = open("/tmp/c1.calls.tsv").read()
for f in want:
    ref = gzip.open(os.path.join(GOLD, f), "rt").read()
    print(f, "identical to the reference's call file:", ref == got, len(got.splitlines()), "lines")
