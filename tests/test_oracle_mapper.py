"""Pins oracle/rvm_oracle.c against outputs of the reference mapper (tests/golden/, made by
tools/make_golden.py which ran /root/reference/phaser/read_variant_map.py in the build container)."""
import ctypes
import json
import os
import subprocess

import pytest

from conftest import GOLD, gz_text
from helpers import variant_table_text


def run_oracle_cli(oracle_dir, sam_text, table_path, out_path, baseq, isize):
    subprocess.run([os.path.join(oracle_dir, "rvm_oracle"), "--variant_table", table_path, "--baseq", str(baseq),
                    "--o", out_path, "--isize_cutoff", str(isize)], input=sam_text.encode(), check=True)
    return open(out_path).read()


def test_kat_micro(oracle_build):
    lib = ctypes.CDLL(os.path.join(oracle_build, "librvm_oracle.so"))
    lib.rvm_oracle_kat.restype = ctypes.c_int
    cases = json.load(open(os.path.join(GOLD, "kat_micro.json")))
    assert len(cases) >= 25
    buf = ctypes.create_string_buffer(4096)
    for c in cases:
        for v in c["variants"]:
            nseg = lib.rvm_oracle_kat(c["pos"], c["seq"].encode(), c["qual"].encode(), c["cigar"].encode(), c["baseq"],
                                      v["pos"], v["ref_len"], buf, 4096)
            assert nseg == len(c["segments"]), c["name"]
            assert buf.value.decode() == "|".join(v["per_segment"]), (c["name"], v["pos"])


def test_mapper_small_bytes(oracle_build, tmp_path):
    d = os.path.join(GOLD, "mapper_small")
    meta = json.load(open(os.path.join(d, "meta.json")))
    sam = gz_text(os.path.join(d, "in.sam.gz"))
    for run in meta["runs"]:
        got = run_oracle_cli(oracle_build, sam, os.path.join(d, "table.tsv"), str(tmp_path / "o.tsv"), run["baseq"], run["isize"])
        assert got == gz_text(os.path.join(d, run["file"])), run


def test_pipe_one_calls_bytes(oracle_build, tmp_path):
    d = os.path.join(GOLD, "pipe_one")
    sam = gz_text(os.path.join(d, "a.chr22.sam.gz"))
    # rebuild the variant table from the VCF exactly as phaser.py:1371-1404 does for SNPs
    rows = []
    for line in open(os.path.join(d, "in.vcf")):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        uid = "_".join([f[0], f[1], f[3]] + f[4].split(","))
        rows.append("\t".join([f[0], f[1], uid, f[2], f[3] + "," + f[4], str(len(f[3])), f[9], "None"]))
    tp = tmp_path / "t.tsv"
    tp.write_text("\n".join(rows) + "\n")
    got = run_oracle_cli(oracle_build, sam, str(tp), str(tmp_path / "o.tsv"), 10, 0)
    assert got == gz_text(os.path.join(d, "calls.a.chr22.tsv.gz"))


def test_c1_calls_bytes(oracle_build, c1_inputs, tmp_path):
    tp = tmp_path / "t.tsv"
    tp.write_text(variant_table_text(c1_inputs["variants"]))
    got = run_oracle_cli(oracle_build, c1_inputs["sam"], str(tp), str(tmp_path / "o.tsv"), 10, 0)
    want = gz_text(os.path.join(GOLD, "c1", "calls.tsv.gz"))
    assert got.count("\n") == c1_inputs["meta"]["call_lines"]
    assert got == want
