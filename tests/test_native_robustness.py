"""The native text parsers must answer malformed input with a status code (a Python exception here), never with a crash."""
import os
import random

import pytest

from conftest import GOLD, gz_text


def _mutations(text, rng, n):
    lines = text.split("\n")
    for _ in range(n):
        ls = list(lines)
        k = rng.randrange(len(ls))
        kind = rng.randrange(6)
        if kind == 0:
            ls[k] = ls[k][:rng.randrange(len(ls[k]) + 1)]                       # truncated line
        elif kind == 1:
            ls[k] = ls[k].replace("\t", " ", rng.randrange(1, 4))               # lost tabs
        elif kind == 2:
            ls[k] = "".join(rng.choice("ACGT|/.:;,=\t0123456789xyz") for _ in range(rng.randrange(0, 80)))
        elif kind == 3:
            ls[k] = ls[k] + "\t" * rng.randrange(1, 5)
        elif kind == 4:
            c = ls[k].split("\t"); rng.shuffle(c); ls[k] = "\t".join(c)
        else:
            ls = ls[:k]                                                          # truncated file, no final newline
        yield "\n".join(ls)


def test_vcf_loader_survives_malformed_lines():
    from phaser_amd import _lib, vcf
    _lib.build()
    text = open(os.path.join(GOLD, "pipe_opts", "in.vcf")).read()
    rng = random.Random(1)
    ok = bad = 0
    for m in _mutations(text, rng, 300):
        try:
            vcf.load_variants(m, gw_phase_method=rng.randrange(2), include_indels=rng.randrange(2), pass_only=rng.randrange(2), threads=2)
            ok += 1
        except (SystemExit, _lib.PhzError):
            bad += 1
    assert ok > 50 and ok + bad == 300


def test_haplotypic_counts_parser_survives_malformed_lines():
    from phaser_amd import _lib, gene_ae
    _lib.build()
    text = gz_text(os.path.join(GOLD, "pipe_two", "out.haplotypic_counts.txt.gz"))
    rng = random.Random(2)
    ok = bad = 0
    for m in _mutations(text, rng, 300):
        try:
            P = gene_ae.ParsedCounts(m.encode(), "_", 2)
            assert P.n_rows >= 0
            ok += 1
        except (SystemExit, _lib.PhzError):
            bad += 1
    assert ok > 30 and ok + bad == 300


def test_phased_vcf_writer_survives_malformed_lines():
    import sys
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from phasing_oracle import bam_display_names
    from phaser_amd import _lib, vcfout
    from test_host_stages import run_host_stages
    d = os.path.join(GOLD, "pipe_one")
    vcf_text = open(os.path.join(d, "in.vcf")).read()
    out, eng = run_host_stages("pipe_one", {}, {}, vcf_text, bam_display_names(["a.bam"]))
    rng = random.Random(3)
    ok = bad = 0
    for m in _mutations(vcf_text, rng, 200):
        try:
            t, up, pc = vcfout.phased_vcf_text(m, 9, eng, gw_phase_vcf=rng.randrange(3), threads=2)
            ok += 1
        except _lib.PhzError:
            bad += 1
    assert ok > 20 and ok + bad == 200
