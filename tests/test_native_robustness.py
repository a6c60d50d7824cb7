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


def test_bam_decoder_survives_corrupt_records(tmp_path):
    """a truncated / corrupt BAM must come back as a status (PhzError), never as an out-of-bounds read.  The inflated
    stream of a small BAM is mutated (header fields, l_read_name / n_cigar / l_seq / block_size of records, random bytes, cut
    tails), re-wrapped as BGZF and decoded with the native reader; tools/asan_check.sh runs this under AddressSanitizer."""
    import gzip
    import struct
    import ctypes as C
    import numpy as np
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    v, gs, ge, w = synth.make_variants("chr22", 1, 2_000_000, 100, 31, n_genes=8)
    rb = synth.make_reads(v, gs, ge, w, 400, 32)
    bam = str(tmp_path / "a.bam")
    bamio.readbatch_to_bam(bam, [rb], [("chr21", 46709983), ("chr22", 50818468)])
    raw = bytearray(gzip.open(bam, "rb").read())          # BGZF members are gzip members
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
    for _ in range(n_ref):
        l = struct.unpack_from("<i", raw, p)[0]; p += 4 + l + 4
    first = p
    recs = []
    while p + 4 <= len(raw):
        recs.append(p); p += 4 + struct.unpack_from("<i", raw, p)[0]
    assert len(recs) > 500
    rng = random.Random(7)

    def write(data, name):
        path = str(tmp_path / name)
        st = _lib.load().phz_bgzf_write(path.encode(), bytes(data), len(data), 2, 1)
        assert st == 0
        return path

    def decode(path):
        interners = {}
        return bamio.shards_from_bam_native(path, interners, 0, False, False, 0.0, threads=rng.choice([0, 2]))
    base = decode(write(raw, "ok.bam"))
    assert sum(s.n for s in base.values()) == len(recs)
    ok = bad = 0
    for it in range(160):
        m = bytearray(raw)
        kind = it % 8
        r = rng.choice(recs)
        if kind == 0:
            struct.pack_into("<i", m, r, rng.choice([-1, 0, 31, 33, 1 << 30, struct.unpack_from("<i", m, r)[0] - 1]))       # block_size
        elif kind == 1:
            m[r + 4 + 8] = rng.choice([0, 1, 255])                                                                          # l_read_name
        elif kind == 2:
            struct.pack_into("<H", m, r + 4 + 12, rng.choice([0, 1000, 65535]))                                             # n_cigar_op
        elif kind == 3:
            struct.pack_into("<i", m, r + 4 + 16, rng.choice([-5, 0, 1 << 20, 0x7fffffff]))                                 # l_seq
        elif kind == 4:
            m = m[:rng.randrange(first, len(m))]                                                                            # cut tail
        elif kind == 5:
            for _ in range(rng.randrange(1, 20)):
                m[rng.randrange(first, len(m))] = rng.randrange(256)                                                        # noise in records
        elif kind == 6:
            struct.pack_into("<i", m, rng.choice([4, 8 + l_text, 8 + l_text + 4]), rng.choice([-1, 0x7fffffff, 1 << 28, 3]))  # header
        else:
            m = m[:rng.randrange(0, first + 8)]                                                                             # cut header
        try:
            decode(write(m, "m%d.bam" % it))
            ok += 1
        except _lib.PhzError:
            bad += 1
    assert ok + bad == 160 and bad >= 60
