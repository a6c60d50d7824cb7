"""BGZF/BAM writer + reader round trip and samtools-equivalent filtering (CPU only)."""
import gzip
import os

import numpy as np
import torch

from conftest import GOLD, gz_text


def _pipe_one_unfiltered():
    from phaser_amd import synth
    v, gs, ge, w = synth.make_variants("chr22", 1, 3_000_000, 300, 201, n_genes=20)
    rb = synth.make_reads(v, gs, ge, w, 9000, 202)
    return v, rb


def test_bam_roundtrip_matches_filtered_sam(tmp_path):
    from phaser_amd import bamio, samio, synth
    v, rb = _pipe_one_unfiltered()
    path = str(tmp_path / "a.bam")
    bamio.readbatch_to_bam(path, [rb], [("chr21", 46709983), ("chr22", 50818468)])
    # gzip-compatible and carries the BGZF EOF marker
    assert gzip.open(path, "rb").read(4) == b"BAM\x01"
    assert open(path, "rb").read()[-28:] == bamio._EOF
    it_b = {}; it_s = {}
    got = bamio.shards_from_bam(path, it_b, 255, True, True)["chr22"]
    want = samio.shards_from_sam(gz_text(os.path.join(GOLD, "pipe_one", "a.chr22.sam.gz")), it_s)["chr22"]
    for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score", "has_as"):
        assert torch.equal(getattr(got, f), getattr(want, f)), f
    assert it_b["chr22"].names == it_s["chr22"].names
    # filter switches: keeping duplicates / unpaired / low MAPQ lets more records through
    n0 = got.n
    assert bamio.shards_from_bam(path, {}, 255, False, True)["chr22"].n > n0
    assert bamio.shards_from_bam(path, {}, 0, True, True)["chr22"].n > n0
    assert bamio.shards_from_bam(path, {}, 255, True, False)["chr22"].n > n0
    assert "chr22" not in bamio.shards_from_bam(path, {}, 255, True, True, chroms={"chr21"})


def test_aux_as_parsing():
    from phaser_amd import bamio
    import struct
    aux = b"NHC\x01" + b"XSZabc\0" + b"ASs" + struct.pack("<h", -7) + b"XBBc" + struct.pack("<i", 2) + b"\x01\x02" + b"ASC\x98"
    assert bamio.aux_AS(aux) == 152
    assert bamio.aux_AS(b"NHC\x01") is None
