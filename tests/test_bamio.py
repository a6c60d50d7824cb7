"""BGZF/BAM writer + reader round trip and samtools-equivalent filtering (CPU only)."""
import gzip
import os

import numpy as np
import pytest
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


def test_native_decoder_matches_python_reader(tmp_path):
    """The C++ BGZF/BAM decoder + packer + interner (host code in libphz.so) gives bit-identical shards."""
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    v, rb = _pipe_one_unfiltered()
    v2, gs, ge, w = synth.make_variants("chr21", 1, 1_000_000, 80, 77, n_genes=6)
    rb2 = synth.make_reads(v2, gs, ge, w, 1500, 78, qname_prefix="s0.b0.r")      # QNAMEs collide with chr22's on purpose
    path = str(tmp_path / "two.bam")
    bamio.readbatch_to_bam(path, [rb2, rb], [("chr21", 46709983), ("chr22", 50818468)])
    for args in [(255, True, True, 0.0), (0, False, False, 0.0), (255, True, True, 260.0)]:
        ip = {}; inn = {}
        want = bamio.shards_from_bam(path, ip, *args)
        got = bamio.shards_from_bam_native(path, inn, *args, threads=3)
        assert list(got) == list(want) == ["chr21", "chr22"]
        for c in want:
            for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score", "has_as"):
                assert torch.equal(getattr(got[c], f), getattr(want[c], f)), (c, f, args)
            assert inn[c].names == ip[c].names
    only = bamio.shards_from_bam_native(path, {}, 255, True, True, chroms={"chr21"})
    assert list(only) == ["chr21"]


def test_native_decoder_odd_records(tmp_path):
    """Malformed / unusual records: SEQ '*', QUAL missing, IUPAC bases, hard clips, padding, CIGAR longer than SEQ, no AS."""
    from phaser_amd import bamio
    recs = [
        {"ref_id": 0, "pos": 100, "mapq": 60, "flag": 0, "tlen": 0, "qname": "r1", "cigar": [(5, 3), (0, 6), (6, 2), (0, 4), (5, 1)],
         "seq": "ACGTNRYACG", "qual": [40] * 10, "tags": {"AS": -5}},
        {"ref_id": 0, "pos": 120, "mapq": 60, "flag": 0, "tlen": 0, "qname": "r2", "cigar": [(0, 10)], "seq": "", "qual": [], "tags": {}},
        {"ref_id": 0, "pos": 130, "mapq": 60, "flag": 0, "tlen": 0, "qname": "r3", "cigar": [(0, 4), (3, 7), (0, 6)], "seq": "ACGTAC", "qual": None,
         "tags": {"NM": 1}},
        {"ref_id": 0, "pos": 140, "mapq": 60, "flag": 0, "tlen": 0, "qname": "r4", "cigar": [(0, 4), (1, 3), (0, 9), (4, 2)], "seq": "ACGTACG",
         "qual": [30] * 7, "tags": {"AS": 70000}},
        {"ref_id": 0, "pos": 150, "mapq": 60, "flag": 0, "tlen": 0, "qname": "r1", "cigar": [], "seq": "AC=D", "qual": [20] * 4, "tags": {}},
    ]
    path = str(tmp_path / "odd.bam")
    bamio.write_bam(path, [("c1", 1000)], recs)
    ip = {}; inn = {}
    want = bamio.shards_from_bam(path, ip, 0, False, False)["c1"]
    got = bamio.shards_from_bam_native(path, inn, 0, False, False)["c1"]
    for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score", "has_as"):
        assert torch.equal(getattr(got, f), getattr(want, f)), f
    assert got.qid.tolist() == [0, 1, 2, 3, 0]


def test_unsorted_inputs_are_refused(tmp_path):
    """The mapper is a merge join over coordinate-sorted records: every packer refuses an inversion (the kernels do not re-check)."""
    import torch
    from phaser_amd import _lib, bamio, soa, synth
    v, gs, ge, w = synth.make_variants("chr22", 1, 2_000_000, 100, 41, n_genes=5)
    rb = synth.make_reads(v, gs, ge, w, 400, 42)
    rf = rb.select(synth.samtools_keep(rb, 255))
    soa.pack_readbatch(rf)                                   # sorted: fine
    import dataclasses
    bad = dataclasses.replace(rf, pos=rf.pos.flip(0))          # positions descending
    with pytest.raises(ValueError):
        soa.pack_readbatch(bad)
    with pytest.raises(ValueError):
        soa.pack_sam([(100, "4M", "ACGT", "IIII"), (50, "4M", "ACGT", "IIII")])
    path = str(tmp_path / "u.bam")
    bamio.readbatch_to_bam(path, [bad], [("chr22", 50818468)])
    _lib.build()
    with pytest.raises(_lib.PhzError):
        bamio.shards_from_bam_native(path, {}, mapq=0, paired_end=False, remove_dups=False)


def test_native_bgzf_write_and_read(tmp_path):
    """phz_bgzf_write / phz_bgzf_read: what we write, the gzip module and our parallel reader both read back; plain gzip falls back."""
    from phaser_amd import _lib, vcf, vcfout
    _lib.build()
    text = open(os.path.join(GOLD, "pipe_two", "in.vcf")).read() * 40
    p = str(tmp_path / "t.vcf.gz")
    vcfout.write_bgzf(p, text, 3)
    assert gzip.open(p, "rt").read() == text
    assert vcf.read_bytes(p, 3).decode() == text
    raw = open(p, "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC" and raw.endswith(bamio_eof())
    q = str(tmp_path / "plain.vcf.gz")
    with gzip.open(q, "wt") as f:
        f.write(text[:50000])
    assert vcf.read_bytes(q).decode() == text[:50000]
    e = str(tmp_path / "e.vcf.gz")
    vcfout.write_bgzf(e, "")
    assert gzip.open(e, "rb").read() == b""


def bamio_eof():
    from phaser_amd import bamio
    return bamio._EOF


def test_native_bam_writer_matches_python_writer(tmp_path):
    """phz_bam_write (used for the at-scale runs) produces the same BAM stream as the Python writer the other tests rely on."""
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    v, gs, ge, w = synth.make_variants("chr22", 1, 3_000_000, 300, 201, n_genes=20)
    rb = synth.make_reads(v, gs, ge, w, 1500, 202)
    refs = [("chr21", 46709983), ("chr22", 50818468)]
    a = str(tmp_path / "a.bam"); b = str(tmp_path / "b.bam")
    bamio.readbatch_to_bam(a, [rb], refs); bamio.readbatch_to_bam_native(b, [rb], refs, 3)
    assert gzip.open(a, "rb").read() == gzip.open(b, "rb").read()


def test_parallel_interner_numbers_by_first_appearance():
    """phz_intern (hash-partitioned, threaded) hands out the ids a sequential dictionary would, across calls."""
    import ctypes as C
    from phaser_amd import _lib, bamio
    _lib.build()
    rng = np.random.default_rng(3)
    it = bamio.NativeInterner()
    table = {}
    for call in range(3):
        n = 150_000
        ids = rng.integers(0, 90_000 * (call + 1), n)
        names = [b"q%d.%d" % (x % 7, x) for x in ids.tolist()]
        off = np.zeros(n + 1, dtype=np.uint32); off[1:] = np.cumsum([len(x) for x in names])
        blob = b"".join(names)
        out = np.zeros(n, dtype=np.int32)
        it.lib.phz_intern(it.h, blob, C.c_void_p(off.ctypes.data), n, C.c_void_p(out.ctypes.data))
        want = [table.setdefault(x, len(table)) for x in names]
        assert out.tolist() == want
        assert len(it) == len(table)
    assert it.names == [k.decode() for k in table]


def test_parallel_record_hop_gives_the_same_shards(tmp_path, monkeypatch):
    """Large BAMs are hopped in segments whose guessed record boundaries are verified against the chain; forced here on a small file."""
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    v, rb = _pipe_one_unfiltered()
    v2, gs, ge, w = synth.make_variants("chr21", 1, 1_000_000, 80, 77, n_genes=6)
    rb2 = synth.make_reads(v2, gs, ge, w, 4000, 78)
    path = str(tmp_path / "two.bam")
    bamio.readbatch_to_bam_native(path, [rb2, rb], [("chr21", 46709983), ("chr22", 50818468)])
    want = bamio.shards_from_bam_native(path, {}, 255, True, True, threads=1)
    monkeypatch.setenv("PHZ_BAM_PAR_MIN", "0")
    for th in (2, 7):
        got = bamio.shards_from_bam_native(path, {}, 255, True, True, threads=th)
        assert list(got) == list(want)
        for c in want:
            for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score", "has_as"):
                assert torch.equal(getattr(got[c], f), getattr(want[c], f)), (c, f, th)


def test_threaded_decoder_on_a_larger_bam(tmp_path, monkeypatch):
    """> 65,536 kept records on one reference: the sliced offset computation and the parallel hop agree with the one-thread run."""
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    v, gs, ge, w = synth.make_variants("chr21", 1, 8_000_000, 400, 91, n_genes=40)
    rb = synth.make_reads(v, gs, ge, w, 90_000, 92)
    path = str(tmp_path / "big.bam")
    bamio.readbatch_to_bam_native(path, [rb], [("chr21", 46709983)], 4)
    want = bamio.shards_from_bam_native(path, {}, 0, False, False, threads=1)["chr21"]
    assert want.n > 150_000
    monkeypatch.setenv("PHZ_BAM_PAR_MIN", "0")
    got = bamio.shards_from_bam_native(path, {}, 0, False, False, threads=6)["chr21"]
    for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score", "has_as"):
        assert torch.equal(getattr(got, f), getattr(want, f)), f


def test_region_open_equals_full_decode(tmp_path):
    """phz_bam_open_refs (only the BGZF members of the wanted chromosomes are inflated) gives exactly the shards of the full decode,
    for every subset of references, incl. references without records and the first / last one; the byte weights are positive for
    populated references and their order of magnitude follows the record counts."""
    import itertools
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    contigs = [("chr19", 58617616), ("chr20", 64444167), ("chr21", 46709983), ("chr22", 50818468), ("chrEmpty", 1000)]
    batches = []
    counts = {}
    for i, (c, ln) in enumerate(contigs[:4]):
        if c == "chr20":
            continue                                            # a reference between two populated ones with no records at all
        v, gs, ge, w = synth.make_variants(c, 1, 3_000_000, 150, 40 + i, n_genes=12)
        rb = synth.make_reads(v, gs, ge, w, 4000 * (i + 1), 50 + i)
        batches.append(rb); counts[c] = len(rb)
    bam = str(tmp_path / "r.bam")
    bamio.readbatch_to_bam(bam, batches, contigs)
    full = bamio.shards_from_bam_native(bam, {}, 0, False, False, 0.0, threads=2)
    assert {c: s.n for c, s in full.items()} == counts
    names = [c for c, _ in contigs]
    for k in range(1, 4):
        for sub in itertools.combinations(names, k):
            got = bamio.shards_from_bam_native(bam, {}, 0, False, False, 0.0, chroms=set(sub), threads=2)
            assert set(got) == set(sub) & set(full), sub
            for c in got:
                for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score"):
                    assert torch.equal(getattr(got[c], f), getattr(full[c], f)), (sub, c, f)
    w = bamio.bam_ref_weights(bam, threads=2)
    assert list(w) == names and w["chr20"] == 0 and w["chrEmpty"] == 0
    assert w["chr19"] > 0 and w["chr22"] > w["chr19"]
    assert abs(sum(w.values()) - os.path.getsize(bam)) < 70000            # everything but the header member and the EOF marker


def _long_read_bam(path, refs, per_ref, L, shuffle=False):
    """records of L bases (tens of kilobytes each): (ref_id, pos) ascending unless shuffle"""
    import random
    import struct
    from phaser_amd import bamio
    rng = random.Random(5)
    recs = []
    for rid, n in enumerate(per_ref):
        for i in range(n):
            seq = "".join(rng.choice("ACGT") for _ in range(64)) * (L // 64)
            recs.append({"ref_id": rid, "pos": 1000 + 37 * i, "mapq": 60, "flag": 0, "tlen": 0, "qname": "lr%d_%d" % (rid, i), "cigar": [(0, len(seq))],
                         "seq": seq, "qual": [30 + (i + k) % 10 for k in range(len(seq))], "tags": {"AS": 100 + i % 50}})
    if shuffle:
        rng.shuffle(recs)
    bamio.write_bam(path, refs, recs)
    return recs


def test_region_open_with_records_larger_than_the_probe_window(tmp_path):
    """Records of ~30 KB and ~100 KB (long reads): the reference-boundary probe needs 12 chained records, i.e. far more than the three
    BGZF members it starts with -- the window grows, and where a boundary cannot be proven the file goes through the full open.  Either
    way the chromosome-restricted open returns exactly the shards of the full decode (nothing may be dropped silently)."""
    from phaser_amd import _lib, bamio
    _lib.build()
    refs = [("chrA", 5_000_000), ("chrB", 5_000_000), ("chrC", 5_000_000)]
    for L, per_ref in ((20_032, (40, 55, 30)), (66_048, (14, 9, 16))):
        bam = str(tmp_path / ("long%d.bam" % L))
        _long_read_bam(bam, refs, per_ref, L)
        full = bamio.shards_from_bam_native(bam, {}, 0, False, False, 0.0, threads=2)
        assert {c: s.n for c, s in full.items()} == {r[0]: n for r, n in zip(refs, per_ref)}
        for sub in ({"chrA"}, {"chrB"}, {"chrC"}, {"chrA", "chrC"}, {"chrB", "chrC"}):
            got = bamio.shards_from_bam_native(bam, {}, 0, False, False, 0.0, chroms=set(sub), threads=2)
            assert set(got) == sub
            for c in got:
                for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "aln_score"):
                    assert torch.equal(getattr(got[c], f), getattr(full[c], f)), (L, sub, c, f)


def test_region_open_of_an_unsorted_file_takes_the_full_open(tmp_path):
    """A file whose records are not grouped by reference cannot be cut at reference boundaries: the chromosome-restricted open must see
    every record of the wanted reference (the refIDs met by the boundary search do not ascend -> full, order-agnostic open), and the decoder
    then refuses the file for being unsorted exactly as it does after a full open."""
    from phaser_amd import _lib, bamio
    _lib.build()
    refs = [("chrA", 5_000_000), ("chrB", 5_000_000)]
    bam = str(tmp_path / "shuffled.bam")
    _long_read_bam(bam, refs, (60, 60), 4_096, shuffle=True)
    with pytest.raises((_lib.PhzError, SystemExit)):
        bamio.shards_from_bam_native(bam, {}, 0, False, False, 0.0, threads=2)
    with pytest.raises((_lib.PhzError, SystemExit)):
        bamio.shards_from_bam_native(bam, {}, 0, False, False, 0.0, chroms={"chrB"}, threads=2)


def test_empty_qname_record(tmp_path):
    """l_read_name == 1 (just the NUL): the name is the empty string; it interns like any other name (this used to read the back of
    an empty arena in the interner)."""
    from phaser_amd import _lib, bamio
    _lib.build()
    recs = [{"ref_id": 0, "pos": 100 + 10 * i, "mapq": 60, "flag": 0, "tlen": 0, "qname": nm, "cigar": [(0, 4)], "seq": "ACGT", "qual": [30] * 4, "tags": {}}
            for i, nm in enumerate(["", "a", "", "b"])]
    path = str(tmp_path / "e.bam")
    bamio.write_bam(path, [("c1", 1000)], recs)
    ip = {}; inn = {}
    want = bamio.shards_from_bam(path, ip, 0, False, False)["c1"]
    got = bamio.shards_from_bam_native(path, inn, 0, False, False)["c1"]
    assert got.qid.tolist() == want.qid.tolist() == [0, 1, 0, 2]
    assert inn["c1"].names == ["", "a", "b"]


def test_segmented_member_walk_gives_the_same_table(tmp_path, monkeypatch):
    """Files >= 256 MB have their BGZF member chain walked in 16 segments with guessed, then verified, first members; forced here."""
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    v, gs, ge, w = synth.make_variants("chr21", 1, 8_000_000, 400, 91, n_genes=40)
    rb = synth.make_reads(v, gs, ge, w, 60_000, 92)
    path = str(tmp_path / "m.bam")
    bamio.readbatch_to_bam_native(path, [rb], [("chr21", 46709983), ("chr22", 50818468)], 4)
    want = bamio.shards_from_bam_native(path, {}, 0, False, False, chroms={"chr21"}, threads=2)["chr21"]
    monkeypatch.setenv("PHZ_BGZF_PAR_MIN", "0")
    got = bamio.shards_from_bam_native(path, {}, 0, False, False, chroms={"chr21"}, threads=2)["chr21"]
    for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score", "has_as"):
        assert torch.equal(getattr(got, f), getattr(want, f)), f
    w0 = bamio.bam_ref_weights(path)
    monkeypatch.delenv("PHZ_BGZF_PAR_MIN")
    assert bamio.bam_ref_weights(path) == w0


def test_member_walk_without_a_mapping(tmp_path, monkeypatch):
    """The device path's plan reads the member headers with pread (one 64-byte read per member) instead of through a mapping of the file;
    PHZ_BGZF_NO_MAP puts the host decoder on the same walk.  Same shards, one-segment and 16-segment walk; a member whose gzip extra field is
    longer than the 64-byte peek (other subfields before BC) is still parsed."""
    import struct, zlib
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    v, gs, ge, w = synth.make_variants("chr21", 1, 8_000_000, 400, 93, n_genes=40)
    rb = synth.make_reads(v, gs, ge, w, 30_000, 94)
    path = str(tmp_path / "m.bam")
    bamio.readbatch_to_bam_native(path, [rb], [("chr21", 46709983), ("chr22", 50818468)], 4)
    want = bamio.shards_from_bam_native(path, {}, 0, False, False, chroms={"chr21"}, threads=2)["chr21"]
    w0 = bamio.bam_ref_weights(path)
    # the same file with every member's extra field padded by a 90-byte subfield in front of BC
    raw = open(path, "rb").read()
    wide = bytearray(); off = 0; n_members = 0
    while off < len(raw):
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        assert xlen == 6 and raw[off + 12:off + 14] == b"BC"
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        pad = b"ZZ" + struct.pack("<H", 90) + bytes(90)
        wide += raw[off:off + 10] + struct.pack("<H", 6 + len(pad)) + pad + b"BC\x02\x00" + struct.pack("<H", bsize + len(pad) - 1) + raw[off + 18:off + bsize]
        off += bsize; n_members += 1
    assert n_members > 20
    wpath = str(tmp_path / "wide.bam")
    open(wpath, "wb").write(bytes(wide))
    monkeypatch.setenv("PHZ_BGZF_NO_MAP", "1")
    for par_min in (None, "0"):
        if par_min is not None:
            monkeypatch.setenv("PHZ_BGZF_PAR_MIN", par_min)
        for pth in (path, wpath):
            got = bamio.shards_from_bam_native(pth, {}, 0, False, False, chroms={"chr21"}, threads=2)["chr21"]
            for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score", "has_as"):
                assert torch.equal(getattr(got, f), getattr(want, f)), (pth, par_min, f)
        assert bamio.bam_ref_weights(path) == w0
    # a file cut in the middle of a member header / of a member: refused or read up to the last whole member exactly as with the mapping
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(raw[:len(raw) // 2])
    def outcome():
        try:
            s = bamio.shards_from_bam_native(cut, {}, 0, False, False, threads=2)
            return sorted((c, x.pos.tolist()[:50], x.n) for c, x in s.items())
        except Exception as e:
            return str(e)
    a = outcome()
    monkeypatch.delenv("PHZ_BGZF_NO_MAP")
    assert outcome() == a


def _damage_bgzf(raw, rng, what):
    """raw: a BGZF file; -> a copy with ONE member damaged below the BAM level: `crc` = a flipped bit in a trailer's CRC32, `payload` = a flipped bit somewhere in a
    deflate stream (often still valid DEFLATE of the right length)."""
    import struct
    members = []; off = 0
    while off + 18 <= len(raw):
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        members.append((off, bsize)); off += bsize
    off, bsize = members[int(rng.integers(1, len(members) - 1))]         # not the header member, not the EOF member
    m = bytearray(raw)
    at = off + bsize - 8 + int(rng.integers(0, 4)) if what == "crc" else off + 18 + int(rng.integers(0, bsize - 26))
    m[at] ^= 1 << int(rng.integers(0, 8))
    return bytes(m)


def test_damaged_bgzf_members_are_refused(tmp_path):
    """htslib checks the CRC32 of every BGZF block it inflates, so `samtools view` (phaser/phaser.py:1346) stops on a file whose bytes were damaged in storage;
    so does the host decoder here: a flipped bit in a trailer's CRC32 or in a deflate stream never yields shards."""
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    v, gs, ge, w = synth.make_variants("chr21", 1, 8_000_000, 400, 95, n_genes=40)
    rb = synth.make_reads(v, gs, ge, w, 20_000, 96)
    path = str(tmp_path / "m.bam")
    bamio.readbatch_to_bam_native(path, [rb], [("chr21", 46709983), ("chr22", 50818468)], 4)
    raw = open(path, "rb").read()
    want = bamio.shards_from_bam_native(path, {}, 0, False, False, threads=2)["chr21"]
    assert want.n > 10_000
    rng = np.random.default_rng(8)
    for trial in range(16):
        bad = str(tmp_path / ("bad%d.bam" % trial))
        open(bad, "wb").write(_damage_bgzf(raw, rng, "crc" if trial % 2 == 0 else "payload"))
        for chroms in (None, {"chr21"}):                       # whole-file open and the member-table open
            with pytest.raises(_lib.PhzError):
                bamio.shards_from_bam_native(bad, {}, 0, False, False, chroms=chroms, threads=2)


def test_bgzf_member_claiming_more_than_64k_is_refused(tmp_path):
    """BGZF members inflate to at most 64 KiB; a trailer that claims more is not trusted (it would size host and device buffers)."""
    import struct
    from phaser_amd import _lib, bamio, synth
    _lib.build()
    v, gs, ge, w = synth.make_variants("chr22", 1, 2_000_000, 100, 51, n_genes=5)
    rb = synth.make_reads(v, gs, ge, w, 300, 52)
    path = str(tmp_path / "s.bam")
    bamio.readbatch_to_bam(path, [rb], [("chr22", 50818468)])
    raw = bytearray(open(path, "rb").read())
    assert raw[:4] == b"\x1f\x8b\x08\x04"
    bsize = struct.unpack_from("<H", raw, 16)[0] + 1            # first member: BSIZE sits at offset 16 (XLEN 6, subfield BC)
    struct.pack_into("<I", raw, bsize - 4, 70000)               # its ISIZE
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(_lib.PhzError):
        bamio.shards_from_bam_native(bad, {}, mapq=0, paired_end=False, remove_dups=False)
    with pytest.raises(_lib.PhzError):
        bamio.shards_from_bam_native(bad, {}, mapq=0, paired_end=False, remove_dups=False, chroms={"chr22"})
