"""Device BAM path (phz_bgzf_inflate_device, phz_bamdev_*): K_inflate against zlib, and the shards decoded on the GPU against the
host decoder (phz_bam_*, itself pinned against the pure-Python BAM reader in tests/test_bamio.py) array by array."""
import ctypes as C
import itertools
import os
import sys
import zlib

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pytestmark = pytest.mark.gpu
FIELDS = ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score", "has_as")


@pytest.fixture(scope="module")
def ctx():
    from phaser_amd.mapper import Mapper
    return Mapper(0).ctx


def _members(buf):
    out = []; off = 0; dst = 0
    while off + 18 <= len(buf):
        xlen = int.from_bytes(buf[off + 10:off + 12], "little")
        x = off + 12; bsize = 0
        while x + 4 <= off + 12 + xlen:
            slen = int.from_bytes(buf[x + 2:x + 4], "little")
            if buf[x:x + 2] == b"BC" and slen == 2:
                bsize = int.from_bytes(buf[x + 4:x + 6], "little") + 1
            x += 4 + slen
        isize = int.from_bytes(buf[off + bsize - 4:off + bsize], "little")
        out.append((off + 12 + xlen, bsize - xlen - 20, isize, dst))
        dst += isize; off += bsize
    return out, dst


def _inflate_on_device(ctx, buf):
    import torch
    from phaser_amd import _lib
    tab, total = _members(buf)
    rec = np.zeros(len(tab), dtype=[("src", "<u8"), ("csize", "<u4"), ("isize", "<u4"), ("dst", "<u8")])
    for i, t in enumerate(tab):
        rec[i] = t
    comp = torch.zeros(len(buf) + 16, dtype=torch.uint8, device="cuda")
    comp[:len(buf)] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    drec = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
    out = torch.zeros(max(1, total), dtype=torch.uint8, device="cuda")
    bad = C.c_int(0)
    ctx.check(ctx.lib.phz_bgzf_inflate_device(ctx.h, C.c_void_p(comp.data_ptr()), C.c_void_p(drec.data_ptr()), len(tab), C.c_void_p(out.data_ptr()), C.byref(bad)))
    return out[:total].cpu().numpy().tobytes(), bad.value, tab


def _bgzf(payloads, level=6, wbits=-15, strategy=zlib.Z_DEFAULT_STRATEGY):
    """BGZF file with one member per payload (<= 65280 bytes each) + the EOF member"""
    out = bytearray()
    for p in list(payloads) + [b""]:
        c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
        body = c.compress(p) + c.flush()
        bsize = len(body) + 25
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + (bsize).to_bytes(2, "little") + body
        out += zlib.crc32(p).to_bytes(4, "little") + len(p).to_bytes(4, "little")
    return bytes(out)


def test_inflate_matches_zlib_on_every_block_type(ctx):
    """Dynamic-Huffman members (text, long matches, distance-1 runs), fixed-Huffman members (tiny inputs, Z_FIXED), stored members
    (level 0, incompressible bytes), empty members, several deflate blocks inside one member."""
    rng = np.random.default_rng(7)
    text = ("".join("chr%d\t%d\trs%d\t%s\t%s\t.\tPASS\tAF=%.3f\n" % (rng.integers(1, 23), rng.integers(1, 10 ** 8), rng.integers(1, 10 ** 7), "ACGT"[rng.integers(4)],
                                                                       "ACGT"[rng.integers(4)], rng.random()) for _ in range(5000))).encode()
    payloads = [text[i:i + 60000] for i in range(0, len(text), 60000)]
    payloads += [b"A" * 65000, b"AB" * 30000, bytes(rng.integers(0, 256, 65000, dtype=np.uint8)), bytes(rng.integers(0, 4, 50000, dtype=np.uint8)),
                 b"x", b"hello hello hello", b"", bytes(range(256)) * 200]
    for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY),
                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
        buf = _bgzf(payloads, level, -15, strategy)
        got, bad, tab = _inflate_on_device(ctx, buf)
        assert bad == 0, (level, strategy)
        assert got == b"".join(payloads), (level, strategy)
    # several deflate blocks in one member: Z_FULL_FLUSH between pieces
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = c.compress(text[:20000]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(text[20000:40000]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(text[40000:60000]) + c.flush()
    p = text[:60000]
    buf = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + (len(body) + 25).to_bytes(2, "little") + body + zlib.crc32(p).to_bytes(4, "little") + len(p).to_bytes(4, "little")
    got, bad, _ = _inflate_on_device(ctx, buf)
    assert bad == 0 and got == p


def test_inflate_reports_damage(ctx):
    """A member that is not valid DEFLATE, or does not produce ISIZE bytes, sets the status word (the caller then uses zlib)."""
    payload = b"the quick brown fox jumps over the lazy dog " * 500
    good = _bgzf([payload])
    tab, _ = _members(good)
    src, csize, isize, _ = tab[0]
    for mutate in ("flip", "isize", "truncate"):
        buf = bytearray(good)
        if mutate == "flip":
            for k in range(src + 20, src + 60):
                buf[k] ^= 0x5A
        elif mutate == "isize":
            end = src + csize + 8
            buf[end - 4:end] = (isize + 7).to_bytes(4, "little")
        else:
            for k in range(src + csize // 2, src + csize):
                buf[k] = 0
        got, bad, _ = _inflate_on_device(ctx, bytes(buf))
        assert bad != 0 or got != payload + b"", mutate      # never a silent wrong answer: either flagged, or (isize case) flagged
        if mutate in ("isize", "flip"):
            assert bad != 0, mutate


def _two_chrom_bam(tmp_path, n1=4000, n2=6000):
    from phaser_amd import bamio, synth
    v2, gs, ge, w = synth.make_variants("chr21", 1, 1_000_000, 80, 77, n_genes=6)
    rb2 = synth.make_reads(v2, gs, ge, w, n1, 78)
    v, gs, ge, w = synth.make_variants("chr22", 1, 2_000_000, 120, 79, n_genes=8)
    rb = synth.make_reads(v, gs, ge, w, n2, 80)
    path = str(tmp_path / "two.bam")
    bamio.readbatch_to_bam_native(path, [rb2, rb], [("chr21", 46709983), ("chr22", 50818468), ("chrEmpty", 1000)])
    return path


def _same(host, dev, what):
    import torch
    assert list(host) == list(dev), what
    for c in host:
        for f in FIELDS:
            assert torch.equal(getattr(host[c], f), getattr(dev[c], f).cpu()), (c, f, what)


def test_device_shards_equal_host_shards(ctx, tmp_path):
    """Every filter combination, whole file and chromosome-restricted: the GPU-decoded shards are the host decoder's, array by array;
    the interner sees the same names in the same order."""
    from phaser_amd import bamio
    path = _two_chrom_bam(tmp_path)
    for mapq, rmdup, paired, isz in itertools.product((0, 255), (False, True), (False, True), (0.0, 300.0)):
        for chroms in (None, {"chr21"}, {"chr22"}, {"chrEmpty"}):
            hi = {}; di = {}
            host = bamio.shards_from_bam_native(path, hi, mapq, rmdup, paired, isz, chroms=chroms, threads=2)
            dev = bamio.shards_from_bam_device(ctx, path, di, mapq, rmdup, paired, isz, chroms=chroms)
            assert dev is not None
            _same(host, dev, (mapq, rmdup, paired, isz, chroms))
            assert sorted(hi) == sorted(di)
            for c in hi:
                assert hi[c].names == di[c].names


def test_device_out_of_memory_falls_back_to_the_host_decoder(ctx, tmp_path, monkeypatch):
    """An allocation failure of the device BAM path (HBM short because earlier BAMs' shards stay resident) is PHZ_E_NOMEM at every
    allocation: the loader answers None, the host decoder takes the file, and the ctx carries no stale HIP error into the next launch
    (K_map right after it).  PHZ_BAMDEV_FORCE_NOMEM takes the out-of-memory exit without exhausting the GPU."""
    import torch
    from phaser_amd import bamio
    from phaser_amd.mapper import Mapper
    path = _two_chrom_bam(tmp_path)
    monkeypatch.setenv("PHZ_BAMDEV_FORCE_NOMEM", "1")
    assert bamio.shards_from_bam_device(ctx, path, {}, 255, True, True, 0.0) is None
    monkeypatch.delenv("PHZ_BAMDEV_FORCE_NOMEM")
    hi = {}; di = {}
    host = bamio.shards_from_bam_native(path, hi, 255, True, True, 0.0, threads=2)
    dev = bamio.shards_from_bam_device(ctx, path, di, 255, True, True, 0.0)       # the same ctx right after the failure
    assert dev is not None
    _same(host, dev, "after a forced out-of-memory")
    m = Mapper(0, ctx=ctx)
    c = next(iter(dev))
    vpos = torch.arange(1000, 2_000_000, 5000, dtype=torch.int32)
    calls = m.map(dev[c], vpos, 10)                                                 # a kernel launch + its error check on this ctx
    assert calls.n >= 0


def test_device_decoder_odd_records(ctx, tmp_path):
    """SEQ '*', QUAL missing, IUPAC bases, hard clips, padding, CIGAR longer than SEQ, no AS, AS in a wide type, B-array tags."""
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
    host = bamio.shards_from_bam_native(path, {}, 0, False, False)
    dev = bamio.shards_from_bam_device(ctx, path, {}, 0, False, False)
    assert dev is not None
    _same(host, dev, "odd records")


@pytest.mark.parametrize("L,pairs", [(30000, 30), (100000, 12)])
def test_long_reads_through_the_device_decoder_and_k_map(ctx, tmp_path, oracle_build, L, pairs):
    """Records of 30,000 / 100,000 bases (45 - 150 KB each: every record spans several BGZF members, a read covers hundreds of het
    SNPs): the device decoder's shards are the host decoder's array by array, and K_map on them gives the C oracle's calls for the
    arrays the BAM was written from."""
    import numpy as np
    import torch
    from helpers import oracle_map_readbatch
    from phaser_amd import bamio, synth
    from phaser_amd.mapper import Mapper
    v, gs, ge, w = synth.make_variants("chr22", 1, 3_000_000, 3000, 5, n_genes=8)
    rb = synth.make_reads(v, gs, ge, w, pairs, 6, L=L)
    path = str(tmp_path / "long.bam")
    bamio.readbatch_to_bam_native(path, [rb], [("chr22", 50818468)])
    host = bamio.shards_from_bam_native(path, {}, 0, False, False, 0.0, threads=2)
    dev = bamio.shards_from_bam_device(ctx, path, {}, 0, False, False, 0.0)
    assert dev is not None
    _same(host, dev, "long reads")
    calls = Mapper(0, ctx=ctx).map(dev["chr22"], v.pos, 10).cpu()
    o_r, o_v, o_c, _ = oracle_map_readbatch(oracle_build, rb, v.pos.numpy(), 10, with_text=False)
    assert calls.n == len(o_r) and calls.n > 50 * len(rb)
    assert np.array_equal(calls.read_idx.numpy(), o_r) and np.array_equal(calls.var_idx.numpy(), o_v) and np.array_equal(calls.code.numpy(), o_c)


def test_device_path_refuses_what_it_cannot_prove(ctx, tmp_path):
    """An unsorted BAM is declined (None: the host path then raises its own error); a truncated record is an error."""
    import dataclasses
    from phaser_amd import _lib, bamio, synth
    v, gs, ge, w = synth.make_variants("chr22", 1, 2_000_000, 100, 41, n_genes=5)
    rb = synth.make_reads(v, gs, ge, w, 400, 42)
    rf = rb.select(synth.samtools_keep(rb, 255))
    bad = dataclasses.replace(rf, pos=rf.pos.flip(0))
    path = str(tmp_path / "u.bam")
    bamio.readbatch_to_bam(path, [bad], [("chr22", 50818468)])
    assert bamio.shards_from_bam_device(ctx, path, {}, 0, False, False) is None
    with pytest.raises(_lib.PhzError):
        bamio.shards_from_bam_native(path, {}, 0, False, False)


def test_device_interning_then_a_second_bam(ctx, tmp_path):
    """First BAM of a chromosome: ids assigned on the GPU, names deferred; a second BAM (shared and new QNAMEs) materialises the
    interner and continues the numbering exactly as the all-host path does."""
    import torch
    from phaser_amd import bamio, synth
    path1 = _two_chrom_bam(tmp_path)
    v, gs, ge, w = synth.make_variants("chr22", 1, 2_000_000, 120, 79, n_genes=8)
    rb_same = synth.make_reads(v, gs, ge, w, 3000, 80)          # same seed: QNAMEs of the first BAM again
    rb_new = synth.make_reads(v, gs, ge, w, 2000, 81, qname_prefix="other.") if "qname_prefix" in synth.make_reads.__code__.co_varnames else synth.make_reads(v, gs, ge, w, 2000, 81)
    path2 = str(tmp_path / "second.bam")
    bamio.readbatch_to_bam_native(path2, [rb_same], [("chr21", 46709983), ("chr22", 50818468), ("chrEmpty", 1000)])
    path3 = str(tmp_path / "third.bam")
    bamio.readbatch_to_bam_native(path3, [rb_new], [("chr21", 46709983), ("chr22", 50818468), ("chrEmpty", 1000)])
    hi = {}; di = {}
    for p in (path1, path2, path3):
        host = bamio.shards_from_bam_native(p, hi, 0, False, False, threads=2)
        dev = bamio.shards_from_bam_device(ctx, p, di, 0, False, False)
        assert dev is not None
        _same(host, dev, p)
        for c in hi:
            assert len(hi[c]) == len(di[c]), (p, c)
    for c in hi:
        assert hi[c].names == di[c].names


def test_device_path_refuses_damaged_bgzf_members(ctx, tmp_path):
    """Bits flipped BELOW the BAM level -- in a member's CRC32 or in its deflate stream, which often stays valid DEFLATE of the right length: the device path checks
    every member's output against its trailer's CRC32 (k_crc32; htslib does the same, so the reference's samtools stops on such a file) and hands the file over
    (None) -- to the host decoder, which refuses it for the same reason.  It never returns shards."""
    import numpy as np
    from test_bamio import _damage_bgzf
    from phaser_amd import _lib, bamio, synth
    v, gs, ge, w = synth.make_variants("chr21", 1, 8_000_000, 400, 95, n_genes=40)
    rb = synth.make_reads(v, gs, ge, w, 20_000, 96)
    path = str(tmp_path / "m.bam")
    bamio.readbatch_to_bam_native(path, [rb], [("chr21", 46709983), ("chr22", 50818468)], 4)
    raw = open(path, "rb").read()
    good = bamio.shards_from_bam_device(ctx, path, {}, 0, False, False, 0.0, device="cuda:0")
    assert good is not None and good["chr21"].n > 10_000
    rng = np.random.default_rng(9)
    n_crc_only = 0
    for trial in range(16):
        bad = str(tmp_path / ("bad%d.bam" % trial))
        open(bad, "wb").write(_damage_bgzf(raw, rng, "crc" if trial % 2 == 0 else "payload"))
        try:
            got = bamio.shards_from_bam_device(ctx, bad, {}, 0, False, False, 0.0, device="cuda:0")
        except _lib.PhzError:
            got = None
        assert got is None, trial
        os.environ["PHZ_BAM_CRC"] = "0"                      # without the check: what DEFLATE alone lets through
        try:
            try:
                n_crc_only += bamio.shards_from_bam_device(ctx, bad, {}, 0, False, False, 0.0, device="cuda:0") is not None
            except _lib.PhzError:
                pass
        finally:
            del os.environ["PHZ_BAM_CRC"]
        with pytest.raises(_lib.PhzError):
            bamio.shards_from_bam_native(bad, {}, 0, False, False, threads=2)
    assert n_crc_only >= 8                                   # (every damaged checksum, and some of the damaged streams)


def test_device_path_on_corrupt_bam_streams(ctx, tmp_path):
    """The mutations of tests/test_native_robustness.py (block_size / l_read_name / n_cigar / l_seq of records, noise, cut tails, header
    fields) through the DEVICE path: it raises, declines (None) or returns exactly what the host decoder returns -- never a crash,
    never a different answer."""
    import gzip, random, struct
    import torch
    from phaser_amd import _lib, bamio, synth
    v, gs, ge, w = synth.make_variants("chr22", 1, 2_000_000, 100, 31, n_genes=8)
    rb = synth.make_reads(v, gs, ge, w, 400, 32)
    bam = str(tmp_path / "a.bam")
    bamio.readbatch_to_bam(bam, [rb], [("chr21", 46709983), ("chr22", 50818468)])
    raw = bytearray(gzip.open(bam, "rb").read())
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]; p += 4
    for _ in range(n_ref):
        l = struct.unpack_from("<i", raw, p)[0]; p += 4 + l + 4
    first = p
    recs = []
    while p + 4 <= len(raw):
        recs.append(p); p += 4 + struct.unpack_from("<i", raw, p)[0]
    rng = random.Random(11)

    def write(data, name):
        path = str(tmp_path / name)
        assert _lib.load().phz_bgzf_write(path.encode(), bytes(data), len(data), 2, 1) == 0
        return path
    outcomes = {"same": 0, "both raise": 0, "device raises, host decodes": 0}
    for it in range(96):
        m = bytearray(raw)
        kind = it % 8
        r = rng.choice(recs)
        if kind == 0:
            struct.pack_into("<i", m, r, rng.choice([-1, 0, 31, 33, 1 << 30, struct.unpack_from("<i", m, r)[0] - 1]))
        elif kind == 1:
            m[r + 4 + 8] = rng.choice([0, 1, 255])
        elif kind == 2:
            struct.pack_into("<H", m, r + 4 + 12, rng.choice([0, 1000, 65535]))
        elif kind == 3:
            struct.pack_into("<i", m, r + 4 + 16, rng.choice([-5, 0, 1 << 20, 0x7fffffff]))
        elif kind == 4:
            m = m[:rng.randrange(first, len(m))]
        elif kind == 5:
            for _ in range(rng.randrange(1, 20)):
                m[rng.randrange(first, len(m))] = rng.randrange(256)
        elif kind == 6:
            struct.pack_into("<i", m, rng.choice([4, 8 + l_text, 8 + l_text + 4]), rng.choice([-1, 0x7fffffff, 1 << 28, 3]))
        else:
            m = m[:rng.randrange(0, first + 8)]
        path = write(m, "m%d.bam" % it)
        try:
            host = bamio.shards_from_bam_native(path, {}, 0, False, False, 0.0, threads=2)
        except _lib.PhzError:
            host = "raise"
        try:
            dev = bamio.shards_from_bam_device(ctx, path, {}, 0, False, False, 0.0)
        except _lib.PhzError:
            dev = "raise"
        if dev is None:
            outcomes["declined (host path decides)"] = outcomes.get("declined (host path decides)", 0) + 1
        elif dev == "raise":
            outcomes["both raise" if host == "raise" else "device raises, host decodes"] += 1
        else:
            assert host != "raise", (it, kind)
            _same(host, dev, (it, kind))
            outcomes["same"] += 1
    torch.cuda.synchronize()
    assert outcomes["same"] >= 10 and outcomes.get("declined (host path decides)", 0) + outcomes["both raise"] >= 20, outcomes


def test_deep_bam_is_decoded_in_chromosome_halves(ctx, tmp_path, monkeypatch):
    """One device call is limited to 2^32 bytes of names / base groups; beyond that the loader splits the chromosomes into halves
    (lowered limit here) and the result is still the host decoder's."""
    from phaser_amd import bamio
    path = _two_chrom_bam(tmp_path)
    host = bamio.shards_from_bam_native(path, {}, 0, False, False, threads=2)
    monkeypatch.setenv("PHZ_BAMDEV_LIMIT", str(300_000))          # either chromosome fits (152k and 228k base groups), both together do not
    di = {}
    dev = bamio.shards_from_bam_device(ctx, path, di, 0, False, False)
    assert dev is not None
    _same(host, {c: dev[c] for c in host}, "split")
    monkeypatch.setenv("PHZ_BAMDEV_LIMIT", "1000")                # nothing fits: declined, the host path takes over
    assert bamio.shards_from_bam_device(ctx, path, {}, 0, False, False) is None
