"""Native SAM-text front end of the mapper seam (phz_sam_parse / phz_sam_calls_tsv) vs the Python reader + packer it replaces
(samio.shards_from_sam -> soa.pack_sam, read_variant_map._allele_text): identical arrays and identical output lines, on the
configs[0] input and on deliberately odd records."""
import ctypes as C
import io
import random

import numpy as np
import pytest
import torch


def _native(sam_text, isize=0.0, threads=3):
    from phaser_amd import _lib, read_variant_map as prvm
    _lib.build()
    return prvm._native_sam(sam_text.encode(), isize, threads)


def _compare(sam_text, isize=0.0):
    from phaser_amd import samio
    owner, contigs, shards = _native(sam_text, isize)
    want = samio.shards_from_sam(sam_text, {}, isize)
    assert [c for c, _ in shards] == list(want)
    for chrom, sh in shards:
        w = want[chrom]
        for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual"):
            assert torch.equal(getattr(sh, f), getattr(w, f)), (chrom, f)
    return owner, contigs, shards, want


def test_c1_sam_arrays_identical(c1_inputs):
    owner, contigs, shards, want = _compare(c1_inputs["sam"])
    assert contigs == ["chr22"] and shards[0][1].n == len(c1_inputs["reads"])
    _compare(c1_inputs["sam"], isize=260.0)


def _odd_sam(rng, n):
    ops = "MMMMMIDNSHP=X"
    lines = ["@HD\tVN:1.6", "@SQ\tSN:chrF\tLN:5000", "@SQ\tSN:chrG:x\tLN:900", "@PG\tID:x"]
    pos = 100
    for i in range(n):
        pos += rng.randint(0, 30)
        cig = "".join("%d%s" % (rng.randint(0, 12), rng.choice(ops)) for _ in range(rng.randint(1, 6)))
        if rng.random() < 0.1:
            cig = "*" if rng.random() < 0.5 else cig + "7"            # '*' and a dangling number
        L = rng.randint(0, 60)
        seq = "".join(rng.choice("ACGTACGTNRYD*") for _ in range(L)) or "*"
        qual = "".join(chr(33 + rng.randint(0, 60)) for _ in range(max(0, L + rng.randint(-3, 3)))) or "*"
        tags = []
        for _ in range(rng.randint(0, 4)):
            tags.append(rng.choice(["NH:i:1", "AS:i:%d" % rng.randint(-20, 160), "XS:Z:AS:", "AS:Z:5:9", "nM:i:0"]))
        tags = [t for t in tags if t != "AS:Z:5:9" or True]
        line = "\t".join(["q%d" % (i // 2), str(rng.choice([99, 147, 355])), rng.choice(["chrF", "chrF", "chrG:x"]), str(pos), "255", cig, "=",
                          str(pos + 100), str(rng.choice([250, -250, 0, 90000])), seq, qual] + tags)
        lines.append(line + rng.choice(["", " ", "\t"]))
    return "\n".join(lines) + "\n"


def test_odd_records_arrays_identical():
    rng = random.Random(5)
    for rep in range(30):
        text = _odd_sam(rng, 120)
        # keep each chromosome coordinate-sorted (both readers refuse inversions): positions only grow in _odd_sam
        _compare(text, isize=rng.choice([0.0, 300.0]))


def test_as_and_counts(c1_inputs):
    from phaser_amd import _lib, samio
    text = _odd_sam(random.Random(9), 300)
    owner, contigs, shards, want = _compare(text)
    assert contigs == ["chrF", "chrG"]                                  # columns[1].split(":")[1]
    lib = _lib.load()
    assert lib.phz_sam_n_records(owner.h) == sum(1 for l in text.split("\n") if l and l[0] != "@")
    for si, (chrom, sh) in enumerate(shards):
        hs = _lib.phz_host_shard()
        lib.phz_sam_shard(owner.h, si, C.byref(hs))
        n = int(hs.n_reads)
        aln = np.ctypeslib.as_array(C.cast(hs.aln_score, C.POINTER(C.c_int32)), (n,)); has = np.ctypeslib.as_array(C.cast(hs.has_as, C.POINTER(C.c_uint8)), (n,))
        assert aln.tolist() == want[chrom].aln_score.tolist() and has.tolist() == want[chrom].has_as.tolist()


def test_malformed_input_is_a_status():
    from phaser_amd import _lib
    for bad in ("q1\t0\tchr1\t10\n", "q1\t0\tchr1\tx\t255\t5M\t=\t1\t0\tACGTA\tIIIII\n", "q1\t0\tchr1\t5\t255\t5M\t=\t1\tzz\tACGTA\tIIIII\n",
                "@SQ\n", "q1\t0\tchr1\t9\t255\t5M\t=\t1\t0\tACGTA\tIIIII\nq2\t0\tchr1\t3\t255\t5M\t=\t1\t0\tACGTA\tIIIII\n"):
        with pytest.raises(_lib.PhzError):
            _native(bad)


def test_tsv_lines_match_python_formatter():
    """phz_sam_calls_tsv vs the Python line formatter on synthetic call lists (single bases, composite texts with inserted bases,
    low-quality characters, IUPAC symbols, a 'D' that gets stripped)."""
    from phaser_amd import _lib, read_variant_map as prvm
    from phaser_amd.vcf import sep_pool
    rng = random.Random(3)
    recs = []
    lines = ["@SQ\tSN:chrT\tLN:1000"]
    for i in range(200):
        L = rng.randint(5, 40)
        seq = "".join(rng.choice("ACGTACGTNRD") for _ in range(L)); qual = "".join(chr(33 + rng.choice([2, 11, 25, 37])) for _ in range(L))
        tag = ["AS:i:%d" % rng.randint(0, 150)] if rng.random() < 0.8 else []
        recs.append(("r%d" % i, seq, qual, tag[0].split(":")[2] if tag else ""))
        lines.append("\t".join(["r%d" % i, "99", "chrT", str(10 + i), "255", "%dM" % L, "=", "1", "0", seq, qual] + tag))
    owner, contigs, shards = _native("\n".join(lines) + "\n")
    nv = 50
    ids = ["chrT_%d_A_C" % v for v in range(nv)]; rs = ["rs%d" % v for v in range(nv)]; gts = [rng.choice(["0|1", "1|0", "0/1"]) for _ in range(nv)]; mafs = ["None"] * nv
    n = 600
    ri = np.sort(np.array([rng.randrange(200) for _ in range(n)], dtype=np.int32)); vi = np.array([rng.randrange(nv) for _ in range(n)], dtype=np.int32)
    code = np.array([rng.choice([0, 1, 2, 3, 4, 4]) for _ in range(n)], dtype=np.uint8)
    a0 = np.zeros(n, dtype=np.uint32); a1 = np.zeros(n, dtype=np.uint32)
    for k in range(n):
        L = len(recs[ri[k]][1])
        a0[k] = rng.randrange(L) if rng.random() < 0.85 else 0xFFFFFFFF
        if code[k] == 4 and rng.random() < 0.6:
            off = rng.randrange(L); ln = rng.randint(1, min(3, L - off))
            a1[k] = (off << 12) | ln
    want = []
    for k in range(n):
        r = recs[ri[k]]
        allele = prvm._allele_text(int(code[k]), int(a0[k]), int(a1[k]), r[1], r[2], 10)
        want.append("\t".join([r[0], ids[vi[k]], rs[vi[k]], allele, r[3], gts[vi[k]], mafs[vi[k]]]))
    lib = _lib.load()
    pools = [sep_pool(x) for x in (ids, rs, gts, mafs)]
    pa = []
    for off, b in pools:
        pa += [C.c_void_p(off.ctypes.data), C.cast(C.c_char_p(b), C.c_void_p)]
    p = C.c_void_p(); m = C.c_int64(0)
    st = lib.phz_sam_calls_tsv(owner.h, 0, n, *[C.c_void_p(x.ctypes.data) for x in (ri, vi, code, a0, a1)], 10, *pa, 3, C.byref(p), C.byref(m))
    assert st == 0
    got = C.string_at(p, m.value).decode(); lib.phz_buf_free(p)
    assert got == "\n".join(want) + "\n"
