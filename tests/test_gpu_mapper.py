"""GPU parity tests of K_map through the C ABI (phz_map_reads) against the pinned oracle and the golden
outputs of the reference mapper."""
import io
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLD, gz_text
from helpers import oracle_map_readbatch, variant_table_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mapper():
    from phaser_amd.mapper import Mapper
    return Mapper(0)


def run_dropin(mapper, sam_text, table_path, out_path, baseq, isize):
    from phaser_amd import read_variant_map
    old = sys.stdin
    sys.stdin = io.StringIO(sam_text)
    try:
        read_variant_map.do_read_variant_map(table_path, baseq, out_path, 1, isize, _mapper=mapper)
    finally:
        sys.stdin = old
    return open(out_path).read()


def test_kat_micro(mapper):
    from phaser_amd import soa
    from phaser_amd.read_variant_map import _allele_text
    cases = json.load(open(os.path.join(GOLD, "kat_micro.json")))
    checked = 0
    for c in cases:
        vs = [v for v in c["variants"] if v["ref_len"] == 1]
        if not vs:
            continue
        vs = sorted(vs, key=lambda v: v["pos"])    # stable: keeps table order of duplicates
        shard = soa.pack_sam([(c["pos"], c["cigar"], c["seq"], c["qual"])])
        calls = mapper.map(shard, torch.tensor([v["pos"] for v in vs], dtype=torch.int32), c["baseq"]).cpu()
        got = []
        for k in range(calls.n):
            a0 = int(calls.aux0[k]) & 0xFFFFFFFF; a1 = int(calls.aux1[k]) & 0xFFFFFFFF
            got.append((int(calls.var_idx[k]), _allele_text(int(calls.code[k]), a0, a1, c["seq"], c["qual"], c["baseq"])))
        want = []
        for s in range(len(c["segments"])):
            for i, v in enumerate(vs):
                if v["per_segment"][s] != "":
                    want.append((i, v["per_segment"][s]))
        if c["name"] == "iupac":
            # documented deviation: an IUPAC 'D' base is treated like 'N' (no call); the reference strips it
            # (read_variant_map.py:254) which is also "no call" for a lone base -- identical here
            pass
        assert got == want, c["name"]
        checked += 1
    assert checked >= 20


def test_mapper_small_bytes(mapper, tmp_path):
    d = os.path.join(GOLD, "mapper_small")
    meta = json.load(open(os.path.join(d, "meta.json")))
    sam = gz_text(os.path.join(d, "in.sam.gz"))
    for run in meta["runs"]:
        got = run_dropin(mapper, sam, os.path.join(d, "table.tsv"), str(tmp_path / "o.tsv"), run["baseq"], run["isize"])
        assert got == gz_text(os.path.join(d, run["file"])), run


def test_c1_calls_bytes(mapper, c1_inputs, tmp_path):
    tp = tmp_path / "t.tsv"
    tp.write_text(variant_table_text(c1_inputs["variants"]))
    got = run_dropin(mapper, c1_inputs["sam"], str(tp), str(tmp_path / "o.tsv"), 10, 0)
    assert got == gz_text(os.path.join(GOLD, "c1", "calls.tsv.gz"))


@pytest.mark.parametrize("n_pairs,n_snps,baseq,seed", [(200_000, 3000, 10, 5), (150_000, 20_000, 30, 6), (1500, 40, 0, 7)])
def test_random_vs_oracle(mapper, oracle_build, n_pairs, n_snps, baseq, seed):
    """Device-resident shard (packed on the GPU) vs the C restatement on identical seeded inputs."""
    from phaser_amd import soa, synth
    v, gs, ge, w = synth.make_variants("chr1", 1, 30_000_000, n_snps, seed, n_genes=max(4, n_snps // 25))
    rb = synth.make_reads(v, gs, ge, w, n_pairs, seed + 100, n_rate=0.002)
    rb = rb.select(synth.samtools_keep(rb, 255))
    o_r, o_v, o_c, o_t = oracle_map_readbatch(oracle_build, rb, v.pos.numpy(), baseq)
    shard = soa.pack_readbatch(rb).to("cuda")
    calls = mapper.map(shard, v.pos, baseq, cap=16).cpu()      # tiny cap: exercises the capacity retry
    assert calls.n == len(o_r)
    assert np.array_equal(calls.read_idx.numpy(), o_r)
    assert np.array_equal(calls.var_idx.numpy(), o_v)
    assert np.array_equal(calls.code.numpy(), o_c)
    # composite calls: text must match too
    from phaser_amd.read_variant_map import _allele_text
    idx = np.nonzero(o_c == 4)[0]
    lut = "ACGTN"
    for k in idx[:2000]:
        r = int(o_r[k])
        seq = "".join(lut[x] for x in rb.seq[r].tolist()); qual = "".join(chr(33 + q) for q in rb.qual[r].tolist())
        txt = _allele_text(4, int(calls.aux0[k]) & 0xFFFFFFFF, int(calls.aux1[k]) & 0xFFFFFFFF, seq, qual, baseq)
        assert txt == o_t[k]


@pytest.mark.parametrize("baseq", [10, 40, 63, 70])
def test_one_byte_plane_equals_two_planes_and_the_oracle(oracle_build, monkeypatch, baseq):
    """K_map's ONE instantiation (base and quality of a call from one byte: soa.bq_plane, phz_reads.bq; PHZ_MAP_ONE_PLANE=1) against its two-plane
    instantiation (seq2 + qual, the default) and against the C oracle, on reads with everything the plane has to escape for --
    N bases, phred values above 62 (the plane holds six quality bits) -- and with --baseq on both sides of 62."""
    from phaser_amd import soa, synth
    from phaser_amd.mapper import Mapper
    v, gs, ge, w = synth.make_variants("chr1", 1, 20_000_000, 4000, 61, n_genes=160)
    rb = synth.make_reads(v, gs, ge, w, 120_000, 62, n_rate=0.01)
    rb = rb.select(synth.samtools_keep(rb, 255))
    g = torch.Generator().manual_seed(63)
    hi = torch.rand(rb.qual.shape, generator=g) < 0.2
    rb.qual[hi] = torch.randint(60, 94, (int(hi.sum()),), generator=g, dtype=rb.qual.dtype)      # phred 60..93: around and beyond the six bits
    o_r, o_v, o_c, _ = oracle_map_readbatch(oracle_build, rb, v.pos.numpy(), baseq, with_text=False)
    shard = soa.pack_readbatch(rb).to("cuda")
    got = {}
    for mode in ("two", "one"):
        if mode == "one":
            monkeypatch.setenv("PHZ_MAP_ONE_PLANE", "1")          # (off by default: the plane buys the production kernel nothing, soa.bq_plane)
            bq = soa.bq_plane(shard)
            assert bq is not None and int(((bq & 63) == 63).sum()) > 1000          # escapes are present
        else:
            monkeypatch.delenv("PHZ_MAP_ONE_PLANE", raising=False)
            assert soa.bq_plane(shard) is None
        m = Mapper(0)
        got[mode] = m.map(shard, v.pos, baseq).cpu()
    for mode, c in got.items():
        assert c.n == len(o_r) and len(o_r) > 10_000, mode
        assert np.array_equal(c.read_idx.numpy(), o_r) and np.array_equal(c.var_idx.numpy(), o_v) and np.array_equal(c.code.numpy(), o_c), mode
    assert np.array_equal(got["one"].aux0.numpy(), got["two"].aux0.numpy()) and np.array_equal(got["one"].aux1.numpy(), got["two"].aux1.numpy())


@pytest.mark.parametrize("n_snps", [1500, 3000, 7800])
def test_dense_windows_vs_oracle(mapper, oracle_build, n_snps):
    """Het SNPs packed into six short genes: the staged window of a tile holds anything from 8 to several thousand entries, so the
    tiles of one shard exercise every entry depth of the unrolled window search (2^d <= window < 2^(d+1), up to the 512 staged
    entries), windows exactly at and beyond that capacity (truncated: the LDS-only walkers are off, the general walker reads the
    table) and records with tens of het SNPs under them."""
    from phaser_amd import soa, synth
    v, gs, ge, w = synth.make_variants("chr1", 1, 2_000_000, n_snps, 11, n_genes=6)
    rb = synth.make_reads(v, gs, ge, w, 60000, 12, n_rate=0.002)
    rb = rb.select(synth.samtools_keep(rb, 255))
    pos = rb.pos.numpy().astype(np.int64); vp = v.pos.numpy().astype(np.int64)
    first = pos[::256]; last = pos[np.minimum(np.arange(255, len(pos) + 255, 256), len(pos) - 1)]
    wl = np.searchsorted(vp, last + 65536) - np.searchsorted(vp, first) + 8
    depths = set(int(x).bit_length() for x in wl if x <= 512)
    assert len(depths) >= 3 and (n_snps < 3000 or (wl > 512).any())    # the inputs do what the docstring says
    o_r, o_v, o_c, o_t = oracle_map_readbatch(oracle_build, rb, v.pos.numpy(), 10)
    calls = mapper.map(soa.pack_readbatch(rb).to("cuda"), v.pos, 10).cpu()
    assert calls.n == len(o_r) and calls.n > 10000
    assert np.array_equal(calls.read_idx.numpy(), o_r) and np.array_equal(calls.var_idx.numpy(), o_v) and np.array_equal(calls.code.numpy(), o_c)


@pytest.mark.parametrize("seed,max_gap_ops,snp_every", [(31, 4, 25), (32, 40, 25), (33, 40, 6)])
def test_many_op_records_vs_oracle(mapper, oracle_build, seed, max_gap_ops, snp_every):
    """Thousands of records made of many short runs (the shape of long, indel-rich reads): M / = / X runs separated by I, D and N in random
    order, soft clips at the ends, 400 bases each.  With up to 40 gaps per record a 256-record tile has ten times the CIGAR words its LDS
    staging holds (the general walker reads the operators from global memory), the runs are a few bases long so insertions sit next to het SNPs
    all the time (composite calls, text compared), and with a het SNP every 6 bases a record has up to ~60 calls.  All of it against the C oracle."""
    from phaser_amd import soa, synth
    from phaser_amd.read_variant_map import _allele_text
    rng = np.random.default_rng(seed)
    L, n = 400, 6000
    OP = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "=": 7, "X": 8}
    words = []; coff = [0]; span = []
    for _ in range(n):
        ops = []
        left = L
        lead = int(rng.integers(0, 15)) if rng.random() < 0.3 else 0
        trail = int(rng.integers(0, 15)) if rng.random() < 0.3 else 0
        if lead: ops.append(("S", lead)); left -= lead
        left -= trail
        gaps = int(rng.integers(1, max_gap_ops + 1))
        ref = 0
        for g in range(gaps):
            run = int(rng.integers(1, max(2, left // (gaps - g + 1) + 1)))
            run = min(run, left - (gaps - g))           # keep a base for every later run
            if run < 1: break
            ops.append((str(rng.choice(["M", "M", "M", "=", "X"])), run)); left -= run; ref += run
            kind = str(rng.choice(["I", "D", "N", "N"]))
            if kind == "I":
                k = int(rng.integers(1, 6)); k = min(k, left - 1)
                if k < 1: continue
                ops.append(("I", k)); left -= k
            elif kind == "D":
                k = int(rng.integers(1, 9)); ops.append(("D", k)); ref += k
            else:
                k = int(rng.integers(20, 3000)); ops.append(("N", k)); ref += k
        if ops and ops[-1][0] in "IDN":                 # a record does not end on a gap
            pass
        ops.append(("M", left)); ref += left
        if trail: ops.append(("S", trail))
        assert sum(k for o, k in ops if o in "MIS=X") == L
        words += [(k << 4) | OP[o] for o, k in ops]; coff.append(len(words)); span.append(ref)
    pos = np.sort(rng.integers(1000, 600_000, n)).astype(np.int32)
    z = torch.zeros(n, dtype=torch.int32)
    rb = synth.ReadBatch("chr1", L, torch.from_numpy(pos), z, torch.full((n,), 255, dtype=torch.uint8), z, z, torch.arange(n, dtype=torch.int32),
                         torch.tensor(coff, dtype=torch.int64), torch.tensor(words, dtype=torch.int64),
                         torch.from_numpy(rng.integers(0, 4, (n, L)).astype(np.uint8)), torch.from_numpy(rng.integers(2, 41, (n, L)).astype(np.uint8)))
    vpos = np.unique(rng.integers(900, 600_000 + max(span) + 100, (600_000 + max(span)) // snp_every)).astype(np.int32)
    o_r, o_v, o_c, o_t = oracle_map_readbatch(oracle_build, rb, vpos, 10)
    calls = mapper.map(soa.pack_readbatch(rb).to("cuda"), torch.from_numpy(vpos), 10).cpu()
    assert calls.n == len(o_r) and (calls.n > 20000 or snp_every > 25)
    assert np.array_equal(calls.read_idx.numpy(), o_r) and np.array_equal(calls.var_idx.numpy(), o_v) and np.array_equal(calls.code.numpy(), o_c)
    comp = np.nonzero(o_c == 4)[0]
    assert len(comp) > 50 or snp_every > 25 or max_gap_ops < 4      # insertions next to het SNPs did occur
    lut = "ACGTN"
    for k in comp[:1500]:
        r = int(o_r[k])
        seq = "".join(lut[x] for x in rb.seq[r].tolist()); qual = "".join(chr(33 + q) for q in rb.qual[r].tolist())
        assert _allele_text(4, int(calls.aux0[k]) & 0xFFFFFFFF, int(calls.aux1[k]) & 0xFFFFFFFF, seq, qual, 10) == o_t[k]


def test_long_records_wide_offsets(mapper, oracle_build):
    """Offsets of the called base at and beyond 2^16 and calls carrying inserted text leave the packed 8-byte staging record
    (side planes, phz_map.hip stage_put): single-run and multi-op records of 200 kb against the oracle, offsets and text included."""
    from phaser_amd import soa, synth
    from phaser_amd.read_variant_map import _allele_text
    from phaser_amd.soa import parse_cigar
    rng = np.random.default_rng(11)
    L = 200000
    cigars = ["200000M", "65530M3I10M2D134457M", "100S65536M5N134364M", "65535M1I134464M", "70000M130000S", "200000M"]
    n = len(cigars)
    ops = [[(ln << 4) | op for op, ln in parse_cigar(c)] for c in cigars]
    coff = np.zeros(n + 1, np.int64); coff[1:] = np.cumsum([len(o) for o in ops])
    z = torch.zeros(n, dtype=torch.int32)
    rb = synth.ReadBatch("chr1", L, torch.tensor([1000, 1000, 1000, 1001, 1500, 70000], dtype=torch.int32), z, torch.full((n,), 255, dtype=torch.uint8),
                         z, z, torch.arange(n, dtype=torch.int32), torch.from_numpy(coff),
                         torch.tensor([x for o in ops for x in o], dtype=torch.int64),
                         torch.from_numpy(rng.integers(0, 4, (n, L)).astype(np.uint8)), torch.from_numpy(rng.integers(2, 41, (n, L)).astype(np.uint8)))
    edge = 1000 + np.array([65528, 65529, 65530, 65531, 65534, 65535, 65536, 65537, 65540, 65541, 65542])
    vpos = np.unique(np.concatenate([rng.integers(1000, 271000, 400), edge])).astype(np.int32)
    baseq = 10
    o_r, o_v, o_c, o_t = oracle_map_readbatch(oracle_build, rb, vpos, baseq)
    calls = mapper.map(soa.pack_readbatch(rb).to("cuda"), torch.from_numpy(vpos), baseq).cpu()
    assert calls.n == len(o_r) and calls.n > 500
    assert np.array_equal(calls.read_idx.numpy(), o_r) and np.array_equal(calls.var_idx.numpy(), o_v)
    assert np.array_equal(calls.code.numpy(), o_c)
    a0 = calls.aux0.numpy().view(np.uint32); a1 = calls.aux1.numpy().view(np.uint32)
    lut = "ACGTN"
    wide = texts = 0
    for k in range(calls.n):
        r = int(o_r[k])
        if o_c[k] < 4:
            assert int(rb.seq[r, int(a0[k])]) == int(o_c[k]) and a1[k] == 0        # the offset names the called base
            wide += int(a0[k]) >= 65536
        else:
            seq = "".join(lut[x] for x in rb.seq[r].tolist()); qual = "".join(chr(33 + q) for q in rb.qual[r].tolist())
            assert _allele_text(4, int(a0[k]), int(a1[k]), seq, qual, baseq) == o_t[k]
            texts += 1
    assert wide > 100 and texts >= 1


def test_empty_and_edge_shards(mapper):
    from phaser_amd import soa
    # no reads
    shard = soa.pack_sam([])
    assert mapper.map(shard, torch.tensor([5], dtype=torch.int32), 10).n == 0
    # no variants
    shard = soa.pack_sam([(100, "10M", "ACGTACGTAC", "I" * 10)])
    assert mapper.map(shard, torch.zeros(0, dtype=torch.int32), 10).n == 0
    # every base of one read is a het site; variants before / after the read too
    vpos = torch.arange(90, 120, dtype=torch.int32)
    calls = mapper.map(shard, vpos, 10).cpu()
    assert calls.var_idx.tolist() == list(range(10, 20))
    assert calls.code.tolist() == [0, 1, 2, 3, 0, 1, 2, 3, 0, 1]
    # the SNP kernel refuses indel variants loudly (they go through map_general instead)
    from phaser_amd import _lib
    with pytest.raises(_lib.PhzError):
        mapper.map(shard, vpos, 10, ref_len=torch.full((30,), 2, dtype=torch.uint8))


@pytest.mark.parametrize("n_snps,span,shift", [(400, 2_000_000, 0), (20_000, 2_000_000, 0), (200_000, 2_000_000, 0), (3000, 3_000_000, 1_200_000_000)])
def test_general_kernel_agrees_with_snp_kernel(mapper, n_snps, span, shift):
    """K_map_general on het SNPs handed over as allele strings must produce K_map's (record, variant) list, with code 5 / 6 where
    the base is allele 0 / 1.  The four shapes walk its fast pass, its staged-window overflow (one SNP per 10 bp: every record is
    handed to the work list), and the hand-over of coordinates beyond 2^30."""
    from phaser_amd import soa, synth
    v, gs, ge, w = synth.make_variants("chr1", 1, span, n_snps, 31, n_genes=max(4, n_snps // 50))
    rb = synth.make_reads(v, gs, ge, w, 60_000, 32, n_rate=0.002)
    rb = rb.select(synth.samtools_keep(rb, 255))
    if shift:
        rb.pos = rb.pos + shift
    vpos = (v.pos + shift).to(torch.int32)
    shard = soa.pack_readbatch(rb).to("cuda")
    base = mapper.map(shard, vpos, 10).cpu()
    nv = len(v)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    ab = np.zeros(2 * nv + 1, dtype=np.uint8); ab[0:2 * nv:2] = letters[v.ref.numpy()]; ab[1:2 * nv:2] = letters[v.alt.numpy()]
    calls, pool = mapper.map_general(shard, vpos, torch.ones(nv, dtype=torch.uint8), torch.arange(2 * nv + 1, dtype=torch.int32),
                                     torch.from_numpy(ab), 10, want_text=True)
    calls = calls.cpu()
    assert base.n > 1000 and calls.n == base.n
    assert torch.equal(calls.read_idx, base.read_idx) and torch.equal(calls.var_idx, base.var_idx)
    bc = base.code.numpy().astype(np.int64); gc = calls.code.numpy().astype(np.int64); vi = base.var_idx.numpy()
    single = bc < 4
    want = np.where(bc == v.ref.numpy()[vi], 5, np.where(bc == v.alt.numpy()[vi], 6, bc))
    assert np.array_equal(gc[single], want[single])
    assert np.all(gc[~single] == 4) or np.all(np.isin(gc[~single], (4, 5, 6)))
    # text pool: one offset per character of every code-4 call, in call order
    toff = pool.call_off.numpy(); n4 = int((gc == 4).sum())
    assert toff[0] == 0 and toff[-1] == len(pool.roff) and np.all(np.diff(toff) >= 0) and (np.diff(toff) > 0).sum() == n4


def test_indel_mode_calls_bytes(mapper, tmp_path):
    """Mapper TSV with indel variants in the table (ref_len > 1, multi-base alleles): byte-identical to the reference."""
    d = os.path.join(GOLD, "pipe_indel")
    got = run_dropin(mapper, gz_text(os.path.join(d, "i.chr22.sam.gz")), os.path.join(d, "table.chr22.tsv"), str(tmp_path / "o.tsv"), 10, 0)
    assert got == gz_text(os.path.join(d, "calls.i.chr22.tsv.gz"))


def test_kat_micro_indel_variants(mapper, tmp_path):
    """The known-answer cases whose variants are indels (ref_len 2..4), one table per case through the drop-in."""
    cases = json.load(open(os.path.join(GOLD, "kat_micro.json")))
    n = 0
    for c in cases:
        vs = [v for v in c["variants"] if v["ref_len"] > 1]
        if not vs:
            continue
        vs = sorted(vs, key=lambda v: v["pos"])
        tp = tmp_path / ("t%d.tsv" % n)
        tp.write_text("".join("\t".join(["1", str(v["pos"]), "1_%d_x" % v["pos"], ".", v["alleles"], str(v["ref_len"]), "0|1", "None"]) + "\n" for v in vs))
        sam = "@SQ\tSN:1\tLN:1000\n" + "\t".join(["r", "0", "1", str(c["pos"]), "255", c["cigar"], "*", "0", "0", c["seq"], c["qual"], "AS:i:9"]) + "\n"
        got = run_dropin(mapper, sam, str(tp), str(tmp_path / "o.tsv"), c["baseq"], 0)
        want = ""
        for s_i in range(len(c["segments"])):
            for v in vs:
                if v["per_segment"][s_i] != "":
                    want += "\t".join(["r", "1_%d_x" % v["pos"], ".", v["per_segment"][s_i], "9", "0|1", "None"]) + "\n"
        assert got == want, c["name"]
        n += 1
    assert n >= 3


# (test_full_size_properties moved to tests/test_gpu_scale.py::test_configs1_chr1_50m_records: configs[1] is checked there on ALL 50M records and against the phasing oracle)

def test_stream_out_of_coordinate_order(mapper, tmp_path):
    """A SAM stream whose records are out of coordinate order.  The reference's variant buffer is forward-only (read_variant_map.py:37-50,
    :88-93, :106-112): a record that steps backwards misses the variants the buffer has let go, and a record far ahead makes the mapper
    skip variants for good.  The native parser declines such a stream; the Python path maps it in sorted order, drops the calls the buffer
    would not have made and puts the lines back into stream order: the REFERENCE's bytes on both fixtures (local disorder; records moved
    far ahead), with and without the isize filter (a filtered record still prunes the buffer)."""
    d = os.path.join(GOLD, "mapper_unsorted")
    meta = json.load(open(os.path.join(d, "meta.json")))
    assert {r["stream"] for r in meta["runs"]} == {"local", "far"}
    for run in meta["runs"]:
        sam = gz_text(os.path.join(d, "in_%s.sam.gz" % run["stream"]))
        got = run_dropin(mapper, sam, os.path.join(GOLD, "mapper_small", "table.tsv"), str(tmp_path / "o.tsv"), run["baseq"], run["isize"])
        assert got == gz_text(os.path.join(d, run["file"])), run


def test_chromosome_coming_back_is_refused(mapper, tmp_path):
    """chrA, chrB, chrA: the reference's variant stream cannot rewind and its allele test compares positions only -- there is nothing sensible
    to reproduce; the drop-in stops with the mapper's kind of error (message + exit status 1)."""
    d = os.path.join(GOLD, "mapper_small")
    sam = gz_text(os.path.join(d, "in.sam.gz"))
    head = [l for l in sam.split("\n") if l.startswith("@")] + ["@SQ\tSN:chrOther\tLN:1000000"]
    recs = [l for l in sam.split("\n") if l and not l.startswith("@")]
    other = recs[10].split("\t"); other[2] = "chrOther"
    text = "\n".join(head + recs[:20] + ["\t".join(other)] + recs[20:40]) + "\n"
    with pytest.raises(SystemExit) as e:
        run_dropin(mapper, text, os.path.join(d, "table.tsv"), str(tmp_path / "o.tsv"), 10, 0)
    assert e.value.code == 1


def test_call_lists_without_the_text_planes(mapper, oracle_build):
    """phz_calls.aux0 / aux1 = NULL (what the phasing stage asks for): the (record, variant, code) planes are those of the full call list and of
    the oracle, through the batched entry point on two shards at once and through phz_map_reads on a host-resident shard."""
    import ctypes as C
    from phaser_amd import _lib, soa, synth
    shards = []; wants = []; vps = []
    for seed, n_snps in ((501, 400), (502, 2500)):
        v, gs, ge, w = synth.make_variants("chr1", 1, 20_000_000, n_snps, seed, n_genes=max(4, n_snps // 25))
        rb = synth.make_reads(v, gs, ge, w, 6000, seed + 100, n_rate=0.002)
        rb = rb.select(synth.samtools_keep(rb, 255))
        wants.append(oracle_map_readbatch(oracle_build, rb, v.pos.numpy(), 10, with_text=False))
        shards.append(soa.pack_readbatch(rb)); vps.append(v.pos)
    dev = [s.to("cuda") for s in shards]
    full = mapper.map_batch(dev, vps, 10)
    lean = mapper.map_batch(dev, vps, 10, aux=False)
    for f, l, (o_r, o_v, o_c, _) in zip(full, lean, wants):
        assert l.aux0 is None and l.aux1 is None and f.aux0 is not None
        assert l.n == f.n == len(o_r)
        for a, b, o in ((l.read_idx, f.read_idx, o_r), (l.var_idx, f.var_idx, o_v), (l.code, f.code, o_c)):
            assert torch.equal(a, b) and np.array_equal(a.cpu().numpy(), o)
    # host-space entry point with NULL planes
    sh = shards[0]; o_r, o_v, o_c, _ = wants[0]
    p = lambda t: C.c_void_p(t.data_ptr())
    vpos = vps[0].to(torch.int32).contiguous()
    r = _lib.phz_reads(sh.n, int(sh.cigar.numel()), int(sh.seq2.numel()), p(sh.pos), p(sh.cigar_off), p(sh.cigar), p(sh.seq_off), p(sh.seq2), p(sh.qual))
    vv = _lib.phz_variants(int(vpos.numel()), p(vpos), None)
    cap = len(o_r) + 8
    bufs = [torch.empty(cap, dtype=torch.int32), torch.empty(cap, dtype=torch.int32), torch.empty(cap, dtype=torch.uint8)]
    c = _lib.phz_calls(cap, p(bufs[0]), p(bufs[1]), p(bufs[2]), None, None)
    n = C.c_int64(0)
    mapper.ctx.check(mapper.ctx.lib.phz_map_reads(mapper.ctx.h, C.byref(r), C.byref(vv), 10, C.byref(c), C.byref(n), _lib.PHZ_HOST))
    assert n.value == len(o_r)
    assert np.array_equal(bufs[0][:n.value].numpy(), o_r) and np.array_equal(bufs[1][:n.value].numpy(), o_v) and np.array_equal(bufs[2][:n.value].numpy(), o_c)


@pytest.mark.parametrize("slot_cap", [8, 1])
def test_dense_tiles_take_the_overflow_area(oracle_build, monkeypatch, slot_cap):
    """K_map stages a tile's calls in a slot sized for the TYPICAL tile; a denser tile takes a stretch of the overflow area behind the slots (round 4 sized every
    slot of a submission for its densest tile).  With slots of 8 calls (and of ONE call) nearly every tile of a dense-variant shard overflows: the first
    submission finds the area too small, is redone with exactly what it needs, the following ones reuse it -- call lists identical to the C oracle every time,
    with and without the text planes, one shard and a batch of three."""
    from phaser_amd import soa, synth
    from phaser_amd.mapper import Mapper
    monkeypatch.setenv("PHZ_MAP_SLOT_CAP", str(slot_cap))
    m = Mapper(0)                                           # its own context: the slot size is fixed at a context's first submission
    shards = []; want = []; vps = []
    for seed, n_snps, pairs in ((41, 3000, 40000), (42, 200, 3000), (43, 6000, 60000)):
        v, gs, ge, w = synth.make_variants("chr1", 1, 2_000_000, n_snps, seed, n_genes=6)
        rb = synth.make_reads(v, gs, ge, w, pairs, seed + 1, n_rate=0.002)
        rb = rb.select(synth.samtools_keep(rb, 255))
        want.append(oracle_map_readbatch(oracle_build, rb, v.pos.numpy(), 10))
        shards.append(soa.pack_readbatch(rb).to("cuda")); vps.append(v.pos)
    def same(calls, o):
        c = calls.cpu()
        return c.n == len(o[0]) and np.array_equal(c.read_idx.numpy(), o[0]) and np.array_equal(c.var_idx.numpy(), o[1]) and np.array_equal(c.code.numpy(), o[2])
    for rep in range(2):
        assert same(m.map(shards[0], vps[0], 10), want[0]), rep             # the densest single shard: area too small the first time
    for aux in (True, False):
        got = m.map_batch(shards, vps, 10, aux=aux)
        assert all(same(g, o) for g, o in zip(got, want)), aux
    assert want[0][0].size > 50000 and max(np.bincount(want[0][0] // 256)) > 8 * 8          # tiles with far more calls than a slot


def test_resident_variant_table(mapper, oracle_build):
    """phz_load_variants: the het-variant table of a chromosome uploaded once into a slot of the ctx, K_map run from the resident pointers
    over two BAMs' shards, a second chromosome in another slot, the first slot reloaded with a longer table -- every call list = the C oracle's."""
    from phaser_amd import _lib, soa, synth
    tabs = {}
    for slot, (chrom, n_snps, seed) in enumerate([("chr1", 3000, 41), ("chr2", 800, 42)]):
        v, gs, ge, w = synth.make_variants(chrom, 1, 20_000_000, n_snps, seed, n_genes=60)
        tabs[slot] = (v, gs, ge, w, mapper.load_variants(slot, v.pos.numpy()))
    for slot, (v, gs, ge, w, res) in tabs.items():
        assert res.n == len(v.pos) and res.pos
        for bam in range(2):
            rb = synth.make_reads(v, gs, ge, w, 40_000, 500 + 10 * slot + bam)
            rb = rb.select(synth.samtools_keep(rb, 255))
            o_r, o_v, o_c, o_t = oracle_map_readbatch(oracle_build, rb, v.pos.numpy(), 10)
            calls = mapper.map(soa.pack_readbatch(rb).to("cuda"), None, 10, resident=res).cpu()
            assert calls.n == len(o_r) and calls.n > 1000
            assert np.array_equal(calls.read_idx.numpy(), o_r) and np.array_equal(calls.var_idx.numpy(), o_v) and np.array_equal(calls.code.numpy(), o_c)
    v, gs, ge, w = synth.make_variants("chr1", 1, 20_000_000, 9000, 43, n_genes=60)          # slot 0 again, three times the size: the slot's buffers grow
    res = mapper.load_variants(0, v.pos.numpy())
    rb = synth.make_reads(v, gs, ge, w, 40_000, 540)
    rb = rb.select(synth.samtools_keep(rb, 255))
    o_r, o_v, o_c, o_t = oracle_map_readbatch(oracle_build, rb, v.pos.numpy(), 10)
    calls = mapper.map(soa.pack_readbatch(rb).to("cuda"), None, 10, resident=res).cpu()
    assert np.array_equal(calls.read_idx.numpy(), o_r) and np.array_equal(calls.var_idx.numpy(), o_v) and np.array_equal(calls.code.numpy(), o_c)
    bad = np.array([5, 3], dtype=np.int32); one = np.ones(2, dtype=np.uint8); out = _lib.phz_variants()
    import ctypes as C
    assert mapper.ctx.lib.phz_load_variants(mapper.ctx.h, 1, C.c_void_p(bad.ctypes.data), C.c_void_p(one.ctypes.data), 2, _lib.PHZ_HOST, C.byref(out)) == _lib.PHZ_E_ARG
