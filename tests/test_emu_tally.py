"""K_tally (phaser_amd/csrc/phz_tally.hip: per-line pass, QNAME groups, variant-pair cells, read lists) executed under the host-side HIP
emulation of tests/hipemu on the call lines of the fixtures written on an MI355X (tests/golden/tally): every result array must equal what the
GPU produced (those results are pinned against the reference through the five files), and the chain K_tally -> device row stage must give
the reference's files.  Kernel LOGIC on the CPU box; the real kernels are checked by the -m gpu tests."""
import ctypes as C
import gzip
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import GOLD, REPO, gz_text
from helpers import OUTPUTS, EmuContext, canonical, emu_library, genome_from_saved
from test_host_stages import _cases


def lines_from_saved(saved, chrom_list, n_bams):
    """phz_lines arrays of every (chromosome, BAM) shard of a fixture: one synthetic record per call line (read_idx = line), the class of a
    line expressed through the general mapper's codes (5 / 6 = allele 0 / 1, 4 = other), dropped lines as records without an AS value."""
    shards = []; keep = []
    vb = qb = 0
    for c in chrom_list:
        R = saved["tally"][c]
        nq = max(1, saved["n_qid"].get(c, int(R["line_qid"].max()) + 1 if len(R["line_qid"]) else 1))
        for b, base, n in R["bam_offsets"]:
            sl = slice(base, base + n)
            cls = R["line_cls"][sl]
            code = np.where(cls == 0, 5, np.where(cls == 1, 6, 4)).astype(np.uint8)
            arr = {"read_idx": np.arange(n, dtype=np.int32), "var_idx": np.ascontiguousarray(R["line_var"][sl], dtype=np.int32), "code": code,
                   "read_qid": np.ascontiguousarray(R["line_qid"][sl], dtype=np.int32), "read_as": np.zeros(max(1, n), dtype=np.int32),
                   "has_as": (cls != 255).astype(np.uint8)}
            keep.append(arr)
            shards.append((arr, n, b, vb, qb))
        vb += R["nv"]; qb += nq
    return shards, keep, vb, qb


def run_tally(ctx, saved, chrom_list, n_bams, as16=False):
    """as16: hand the AS column over as the 2-byte plane (phz_lines.read_as16, the form the Engine uses) instead of read_as + read_has_as"""
    from phaser_amd import _lib
    shards, keep, NV, NQ = lines_from_saved(saved, chrom_list, n_bams)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    if as16:
        for a, n, b, v0, q0 in shards:
            a["as16"] = np.where(a["has_as"] != 0, a["read_as"][:len(a["has_as"])].astype(np.int16), np.int16(-32768)).astype(np.int16)
            if len(a["as16"]) == 0:
                a["as16"] = np.zeros(1, np.int16)
        arr = (_lib.phz_lines * max(1, len(shards)))(*[_lib.phz_lines(n, vp(a["read_idx"]), vp(a["var_idx"]), vp(a["code"]), max(1, n), vp(a["read_qid"]), None, None,
                                                                      0.0, 1, b, v0, q0, vp(a["as16"])) for a, n, b, v0, q0 in shards])
    else:
        arr = (_lib.phz_lines * max(1, len(shards)))(*[_lib.phz_lines(n, vp(a["read_idx"]), vp(a["var_idx"]), vp(a["code"]), max(1, n), vp(a["read_qid"]), vp(a["read_as"]),
                                                                      vp(a["has_as"]), 0.0, 1, b, v0, q0) for a, n, b, v0, q0 in shards])
    a0 = np.full(max(1, NV), 255, dtype=np.uint8)
    sz = _lib.phz_tally_sizes()
    ctx.check(ctx.lib.phz_tally(ctx.h, arr, len(shards), NV, vp(a0), vp(a0), NQ, n_bams, C.byref(sz), _lib.PHZ_HOST))
    ne = int(sz.n_edges); nrl = int(sz.n_read_list)
    out = {"var_count": np.zeros(NV * 3, np.int32), "var_first": np.zeros(NV, np.int64), "var_distinct": np.zeros(NV * 3, np.int32), "var_rank": np.zeros(NV, np.uint64),
           "line_cls": np.zeros(max(1, int(sz.n_lines)), np.uint8), "ea": np.zeros(ne, np.int32), "eb": np.zeros(ne, np.int32), "cells": np.zeros(ne * 9, np.int32),
           "linked": np.zeros(ne, np.uint8), "cto": np.zeros(ne * 3, np.int32), "rl_start": np.zeros(NV * 2 * n_bams + 1, np.uint32), "rl_qid": np.zeros(nrl, np.int32),
           "stats": np.zeros(ne * 5, np.int32)}
    p = lambda k: vp(out[k]) if out[k].size else None
    o = _lib.phz_tally_out(p("var_count"), p("var_first"), p("var_distinct"), p("var_rank"), p("line_cls"), p("ea"), p("eb"), p("cells"), p("linked"), p("cto"),
                           p("rl_start"), p("rl_qid"), p("stats"))
    ctx.check(ctx.lib.phz_tally_fetch(ctx.h, C.byref(o), _lib.PHZ_HOST))
    return out, sz


@pytest.mark.parametrize("tile", [0, 256])      # 256: tiles of 256 lines, so that the fixtures' QNAMEs straddle tiles (spill path of k_tile)
@pytest.mark.parametrize("case", sorted(set(c[0] for c in _cases())))
def test_tally_kernels_reproduce_the_gpu_fixture(case, tile):
    ctx = EmuContext(emu_library(tile))
    saved = pickle.load(gzip.open(os.path.join(GOLD, "tally", case + ".pkl.gz"), "rb"))
    chroms = list(saved["tally"])
    nb = 1 + max(b for c in chroms for b, _, _ in saved["tally"][c]["bam_offsets"])
    want = genome_from_saved(saved, chroms, nb)
    for rep in range(2):                       # twice on the same ctx: the persistent per-QNAME counters and the pair table must come back clean
        got, sz = run_tally(ctx, saved, chroms, nb)
        NV = want["nv"]
        assert np.array_equal(got["var_count"].reshape(NV, 3), want["var_count"])
        assert np.array_equal(got["var_first"], want["var_first"])
        assert np.array_equal(got["var_distinct"].reshape(NV, 3), want["var_distinct"])
        assert np.array_equal(got["var_rank"], want["var_rank"])
        assert np.array_equal(got["ea"], want["ea"]) and np.array_equal(got["eb"], want["eb"])
        assert np.array_equal(got["linked"], want["linked"])
        assert np.array_equal(got["cto"].reshape(-1, 3), want["cto"])
        assert np.array_equal(got["stats"].reshape(5, -1), want["stats"])
        assert np.array_equal(got["rl_start"], want["rl_start"]) and np.array_equal(got["rl_qid"], want["rl_qid"])
        assert (int(sz.noise_match), int(sz.noise_mismatch)) == want["noise"] and int(sz.n_kept) == want["n_kept"]


def test_tally_kernels_on_skewed_synthetic_lines():
    """Read lists of thousands of entries (workgroup bitonic sort, device radix sort), QNAMEs shared by two BAMs (owner = last BAM), dropped
    lines: against a direct restatement with Python sets / sorts on the same lines."""
    rng = np.random.default_rng(7)
    nv = 24; nq = 3000
    saved = {"tally": {}, "n_qid": {"chrS": nq}}
    lines = []
    for b in range(2):
        n = 30000 if b == 0 else 7000
        var = np.sort(rng.choice(nv, size=n, p=np.array([0.55, 0.2] + [0.25 / 22] * 22)))          # variant 0: thousands of lines
        qid = rng.integers(0, nq, size=n)
        cls = rng.choice([0, 1, 2, 255], size=n, p=[0.45, 0.4, 0.1, 0.05]).astype(np.uint8)
        lines.append((var.astype(np.int32), qid.astype(np.int32), cls, np.full(n, b, np.int32)))
    R = {"nv": nv, "line_var": np.concatenate([l[0] for l in lines]), "line_qid": np.concatenate([l[1] for l in lines]), "line_cls": np.concatenate([l[2] for l in lines]),
         "line_bam": np.concatenate([l[3] for l in lines]), "bam_offsets": [(0, 0, 30000), (1, 30000, 7000)]}
    _check_against_sets(R, nv, nq, 2)


def _check_against_sets(R, nv, nq, nb, tile=0, as16=False):
    saved = {"tally": {"chrS": R}, "n_qid": {"chrS": nq}}
    ctx = EmuContext(emu_library(tile))
    got, sz = run_tally(ctx, saved, ["chrS"], nb, as16=as16)
    var, qid, cls, bam = R["line_var"], R["line_qid"], R["line_cls"], R["line_bam"]
    kept = cls != 255
    # per-variant counters
    vc = np.zeros((nv, 3), np.int64)
    np.add.at(vc, (var[kept], cls[kept]), 1)
    assert np.array_equal(got["var_count"].reshape(nv, 3), vc)
    sets = [[set(), set(), set()] for _ in range(nv)]
    for v, q, c in zip(var[kept], qid[kept], cls[kept]):
        sets[v][c].add(int(q))
    assert np.array_equal(got["var_distinct"].reshape(nv, 3), np.array([[len(s) for s in row] for row in sets]))
    # read lists in line order per (variant, allele, BAM)
    rs = got["rl_start"]
    for v in range(nv):
        for k in range(2):
            for b in range(nb):
                e = (2 * v + k) * nb + b
                want = qid[kept & (var == v) & (cls == k) & (bam == b)]
                assert np.array_equal(got["rl_qid"][rs[e]:rs[e + 1]], want), (v, k, b)
    assert int(rs[-1]) == int((kept & (cls < 2)).sum())
    # nine cells of every variant pair = sizes of the set intersections (phaser.py:1602-1632)
    cells = got["cells"].reshape(-1, 9)
    key = got["ea"].astype(np.int64) * (1 << 32) + got["eb"].astype(np.int64)
    assert np.all(np.diff(key) > 0)                      # the pair list is in (a, b) order, every pair once
    seen = {}
    for i, (a, b2) in enumerate(zip(got["ea"], got["eb"])):
        seen[(int(a), int(b2))] = cells[i]
    for a in range(nv):
        for b2 in range(a + 1, nv):
            want = [len(sets[a][i] & sets[b2][j]) for i in range(3) for j in range(3)]
            if sum(want) == 0:
                assert (a, b2) not in seen
            else:
                assert list(seen[(a, b2)]) == want, (a, b2)


@pytest.mark.parametrize("tile", [0, 256])
def test_read_lists_with_far_lines_are_sorted_afterwards(tile):
    """A read spliced over hundreds of het SNPs puts call lines outside the variant window of the tile they arrive in (far lines): their read
    lists are filled through cursors and put into line order afterwards -- a short one in LDS, one of > 4,096 entries by the device radix
    sort -- while every other list is written in place, in order, with no sort.  The AS column travels as the 2-byte plane here."""
    rng = np.random.default_rng(23)
    nv = 900; nq = 5000
    var = [np.zeros(6000, np.int64)]                                   # variant 0: 6,000 lines in a row (six tiles with the same window) ...
    later = np.sort(rng.integers(500, 560, size=3000))                 # ... then lines of variants 500-559, the windows have moved on ...
    far0 = rng.choice(3000, size=12, replace=False); later[far0] = 0   # ... with twelve more lines of variant 0 among them (far: list (0, *) > 4,096 entries)
    far7 = rng.choice(3000, size=5, replace=False); later[far7] = 7    # and five of variant 7, which has no other line (a short dirty list)
    var.append(later)
    var.append(np.sort(rng.integers(560, 900, size=2500)))
    var = np.concatenate(var).astype(np.int32)
    n = len(var)
    qid = rng.integers(0, nq, size=n).astype(np.int32)
    cls = rng.choice([0, 1, 2, 255], size=n, p=[0.5, 0.4, 0.05, 0.05]).astype(np.uint8)
    R = {"nv": nv, "line_var": var, "line_qid": qid, "line_cls": cls, "line_bam": np.zeros(n, np.int32), "bam_offsets": [(0, 0, n)]}
    _check_against_sets(R, nv, nq, 1, tile, as16=True)


@pytest.mark.parametrize("tile", [0, 256])
def test_tally_kernels_on_long_groups(tile):
    """QNAMEs with more than 64 distinct (variant, class) items: groups that sit inside one tile (finished in place; their later items come
    from the 64-item look-ahead and, beyond it, from memory) and groups that straddle tiles (spill path)."""
    rng = np.random.default_rng(11)
    nv = 160; nq = 60
    var = []; qid = []
    for q in range(40):                                   # 40 QNAMEs x 110 lines on 110 different variants, contiguous: tile-local groups
        v = np.sort(rng.choice(nv, size=110, replace=False))
        var.append(v); qid.append(np.full(110, q))
    v2 = rng.integers(0, nv, size=6000); q2 = rng.integers(40, nq, size=6000)      # 20 QNAMEs x ~300 lines spread over everything
    var.append(v2); qid.append(q2)
    var = np.concatenate(var).astype(np.int32); qid = np.concatenate(qid).astype(np.int32)
    n = len(var)
    cls = rng.choice([0, 1, 2, 255], size=n, p=[0.45, 0.4, 0.1, 0.05]).astype(np.uint8)
    R = {"nv": nv, "line_var": var, "line_qid": qid, "line_cls": cls, "line_bam": np.zeros(n, np.int32), "bam_offsets": [(0, 0, n)]}
    _check_against_sets(R, nv, nq, 1, tile)


def test_scan_over_many_tiles():
    """The single-launch scan (look-back over the predecessors' status words) on a read-list table of 13 tiles -- more than one round of eight predecessors --,
    twice on one context (the status words of the first scan must read as stale in the second).  (Until round 6: 100 tiles, 4.5 minutes of fibers since the
    tally's column scan walks every variant block; the GPU suite runs the scan over thousands of tiles.)"""
    rng = np.random.default_rng(5)
    nv = 26000; nq = 500; n = 3000
    var = np.sort(rng.integers(0, nv, size=n)).astype(np.int32); qid = rng.integers(0, nq, size=n).astype(np.int32)
    cls = rng.choice([0, 1, 2, 255], size=n, p=[0.45, 0.4, 0.1, 0.05]).astype(np.uint8)
    R = {"nv": nv, "line_var": var, "line_qid": qid, "line_cls": cls, "line_bam": np.zeros(n, np.int32), "bam_offsets": [(0, 0, n)]}
    saved = {"tally": {"chrS": R}, "n_qid": {"chrS": nq}}
    ctx = EmuContext(emu_library())
    for rep in range(2):
        got, sz = run_tally(ctx, saved, ["chrS"], 1)
        cnt = np.zeros(nv * 2, np.int64)
        k = cls < 2
        np.add.at(cnt, var[k].astype(np.int64) * 2 + cls[k], 1)
        assert np.array_equal(got["rl_start"], np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32))
        for e in np.flatnonzero(cnt)[:200]:
            assert np.array_equal(got["rl_qid"][got["rl_start"][e]:got["rl_start"][e + 1]], qid[k & (var == e // 2) & (cls == e % 2)])


def test_tally_on_empty_and_fully_dropped_input():
    """No shard at all, a shard without lines, a shard whose lines are all dropped by the AS cutoff: empty results, and the context stays usable."""
    from phaser_amd import _lib
    ctx = EmuContext(emu_library())
    nv = 50; nq = 10
    empty = {"nv": nv, "line_var": np.zeros(0, np.int32), "line_qid": np.zeros(0, np.int32), "line_cls": np.zeros(0, np.uint8), "line_bam": np.zeros(0, np.int32),
             "bam_offsets": [(0, 0, 0)]}
    got, sz = run_tally(ctx, {"tally": {"chrS": empty}, "n_qid": {"chrS": nq}}, ["chrS"], 1)
    assert int(sz.n_lines) == 0 and int(sz.n_edges) == 0 and int(sz.n_kept) == 0 and not got["var_count"].any() and not got["rl_start"].any()
    n = 300
    rng = np.random.default_rng(2)
    dropped = {"nv": nv, "line_var": np.sort(rng.integers(0, nv, n)).astype(np.int32), "line_qid": rng.integers(0, nq, n).astype(np.int32),
               "line_cls": np.full(n, 255, np.uint8), "line_bam": np.zeros(n, np.int32), "bam_offsets": [(0, 0, n)]}
    got, sz = run_tally(ctx, {"tally": {"chrS": dropped}, "n_qid": {"chrS": nq}}, ["chrS"], 1)
    assert int(sz.n_lines) == n and int(sz.n_kept) == 0 and int(sz.n_edges) == 0 and not got["var_count"].any() and int(got["rl_start"][-1]) == 0
    assert np.all(got["var_first"] == -1) and np.all(got["line_cls"][:n] == 255)
    # ... and a normal call afterwards
    live = dict(dropped); live["line_cls"] = rng.choice([0, 1, 2], size=n).astype(np.uint8)
    got, sz = run_tally(ctx, {"tally": {"chrS": live}, "n_qid": {"chrS": nq}}, ["chrS"], 1)
    assert int(sz.n_kept) == n and int(got["var_count"].sum()) == n


def test_as_cutoff_is_numpy_percentile():
    """phz_as_cutoff (AS histogram on the device, occupied bins back, numpy.percentile's linear formula in the library -- phaser.py:545-553 in one
    call) against numpy.percentile itself on the scores of the call lines, bit for bit: narrow and wide score bands, ties at the quantile, a
    single line, records without an AS tag, every quantile the option accepts in practice."""
    from phaser_amd import _lib
    ctx = EmuContext(emu_library())
    rng = np.random.default_rng(41)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    for trial in range(9):
        n_reads = int(rng.integers(1, 800)); n_lines = int(rng.integers(1, 900))
        spread = int(rng.choice([1, 3, 40, 1500]))
        aln = rng.integers(-spread, spread + 1, size=n_reads).astype(np.int32) - int(rng.integers(0, 200))
        has = (rng.random(n_reads) > (0.0 if trial % 3 else 0.3)).astype(np.uint8)
        read_idx = np.sort(rng.integers(0, n_reads, size=n_lines)).astype(np.int32)
        as16 = np.where(has != 0, aln, -32768).astype(np.int16)
        z32 = np.zeros(n_lines, np.int32); z8 = np.zeros(n_lines, np.uint8); qid = np.zeros(n_reads, np.int32)
        for use16 in (False, True):
            ln = _lib.phz_lines(n_lines, vp(read_idx), vp(z32), vp(z8), n_reads, vp(qid), None if use16 else vp(aln), None if use16 else vp(has), 0.0, 0, 0, 0, 0,
                                vp(as16) if use16 else None)
            arr = (_lib.phz_lines * 1)(ln)
            scores = aln[read_idx][has[read_idx] != 0].astype(np.int64)
            for q in ((5.0, 0.0, 100.0, 0.05 * 100) if use16 else (5.0, 50.0, 99.9, 12.5)):
                val = C.c_double(-1.0); found = C.c_int32(-1)
                ctx.check(ctx.lib.phz_as_cutoff(ctx.h, arr, 1, float(q), C.byref(val), C.byref(found)))
                if len(scores) == 0:
                    assert found.value == 0
                else:
                    assert found.value == 1 and val.value == float(np.percentile(scores, q)), (trial, q, val.value, float(np.percentile(scores, q)))
                # the same on the device, no host wait (phz_as_cutoff_enqueue: one workgroup scans the 64 Ki bins and interpolates): identical bits
                if not use16:
                    continue          # (1,024 fibers per call: the enqueue form is checked on the 2-byte plane, 36 of the 72 cases)
                blk = np.full(4, -7.0, dtype=np.float64)
                ctx.check(ctx.lib.phz_as_cutoff_enqueue(ctx.h, arr, 1, float(q), vp(blk)))
                ctx.check(ctx.lib.phz_ctx_sync(ctx.h))
                assert blk[2] == 0.0 and blk[3] == float(len(scores))
                if len(scores) == 0:
                    assert blk[1] == 0.0
                else:
                    assert blk[1] == 1.0 and blk[0] == float(np.percentile(scores, q)), (trial, q, blk[0], float(np.percentile(scores, q)))
