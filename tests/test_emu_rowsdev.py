"""The DEVICE row stage (phaser_amd/csrc/phz_rowsdev.hip: pair-test bookkeeping, pruning, components, ordering, phase_v3, haplotype
read sets, the text of the five files) executed under the host-side HIP emulation of tests/hipemu -- kernel LOGIC on the CPU box -- from
the K_tally fixtures written on an MI355X (tests/golden/tally), against the reference's own output files.  The real kernels are
checked on the GPU by tests/test_gpu_pipeline.py; nothing here is a product path."""
import gzip
import json
import os
import pickle
import sys

import pytest

from conftest import GOLD, REPO, gz_text
from helpers import OUTPUTS, EmuContext, canonical, emu_library, option_case_kwargs, stub_emu_stages, stub_gpu_stages
from test_host_stages import _cases


@pytest.fixture(scope="module", params=[None, 0, "stat2"], ids=["rows_by_thread", "rows_by_wave", "blocks_beyond_tables"])
def emu(request):
    """None: the product's thresholds; 0: the variant whose row stage formats EVERY block row by a wave (the fixtures' blocks are small:
    with the product's threshold they would all take the one-thread-per-row path); "stat2": the variant whose gwStat table / LDS piece arrays
    stop at blocks of 2 variants, so that every longer block of the fixtures is "a block of more than 512 variants" of the product (gwStat text
    formatted by the host inside the run, read-set pieces in the global pool: round-4 verdict, missing #4 -- that used to cost the whole pass)"""
    if request.param == "stat2":
        return emu_library(row_wave_min=0, stat_n=2)
    return emu_library(row_wave_min=request.param)


def run_stages(emu, case, load, cfg, vcf_text, bam_names, device_rows=True, **extra):
    from phaser_amd import vcf
    from phaser_amd.engine import Config, Engine
    load = dict(load); cfg = dict(cfg)
    inc = load.pop("include_indels", 0); cfg.pop("include_indels", None)
    vs = vcf.load_variants(vcf_text, include_indels=inc, **load)
    saved = pickle.load(gzip.open(os.path.join(GOLD, "tally", case + ".pkl.gz"), "rb"))

    class _M:
        ctx = EmuContext(emu)
        device = None
    eng = Engine(vs, bam_names, Config(include_indels=inc, device_rows=device_rows, **cfg, **extra), mapper=_M())
    eng.n_qid.update(saved["n_qid"]); eng.qnames.update(saved["qnames"])
    if device_rows:
        stub_emu_stages(eng, saved)
    else:
        stub_gpu_stages(eng, saved)
    return eng.finish(), eng


def _inputs(case, gold, c1_inputs):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from phasing_oracle import bam_display_names          # naming helper only
    d = os.path.join(GOLD, gold)
    if case == "c1":
        vcf_text = c1_inputs["vcf"]; bams = ["c1.bam"]
    elif case.startswith("opts_"):
        vcf_text = open(os.path.join(GOLD, "pipe_opts", "in.vcf")).read(); bams = ["o1.bam", "o2.bam"]
    else:
        vcf_text = open(os.path.join(d, "in.vcf")).read()
        bams = {"pipe_one": ["a.bam"], "pipe_two": ["t1.bam", "t2.bam"], "pipe_sparse": ["s1.bam", "s2.bam", "s3.bam"], "pipe_indel": ["i.bam"]}.get(case, ["n.bam"])
    return d, vcf_text, bam_display_names(bams)


@pytest.mark.parametrize("case,gold,load,cfg", _cases(), ids=[c[0] for c in _cases()])
def test_device_rows_match_reference(emu, case, gold, load, cfg, c1_inputs):
    d, vcf_text, bams = _inputs(case, gold, c1_inputs)
    out, eng = run_stages(emu, case, load, cfg, vcf_text, bams)
    assert eng.rows_path == "device", getattr(eng, "rows_fallback", "")          # no option is declined (--output_read_ids 1 included, since round 6)
    for name in OUTPUTS:
        want = gz_text(os.path.join(d, "out.%s.txt.gz" % name))
        assert canonical(name, out[name]) == canonical(name, want), name
    # the two row stages agree byte for byte, row order included (and the order of the QNAME lists of --output_read_ids 1: first appearance)
    host, heng = run_stages(emu, case, load, cfg, vcf_text, bams, device_rows=False)
    assert heng.rows_path == "host"
    for name in OUTPUTS:
        assert out[name] == host[name], name
    assert eng.phased == heng.phased and eng.log == heng.log


def test_rows_of_a_block_of_150_variants_by_wave_and_by_thread():
    """A clean component of 150 variants stays ONE block whatever --max_block_size says (phaser.py:2116-2136): its rows of haplotypes.txt /
    haplotypic_counts.txt are joins of 150 elements, i.e. three rounds of 64 lanes in the wave sinks.  Built from synthetic call lines through
    the emulated K_tally; the wave-formatted rows must equal the thread-formatted ones and the host row stage's, byte for byte."""
    import numpy as np
    from phaser_amd import vcf
    from phaser_amd.engine import Config, Engine
    from test_emu_tally import run_tally
    rng = np.random.default_rng(3)
    nv = 150
    truth = rng.integers(0, 2, size=nv)
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
    vcf_text = head + "".join("chrS\t%d\trs%d\tA\tG\t.\tPASS\t.\tGT\t%s\n" % (1000 + 40 * i, i, "0|1" if i % 3 else "1|0") for i in range(nv))
    var = []; qid = []; cls = []
    q = 0
    for i in range(nv - 1):
        for _ in range(4):
            hap = int(rng.integers(0, 2))
            for v in (i, i + 1):
                var.append(v); qid.append(q); cls.append(int(truth[v]) ^ hap)
            q += 1
    n = len(var)
    R = {"nv": nv, "line_var": np.asarray(var, np.int32), "line_qid": np.asarray(qid, np.int32), "line_cls": np.asarray(cls, np.uint8),
         "line_bam": np.zeros(n, np.int32), "bam_offsets": [(0, 0, n)]}
    saved = {"tally": {"chrS": R}, "n_qid": {"chrS": q}, "qnames": {"chrS": ["q%d" % i for i in range(q)]}}
    got, sz = run_tally(EmuContext(emu_library()), saved, ["chrS"], 1)
    R.update({"var_count": got["var_count"], "var_first": got["var_first"], "var_distinct": got["var_distinct"], "var_rank": got["var_rank"], "ea": got["ea"], "eb": got["eb"],
              "cells": got["cells"], "linked": got["linked"]})
    vs = vcf.load_variants(vcf_text)
    outs = []
    for wave_min, device_rows in ((0, True), (100000, True), (None, False)):
        lib = emu_library(row_wave_min=wave_min)

        class _M:
            ctx = EmuContext(lib)
            device = None
        eng = Engine(vs, ["s"], Config(device_rows=device_rows), mapper=_M())
        eng.n_qid.update(saved["n_qid"]); eng.qnames.update(saved["qnames"])
        (stub_emu_stages if device_rows else stub_gpu_stages)(eng, saved)
        outs.append(eng.finish())
        assert eng.rows_path == ("device" if device_rows else "host")
    hap_rows = outs[0]["haplotypes"].split("\n")
    assert len(hap_rows) == 3 and hap_rows[1].split("\t")[4] == "150"          # header, ONE block of 150 variants, trailing newline
    for name in OUTPUTS:
        assert outs[0][name] == outs[1][name] == outs[2][name], name


@pytest.mark.parametrize("env", [{"PHZ_ROWS_SORT64": "1"}, {"PHZ_ROWS_FAKE_LINE_BITS": "32"}, {"PHZ_ROWS_FAKE_LINE_BITS": "31"}, {"PHZ_ROWS_NO_PRESTAGE": "1"}, {"PHZ_SORT_ONE_LAUNCH": "1"},
                                 {"PHZ_ROWS_NO_PREKEYS": "1"}],
                         ids=["sort64", "line_bits_32", "line_bits_31", "no_prestage", "one_launch_sort", "keys_in_the_second_stage"])
def test_key_layouts_of_the_ordering_sorts(env, monkeypatch, c1_inputs):
    """The rank order of the variants and the first-appearance order of the covered variants sort 32-bit keys when (line, gap) / (BAM, line) fit and 64-bit
    keys otherwise.  A BAM with more than 2^31 call lines has 32 line bits: a 32-bit (BAM, line) key would shift a 32-bit word by 32 (round-4 advisor
    finding), so that size must take the 64-bit keys; forced here through PHZ_ROWS_FAKE_LINE_BITS on the two-BAM fixture.  Also: the p-value-independent sorts
    enqueued by the first stage (default) or by the second (PHZ_ROWS_NO_PRESTAGE), the first-appearance keys prepared by the first stage (default, the handle knows the
    tally's shards) or by the second (PHZ_ROWS_NO_PREKEYS), and the one-launch sort passes.  Every variant gives the same bytes."""
    lib = emu_library()
    case, gold, load, cfg = next(c for c in _cases() if c[0] == "pipe_two")
    d, vcf_text, bams = _inputs(case, gold, c1_inputs)
    base, _ = run_stages(lib, case, load, cfg, vcf_text, bams)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    out, eng = run_stages(lib, case, load, cfg, vcf_text, bams)
    assert eng.rows_path == "device"
    for name in OUTPUTS:
        assert out[name] == base[name], name


def test_pair_key_table_grows_instead_of_declining_the_pass(monkeypatch, c1_inputs):
    """More distinct (supporting, total) read-count pairs than the pair-key table holds used to send the WHOLE pass to the host stage (round-4 verdict,
    missing #4).  Now phz_rowsdev_pair_keys reports PHZ_E_CAPACITY, the host quadruples the table and redoes the stage: started from a 16-slot table the
    noisy fixture (dozens of distinct pairs) must still come out byte for byte, on the device."""
    lib = emu_library()
    case, gold, load, cfg = next(c for c in _cases() if c[0] == "pipe_noisy_a")
    d, vcf_text, bams = _inputs(case, gold, c1_inputs)
    base, _ = run_stages(lib, case, load, cfg, vcf_text, bams)
    monkeypatch.setenv("PHZ_ROWS_PAIR_SLOTS", "16")
    out, eng = run_stages(lib, case, load, cfg, vcf_text, bams)
    assert eng.rows_path == "device" and eng.stats.get("rowsdev_n_pair_table_growths", 0) >= 1
    for name in OUTPUTS:
        assert out[name] == base[name], name


@pytest.mark.parametrize("case", ["pipe_two", "opts_read_ids", "pipe_sparse"])
def test_hap_counts_are_the_distinct_reads_per_list(case, c1_inputs):
    """phz_hap_counts: distinct QNAMEs per (variant, allele, BAM) read list of the resident tally = len(set(haplo_reads[allele][bam]))
    (phaser.py:1196-1204), computed by the read-set kernels of the row stage; against numpy on the fixture's read lists."""
    import ctypes
    import numpy as np
    from phaser_amd import _lib
    lib = emu_library()
    c, gold, load, cfg = next(x for x in _cases() if x[0] == case)
    d, vcf_text, bams = _inputs(case, gold, c1_inputs)
    from phaser_amd import vcf
    from phaser_amd.engine import Config, Engine
    load = dict(load); cfg = dict(cfg)
    inc = load.pop("include_indels", 0); cfg.pop("include_indels", None)
    vs = vcf.load_variants(vcf_text, include_indels=inc, **load)
    saved = pickle.load(gzip.open(os.path.join(GOLD, "tally", case + ".pkl.gz"), "rb"))

    class _M:
        ctx = EmuContext(lib)
        device = None
    eng = Engine(vs, bams, Config(include_indels=inc, **cfg), mapper=_M())
    eng.n_qid.update(saved["n_qid"]); eng.qnames.update(saved["qnames"])
    stub_emu_stages(eng, saved)
    G = eng._tally_genome()
    nseg = G["nv"] * 2 * G["nb"]
    got = np.full(nseg, -1, dtype=np.int32)
    eng.ctx.check(lib.phz_hap_counts(eng.ctx.h, ctypes.c_void_p(got.ctypes.data), nseg, _lib.PHZ_HOST))
    rs = G["rl_start"].astype(np.int64); rq = G["rl_qid"]
    want = np.array([len(set(rq[rs[e]:rs[e + 1]].tolist())) for e in range(nseg)], dtype=np.int32)
    assert np.array_equal(got, want) and int(want.sum()) > 0
    assert lib.phz_hap_counts(eng.ctx.h, ctypes.c_void_p(got.ctypes.data), nseg + 1, _lib.PHZ_HOST) == _lib.PHZ_E_ARG


@pytest.mark.parametrize("shards", ["right", "wrong"])
def test_first_stage_issued_with_the_tally_and_keys_for_other_shards(shards, c1_inputs):
    """The fused order of the product (phz_tally_pairs: stage 1 of the row stage issued by the same native call as the tally; Engine hands its result to rowsdev.run as
    G["pair_stage"]) replayed under the emulation on the three-BAM fixture -- with the shard table the first stage was told equal to the run's ("right": the first-appearance
    keys prepared by stage 1 are used) and different from it ("wrong": phz_rowsdev_run must notice and prepare them again).  Same bytes as the plain order either way."""
    import numpy as np
    from phaser_amd import _lib, rowsdev
    lib = emu_library()
    case, gold, load, cfg = next(c for c in _cases() if c[0] == "pipe_sparse")
    d, vcf_text, bams = _inputs(case, gold, c1_inputs)
    base, _ = run_stages(lib, case, load, cfg, vcf_text, bams)
    from phaser_amd import vcf
    from phaser_amd.engine import Config, Engine
    vs = vcf.load_variants(vcf_text, **{k: v for k, v in load.items() if k != "include_indels"})
    saved = pickle.load(gzip.open(os.path.join(GOLD, "tally", case + ".pkl.gz"), "rb"))

    class _M:
        ctx = EmuContext(lib)
        device = None
    eng = Engine(vs, bams, Config(**{k: v for k, v in cfg.items() if k != "include_indels"}), mapper=_M())
    eng.n_qid.update(saved["n_qid"]); eng.qnames.update(saved["qnames"])
    stub_emu_stages(eng, saved)
    plain = eng._tally_genome

    def fused():
        G = plain()
        eng.G = G
        T, keys, n_slots = rowsdev.pair_stage_inputs(eng)
        sh = sorted(((b0, b0 + n, b) for (c, b), (b0, n) in G["line_base"].items()))
        lo = np.array([x[0] for x in sh], np.int64); hi = np.array([x[1] for x in sh], np.int64); sb = np.array([x[2] for x in sh], np.int32)
        if shards == "wrong":
            sb = (sb + 1) % max(1, G["nb"]); hi = hi.copy(); hi[-1] += 1
        vp = lambda a: C.c_void_p(a.ctypes.data)
        eng.ctx.check(lib.phz_rowsdev_set_shards(T.h, len(sh), vp(lo), vp(hi), vp(sb)))
        st = lib.phz_rowsdev_pair_keys(eng.ctx.h, T.h, vp(keys))
        G["pair_stage"] = (keys, n_slots, int(st), T)
        return G
    import ctypes as C
    eng._tally_genome = fused
    out = eng.finish()
    assert eng.rows_path == "device" and "pair_stage" not in eng.G
    for name in OUTPUTS:
        assert out[name] == base[name], name


def test_copy_as_written_gives_the_same_text(c1_inputs, monkeypatch):
    """Second and later passes over a variant set hand phz_rowsdev_run a host region sized by the previous pass: the run copies every finished text there itself (on a
    second stream beside its last writers; largest file first) and reports the offsets.  Same bytes as the first pass (texts fetched afterwards), also when the region is
    too small for the pass (the guess of a smaller previous pass: every text is fetched as before) and with the path switched off."""
    from phaser_amd import rowsdev, vcf
    from phaser_amd.engine import Config, Engine
    lib = emu_library()
    case, gold, load, cfg = next(c for c in _cases() if c[0] == "pipe_two")
    d, vcf_text, bams = _inputs(case, gold, c1_inputs)
    vs = vcf.load_variants(vcf_text)
    saved = pickle.load(gzip.open(os.path.join(GOLD, "tally", case + ".pkl.gz"), "rb"))

    def one_pass():
        class _M:
            ctx = EmuContext(lib)
            device = None
        eng = Engine(vs, bams, Config(), mapper=_M())
        eng.n_qid.update(saved["n_qid"]); eng.qnames.update(saved["qnames"])
        stub_emu_stages(eng, saved)
        out = eng.finish()
        return out, eng
    first, e1 = one_pass()
    T = rowsdev.tables_for(e1)
    assert T.__dict__.get("_text_total", 0) > 0
    seen = []
    real = lib.phz_rowsdev_run

    def spy(ctx_h, th, o, *rest):
        seen.append(int(o._obj.host_text_cap))
        return real(ctx_h, th, o, *rest)
    monkeypatch.setattr(lib, "phz_rowsdev_run", spy, raising=False)
    second, _ = one_pass()
    assert seen[-1] > 0
    T.__dict__["_text_total"] = 4096          # a region far too small: the run declines it, the texts are fetched
    third, _ = one_pass()
    monkeypatch.setenv("PHZ_ROWS_COPY_AS_WRITTEN", "0")
    fourth, _ = one_pass()
    assert seen[-1] == 0
    for name in OUTPUTS:
        assert first[name] == second[name] == third[name] == fourth[name], name
        assert canonical(name, first[name]) == canonical(name, gz_text(os.path.join(d, "out.%s.txt.gz" % name))), name
