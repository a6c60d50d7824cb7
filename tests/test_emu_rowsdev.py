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


@pytest.fixture(scope="module")
def emu():
    return emu_library()


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
        bams = {"pipe_one": ["a.bam"], "pipe_two": ["t1.bam", "t2.bam"], "pipe_indel": ["i.bam"]}.get(case, ["n.bam"])
    return d, vcf_text, bam_display_names(bams)


@pytest.mark.parametrize("case,gold,load,cfg", _cases(), ids=[c[0] for c in _cases()])
def test_device_rows_match_reference(emu, case, gold, load, cfg, c1_inputs):
    d, vcf_text, bams = _inputs(case, gold, c1_inputs)
    out, eng = run_stages(emu, case, load, cfg, vcf_text, bams)
    declined = cfg.get("gw_phase_method", 0) == 1 or cfg.get("output_read_ids", 0) == 1
    assert eng.rows_path == ("host" if declined else "device"), getattr(eng, "rows_fallback", "")
    for name in OUTPUTS:
        want = gz_text(os.path.join(d, "out.%s.txt.gz" % name))
        assert canonical(name, out[name]) == canonical(name, want), name
    if not declined:
        # the two row stages agree byte for byte, row order included
        host, heng = run_stages(emu, case, load, cfg, vcf_text, bams, device_rows=False)
        assert heng.rows_path == "host"
        for name in OUTPUTS:
            assert out[name] == host[name], name
        assert eng.phased == heng.phased and eng.log == heng.log
