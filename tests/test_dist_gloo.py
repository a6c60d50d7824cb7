"""Multi-rank path on CPU: 2 gloo processes own one chromosome each (LPT assignment), all-reduce the per-BAM
AS histograms and the noise counters, run the host stages (native block phasing + row writer) for their own
chromosome, gather the per-chromosome fragments to rank 0 and assemble the files.
GPU stage results come from fixtures written on an MI355X: tests/golden/tally/pipe_two.pkl.gz (K_tally arrays,
component labels; tools/make_tally_fixture.py) and tests/golden/frags_pipe_two.json.gz (per-chromosome AS
histograms; tools/make_frag_fixture.py).  The assembled files must equal what the reference wrote for the same inputs."""
import gzip
import json
import os
import pickle
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLD, REPO, gz_text
from helpers import OUTPUTS, canonical


def _worker(rank, world, port, result_path, private_spool=False, case="pipe_two", bam_names=("t1", "t2")):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phaser_amd import dist as pdist
    from phaser_amd import vcf
    from phaser_amd.engine import Config, Engine
    fx = json.load(gzip.open(os.path.join(GOLD, "frags_pipe_two.json.gz"), "rt")) if case == "pipe_two" else None
    saved = pickle.load(gzip.open(os.path.join(GOLD, "tally", case + ".pkl.gz"), "rb"))
    vs = vcf.load_variants(open(os.path.join(GOLD, case, "in.vcf")).read())
    chroms = list(vs.chroms)
    weights = {c: float(len(saved["tally"][c]["line_cls"]) + i) for i, c in enumerate(chroms)}
    owner = pdist.assign_chromosomes(weights, world)
    assert sorted(set(owner.values())) == list(range(min(world, len(chroms))))
    mine = [c for c in chroms if owner[c] == rank]
    cutoffs = []
    for bam in ((0, 1) if fx is not None else ()):
        h = torch.zeros(65536, dtype=torch.int64)
        for c in mine:
            for i, n in fx["hists"]["%d:%s" % (bam, c)]:
                h[i] += n
        pdist.allreduce_sum_(h)
        hh = h.numpy(); nz = np.nonzero(hh)[0]
        cutoffs.append(float(np.percentile(np.repeat(nz.astype(np.int64) - 32768, hh[nz]), 5.0)))

    class _M:                             # the host stages never touch the mapper or the GPU context
        class ctx:
            lib = None
        device = None
    eng = Engine(vs, list(bam_names), Config(), mapper=_M())
    eng.set_owned(mine)
    eng.n_qid.update(saved["n_qid"])
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import stub_gpu_stages
    stub_gpu_stages(eng, saved)
    if private_spool:                     # no directory shared by the ranks: rank 0 cannot see the other spool files, the bytes travel through the group
        eng.spool_dir = os.path.join(os.path.dirname(result_path), "spool%d" % rank); os.makedirs(eng.spool_dir, exist_ok=True)
    out = eng.finish()                    # all-reduce of the noise counters + gather of the fragments inside
    if rank == 0:
        json.dump({"out": out, "cutoffs": cutoffs, "noise": eng.noise, "log": eng.log, "phased": eng.phased}, open(result_path, "w"))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("private_spool", [False, True])
def test_two_rank_reduce_gather_merge(tmp_path, private_spool):
    port = 29500 + (os.getpid() % 2000) + (7 if private_spool else 0)
    res = str(tmp_path / "res.json")
    mp.spawn(_worker, args=(2, port, res, private_spool), nprocs=2, join=True)
    if private_spool:
        assert not os.listdir(str(tmp_path / "spool0")) and not os.listdir(str(tmp_path / "spool1")), "spool files left behind"
    r = json.load(open(res))
    d = os.path.join(GOLD, "pipe_two")
    assert r["phased"] == 229
    for name in OUTPUTS:
        want = gz_text(os.path.join(d, "out.%s.txt.gz" % name))
        assert canonical(name, r["out"][name]) == canonical(name, want), name
    log = gz_text(os.path.join(d, "out.log.txt.gz"))
    for c in r["cutoffs"]:
        assert ("using alignment score cutoff of %d" % c) in log
    for line in r["log"]:
        assert line in log, line


def test_three_ranks_block_order_when_the_first_bam_misses_chromosomes(tmp_path):
    """tests/golden/pipe_sparse on three ranks (one chromosome each): the place of a chromosome in the block files -- the first BAM with a kept line on it, VCF order
    inside a BAM (phaser.py:1299, :573-574) -- is known to the rank that owns it and travels to rank 0 in the fragment table; the merged files are the
    reference's, allele_config.txt in FILE order (chr11, chr3, chr19 although the VCF says chr3, chr11, chr19)."""
    port = 29500 + (os.getpid() % 2000) + 23
    res = str(tmp_path / "res.json")
    mp.spawn(_worker, args=(3, port, res, False, "pipe_sparse", ("s1", "s2", "s3")), nprocs=3, join=True)
    r = json.load(open(res))
    d = os.path.join(GOLD, "pipe_sparse")
    for name in OUTPUTS:
        assert canonical(name, r["out"][name]) == canonical(name, gz_text(os.path.join(d, "out.%s.txt.gz" % name))), name
    first = [l.split("\t")[0].split("_")[0] for l in r["out"]["allele_config"].split("\n")[1:] if l]
    assert [c for i, c in enumerate(first) if i == 0 or first[i - 1] != c] == ["chr11", "chr3", "chr19"]


def test_a_rank_without_chromosomes(tmp_path):
    """Three ranks, two chromosomes: the third rank owns nothing, still takes part in every collective (AS histograms, noise counters,
    the gather) and the assembled files are the reference's."""
    port = 29500 + (os.getpid() % 2000) + 13
    res = str(tmp_path / "res.json")
    mp.spawn(_worker, args=(3, port, res, False), nprocs=3, join=True)
    r = json.load(open(res))
    d = os.path.join(GOLD, "pipe_two")
    assert r["phased"] == 229
    for name in OUTPUTS:
        assert canonical(name, r["out"][name]) == canonical(name, gz_text(os.path.join(d, "out.%s.txt.gz" % name))), name


def _worker_forced(rank, world, port, result_path):
    os.environ["PHZ_DIST_FORCE_COLLECTIVES"] = "1"
    _worker(rank, world, port, result_path)


def test_one_rank_forced_through_the_collectives(tmp_path):
    """PHZ_DIST_FORCE_COLLECTIVES=1: ONE rank takes the multi-rank path (all-reduces, broadcasts, all-gathers, spool file + byte-range splice,
    barrier) -- the switch behind the GPU test that runs the same path over backend nccl on a one-GPU box; without it a single rank skips
    every collective."""
    port = 29500 + (os.getpid() % 2000) + 31
    res = str(tmp_path / "res.json")
    mp.spawn(_worker_forced, args=(1, port, res), nprocs=1, join=True)
    r = json.load(open(res))
    d = os.path.join(GOLD, "pipe_two")
    assert r["phased"] == 229
    for name in OUTPUTS:
        assert canonical(name, r["out"][name]) == canonical(name, gz_text(os.path.join(d, "out.%s.txt.gz" % name))), name
    from phaser_amd import dist as pdist
    assert not pdist.collectives_live()                     # (no process group in this process)


def test_write_files_cuts_a_failed_write_to_what_was_written(tmp_path):
    """Round-5 advisor: outputs are overwritten in place (no O_TRUNC); a write that fails half way must not leave the new head followed by the old tail."""
    from phaser_amd import dist as pdist
    p = str(tmp_path / "f.txt")
    open(p, "wb").write(b"OLD" * 1000)
    with pytest.raises(IOError):
        pdist.write_files([(p, [b"new-head", pdist.FileSpan(str(tmp_path / "missing.bin"), 0, 10)])])
    assert open(p, "rb").read() == b"new-head"


def test_lpt_assignment_balanced_and_deterministic():
    from phaser_amd import dist as pdist
    w = {"chr%d" % i: float(250 - 10 * i) for i in range(1, 23)}
    a = pdist.assign_chromosomes(w, 8); b = pdist.assign_chromosomes(dict(reversed(list(w.items()))), 8)
    assert a == b
    load = [sum(w[c] for c in w if a[c] == r) for r in range(8)]
    assert max(load) / (sum(load) / 8) < 1.25
    assert pdist.assign_chromosomes(w, 1) == {c: 0 for c in w}
