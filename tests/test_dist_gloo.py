"""Multi-rank path on CPU: 2 gloo processes own one chromosome each (LPT assignment), all-reduce the per-BAM
AS histograms and the noise counters, gather the per-chromosome fragments to rank 0 and assemble the files.
Per-chromosome stage results come from tests/golden/frags_pipe_two.json.gz (produced on an MI355X by
tools/make_frag_fixture.py); the assembled files must equal what the reference wrote for the same inputs."""
import gzip
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLD, REPO, gz_text
from helpers import OUTPUTS, canonical


def _worker(rank, world, port, result_path):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phaser_amd import dist as pdist
    from phaser_amd.engine import Config, Engine, merge_fragments
    fx = json.load(gzip.open(os.path.join(GOLD, "frags_pipe_two.json.gz"), "rt"))
    chroms = fx["chroms"]
    weights = {c: float(len(fx["frags"][c]["allelic"]) + 1 + i) for i, c in enumerate(chroms)}
    owner = pdist.assign_chromosomes(weights, world)
    assert sorted(set(owner.values())) == list(range(world))
    mine = [c for c in chroms if owner[c] == rank]
    cutoffs = []
    for bam in (0, 1):
        h = torch.zeros(65536, dtype=torch.int64)
        for c in mine:
            for i, n in fx["hists"]["%d:%s" % (bam, c)]:
                h[i] += n
        pdist.allreduce_sum_(h)
        hh = h.numpy(); nz = np.nonzero(hh)[0]
        cutoffs.append(float(np.percentile(np.repeat(nz.astype(np.int64) - 32768, hh[nz]), 5.0)))
    match = sum(fx["counts"][c][0] for c in mine); mism = sum(fx["counts"][c][1] for c in mine)
    match, mism = pdist.allreduce_counts(match, mism)
    noise = Engine.noise_from_counts(match, mism)
    frags = pdist.gather_fragments({c: fx["frags"][c] for c in mine})
    if rank == 0:
        out, summary = merge_fragments(frags, [c for c in chroms if c in frags], Config(), noise)
        json.dump({"out": out, "cutoffs": cutoffs, "noise": noise, "log": summary["log"], "n_frags": len(frags)}, open(result_path, "w"))
    else:
        assert frags is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_gather_merge(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    res = str(tmp_path / "res.json")
    mp.spawn(_worker, args=(2, port, res), nprocs=2, join=True)
    r = json.load(open(res))
    d = os.path.join(GOLD, "pipe_two")
    assert r["n_frags"] == 2
    for name in OUTPUTS:
        want = gz_text(os.path.join(d, "out.%s.txt.gz" % name))
        assert canonical(name, r["out"][name]) == canonical(name, want), name
    log = gz_text(os.path.join(d, "out.log.txt.gz"))
    for c in r["cutoffs"]:
        assert ("using alignment score cutoff of %d" % c) in log
    for line in r["log"]:
        assert line in log, line


def test_lpt_assignment_balanced_and_deterministic():
    from phaser_amd import dist as pdist
    w = {"chr%d" % i: float(250 - 10 * i) for i in range(1, 23)}
    a = pdist.assign_chromosomes(w, 8); b = pdist.assign_chromosomes(dict(reversed(list(w.items()))), 8)
    assert a == b
    load = [sum(w[c] for c in w if a[c] == r) for r in range(8)]
    assert max(load) / (sum(load) / 8) < 1.25
    assert pdist.assign_chromosomes(w, 1) == {c: 0 for c in w}
