#!/usr/bin/env python3
"""Runs ON THE GPU BOX (gpurun): per-chromosome AS histograms and noise counters of the pipe_two fixture, saved so
that the CPU-only gloo tests can exercise the multi-rank all-reduce path without a GPU (the K_tally arrays the host
stages start from come from tools/make_tally_fixture.py).  Writes gpurun_out/frags_pipe_two.json; gzip it into tests/golden/."""
import ctypes as C
import gzip
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from phaser_amd import _lib, samio, vcf
from phaser_amd.engine import Config, Engine, _p

d = os.path.join(REPO, "tests", "golden", "pipe_two")
gz = lambda p: gzip.open(p, "rt").read()
bams = {b + ".bam": {c: gz(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")} for b in ("t1", "t2")}
vs = vcf.load_variants(open(os.path.join(d, "in.vcf")).read())
eng = Engine(vs, ["t1", "t2"], Config())
interners = {}
hists = {}
for bi, (bam, per_chrom) in enumerate(bams.items()):
    for chrom in vs.chroms:
        for c2, sh in samio.shards_from_sam(per_chrom[chrom], interners).items():
            eng.add_shard(bi, c2, sh.to("cuda"), len(interners[c2]), interners[c2].names)
    for c2 in interners:
        eng.n_qid[c2] = len(interners[c2])
    # per-chromosome AS histograms of this BAM (what each rank would contribute)
    for c2 in vs.chroms:
        sh = eng.shards[c2][bi]
        h = torch.zeros(_lib.PHZ_AS_BINS, dtype=torch.int64, device="cuda")
        ln = eng._lines(sh, bi)
        eng.ctx.check(eng.lib.phz_as_histogram(eng.ctx.h, C.byref(ln), _p(h), _lib.PHZ_DEVICE))
        hh = h.cpu().numpy(); nz = np.nonzero(hh)[0]
        hists["%d:%s" % (bi, c2)] = [[int(i), int(hh[i])] for i in nz]
    eng.close_bam(bi)
counts = {}
match = mism = 0
all_chroms = list(eng.chrom_list)
for c in list(vs.chroms):
    eng.chrom_list = [c]                 # what the rank owning c alone would contribute
    m, mm = eng.tally_all()
    counts[c] = [m, mm]
    match += m; mism += mm
eng.chrom_list = all_chroms
noise = Engine.noise_from_counts(match, mism)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump({"hists": hists, "counts": counts, "chroms": list(vs.chroms), "log": eng.log},
          open(os.path.join(REPO, "gpurun_out", "frags_pipe_two.json"), "w"))
print("wrote histograms / counts for", list(vs.chroms), "noise", noise)
