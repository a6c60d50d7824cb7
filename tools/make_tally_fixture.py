#!/usr/bin/env python3
"""Runs ON THE GPU BOX (gpurun): for every golden pipeline case, the GPU stage results the host stages start from
(K_tally arrays per chromosome, component labels, AS-cutoff log lines), saved so that the CPU-only tests can run the
host stages (ordering rules, pair tests, block phasing, row formatting, merge) against the reference's five files
without a GPU.  Writes gpurun_out/tally/<case>.pkl.gz; copy them to tests/golden/tally/."""
import gzip
import json
import os
import pickle
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from phaser_amd import samio, vcf
from phaser_amd.engine import Config, Engine
from phaser_amd.mapper import Mapper
from helpers import option_case_kwargs

GOLD = os.path.join(REPO, "tests", "golden")
gz = lambda p: gzip.open(p, "rt").read()
mapper = Mapper(0)


def cases():
    d = os.path.join(GOLD, "pipe_one")
    yield "pipe_one", open(os.path.join(d, "in.vcf")).read(), {"a.bam": {"chr22": gz(os.path.join(d, "a.chr22.sam.gz"))}}, {}, {}, 0.0
    d = os.path.join(GOLD, "pipe_two")
    yield ("pipe_two", open(os.path.join(d, "in.vcf")).read(),
           {b + ".bam": {c: gz(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")} for b in ("t1", "t2")}, {}, {}, 0.0)
    d = os.path.join(GOLD, "pipe_sparse")
    yield ("pipe_sparse", open(os.path.join(d, "in.vcf")).read(),
           {b + ".bam": {c: gz(os.path.join(d, "%s.%s.sam.gz" % (b, c))) for c in ("chr3", "chr11", "chr19")} for b in ("s1", "s2", "s3")}, {}, {}, 0.0)
    for tag in "abc":
        d = os.path.join(GOLD, "pipe_noisy_" + tag)
        meta = json.load(open(os.path.join(d, "meta.json")))
        yield ("pipe_noisy_" + tag, open(os.path.join(d, "in.vcf")).read(), {"n.bam": {"chr22": gz(os.path.join(d, "n.chr22.sam.gz"))}},
               {}, {"max_block_size": meta["max_block_size"]}, 0.0)
    from phaser_amd import synth
    g = json.load(open(os.path.join(GOLD, "c1", "meta.json")))["gen"]
    v, gs, ge, w = synth.make_variants(g["region"][0], g["region"][1], g["region"][2], g["n_snps"], g["vseed"], n_genes=g["n_genes"])
    rb = synth.make_reads(v, gs, ge, w, g["n_pairs"], g["rseed"])
    rf = rb.select(synth.samtools_keep(rb, g["mapq"]))
    yield ("c1", "\n".join(synth.vcf_lines([v])) + "\n", {"c1.bam": {"chr22": "\n".join(synth.sam_lines(rf, [("chr22", 50818468)])) + "\n"}},
           {}, {}, 0.0)
    d = os.path.join(GOLD, "pipe_indel")
    yield ("pipe_indel", open(os.path.join(d, "in.vcf")).read(), {"i.bam": {"chr22": gz(os.path.join(d, "i.chr22.sam.gz"))}},
           {"include_indels": 1}, {"include_indels": 1}, 0.0)
    d0 = os.path.join(GOLD, "pipe_opts")
    meta = json.load(open(os.path.join(d0, "cases.json")))
    bams = {b + ".bam": {c: gz(os.path.join(d0, "%s.%s.sam.gz" % (b, c))) for c in ("chr21", "chr22")} for b in ("o1", "o2")}
    for name in meta["cases"]:
        load, cfg, baseq, isize = option_case_kwargs(name, meta["cases"][name], meta["blacklist"])
        yield "opts_" + name, open(os.path.join(d0, "in.vcf")).read(), bams, load, cfg, isize


os.makedirs(os.path.join(REPO, "gpurun_out", "tally"), exist_ok=True)
ONLY = sys.argv[1:]
for name, vcf_text, bams, load, cfg, isize in cases():
    if ONLY and name not in ONLY:
        continue
    load = dict(load); cfgk = dict(cfg)
    inc = load.pop("include_indels", 0); cfgk.pop("include_indels", None)
    vs = vcf.load_variants(vcf_text, include_indels=inc, **load)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from phasing_oracle import bam_display_names
    eng = Engine(vs, bam_display_names(list(bams.keys())), Config(include_indels=inc, device_rows=False, **cfgk), mapper=mapper)
    interners = {}
    for bi, (bam, per_chrom) in enumerate(bams.items()):
        for chrom in vs.chroms:
            if chrom not in per_chrom:
                continue
            for c2, sh in samio.shards_from_sam(per_chrom[chrom], interners, isize).items():
                eng.add_shard(bi, c2, sh.to("cuda"), len(interners[c2]), interners[c2].names)
        for c2 in interners:
            eng.n_qid[c2] = len(interners[c2])
        eng.close_bam(bi)
    out = eng.finish()
    # per chromosome, the GPU stage results in the fixture format of tests/helpers.genome_from_saved: K_tally arrays of that
    # chromosome alone + its kept call lines (variant, QNAME id, BAM, class)
    import ctypes as C
    from phaser_amd import _lib
    tally = {}
    all_chroms = list(eng.chrom_list)
    for c in all_chroms:
        eng.chrom_list = [c]
        G = eng.G = eng._tally_genome()
        eng._fetch_tally()
        nv = G["nv"]; ne = len(G["ea"]); nl = G["n_lines"]
        cls = np.zeros(max(1, nl), dtype=np.uint8); cells = np.zeros(max(1, ne * 9), dtype=np.int32)
        o = _lib.phz_tally_out(None, None, None, None, C.c_void_p(cls.ctypes.data), None, None, C.c_void_p(cells.ctypes.data), None, None, None, None)
        eng.ctx.check(eng.lib.phz_tally_fetch(eng.ctx.h, C.byref(o), _lib.PHZ_HOST))
        lv = []; lq = []; lb = []; offs = []
        base = 0
        for b, sh in enumerate(eng.shards[c]):
            if sh is None:
                continue
            offs.append((b, base, sh.calls.n))
            lv.append(sh.calls.var_idx.cpu().numpy()); lq.append(sh.qid[sh.calls.read_idx.long()].cpu().numpy())
            lb.append(np.full(sh.calls.n, b, dtype=np.int32)); base += sh.calls.n
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
        tally[c] = {"nv": nv, "var_count": G["var_count"].copy(), "var_first": G["var_first"].copy(), "var_distinct": G["var_distinct"].copy(),
                    "line_cls": cls[:nl].copy(), "ea": G["ea"].copy(), "eb": G["eb"].copy(), "cells": cells[:ne * 9].reshape(ne, 9).copy(),
                    "linked": G["linked"].astype(bool), "var_rank": G["var_rank"].copy(), "line_var": cat(lv, np.int32),
                    "line_qid": cat(lq, np.int32), "line_bam": cat(lb, np.int32), "bam_offsets": offs}
    eng.chrom_list = all_chroms
    labels = {}
    rec_ = {"tally": tally, "labels": labels, "n_qid": dict(eng.n_qid), "qnames": dict(eng.qnames), "as_log": [l for l in eng.log if "alignment score" in l]}
    with gzip.open(os.path.join(REPO, "gpurun_out", "tally", name + ".pkl.gz"), "wb") as f:
        pickle.dump(rec_, f, protocol=4)
    print(name, {c: int(R["line_cls"].shape[0]) for c, R in tally.items()}, "phased", eng.phased)
