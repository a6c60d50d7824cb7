#!/usr/bin/env python3
"""GPU-box check of the device BAM path: every array of every shard equal to the host decoder's (phz_bam_*), several filter settings,
whole file and chromosome-restricted.  usage: tools/bamdev_check.py [file.bam] (default /tmp/cli_scale.bam)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import bamio
from phaser_amd.mapper import Mapper
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/cli_scale.bam"
ctx = Mapper(0).ctx
ok = True
for (mapq, rmdup, paired, isz, chroms) in [(255, True, True, 0.0, None), (0, False, False, 0.0, None), (255, True, True, 300.0, ["chr2", "chr3", "chr21"]),
                                           (1, False, True, 0.0, ["chr22"])]:
    t0 = time.perf_counter()
    hi = {}; di = {}
    host = bamio.shards_from_bam_native(path, hi, mapq, rmdup, paired, isz, chroms=chroms, threads=16)
    t1 = time.perf_counter()
    dev = bamio.shards_from_bam_device(ctx, path, di, mapq, rmdup, paired, isz, chroms=chroms)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if dev is None:
        print("device path declined", (mapq, rmdup, paired, isz, chroms)); ok = False; continue
    bad = []
    if sorted(host) != sorted(dev):
        bad.append("shard sets differ: %s vs %s" % (sorted(host), sorted(dev)))
    for c in host:
        if c not in dev:
            continue
        for f in ("pos", "cigar_off", "cigar", "seq_off", "seq2", "qual", "qid", "aln_score", "has_as"):
            a = getattr(host[c], f); b = getattr(dev[c], f).cpu()
            if a.shape != b.shape or not torch.equal(a, b):
                bad.append("%s.%s differs (%s vs %s)" % (c, f, tuple(a.shape), tuple(b.shape)))
    n = sum(s.n for s in host.values())
    print("mapq %s rmdup %s paired %s isize %s chroms %s: %d shards, %d records | host %.2f s, device %.2f s | %s"
          % (mapq, rmdup, paired, isz, chroms, len(host), n, t1 - t0, t2 - t1, "IDENTICAL" if not bad else bad[:6]), flush=True)
    ok &= not bad
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
