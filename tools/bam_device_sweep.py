#!/usr/bin/env python3
"""Runs ON THE GPU BOX: file -> resident shards of the whole-genome BAM of configs[2] (written to /tmp first, not timed) under the switches of the device BAM
decoder: PHZ_BAM_REGISTER (1: the mapped file registered with the runtime, DMA straight out of the page cache; 0: pread into page-locked staging on host
threads) x PHZ_BAM_CHUNK_MB (compressed bytes per K_inflate launch).  Best of three per setting; the stage lines of the last run with PHZ_TIMING=1.
usage: tools/bam_device_sweep.py [records=80000000]"""
import argparse, os, sys, tempfile, time, shutil
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
import bench
from phaser_amd import bamio
from phaser_amd.mapper import Mapper

a = argparse.Namespace(records=int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000, snps=1_500_000, baseq=10)
tmp = tempfile.mkdtemp(prefix="phz_bam_sweep_")
try:
    path, vcfgz, vsets, nrec, t_write = bench.write_genome_files(tmp, a, "cuda:0")
    print("BAM %d records, %.2f GB, written in %.1f s" % (nrec, os.path.getsize(path) / 1e9, t_write), flush=True)
    ctx = Mapper(0).ctx
    settings = [(r, "1", mb) for r in ("1", "0") for mb in ("1280", "640", "2048", "4096")]
    if os.environ.get("PHZ_SWEEP") == "streams":        # K_inflate launches of consecutive chunks on several streams (they fit the chip together since the hot/cold symbol tables)
        settings = [("0", st, mb) for st, mb in (("1", "4096"), ("1", "1280"), ("3", "1280"), ("3", "640"), ("4", "320"))]
    if os.environ.get("PHZ_SWEEP") == "streams2":
        settings = [("0", st, mb) for st, mb in (("3", "1280"), ("2", "1280"), ("2", "2048"), ("3", "960"), ("4", "960"), ("4", "1280"), ("3", "1600"))]
    for reg, nst, mb in settings:
        if True:
            os.environ["PHZ_BAM_REGISTER"] = reg; os.environ["PHZ_BAM_CHUNK_MB"] = mb; os.environ["PHZ_BAM_INFLATE_STREAMS"] = nst
            best = None
            for rep in range(3):
                if rep == 2:
                    os.environ["PHZ_TIMING"] = "1"
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sh = bamio.shards_from_bam_device(ctx, path, {}, 255, True, True, 0.0, device="cuda:0")
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                os.environ.pop("PHZ_TIMING", None)
                assert sh is not None
                kept = sum(s.n for s in sh.values())
                del sh
                best = dt if best is None else min(best, dt)
            print("PHZ_BAM_REGISTER=%s PHZ_BAM_INFLATE_STREAMS=%s PHZ_BAM_CHUNK_MB=%-5s file -> shards %.3f s (%.1f M BAM records/s, %.1f GB/s of BGZF), %d kept" %
                  (reg, nst, mb, best, nrec / best / 1e6, os.path.getsize(path) / best / 1e9, kept), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
