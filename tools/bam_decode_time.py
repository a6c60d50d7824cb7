#!/usr/bin/env python3
"""GPU-box (host-only) timing of the native BAM path on the file tools/run_cli_scale.py leaves in /tmp, per thread count.
usage: tools/bam_decode_time.py [threads ...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
os.environ["PHZ_TIMING"] = "1"
from phaser_amd import bamio
for th in [int(x) for x in sys.argv[1:]] or [32]:
    t0 = time.perf_counter()
    sh = bamio.shards_from_bam_native("/tmp/cli_scale.bam", {}, 255, True, True, 0.0, threads=th)
    print("threads %d: %d shards, %d kept records, %.2f s" % (th, len(sh), sum(s.n for s in sh.values()), time.perf_counter() - t0), flush=True)
    del sh
