#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the memory-side calibration patterns of libphz.so (phz_membench, phaser_amd/csrc/phz_ubench.hip) -- coalesced streams,
1-byte gathers at one load per 32..512-byte unit (permuted / address order) and coalesced stores, each with a KNOWN byte count -- one JSON line
per pattern.  tools/prof_calib.sh runs this under rocprofv3 --pmc passes and divides the counters by these byte counts."""
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phaser_amd import _lib

PATTERNS = [("stream4", 0, 0), ("stream16", 1, 0)] + [("gather_perm%d" % u, 2, u) for u in (32, 64, 128, 256, 512)] + \
           [("gather_seq%d" % u, 3, u) for u in (32, 64, 128, 256, 512)] + [("write4", 4, 0), ("write8", 5, 0), ("write16", 6, 0)]


def kernel_name(kind, unit):
    if kind < 2:
        return "k_mb_stream<%d>" % (4 if kind == 0 else 16)
    if kind < 4:
        return "k_mb_gather<%d, %d>" % (unit, 1 if kind == 2 else 0)
    return "k_mb_write<%d>" % {4: 4, 5: 8, 6: 16}[kind]


def main():
    log2_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ctx = _lib.Context(0)
    for name, kind, unit in PATTERNS:
        sec = C.c_double(0); known = C.c_int64(0); req = C.c_int64(0)
        ctx.check(ctx.lib.phz_membench(ctx.h, kind, log2_bytes, unit or 32, reps, C.byref(sec), C.byref(known), C.byref(req)))
        print(json.dumps({"pattern": name, "kernel": kernel_name(kind, unit), "array_bytes": 1 << log2_bytes, "unit": unit, "lane_bytes": known.value,
                          "requests": req.value, "seconds": sec.value, "launches": reps,
                          "lane_GBps": known.value / sec.value / 1e9 if sec.value else None,
                          "requests_per_s": req.value / sec.value if sec.value else None}), flush=True)


if __name__ == "__main__":
    main()
