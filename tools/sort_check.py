#!/usr/bin/env python3
"""Runs ON THE GPU BOX: phz_selftest_sort (the device radix sort of phz_sort.h) against numpy's stable sort, one-launch-per-pass and three-launch passes,
over sizes / key widths; prints where the first mismatch is.  usage: tools/sort_check.py"""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import numpy as np
from phaser_amd import _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(5)
vp = lambda a: C.c_void_p(a.ctypes.data)
bad = 0
for dtype, ranges in (("uint32", [(0, 21)]), ("uint64", [(0, 21), (32, 54)]), ("uint32", [(0, 8)]), ("uint32", [(0, 32)])):
    for n in (4096, 8192, 100_000, 1_500_000, 3_000_001):
        mask = 0
        for lo_, hi_ in ranges:
            mask |= ((1 << (hi_ - lo_)) - 1) << lo_
        keys = (rng.integers(0, 1 << 62, size=n, dtype=np.uint64) & np.uint64(mask)).astype(dtype)
        if n >= 100_000:
            keys[rng.integers(0, n, n // 4)] = keys[0]          # a quarter of the keys equal: one digit carries most of every tile
        vals = np.arange(n, dtype=np.uint32)
        order = np.arange(n)
        for lo_, hi_ in ranges:
            d = (keys[order].astype(np.uint64) >> np.uint64(lo_)) & np.uint64((1 << (hi_ - lo_)) - 1)
            order = order[np.argsort(d, kind="stable")]
        rg = np.asarray(ranges, dtype=np.int32).reshape(-1)
        for three in (0, 1):
            for rep in range(3):
                ko = np.empty_like(keys); vo = np.empty_like(vals)
                t0 = time.perf_counter()
                ctx.check(ctx.lib.phz_selftest_sort(ctx.h, keys.dtype.itemsize, vp(keys), vp(vals), n, vp(rg), len(ranges), three, vp(ko), vp(vo)))
                dt = time.perf_counter() - t0
                okv = np.array_equal(vo, vals[order]); okk = np.array_equal(ko, keys[order])
                if not (okv and okk):
                    bad += 1
                    w = np.flatnonzero(vo != vals[order])
                    srt = bool((np.diff(ko.astype(np.int64)) >= 0).all()) if len(ranges) == 1 else None
                    print("MISMATCH %s %s n=%d three=%d rep=%d: %d of %d values differ, first at %d; keys sorted: %s; is a permutation: %s" %
                          (dtype, ranges, n, three, rep, len(w), n, w[0] if len(w) else -1, srt, bool(np.array_equal(np.sort(vo), vals))), flush=True)
        print("%s %s n=%d done (%.1f ms per call incl. copies)" % (dtype, ranges, n, dt * 1e3), flush=True)
print("BAD" if bad else "ALL EQUAL")
