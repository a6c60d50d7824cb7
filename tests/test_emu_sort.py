"""The device radix sort of phaser_amd/csrc/phz_sort.h (one launch per pass, decoupled look-back over the tiles' digit counts) under the host-side HIP
emulation: keys of 4 and 8 bytes, one and two bit ranges, sizes from one key to several tiles of 4,096, against numpy's stable sort and against the
three-launch passes it replaced.  The GPU run of the same checks is tests/test_gpu_pipeline.py::test_device_sort_matches_a_stable_host_sort."""
import ctypes as C

import numpy as np
import pytest

from helpers import EmuContext, emu_library


def device_sort(ctx, keys, vals, ranges, three_launch=0):
    kb = keys.dtype.itemsize
    ko = np.empty_like(keys); vo = np.empty_like(vals)
    rg = np.asarray(ranges, dtype=np.int32).reshape(-1)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    ctx.check(ctx.lib.phz_selftest_sort(ctx.h, kb, vp(keys), vp(vals), len(keys), vp(rg), len(ranges), three_launch, vp(ko), vp(vo)))
    return ko, vo


def host_sort(keys, vals, ranges):
    order = np.arange(len(keys))
    for lo, hi in ranges:
        d = (keys[order].astype(np.uint64) >> np.uint64(lo)) & np.uint64((1 << (hi - lo)) - 1)
        order = order[np.argsort(d, kind="stable")]
    return keys[order], vals[order]


CASES = [(np.uint32, [(0, 21)]), (np.uint32, [(0, 32)]), (np.uint64, [(0, 21), (32, 54)]), (np.uint64, [(0, 13)]), (np.uint32, [(3, 9)])]


@pytest.mark.parametrize("dtype,ranges", CASES, ids=[str(i) for i in range(len(CASES))])
def test_sort_matches_a_stable_host_sort(dtype, ranges):
    ctx = EmuContext(emu_library())
    rng = np.random.default_rng(17)
    for n in (1, 2, 63, 64, 1000, 4096, 4097, 9000, 20000):
        hi = max(h for _, h in ranges)
        mask = 0
        for lo_, hi_ in ranges:
            mask |= ((1 << (hi_ - lo_)) - 1) << lo_
        # bits outside the ranges are zero, as in every caller: a pass takes whole 8-bit digits, so the last digit of a range may reach beyond it
        keys = (rng.integers(0, 1 << min(hi, 62), size=n, dtype=np.uint64) & np.uint64(mask)).astype(dtype)
        if n >= 1000:
            keys[rng.integers(0, n, n // 3)] = keys[0]          # long runs of one digit: the per-wave cursors and the look-back carry them
        vals = np.arange(n, dtype=np.uint32)
        want = host_sort(keys, vals, ranges)
        for three in (0, 1):
            got = device_sort(ctx, keys, vals, ranges, three)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), (n, three)
