#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the issue-rate microbenchmarks of libphz.so (phz_microbench, phaser_amd/csrc/phz_ubench.hip) -- wave64 VALU, SALU and
LDS instructions per second over the chip at 1 / 2 / 4 / 8 waves per SIMD -- as one JSON object (kept under profiles/ and read by bench.py)."""
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from phaser_amd import _lib


def measure(ctx=None, iters=20000):
    ctx = ctx or _lib.Context(0)
    out = {"unit": "wave64 instructions per second, whole chip", "kinds": {}}
    for kind, name in ((0, "valu_v_add_u32"), (1, "salu_s_add_u32"), (2, "lds_ds_read_b32")):
        rates = {}
        for w in (1, 2, 4, 8):
            r = C.c_double(0); cu = C.c_int(0); mhz = C.c_int(0)
            ctx.check(ctx.lib.phz_microbench(ctx.h, kind, w, iters, C.byref(r), C.byref(cu), C.byref(mhz)))
            rates[str(w)] = r.value
            out["n_cu"] = cu.value; out["clock_mhz"] = mhz.value
        best = max(rates.values())
        hz = out["clock_mhz"] * 1e6
        out["kinds"][name] = {"by_waves_per_simd": rates, "peak": best,
                              "cycles_per_inst_per_simd": 4 * out["n_cu"] * hz / best if best else None,
                              "insts_per_cycle_per_cu": best / (out["n_cu"] * hz) if best else None}
    return out


if __name__ == "__main__":
    print(json.dumps(measure(), indent=1))
