"""k_phase_pair / k_phase_general (phz_rowsdev.hip: phase_v3 on the GPU, one wave per component) under the host-side HIP emulation,
against the native host routine phz_phase_block on random connected components -- conflicts, ties, weak points, brute force,
stitching, components beyond the kernel's limits (they take the host path inside the call).  phz_phase_block itself is pinned against
the reference restatement in tests/test_phase_block.py."""
import ctypes as C
import random

import numpy as np
import pytest

from helpers import EmuContext, emu_library
from test_phase_block import _random_component


def host_phase(lib, n, edges, mbs):
    ei = np.asarray([e[0] for e in edges], dtype=np.int32); ej = np.asarray([e[1] for e in edges], dtype=np.int32)
    ek = np.asarray([e[2] for e in edges], dtype=np.int8)
    first = np.zeros(n + 1, dtype=np.int32); ln = np.zeros(n + 1, dtype=np.int32); cfg = np.zeros(n + 1, dtype=np.uint8); ns = C.c_int32(0)
    st = lib.phz_phase_block(n, len(edges), ei.ctypes.data, ej.ctypes.data, ek.ctypes.data, mbs, first.ctypes.data, ln.ctypes.data, cfg.ctypes.data, C.byref(ns))
    if st != 0:
        return None
    sub = np.full(n, -1, dtype=np.int16); al = np.zeros(n, dtype=np.uint8)
    w = 0; k2 = 0
    for k in range(ns.value):
        if ln[k] <= 0:
            continue
        for t in range(int(ln[k])):
            sub[first[k] + t] = k2; al[first[k] + t] = 1 if cfg[w] == ord("1") else 0; w += 1
        k2 += 1
    return sub, al, k2


def batch_phase(ctx, comps, mbs):
    cs = np.zeros(len(comps) + 1, dtype=np.uint32); es = np.zeros(len(comps) + 1, dtype=np.uint32)
    pi = []; pj = []; pk = []
    for c, (n, edges) in enumerate(comps):
        cs[c + 1] = cs[c] + n; es[c + 1] = es[c] + len(edges)
        pi += [e[0] for e in edges]; pj += [e[1] for e in edges]; pk += [e[2] for e in edges]
    pi = np.asarray(pi, dtype=np.int32); pj = np.asarray(pj, dtype=np.int32); pk = np.asarray(pk, dtype=np.int8)
    sub = np.zeros(int(cs[-1]), dtype=np.int32); al = np.zeros(int(cs[-1]), dtype=np.uint8); ns = np.zeros(len(comps), dtype=np.uint32)
    vp = lambda a: C.c_void_p(a.ctypes.data)
    ctx.check(ctx.lib.phz_phase_components(ctx.h, len(comps), vp(cs), vp(es), vp(pi), vp(pj), vp(pk), mbs, vp(sub), vp(al), vp(ns)))
    return cs, sub, al, ns


@pytest.mark.parametrize("seed,mbs", [(1, 3), (2, 5), (3, 9), (4, 15), (5, 0), (6, 4), (7, 18), (8, 12)])
def test_device_phase_matches_host_routine(seed, mbs):
    lib = emu_library()
    ctx = EmuContext(lib)
    rng = random.Random(4200 + seed)
    comps = []; want = []
    for t in range(160):
        n = rng.choice([2, 2, 2, 3, 3, 4, 5, 6, 8, 11, 14, 19, 26, 40]) if mbs else rng.choice([2, 3, 4, 6, 9, 12])
        if t % 40 == 7 and mbs:
            n = rng.choice([270, 300])                                   # beyond PH_NMAX: host path inside the call
        edges = _random_component(rng, n, rng.randint(0, n), rng.choice([0.0, 0.05, 0.15, 0.3]), rng.choice([0.0, 0.05, 0.2]))
        w = host_phase(lib, n, edges, mbs)
        if w is None:
            continue
        comps.append((n, edges)); want.append(w)
    cs, sub, al, ns = batch_phase(ctx, comps, mbs)
    split = 0
    for c, ((n, edges), (wsub, wal, wns)) in enumerate(zip(comps, want)):
        lo, hi = int(cs[c]), int(cs[c + 1])
        assert int(ns[c]) == wns, (n, mbs, edges)
        assert np.array_equal(sub[lo:hi], wsub), (n, mbs, edges)
        live = wsub >= 0
        assert np.array_equal(al[lo:hi][live], wal[live]), (n, mbs, edges)
        split += wns != 1 or not live.all()
    assert len(comps) > 100 and split > 10
