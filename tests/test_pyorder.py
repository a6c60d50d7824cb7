"""Raw-byte tier: with Config.py_hash_order=1 and PYTHONHASHSEED=0 the five files equal the reference's files BYTE FOR BYTE -- row
order of variant_connections / singleton rows, aReads / bReads labels, blacklisted-variant order included -- on every golden fixture (the goldens were
written by the reference under PYTHONHASHSEED=0, CPython 3.10).  The replay of the reference's set constructions is phaser_amd/pyorder.py; the host
stages run on the GPU-stage fixtures (tests/golden/tally)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO
from test_host_stages import _cases


@pytest.mark.parametrize("impl", ["native", "python"])
@pytest.mark.parametrize("case", sorted(set(c[0] for c in _cases())))
def test_five_files_raw_bytes_under_hashseed0(case, impl):
    """impl native: libphz's restatement of the str hash and the set (phz_pyorder.cpp), the default of --py_hash_order 1; impl python: the pure-Python
    twin with real set objects (PHZ_PYORDER_PYTHON=1)."""
    if impl == "python" and sys.version_info[:2] != (3, 10):
        pytest.skip("the goldens carry CPython 3.10's set order")
    env = dict(os.environ, PYTHONHASHSEED="0")
    if impl == "python":
        env["PHZ_PYORDER_PYTHON"] = "1"
    else:
        env.pop("PHZ_PYORDER_PYTHON", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "pyorder_worker.py"), case], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().split("\n")[-1])
    assert all(v == "raw" for v in res.values()), res


@pytest.mark.parametrize("case", ["pipe_two", "pipe_sparse", "opts_blacklist"])
def test_native_raw_bytes_do_not_depend_on_the_interpreters_hash_seed(case):
    """The native tier restates CPython 3.10's seed-0 hash itself: the reference's bytes come out whatever seed (or version) the interpreter around it has."""
    env = dict(os.environ, PYTHONHASHSEED="12345")
    env.pop("PHZ_PYORDER_PYTHON", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "pyorder_worker.py"), case], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().split("\n")[-1])
    assert all(v == "raw" for v in res.values()), res


def test_python_twin_refuses_a_randomised_interpreter():
    env = dict(os.environ, PYTHONHASHSEED="12345", PHZ_PYORDER_PYTHON="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "pyorder_worker.py"), "pipe_one"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "PYTHONHASHSEED=0" in (r.stderr + r.stdout)


_SET_CHECK = r'''
import sys, random, ctypes as C
import numpy as np
sys.path.insert(0, %r)
from phaser_amd import _lib
lib = _lib.load()
assert sys.flags.hash_randomization == 0
rng = random.Random(int(sys.argv[1]))
def pool(items):
    b = "".join(items).encode(); off = np.zeros(len(items) + 1, np.uint32); off[1:] = np.cumsum([len(x) for x in items]) if items else 0
    return b, off
for n in list(range(0, 40)) + [100, 1000]:
    s = "".join(rng.choice("ACGTchr_0123456789.:|abcXYZ") for _ in range(n))
    assert lib.phz_py_str_hash(s.encode(), len(s)) == hash(s), s
def order(items, n_a=None):
    b, off = pool(items)
    out = np.zeros(max(1, len(items)), np.int32)
    k = lib.phz_py_set_order(b, C.c_void_p(off.ctypes.data), len(items), len(items) if n_a is None else n_a, 0 if n_a is None else 1, C.c_void_p(out.ctypes.data))
    return [items[i] for i in out[:k]]
for trial in range(1200):
    n = rng.choice([0, 1, 2, 5, 8, 9, 20, 33, 100, 500, 3000, 60000 if trial %% 400 == 0 else 7])
    universe = ["chr%%d_%%d_%%s_%%s" %% (rng.randint(1, 22), rng.randint(1, 10 ** rng.randint(2, 8)), rng.choice("ACGT"), rng.choice("ACGT")) for _ in range(max(1, n // rng.choice([1, 1, 2, 5])))]
    items = [rng.choice(universe) for _ in range(n)]
    assert order(items) == list(set(items)), ("set", n)
    na = rng.randint(0, n)
    a, b = items[:na], items[na:]
    if trial %% 3 == 0 and len(a) > 8:          # the copy-and-discard path of set_difference: len(a) // 4 > len(b)
        b = [rng.choice(a) for _ in range(max(0, len(set(a)) // 4 - 1 - rng.randint(0, 3)))]
    assert order(a + b, len(a)) == list(set(a) - set(b)), ("difference", len(a), len(b))
print("ok")
'''


def test_restated_str_hash_and_set_equal_the_interpreters():
    """phz_py_str_hash = hash(str) of a CPython 3.10 started with PYTHONHASHSEED=0, and phz_py_set_order = the iteration order of set(items) and of
    set(a) - set(b) (both paths of set_difference) on random id-like strings, sizes from 0 to 60,000 -- against the real objects of the running interpreter."""
    if sys.version_info[:2] != (3, 10):
        pytest.skip("pins CPython 3.10's hash and set")
    env = dict(os.environ, PYTHONHASHSEED="0")
    r = subprocess.run([sys.executable, "-c", _SET_CHECK % REPO, "5"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
