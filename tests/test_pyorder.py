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


@pytest.mark.parametrize("case", sorted(set(c[0] for c in _cases())))
def test_five_files_raw_bytes_under_hashseed0(case):
    if sys.version_info[:2] != (3, 10):
        pytest.skip("the goldens carry CPython 3.10's set order")
    env = dict(os.environ, PYTHONHASHSEED="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "pyorder_worker.py"), case], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().split("\n")[-1])
    assert all(v == "raw" for v in res.values()), res


def test_py_hash_order_refuses_a_randomised_interpreter():
    env = dict(os.environ, PYTHONHASHSEED="12345")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "pyorder_worker.py"), "pipe_one"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "PYTHONHASHSEED=0" in (r.stderr + r.stdout)
