"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol."""
import os
import re

import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def lib():
    from phaser_amd import _lib
    _lib.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from phaser_amd import _lib
    hdr = open(os.path.join(REPO, "include", "phz.h")).read()
    declared = set(re.findall(r"\b(phz_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_strerror(lib):
    assert lib.phz_version() >= 100
    assert lib.phz_strerror(0) == b"ok"
    assert b"capacity" in lib.phz_strerror(-3)


def test_product_path_refuses_without_gpu():
    """No silent CPU fallback: without a HIP device the product path must raise."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from phaser_amd import _lib
    with pytest.raises(_lib.PhzError):
        _lib.Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "phaser_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "rvm_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_package_asks_for_hardware_queues_unless_the_user_did():
    """phaser_amd/__init__.py: GPU_MAX_HW_QUEUES=16 before the ROCm runtime starts (the device BAM decoder overlaps K_inflate launches and copies on
    separate hardware queues), never over a value the user exported."""
    import subprocess, sys
    code = "import os, phaser_amd; print(os.environ.get('GPU_MAX_HW_QUEUES'))"
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env["PYTHONPATH"] = REPO
    assert subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip() == "16"
    env["GPU_MAX_HW_QUEUES"] = "4"
    assert subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip() == "4"


def test_hardware_queues_are_not_claimed_after_the_runtime_started():
    """(round-5 advisor) An application that initialised the GPU runtime BEFORE importing the package keeps the runtime's four hardware queues whatever the
    variable says afterwards: the package then leaves the variable alone and tells the BAM decoder (PHZ_HW_QUEUES_LATE) to keep its one-launch policy.
    The started runtime is simulated by a torch stand-in whose cuda.is_initialized() answers True."""
    import subprocess, sys
    code = ("import sys, types, os; t = types.ModuleType('torch'); t.cuda = types.SimpleNamespace(is_initialized=lambda: True); sys.modules['torch'] = t; "
            "import phaser_amd; print(os.environ.get('GPU_MAX_HW_QUEUES'), os.environ.get('PHZ_HW_QUEUES_LATE'))")
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "PHZ_HW_QUEUES_LATE")}
    env["PYTHONPATH"] = REPO
    assert subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip() == "None 1"
    env["GPU_MAX_HW_QUEUES"] = "16"          # exported by the user before the application started: trusted
    assert subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip() == "16 None"
