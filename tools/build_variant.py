#!/usr/bin/env python3
"""Experiment builds: libphz with one translation unit compiled under extra -D flags, linked with the regular objects of the others.
  python tools/build_variant.py NAME phz_map.hip -DPHZ_MAP_WIN=256 ...   ->  phaser_amd/variants/libphz_NAME.so
Select it on the GPU box with PHZ_LIB_PATH=phaser_amd/variants/libphz_NAME.so (see _lib.load); tools/ab_env.sh compares settings."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from phaser_amd import _lib
name, unit, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
_lib.build()
csrc = _lib.CSRC; bdir = os.path.join(csrc, "build"); vdir = os.path.join(REPO, "phaser_amd", "variants"); os.makedirs(vdir, exist_ok=True)
obj = os.path.join(vdir, "%s.%s.o" % (name, unit))
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + csrc]
subprocess.check_call(["hipcc"] + flags + extra + ["-c", os.path.join(csrc, unit), "-o", obj])
objs = [obj if os.path.basename(s) == unit else os.path.join(bdir, os.path.basename(s) + ".o") for s in _lib.hip_sources()]
out = os.path.join(vdir, "libphz_%s.so" % name)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", out, "-lz", "-lpthread"])
print(out)
