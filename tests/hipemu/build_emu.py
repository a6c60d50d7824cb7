"""TEST INFRASTRUCTURE ONLY: builds tests/hipemu/_build/libphz_emu.so -- the translation units of libphz that hold no gfx950
intrinsics, compiled by g++ against the host-side HIP emulation (hipemu.h) -- so that the CPU suite can run kernel LOGIC without
a GPU.  Never loaded by phaser_amd (the product path raises without libphz.so + a GPU)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "phaser_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libphz_emu.so")
UNITS = ["phz_api.hip", "phz_tally.hip", "phz_rowsdev.hip", "phz_rows.cpp"]


def build(verbose=False, tally_tile=0):
    """tally_tile: build the variant libphz_emu_t<N>.so whose K_tally groups tiles of N lines (256 / 512): the small fixtures then
    straddle tiles, which is what sends QNAMEs through the spill path of k_tile"""
    os.makedirs(OUT, exist_ok=True)
    LIB = os.path.join(OUT, "libphz_emu_t%d.so" % tally_tile if tally_tile else "libphz_emu.so")
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(REPO, "include", "phz.h"),
                                                                                  os.path.join(HERE, "hipemu.h"), os.path.join(HERE, "hipemu.cpp")]
    newest_hdr = max(os.path.getmtime(h) for h in hdr)
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-I" + os.path.join(HERE, "include"), "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
    jobs = []; objs = []
    for u in UNITS + ["hipemu.cpp"]:
        src = os.path.join(HERE if u == "hipemu.cpp" else CSRC, u)
        variant = tally_tile and u == "phz_tally.hip"
        obj = os.path.join(OUT, u + (".t%d" % tally_tile if variant else "") + ".o"); objs.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            jobs.append(["g++"] + flags + (["-DPHZ_TALLY_TILE=%d" % tally_tile] if variant else []) + ["-x", "c++", "-c", src, "-o", obj])
    if jobs or not os.path.exists(LIB):
        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(run, jobs))
        run(["g++", "-shared", "-fPIC"] + objs + ["-o", LIB, "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
