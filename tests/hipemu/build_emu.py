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
UNITS = ["phz_api.hip", "phz_tally.hip", "phz_rowsdev.hip", "phz_rows.cpp", "phz_inflate.hip"]


def build(verbose=False, tally_tile=0, row_wave_min=None, stat_n=None):
    """tally_tile: build the variant libphz_emu_t<N>.so whose K_tally groups tiles of N lines (256 / 512): the small fixtures then
    straddle tiles, which is what sends QNAMEs through the spill path of k_tile.
    row_wave_min: the variant libphz_emu_w<N>.so whose row stage formats the rows of blocks with more than N variants by a wave each
    (0: every block row, so that the fixtures exercise the wave sinks)
    stat_n: the variant libphz_emu_s<N>.so whose gwStat table and LDS piece arrays cover blocks of up to N variants only (product: 512), so that the
    fixtures' blocks take the paths of a block beyond them (host-formatted gwStat text, piece arrays in the global pool)"""
    os.makedirs(OUT, exist_ok=True)
    import fcntl
    with open(os.path.join(OUT, ".lock"), "w") as lk:        # pytest-xdist workers build the same files: one at a time
        fcntl.flock(lk, fcntl.LOCK_EX)
        return _build_locked(verbose, tally_tile, row_wave_min, stat_n)


def _build_locked(verbose, tally_tile, row_wave_min, stat_n=None):
    tag = ("_t%d" % tally_tile if tally_tile else "") + ("_w%d" % row_wave_min if row_wave_min is not None else "") + ("_s%d" % stat_n if stat_n is not None else "")
    extra = os.environ.get("PHZ_EMU_EXTRA_DEFS", "").split()      # experiment builds: extra -D flags for every unit, e.g. "-DPHZ_TILE_TB=512"
    if extra:
        import hashlib
        tag += "_x" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8]
    LIB = os.path.join(OUT, "libphz_emu%s.so" % tag)
    variant_defs = {}
    if tally_tile:
        variant_defs["phz_tally.hip"] = ["-DPHZ_TALLY_TILE=%d" % tally_tile, "-DPHZ_RL_STAGE=8"]      # + a tiny read-list stage: the fallback of k_rl_sort
    if row_wave_min is not None:
        variant_defs["phz_rowsdev.hip"] = ["-DPHZ_ROW_WAVE_MIN=%d" % row_wave_min]
    if stat_n is not None:
        variant_defs["phz_rowsdev.hip"] = variant_defs.get("phz_rowsdev.hip", []) + ["-DPHZ_STAT_N=%d" % stat_n]
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(REPO, "include", "phz.h"),
                                                                                  os.path.join(HERE, "hipemu.h"), os.path.join(HERE, "hipemu.cpp")]
    newest_hdr = max(os.path.getmtime(h) for h in hdr)
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-I" + os.path.join(HERE, "include"), "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
    jobs = []; objs = []
    for u in UNITS + ["hipemu.cpp"]:
        src = os.path.join(HERE if u == "hipemu.cpp" else CSRC, u)
        defs = variant_defs.get(u, []) + extra
        obj = os.path.join(OUT, u + (tag if defs else "") + ".o"); objs.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            jobs.append(["g++"] + flags + defs + ["-x", "c++", "-c", src, "-o", obj])
    if jobs or not os.path.exists(LIB):
        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(run, jobs))
        run(["g++", "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp", "-lpthread"])
        os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
