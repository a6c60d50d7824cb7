#!/usr/bin/env python3
"""GPU-box check of K_inflate: every BGZF member of a file inflated on the device, compared byte for byte with zlib on the host.
usage: tools/inflate_check.py [file.bam|file.gz ...]   (default: a generated BGZF text file + /tmp/cli_scale.bam if present)"""
import ctypes as C, os, sys, time, zlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import numpy as np, torch
from phaser_amd import _lib, vcfout
from phaser_amd.mapper import Mapper

mapper = Mapper(0)
lib = mapper.lib if hasattr(mapper, "lib") else _lib.load()
ctx = mapper.ctx


def members(buf):
    """member table of a BGZF byte string: (src, csize, isize, dst) per member"""
    out = []; off = 0; dst = 0; n = len(buf)
    mv = memoryview(buf)
    while off + 18 <= n:
        assert buf[off] == 0x1f and buf[off + 1] == 0x8b
        xlen = int.from_bytes(mv[off + 10:off + 12], "little")
        x = off + 12; bsize = 0
        while x + 4 <= off + 12 + xlen:
            slen = int.from_bytes(mv[x + 2:x + 4], "little")
            if buf[x] == 66 and buf[x + 1] == 67 and slen == 2:
                bsize = int.from_bytes(mv[x + 4:x + 6], "little") + 1
            x += 4 + slen
        assert bsize
        isize = int.from_bytes(mv[off + bsize - 4:off + bsize], "little")
        out.append((off + 12 + xlen, bsize - xlen - 20, isize, dst))
        dst += isize; off += bsize
    return out, dst


def check(path, verify_limit=None):
    raw = np.fromfile(path, dtype=np.uint8)
    t0 = time.perf_counter()
    tab, total = members(raw.tobytes() if len(raw) < (1 << 28) else raw)
    t1 = time.perf_counter()
    rec = np.zeros(len(tab), dtype=[("src", "<u8"), ("csize", "<u4"), ("isize", "<u4"), ("dst", "<u8")])
    for i, (a, b, c, d) in enumerate(tab):
        rec[i] = (a, b, c, d)
    comp = torch.zeros(len(raw) + 16, dtype=torch.uint8, device="cuda")
    comp[:len(raw)] = torch.from_numpy(raw).cuda()
    drec = torch.from_numpy(rec.view(np.uint8)).cuda()
    out = torch.empty(max(1, total), dtype=torch.uint8, device="cuda")
    bad = C.c_int(0)
    for rep in range(3):
        torch.cuda.synchronize(); t2 = time.perf_counter()
        st = lib.phz_bgzf_inflate_device(ctx.h, C.c_void_p(comp.data_ptr()), C.c_void_p(drec.data_ptr()), len(tab), C.c_void_p(out.data_ptr()), C.byref(bad))
        torch.cuda.synchronize(); t3 = time.perf_counter()
        ctx.check(st)
        print("%s: %d members, %.1f MB -> %.1f MB | member scan (python) %.2f s | K_inflate %.1f ms wall, %.1f ms kernel, %.2f GB/s out | status %d"
              % (os.path.basename(path), len(tab), len(raw) / 1e6, total / 1e6, t1 - t0, (t3 - t2) * 1e3, ctx.timing(_lib.PHZ_T_INFLATE)[0],
                 total / max(1e-9, ctx.timing(_lib.PHZ_T_INFLATE)[0] / 1e3) / 1e9, bad.value), flush=True)
    got = out.cpu().numpy()
    nver = len(tab) if verify_limit is None else min(len(tab), verify_limit)
    step = max(1, len(tab) // nver)
    mism = 0; checked = 0
    for i in range(0, len(tab), step):
        a, b, c, d = tab[i]
        want = zlib.decompress(raw[a:a + b].tobytes(), -15)
        checked += 1
        if len(want) != c or want != got[d:d + c].tobytes():
            mism += 1
            if mism <= 3:
                g = got[d:d + c].tobytes()
                k = next((j for j in range(min(len(want), len(g))) if want[j] != g[j]), -1)
                print("  member %d differs at byte %d of %d" % (i, k, c))
    print("  verified %d members against zlib: %d differ" % (checked, mism), flush=True)
    return mism == 0 and bad.value == 0


ok = True
paths = sys.argv[1:]
if not paths:
    import random
    rng = random.Random(5)
    # three shapes of content: text-like (dynamic codes, long matches), random bytes (stored / near-stored), tiny members
    text = "".join("chr%d\t%d\trs%d\t%s\t%s\t.\tPASS\tAF=%.3f\tGT\t0|1\n" % (rng.randint(1, 22), rng.randint(1, 10 ** 8), rng.randint(1, 10 ** 7), rng.choice("ACGT"), rng.choice("ACGT"), rng.random()) for _ in range(400000))
    vcfout.write_bgzf("/tmp/inf_text.gz", text, 8)
    open("/tmp/inf_rand.bin", "wb").write(os.urandom(3_000_000))
    vcfout.write_bgzf("/tmp/inf_rand.gz", open("/tmp/inf_rand.bin", "rb").read().decode("latin1"), 8) if False else None
    paths = ["/tmp/inf_text.gz"] + (["/tmp/cli_scale.bam"] if os.path.exists("/tmp/cli_scale.bam") else [])
for p in paths:
    ok &= check(p, verify_limit=20000)
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
