"""K_inflate (phaser_amd/csrc/phz_inflate.hip: one lane per BGZF member, hot / cold symbol tables, literals stored eight at a time, LZ77 copies in 16-byte
requests) under the host-side HIP emulation, against zlib: stored, fixed-code and dynamic-code blocks, run-length and far matches, alphabets wider than
the 32 hot symbols, members from one byte to 64 KB, empty members, odd source offsets, several workgroups -- and damaged streams, which must end in a
status code (from the decoder or from the CRC-32 check against the member's trailer, k_crc32), never in a wrong answer that looks right or an access
outside the member.  The GPU runs of the same kernel: tests/test_gpu_bamdev.py,
tools/inflate_check.py (a whole-genome BAM against zlib)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from helpers import EmuContext, emu_library

MEMBER = np.dtype([("src", "<u8"), ("csize", "<u4"), ("isize", "<u4"), ("dst", "<u8")])


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return co.compress(data) + co.flush()


def inflate_device(ctx, streams, sizes, pad_between=0, crcs=None):
    """streams: raw deflate byte strings; -> (status code of the call, output bytes per member)"""
    comp = bytearray(); rec = np.zeros(len(streams), dtype=MEMBER); dst = 0
    for i, (s, n) in enumerate(zip(streams, sizes)):
        comp += b"\xAA" * ((pad_between * (i + 1)) % 7)           # members start at odd offsets, as in a BGZF file (18-byte headers, 8-byte trailers)
        rec[i] = (len(comp), len(s), n, dst)
        comp += s
        dst += n
    comp += bytes(32)                                              # readable past the last member (the API asks for 16 bytes)
    cbuf = np.frombuffer(bytes(comp), dtype=np.uint8).copy()
    out = np.full(max(1, dst), 0xEE, dtype=np.uint8)
    bad = C.c_int(0)
    ctx.check(ctx.lib.phz_bgzf_inflate_device(ctx.h, C.c_void_p(cbuf.ctypes.data), C.c_void_p(rec.ctypes.data), len(streams), C.c_void_p(out.ctypes.data), C.byref(bad)))
    outs = []; at = 0
    for n in sizes:
        outs.append(out[at:at + n].tobytes()); at += n
    if crcs is not None and bad.value == 0:                        # the trailers' CRC32s against what was inflated (k_crc32: what htslib checks per block)
        want = np.asarray(crcs, dtype=np.uint32)
        ctx.check(ctx.lib.phz_bgzf_crc_device(ctx.h, C.c_void_p(out.ctypes.data), C.c_void_p(rec.ctypes.data), len(streams), C.c_void_p(want.ctypes.data), C.byref(bad)))
    return bad.value, outs


def payloads(rng):
    text = b"".join(b"chr%d\t%d\trs%d\tA\tG\t.\tPASS\tAF=0.%03d\tGT\t0|1\n" % (rng.integers(1, 23), rng.integers(1, 10**8), rng.integers(1, 10**7), rng.integers(0, 999)) for _ in range(1500))
    rnd = rng.integers(0, 256, 65280, dtype=np.uint8).tobytes()
    skew = rng.choice(np.arange(256, dtype=np.uint8), size=60000, p=np.r_[np.full(16, 0.03), np.full(240, 0.52 / 240)]).tobytes()      # 256 literals in use: code-order slots far beyond the hot 32
    runs = b"".join(bytes([rng.integers(0, 256)]) * int(rng.integers(1, 600)) for _ in range(300))[:65280]                                  # distance-1 matches of every length
    far = (rnd[:700] + text[:900]) * 40                                                                                                     # matches 1,600 bytes back, up to 258 long
    quals = rng.choice(np.frombuffer(b"FFFFFF:,#AAE/<", dtype=np.uint8), size=65000).tobytes()
    return {"text": text[:65280], "random": rnd, "skewed": skew, "runs": runs, "far": far[:65280], "quals": quals}


def test_members_against_zlib():
    ctx = EmuContext(emu_library())
    rng = np.random.default_rng(5)
    P = payloads(rng)
    streams = []; want = []
    for name, data in P.items():
        for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED),
                                (6, zlib.Z_RLE), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_FILTERED)):
            streams.append(deflate(data, level, strategy)); want.append(data)
        for cut in (1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 300):                 # short members: every tail of the 8- and 16-byte stores
            streams.append(deflate(data[:cut])); want.append(data[:cut])
        streams.append(deflate(data, 6, zlib.Z_DEFAULT_STRATEGY, 1)); want.append(data)        # memLevel 1: many small blocks, a new code every ~1 KB
    streams.append(deflate(b"")); want.append(b"")                                # the BGZF end-of-file member
    streams.append(deflate(bytes(65536), 9)); want.append(bytes(65536))           # the largest member there is
    streams.append(deflate(b"x")); want.append(b"x")
    assert len(streams) > 128                                                     # more than two workgroups
    order = rng.permutation(len(streams))
    streams = [streams[i] for i in order]; want = [want[i] for i in order]
    bad, got = inflate_device(ctx, streams, [len(w) for w in want], pad_between=3, crcs=[zlib.crc32(w) for w in want])
    assert bad == 0
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, "member %d (%d bytes) differs" % (i, len(w))
    # one wrong checksum among them is found
    crcs = [zlib.crc32(w) for w in want]; crcs[len(crcs) // 2] ^= 0x10
    assert inflate_device(ctx, streams, [len(w) for w in want], pad_between=3, crcs=crcs)[0] == 7


def test_damaged_members_end_in_a_status_code():
    ctx = EmuContext(emu_library())
    rng = np.random.default_rng(6)
    P = payloads(rng)
    n_flagged = 0; n_crc = 0
    for name in ("text", "skewed", "far"):
        data = P[name][:20000]
        good = deflate(data)
        for trial in range(12):
            s = bytearray(good)
            kind = trial % 4
            if kind == 0:
                s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))            # one flipped bit
            elif kind == 1:
                s = s[:int(rng.integers(1, len(s)))]                                        # cut short
            elif kind == 2:
                p = int(rng.integers(0, len(s) - 8)); s[p:p + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()
            size = len(data) if kind != 3 else len(data) + int(rng.choice([-1, 1, 100]))    # kind 3: the trailer lies about the size
            bad, got = inflate_device(ctx, [bytes(s), good], [size, len(data)], crcs=[zlib.crc32(data), zlib.crc32(data)])
            assert got[1] == data or bad != 0
            if bad == 0:
                assert got[0] == data and size == len(data), (name, trial)                  # (a flipped padding bit changes nothing)
            else:
                n_flagged += 1
                n_crc += bad == 7
    assert n_flagged >= 24 and n_crc >= 1          # some damage is still valid DEFLATE of the right length: only the checksum finds it
