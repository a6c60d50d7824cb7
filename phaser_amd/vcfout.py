"""Phased VCF output -- phaser/phaser.py:1661-1855 (`write_vcf`), SURVEY.md 8(f) next-2.

The text is produced by the native writer (phz_vcf_phase_text, phaser_amd/csrc/phz_vcfout.cpp): input is the sample's VCF
(the reference feeds `gunzip -c | cut -f 1-9,S`; here the original text + the sample's column) and, per chromosome, the
block arrays the row writer returned (engine.vcf_blocks) with the variant table's string pools.  Output text is what the
reference writes to <o>.vcf before compressing it; we compress it as BGZF ourselves (phz_bgzf_write) and write the tabix
index with phz_tabix_build.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib


def phased_vcf_text(vcf_text, sample_column: int, eng, id_separator: str = "_", chromosome_of_interest: str = "",
                    gw_phase_vcf: int = 0, min_confidence: float = 0.9, threads: int = 8, as_bytes: bool = False) -> Tuple[str, int, int]:
    """-> (vcf text, unphased_phased, phase_corrections).  eng: the Engine whose finish() ran with want_vcf (its vcf_blocks).
    as_bytes: return the text as bytes (what write_bgzf takes as it is: no decode / encode round trip over ~100 MB)."""
    import os, sys, time
    t0 = time.perf_counter()
    lib = _lib.load()
    from .vcf import text_ptr
    data_p, data_n, data = text_ptr(vcf_text)
    keep = []
    arr = (_lib.phz_vcfout_chrom * max(1, len(eng.vcf_blocks)))()

    def P(b):
        keep.append(b)
        return C.cast(C.c_char_p(b), C.c_void_p)

    def A(a, dt):
        a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
        return C.c_void_p(a.ctypes.data)
    for k, (c, v, first) in enumerate(eng.vcf_blocks):
        cv = eng.vs.chroms[c]; raw = cv._raw
        x = arr[k]
        x.uid = P(raw["uid"]); x.uid_len = len(raw["uid"]); x.rsid = P(raw["rsid"]); x.rsid_len = len(raw["rsid"])
        x.alleles = P(raw["alleles"]); x.alleles_len = len(raw["alleles"]); x.maf_str = P(raw["maf_str"]); x.maf_str_len = len(raw["maf_str"])
        x.n_blocks = len(v["size"]); x.n_blk_vars = len(v["var"]); x.first_block_index = first
        x.blk_size = A(v["size"], np.int32); x.blk_var = A(v["var"], np.int32); x.blk_maxmaf = A(v["maxmaf"], np.int32)
        x.blk_hap = A(v["hap"], np.uint8); x.blk_stat_int = A(v["stat_int"], np.uint8); x.blk_cor = A(v["cor"], np.int8)
        x.blk_stat = A(v["stat"], np.float64)
    out = C.c_void_p(); n = C.c_int64(0); up = C.c_int64(0); pc = C.c_int64(0)
    t1 = time.perf_counter()
    st = lib.phz_vcf_phase_text(data_p, data_n, int(sample_column), id_separator.encode(),
                                chromosome_of_interest.encode(), int(gw_phase_vcf), float(min_confidence), arr, len(eng.vcf_blocks),
                                max(1, int(threads)), C.byref(out), C.byref(n), C.byref(up), C.byref(pc))
    if st != _lib.PHZ_OK:
        raise _lib.PhzError(st, "phz_vcf_phase_text failed (malformed VCF line?)")
    t2 = time.perf_counter()
    if as_bytes:            # the native buffer itself (a uint8 array that frees it when it dies): no copy of ~100 MB
        class _Owner:
            def __init__(self, lib, ptr):
                self.lib = lib; self.ptr = ptr

            def __del__(self):
                try:
                    self.lib.phz_buf_free(self.ptr)
                except Exception:
                    pass
        raw = _lib.native_view(out.value, n.value, C.c_uint8, _Owner(lib, out))
        if os.environ.get("PHZ_TIMING"):
            sys.stderr.write("[phz timing]     vcf out (python): inputs %.3f s, native %.3f s, copy out %.3f s\n" % (t1 - t0, t2 - t1, time.perf_counter() - t2))
        return raw, int(up.value), int(pc.value)
    try:
        raw = C.string_at(out, n.value)
        if os.environ.get("PHZ_TIMING"):
            sys.stderr.write("[phz timing]     vcf out (python): inputs %.3f s, native %.3f s, copy out %.3f s\n" % (t1 - t0, t2 - t1, time.perf_counter() - t2))
        return raw.decode(), int(up.value), int(pc.value)
    finally:
        lib.phz_buf_free(out)


def write_bgzf(path: str, text, threads: int = 0, index: str = None) -> bool:
    """BGZF-compress the phased VCF text (what the reference gets from `bgzip`, phaser.py:1851) with the native parallel writer.
    index: "vcf" / "bed" also writes <path>.tbi (`tabix -p ... -f`) from the text in memory while it is being compressed; returns
    False when that index was refused (text not position-sorted: the .gz is written, tabix refuses such files as well)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    import numpy as np
    if isinstance(text, np.ndarray):            # what phased_vcf_text(as_bytes=True) returns: compressed where it lies
        data = np.ascontiguousarray(text, dtype=np.uint8)
        ptr, n = C.c_void_p(data.ctypes.data), int(data.size)
    else:
        data = text.encode() if isinstance(text, str) else bytes(text)
        ptr, n = C.cast(C.c_char_p(data), C.c_void_p), len(data)
    if index is not None:
        st = lib.phz_bgzf_write_indexed(path.encode(), ptr, n, int(threads), 6, {"vcf": 0, "bed": 1}[index])
        if st == _lib.PHZ_E_UNSUPPORTED:
            return False
    else:
        st = lib.phz_bgzf_write(path.encode(), ptr, n, int(threads), 6)
    if st != _lib.PHZ_OK:
        raise _lib.PhzError(st, "phz_bgzf_write(%s) failed" % path)
    return True


def tabix_index(path: str, preset: str = "vcf", threads: int = 0) -> bool:
    """Write <path>.tbi (what `tabix -p vcf|bed -f <path>` does, phaser.py:1851).  Returns False when the file is not position-sorted
    (tabix refuses those as well)."""
    lib = _lib.load()
    st = lib.phz_tabix_build(path.encode(), {"vcf": 0, "bed": 1}[preset], int(threads))
    if st == _lib.PHZ_E_UNSUPPORTED:
        return False
    if st != _lib.PHZ_OK:
        raise _lib.PhzError(st, "phz_tabix_build(%s) failed" % path)
    return True
