"""Glue for the native host stage C2 (phz_rows_format in libphz.so): block phasing + the text rows of the five
output files for one chromosome (phaser/phaser.py:2107-2324 phase_v3, :865-1239 output loops, :691-695, :737-749).

The engine hands over plain arrays (K_tally results, tested pairs, components, first-appearance keys); the
library returns the chromosome's row text per file, already in the reference's order, plus per-block arrays
for write_vcf.  Everything is bytes until a file (or a test) asks for text.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np

from . import _lib


def _arr(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


def _vp(a):
    return C.c_void_p(a.ctypes.data) if isinstance(a, np.ndarray) else C.cast(C.c_char_p(a), C.c_void_p)


def format_chrom(eng, c: str, threads: int) -> Dict:
    """-> the chromosome's fragment fields produced by stage C2 (bytes row text, counts, write_vcf arrays)."""
    import time as _t
    t0 = _t.perf_counter()
    lib = _lib.load()
    cfg = eng.cfg
    P = eng._pre[c]; R = eng.tally[c]; cv = eng.vs.chroms[c]
    nv = R["nv"]
    pools = cv.pools()
    keep = []                      # keeps every buffer alive across the call

    def A(x, dt):
        a = _arr(x, dt); keep.append(a); return _vp(a)

    def B(b):
        keep.append(b); return _vp(b)

    I = _lib.phz_rows_in()
    I.chrom = c.encode(); I.nv = nv
    I.pos = A(cv.pos, np.int32)
    I.uid_off = A(pools["uid"][0], np.uint32); I.uid = B(pools["uid"][1])
    I.rsid_off = A(pools["rsid"][0], np.uint32); I.rsid = B(pools["rsid"][1])
    I.allele_off = A(pools["allele"][0], np.uint32); I.allele = B(pools["allele"][1])
    I.maf_off = A(pools["maf"][0], np.uint32); I.maf_txt = B(pools["maf"][1])
    I.maf = A(pools["maf_val"], np.float64)
    I.is_ref = A(cv.is_ref, np.uint8); I.phase_idx = A(cv.phase_idx, np.int8)
    if cfg.haplo_blacklist:
        bl = np.fromiter((c + "_" + str(int(p)) in cfg.haplo_blacklist for p in cv.pos), dtype=np.uint8, count=nv)
        I.blacklisted = A(bl, np.uint8)
    I.var_count = A(R["var_count"], np.int32); I.var_distinct = A(R["var_distinct"], np.int32)
    I.n_lines = len(R["line_cls"])
    I.line_var = A(R["line_var"], np.int32); I.line_qid = A(R["line_qid"], np.int32); I.line_bam = A(R["line_bam"], np.int32)
    I.line_cls = A(R["line_cls"], np.uint8)
    I.n_edges = len(P["eorder"])
    I.va = A(P["va"], np.int32); I.vb = A(P["vb"], np.int32); I.ea = A(P["ea"], np.int32); I.eb = A(P["eb"], np.int32)
    for k in ("sup", "tot", "cis", "trans", "cfgv", "eorder"):
        setattr(I, k, A(P[k], np.int64))
    I.pv = A(P["pv"], np.float64)
    I.ncomp = P["ncomp"]
    if P["ncomp"]:
        I.mem_s = A(P["mem_s"], np.int32)
        for k, src in (("comp_starts", "starts"), ("comp_ends", "ends"), ("comp_order", "comp_order"), ("e_keep", "e_keep"), ("eo", "eo"),
                       ("e_starts", "e_starts"), ("e_ends", "e_ends")):
            setattr(I, k, A(P[src], np.int64))
    I.n_keys = len(P["key_g"])
    I.key_bam = A(P["key_bam"], np.int64); I.key_g = A(P["key_g"], np.int64)
    nb = len(eng.bam_names)
    names = (C.c_char_p * nb)(*[b.encode() for b in eng.bam_names]); keep.append(names)
    I.nb = nb; I.bam_names = names
    if cfg.haplo_count_bam_exclude:
        ex = np.zeros(nb, dtype=np.uint8)
        for b in cfg.haplo_count_bam_exclude:
            if 0 <= b < nb:
                ex[b] = 1
        I.bam_excluded = A(ex, np.uint8)
    I.unique_ids = int(cfg.unique_ids); I.gw_phase_method = int(cfg.gw_phase_method); I.output_read_ids = int(cfg.output_read_ids)
    I.unphased_vars = int(cfg.unphased_vars); I.max_block_size = int(cfg.max_block_size); I.want_vcf = 1 if cfg.want_vcf else 0
    I.threads = max(1, int(threads))
    if cfg.output_read_ids == 1:
        from .vcf import sep_pool
        qoff, qb = sep_pool(list(eng.qnames[c]))
        I.qname_off = A(qoff, np.uint32); I.qname = B(qb)
    O = _lib.phz_rows_out()
    t1 = _t.perf_counter()
    st = lib.phz_rows_format(C.byref(I), C.byref(O))
    eng.stats["rows_glue_s"] = eng.stats.get("rows_glue_s", 0.0) + t1 - t0
    eng.stats["rows_native_s"] = eng.stats.get("rows_native_s", 0.0) + _t.perf_counter() - t1
    if st != _lib.PHZ_OK:
        raise _lib.PhzError(st, "phz_rows_format(%s): %s" % (c, lib.phz_strerror(st).decode()))
    owner = _NativeRows(lib, O)

    def seg(name):
        p = getattr(O, name)
        return [int(p[i]) for i in range(nb + 1)]

    def vec(name, dt, n):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.frombuffer(C.string_at(getattr(O, name), n * np.dtype(dt).itemsize), dtype=dt).copy()
    out = {"conn": owner.text("conn"), "hap": owner.text("hap"), "ase": owner.text("ase"), "cfg": owner.text("cfg"),
           "allelic": owner.text("allelic"), "allelic_seg": seg("allelic_seg"), "allelic_rows": int(O.allelic_rows),
           "single_ase": owner.text("single_ase"), "single_ase_seg": seg("single_ase_seg"),
           "single_hap": owner.text("single_hap"), "single_hap_seg": seg("single_hap_seg"),
           "n_blocks": int(O.n_blocks), "phased": int(O.phased), "vcf": None}
    if cfg.want_vcf:
        nbk = int(O.n_blocks); nvv = int(O.n_blk_vars)
        out["vcf"] = {"size": vec("blk_size", np.int32, nbk), "var": vec("blk_var", np.int32, nvv), "hap": vec("blk_hap", np.uint8, nvv),
                      "cor": vec("blk_cor", np.int8, 2 * nvv), "stat": vec("blk_stat", np.float64, nbk),
                      "stat_int": vec("blk_stat_int", np.uint8, nbk), "maxmaf": vec("blk_maxmaf", np.int32, nbk)}
    return out


class _NativeRows:
    """Owns one phz_rows_out: hands out zero-copy views of its text buffers and frees them when the last view dies."""

    def __init__(self, lib, O):
        self.lib = lib; self.O = O

    def text(self, name):
        n = getattr(self.O, name + "_len")
        if not n:
            return b""
        return memoryview(_lib.native_view(getattr(self.O, name), n, C.c_uint8, self))

    def __del__(self):
        try:
            self.lib.phz_rows_free(C.byref(self.O))
        except Exception:
            pass
