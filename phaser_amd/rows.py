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


def _fill(eng, c: str, keep: list) -> "_lib.phz_rows_in":
    """phz_rows_in of one chromosome: slices / views of the engine's genome-wide arrays (nothing is copied but small vectors)."""
    cfg = eng.cfg
    P = eng._pre[c]; G = eng.G; cv = eng.vs.chroms[c]
    nv = P["nv"]; v0 = P["v0"]; nb = G["nb"]
    pools = cv.pools()

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
    # --haplo_count_blacklist: the loader marked the variants under the BED intervals; Config.haplo_blacklist ("chrom_pos" names,
    # phaser.py:1070) adds to it
    bl = cv.blacklisted if getattr(cv, "blacklisted", None) is not None and len(cv.blacklisted) == nv and cv.blacklisted.any() else None
    if cfg.haplo_blacklist:
        named = np.fromiter((c + "_" + str(int(p)) in cfg.haplo_blacklist for p in cv.pos), dtype=np.uint8, count=nv)
        bl = named if bl is None else (bl | named)
    if bl is not None:
        I.blacklisted = A(bl, np.uint8)
    I.var_count = A(G["var_count"][v0:v0 + nv], np.int32); I.var_distinct = A(G["var_distinct"][v0:v0 + nv], np.int32)
    rs = G["rl_start"][2 * nb * v0: 2 * nb * (v0 + nv) + 1]          # this chromosome's entries; values index the whole rl_qid
    I.rl_start = A(rs, np.uint32); I.rl_qid = A(G["rl_qid"], np.int32)
    # raw mode: tested pairs + labels + first-appearance numbers; the library derives every order itself
    I.raw = 1; I.v0 = v0
    I.n_edges = len(P["ea"])
    I.ea = A(P["ea"], np.int32); I.eb = A(P["eb"], np.int32)
    for k in ("sup", "tot", "cis", "trans", "cfgv"):
        setattr(I, k, A(P[k], np.int32))
    I.pv = A(P["pv"], np.float64); I.keep = A(P["keep"], np.uint8)
    I.rank = A(G["var_rank"][v0:v0 + nv], np.uint64); I.label = A(eng._label_all[v0:v0 + nv], np.int32)
    I.var_first = A(G["var_first"][v0:v0 + nv], np.int64)
    lo_ = np.full(nb, -1, dtype=np.int64); hi_ = np.full(nb, -1, dtype=np.int64)
    for b in range(nb):
        if (c, b) in G["line_base"]:
            base, n = G["line_base"][(c, b)]
            lo_[b] = base; hi_[b] = base + n
    I.bam_line_lo = A(lo_, np.int64); I.bam_line_hi = A(hi_, np.int64)
    names = (C.c_char_p * nb)(*[b.encode() for b in eng.bam_names]); keep.append(names)
    I.nb = nb; I.bam_names = names
    if cfg.haplo_count_bam_exclude:
        ex = np.zeros(nb, dtype=np.uint8)
        for b in cfg.haplo_count_bam_exclude:
            if 0 <= b < nb:
                ex[b] = 1
        I.bam_excluded = A(ex, np.uint8)
    I.unique_ids = int(cfg.unique_ids); I.gw_phase_method = int(cfg.gw_phase_method); I.output_read_ids = int(cfg.output_read_ids)
    I.unphased_vars = int(cfg.unphased_vars); I.max_block_size = int(cfg.max_block_size); I.want_vcf = 1 if (cfg.want_vcf or cfg.py_hash_order) else 0
    I.threads = 1
    if cfg.output_read_ids == 1:
        from .vcf import sep_pool
        qoff, qb = sep_pool(list(eng.qnames.get(c) or []))          # (a chromosome without a read in any BAM has no QNAME table)
        I.qname_off = A(qoff, np.uint32); I.qname = B(qb)
    return I


def _first_bam1(eng, c) -> int:
    """1 + the first BAM whose call lines on chromosome c include a kept one, 0 if there is none: the chromosome's place in the reference's block order (read_vars
    is keyed by the chromosome process_mapping_result returns, "" for a call file without kept lines: phaser.py:1299, :573-574)."""
    G = eng.G; P = eng._pre[c]
    vf = G["var_first"][P["v0"]:P["v0"] + P["nv"]]
    vf = vf[vf >= 0]
    if vf.size == 0:
        return 0
    # the earliest kept line of every BAM: a variant's var_first is its first kept line over all BAMs, so walk the BAMs and ask whether any variant starts there
    for b in range(G["nb"]):
        if (c, b) in G["line_base"]:
            base, n = G["line_base"][(c, b)]
            if ((vf >= base) & (vf < base + n)).any():
                return b + 1
    return 0


def format_chroms(eng, chroms, threads: int) -> Dict[str, Dict]:
    """-> per chromosome the fragment fields produced by stage C2 (bytes row text, counts, write_vcf arrays).  All chromosomes go
    through one native call (phz_rows_format_multi): their block / row chunks share one pool of `threads` workers."""
    import time as _t
    t0 = _t.perf_counter()
    lib = _lib.load()
    cfg = eng.cfg
    n = len(chroms)
    if n == 0:
        return {}
    keep = []                      # keeps every buffer alive across the call
    IN = (_lib.phz_rows_in * n)(*[_fill(eng, c, keep) for c in chroms])
    OUT = (_lib.phz_rows_out * n)()
    t1 = _t.perf_counter()
    st = lib.phz_rows_format_multi(IN, n, OUT, max(1, int(threads)))
    eng.stats["rows_glue_s"] = eng.stats.get("rows_glue_s", 0.0) + t1 - t0
    eng.stats["rows_native_s"] = eng.stats.get("rows_native_s", 0.0) + _t.perf_counter() - t1
    if st != _lib.PHZ_OK:
        raise _lib.PhzError(st, "phz_rows_format_multi: %s" % lib.phz_strerror(st).decode())
    res = {}
    for i, c in enumerate(chroms):
        O = OUT[i]
        owner = _NativeRows(lib, O)

        def vec(name, dt, cnt, O=O):
            if cnt == 0:
                return np.zeros(0, dtype=dt)
            return np.frombuffer(C.string_at(getattr(O, name), cnt * np.dtype(dt).itemsize), dtype=dt).copy()
        out = {"allelic_rows": int(O.allelic_rows), "n_blocks": int(O.n_blocks), "phased": int(O.phased), "vcf": None, "first_bam1": _first_bam1(eng, c)}
        for name in ("conn", "hap", "ase", "cfg"):
            out[name] = owner.parts(name)[0]
        for name in ("allelic", "single_ase", "single_hap"):
            out[name], out[name + "_bam"] = owner.parts(name)
        if cfg.want_vcf or cfg.py_hash_order:
            nbk = int(O.n_blocks); nvv = int(O.n_blk_vars)
            out["vcf"] = {"size": vec("blk_size", np.int32, nbk), "var": vec("blk_var", np.int32, nvv), "hap": vec("blk_hap", np.uint8, nvv),
                          "cor": vec("blk_cor", np.int8, 2 * nvv), "stat": vec("blk_stat", np.float64, nbk),
                          "stat_int": vec("blk_stat_int", np.uint8, nbk), "maxmaf": vec("blk_maxmaf", np.int32, nbk)}
        res[c] = out
    return res


class _NativeRows:
    """Owns one phz_rows_out: hands out zero-copy views of its text buffers and frees them when the last view dies."""

    def __init__(self, lib, O):
        self.lib = lib
        self.O = _lib.phz_rows_out()
        C.memmove(C.byref(self.O), C.byref(O), C.sizeof(_lib.phz_rows_out))      # own copy of the descriptor (O lives in an array)

    def parts(self, name):
        """-> (list of memoryviews over the native chunk buffers in output order, list of their BAM keys or None)"""
        P = getattr(self.O, name)
        views = [memoryview(_lib.native_view(P.ptr[i], int(P.len[i]), C.c_uint8, self)) for i in range(int(P.n))]
        bams = [int(P.bam[i]) for i in range(int(P.n))] if P.bam else None
        return views, bams

    def __del__(self):
        try:
            self.lib.phz_rows_free(C.byref(self.O))
        except Exception:
            pass
