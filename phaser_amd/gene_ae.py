"""phaser_gene_ae on the GPU -- drop-in for phaser_gene_ae/phaser_gene_ae.py.

Same command line (`--haplotypic_counts --features --o [--id_separator --gw_cutoff --min_cov --min_haplo_maf]`),
same output table.  Pipeline:
  phz_hc_parse   (libphz.so, host threads)  haplotypic_counts.txt -> row / variant / read-label arrays      (:78, :172-204)
  pair finding   (numpy)                    rows x features overlap, the intervaltree query of :101          (half-open overlap)
  phz_gene_counts (HIP, K_genes)            distinct reads per haplotype of every (row, feature) pair         (:172-219)
  aggregation    (numpy)                    phased sums / best unphased block per feature and BAM             (:104-139)
  output         (Python)                   rows formatted with the reference's own expressions               (:147-163)
The per-BAM sections come out in first-appearance order of the BAM names (the reference iterates a Python set).
"""
from __future__ import annotations

import argparse
import ctypes as C
import math
import sys
from typing import Optional

import numpy as np

from . import _lib

ITEM_LABELS = 4096
HEADER = ["contig", "start", "stop", "name", "aCount", "bCount", "totalCount", "log2_aFC", "n_variants", "variants", "gw_phased", "bam"]


def _zero_divide(a, b):
    return float("inf") if b == 0 else float(a) / float(b)


def _zero_log(value, base):
    return float("-inf") if value == 0 else math.log(value, base)


class ParsedCounts:
    """Zero-copy numpy views of a phz_hc handle (freed with the object)."""

    def __init__(self, text: bytes, id_separator: str, threads: int):
        self.lib = _lib.load()
        self.text = text
        self.h = C.c_void_p()
        st = self.lib.phz_hc_parse(C.cast(C.c_char_p(text), C.c_void_p), len(text), id_separator.encode(), threads, C.byref(self.h))
        if st != _lib.PHZ_OK:
            msg = (self.lib.phz_hc_error(self.h) or b"").decode()
            self.lib.phz_hc_free(self.h); self.h = None
            if msg.startswith("ERROR"):
                print(msg)
                raise SystemExit(1)
            raise _lib.PhzError(st, msg or "phz_hc_parse failed")
        v = _lib.phz_hc_arrays()
        self.lib.phz_hc_view(self.h, C.byref(v))
        self.n_rows = v.n_rows; self.n_vars = v.n_vars; self.has_maf = bool(v.has_maf)

        def arr(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,))
        n = self.n_rows
        for k in ("contig", "start", "stop", "a_count", "b_count", "total", "bam", "phase"):
            setattr(self, k, arr(getattr(v, k), n, np.int32))
        self.gw_stat = arr(v.gw_stat, n, np.float64); self.maf = arr(v.maf, n, np.float64)
        self.var_off = arr(v.var_off, n + 1, np.int64); self.var_pos = arr(v.var_pos, v.n_vars, np.int32)
        self.var_id_off = arr(v.var_id_off, v.n_vars, np.int64); self.var_id_len = arr(v.var_id_len, v.n_vars, np.int32)
        self.lab_off = [arr(v.lab_off_a, n + 1, np.int64), arr(v.lab_off_b, n + 1, np.int64)]
        self.n_lab = [int(v.n_lab_a), int(v.n_lab_b)]
        self.lab_pos = [arr(v.lab_pos_a, v.n_lab_a, np.int32), arr(v.lab_pos_b, v.n_lab_b, np.int32)]
        self.lab_prev = [arr(v.lab_prev_a, v.n_lab_a, np.int32), arr(v.lab_prev_b, v.n_lab_b, np.int32)]
        noff = arr(v.names_off, v.n_contigs + v.n_bams + 1, np.int64)
        blob = C.string_at(v.names, int(noff[-1])) if len(noff) else b""
        names = [blob[noff[i]:noff[i + 1] - 1].decode() for i in range(len(noff) - 1)]
        self.contig_names = names[:v.n_contigs]; self.bam_names = names[v.n_contigs:]

    def var_id(self, i: int) -> str:
        o = int(self.var_id_off[i])
        return self.text[o:o + int(self.var_id_len[i])].decode()

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.phz_hc_free(self.h)
            self.h = None


def _expand(counts):
    """[c0, c1, ...] -> (owner index repeated, running index inside each owner)"""
    counts = counts.astype(np.int64)
    total = int(counts.sum())
    owner = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    first = np.cumsum(counts) - counts
    return owner, np.arange(total, dtype=np.int64) - np.repeat(first, counts)


def gene_ae(hc_text: bytes, features_text: str, id_separator: str = "_", gw_cutoff: float = 0.9, min_cov: int = 0,
            min_haplo_maf: float = 0.0, ctx: Optional[_lib.Context] = None, threads: int = 8, stats: Optional[dict] = None,
            _pair_counts=None) -> str:
    """_pair_counts: test hook replacing the K_genes launch (the CPU-only tests check the host stages with it); the product
    path always runs the kernel and raises without a GPU."""
    import time as _t
    if ctx is None and _pair_counts is None:
        ctx = _lib.Context(0)                      # raises without a GPU: there is no CPU path
    t0 = _t.perf_counter()
    # ---- features (:36-55)
    f_chr = []; f_start = []; f_stop = []; f_name = []
    for line in features_text.split("\n"):
        if not line:
            continue
        c = line.rstrip().split("\t")
        if int(c[1]) >= int(c[2]):
            raise ValueError("IntervalTree: Null Interval objects not allowed in IntervalTree")
        f_chr.append(c[0]); f_start.append(int(c[1])); f_stop.append(int(c[2])); f_name.append(c[3])
    nf = len(f_chr)
    f_start_a = np.asarray(f_start, dtype=np.int64); f_stop_a = np.asarray(f_stop, dtype=np.int64)
    P = ParsedCounts(hc_text, id_separator, threads)
    t1 = _t.perf_counter()
    nb = len(P.bam_names)
    # ---- rows x features (:96-103): half-open overlap of [start-1, stop) with [f.start, f.stop)
    pr_row = []; pr_feat = []
    cid = {n: i for i, n in enumerate(P.contig_names)}
    feats_of = {}
    for i, cname in enumerate(f_chr):
        feats_of.setdefault(cname, []).append(i)
    live = P.total > 0
    for cname, flist in feats_of.items():
        if cname not in cid:
            continue
        rows = np.nonzero(live & (P.contig == cid[cname]))[0]
        if not len(rows):
            continue
        fl = np.asarray(flist, dtype=np.int64)
        o = np.argsort(f_start_a[fl], kind="stable")
        fs = f_start_a[fl][o]; fe = f_stop_a[fl][o]; fidx = fl[o]
        pmax = np.maximum.accumulate(fe)
        qa = P.start[rows].astype(np.int64) - 1; qb = P.stop[rows].astype(np.int64)
        hi = np.searchsorted(fs, qb, side="left")
        lo = np.minimum(np.searchsorted(pmax, qa, side="right"), hi)
        owner, k = _expand(hi - lo)
        cand = lo[owner] + k
        ok = fe[cand] > qa[owner]
        pr_row.append(rows[owner[ok]]); pr_feat.append(fidx[cand[ok]])
    pr_row = np.concatenate(pr_row) if pr_row else np.zeros(0, np.int64)
    pr_feat = np.concatenate(pr_feat) if pr_feat else np.zeros(0, np.int64)
    order = np.argsort(pr_row, kind="stable")                      # file order of the rows drives every accumulation
    pr_row = pr_row[order]; pr_feat = pr_feat[order]
    npairs = len(pr_row)
    p_begin = f_start_a[pr_feat].astype(np.int32); p_end = f_stop_a[pr_feat].astype(np.int32)
    nvar = (P.var_off[1:] - P.var_off[:-1])[pr_row]
    # ---- used variants per pair (:183-191)
    owner, k = _expand(nvar)
    vglob = P.var_off[pr_row][owner] + k
    x = P.var_pos[vglob].astype(np.int64) - 1
    inside = (x >= p_begin[owner]) & (x <= p_end[owner])
    u_pair = owner[inside]; u_var = vglob[inside]
    n_used = np.bincount(u_pair, minlength=npairs)
    t2 = _t.perf_counter()
    # ---- distinct reads per pair and haplotype (:193-216): single-variant rows use aCount/bCount, the rest goes to the GPU
    counts = np.zeros((npairs, 2), dtype=np.int64)
    single = nvar == 1
    counts[single, 0] = np.where(n_used[single] > 0, P.a_count[pr_row[single]], 0)
    counts[single, 1] = np.where(n_used[single] > 0, P.b_count[pr_row[single]], 0)
    multi = np.nonzero(~single)[0]
    if len(multi):
        items = {"lo": [], "n": [], "run": [], "pair": [], "hap": []}
        for hap in (0, 1):
            run0 = P.lab_off[hap][pr_row[multi]]; ln = P.lab_off[hap][pr_row[multi] + 1] - run0
            nch = (ln + ITEM_LABELS - 1) // ITEM_LABELS
            owner2, j = _expand(nch)
            lo2 = run0[owner2] + j * ITEM_LABELS
            items["lo"].append(lo2); items["n"].append(np.minimum(ITEM_LABELS, run0[owner2] + ln[owner2] - lo2))
            items["run"].append(run0[owner2]); items["pair"].append(multi[owner2]); items["hap"].append(np.full(len(owner2), hap, dtype=np.uint8))
        it_lo = np.ascontiguousarray(np.concatenate(items["lo"]), dtype=np.int64); it_n = np.ascontiguousarray(np.concatenate(items["n"]), dtype=np.int32)
        it_run = np.ascontiguousarray(np.concatenate(items["run"]), dtype=np.int64)
        it_pair = np.ascontiguousarray(np.concatenate(items["pair"]), dtype=np.int32); it_hap = np.ascontiguousarray(np.concatenate(items["hap"]), dtype=np.uint8)
        w = _lib.phz_gene_work()
        keep = [it_lo, it_n, it_run, it_pair, it_hap, p_begin, p_end]
        vp = lambda a: C.c_void_p(a.ctypes.data) if len(a) else None
        w.n_items = len(it_lo); w.item_lo = vp(it_lo); w.item_n = vp(it_n); w.item_run = vp(it_run); w.item_pair = vp(it_pair); w.item_hap = vp(it_hap)
        w.n_pairs = npairs; w.pair_begin = vp(p_begin); w.pair_end = vp(p_end)
        w.n_lab_a = P.n_lab[0]; w.n_lab_b = P.n_lab[1]
        w.lab_pos_a = vp(P.lab_pos[0]); w.lab_prev_a = vp(P.lab_prev[0]); w.lab_pos_b = vp(P.lab_pos[1]); w.lab_prev_b = vp(P.lab_prev[1])
        gc = np.zeros((npairs, 2), dtype=np.int32)
        if _pair_counts is not None:
            _pair_counts(P, it_lo, it_n, it_run, it_pair, it_hap, p_begin, p_end, gc)
        else:
            ctx.check(ctx.lib.phz_gene_counts(ctx.h, C.byref(w), C.c_void_p(gc.ctypes.data), _lib.PHZ_HOST))
        counts[multi] = gc[multi]
        del keep
        if stats is not None:
            stats["k_genes_ms"] = ctx.timing(_lib.PHZ_T_GENES)[0] if ctx is not None else None
            stats["labels_visited"] = int(it_n.sum()); stats["items"] = len(it_lo)
    if stats is not None:
        stats.update({"rows": int(P.n_rows), "pairs": npairs, "features": nf, "bams": nb})
    t3 = _t.perf_counter()
    # ---- accumulation per (BAM, feature) (:104-139)
    row = pr_row
    key = P.bam[row].astype(np.int64) * nf + pr_feat
    tot = counts[:, 0] + counts[:, 1]
    phased_row = (P.phase[row] != 0) & (P.gw_stat[row] >= gw_cutoff)
    lowmaf = phased_row & (min_haplo_maf > 0) & P.has_maf & (P.maf[row] < min_haplo_maf)
    as_phased = phased_row & ~lowmaf
    ph = P.phase[row]
    add_a = np.where(ph == 1, counts[:, 0], np.where(ph == 2, counts[:, 1], 0)); add_b = np.where(ph == 1, counts[:, 1], np.where(ph == 2, counts[:, 0], 0))
    size = nb * nf
    A = np.bincount(key[as_phased], weights=add_a[as_phased], minlength=size).astype(np.int64)
    B = np.bincount(key[as_phased], weights=add_b[as_phased], minlength=size).astype(np.int64)
    # best unphased block: first pair (row order) with the largest total, only if that total is > 0
    un = np.nonzero(~as_phased & (tot > 0))[0]
    best = np.full(size, -1, dtype=np.int64)
    if len(un):
        o = np.lexsort((un, -tot[un], key[un]))
        ks = key[un][o]
        firsts = np.r_[True, ks[1:] != ks[:-1]]
        best[ks[firsts]] = un[o][firsts]
    UA = np.where(best >= 0, counts[np.maximum(best, 0), 0], 0); UB = np.where(best >= 0, counts[np.maximum(best, 0), 1], 0)
    # variant lists: phased = used variants of every phased pair in row order; unphased = those of the best pair
    up_phased = as_phased[u_pair]
    pk = key[u_pair[up_phased]]
    po = np.argsort(pk, kind="stable")
    pv_sorted = u_var[up_phased][po]; pk_sorted = pk[po]
    pv_lo = np.searchsorted(pk_sorted, np.arange(size), side="left"); pv_hi = np.searchsorted(pk_sorted, np.arange(size), side="right")
    t4 = _t.perf_counter()
    # BAMs in first-appearance order
    if P.n_rows:
        _, first_idx = np.unique(P.bam, return_index=True)
        bam_order = P.bam[np.sort(first_idx)].astype(np.int32)
    else:
        bam_order = np.zeros(0, dtype=np.int32)
    has_best = best >= 0
    best_lo = np.zeros(size, dtype=np.int64); best_hi = np.zeros(size, dtype=np.int64)
    best_lo[has_best] = np.searchsorted(u_pair, best[has_best], side="left"); best_hi[has_best] = np.searchsorted(u_pair, best[has_best], side="right")
    keep = []

    def A_(x, dt):
        a = np.ascontiguousarray(x, dtype=dt); keep.append(a)
        return C.c_void_p(a.ctypes.data)

    def B_(b):
        keep.append(b)
        return C.cast(C.c_char_p(b), C.c_void_p)
    pool = lambda items: ("\n".join(items) + "\n").encode() if items else b""
    R = _lib.phz_gene_rows_in()
    cb = pool(f_chr); nbp = pool(f_name); bnp = pool(P.bam_names)
    R.n_features = nf; R.feat_chr = B_(cb); R.feat_chr_len = len(cb); R.feat_name = B_(nbp); R.feat_name_len = len(nbp)
    R.feat_start = A_(f_start_a, np.int64); R.feat_stop = A_(f_stop_a, np.int64)
    R.n_bam_order = len(bam_order); R.bam_order = A_(bam_order, np.int32); R.bam_names = B_(bnp); R.bam_names_len = len(bnp)
    R.A = A_(A, np.int64); R.B = A_(B, np.int64); R.UA = A_(UA, np.int64); R.UB = A_(UB, np.int64)
    R.pv_lo = A_(pv_lo, np.int64); R.pv_hi = A_(pv_hi, np.int64); R.pv_sorted = A_(pv_sorted, np.int64)
    R.best_lo = A_(best_lo, np.int64); R.best_hi = A_(best_hi, np.int64); R.u_var = A_(u_var, np.int64)
    R.text = B_(P.text); R.var_id_off = A_(P.var_id_off, np.int64); R.var_id_len = A_(P.var_id_len, np.int32)
    R.min_cov = int(min_cov); R.threads = max(1, int(threads))
    optr = C.c_void_p(); olen = C.c_int64(0)
    st = P.lib.phz_gene_rows(C.byref(R), C.byref(optr), C.byref(olen))
    if st != _lib.PHZ_OK:
        raise _lib.PhzError(st, "phz_gene_rows failed")
    try:
        body = C.string_at(optr, olen.value).decode()
    finally:
        P.lib.phz_buf_free(optr)
    out = ["\t".join(HEADER) + "\n", body]
    if stats is not None:
        stats["seconds"] = {"parse": round(t1 - t0, 3), "pairs_and_variants": round(t2 - t1, 3), "counts_incl_copies": round(t3 - t2, 3),
                            "aggregate": round(t4 - t3, 3), "format": round(_t.perf_counter() - t4, 3)}
    return "".join(out)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--haplotypic_counts", required=True); ap.add_argument("--features", required=True); ap.add_argument("--o", required=True)
    ap.add_argument("--id_separator", default="_"); ap.add_argument("--gw_cutoff", type=float, default=0.9)
    ap.add_argument("--min_cov", type=int, default=0); ap.add_argument("--min_haplo_maf", type=float, default=0)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args(argv)
    print(""); print("##################################################")
    print("          Welcome to phASER Gene AE v1.2.0 (phaser_amd, MI355X)")
    print("##################################################"); print("")
    if args.min_haplo_maf < 0 or args.min_haplo_maf > 0.5:
        print("ERROR - invalid value for min_haplo_maf specified. Value must be between 0 and 0.5.")
        return 1
    print("#1 Loading features...")
    feats = open(args.features).read()
    print("#2 Loading haplotype counts...")
    if args.haplotypic_counts.endswith(".gz"):
        import gzip
        text = gzip.open(args.haplotypic_counts, "rb").read()
    else:
        text = open(args.haplotypic_counts, "rb").read()
    print("#3 Processing results...")
    body = gene_ae(text, feats, args.id_separator, args.gw_cutoff, args.min_cov, args.min_haplo_maf, threads=args.threads)
    with open(args.o, "w") as f:
        f.write(body)
    return 0


if __name__ == "__main__":
    sys.exit(main())
