"""Raw-byte tier (SURVEY.md 8(a) "T1"): rows and read labels in the order CPython gives the reference's `set` objects.

Three of phASER's five files depend on the iteration order of Python sets of strings (phaser/phaser.py):
  variant_connections.txt   row order and the orientation of every pair: `for variant_b in dict_variant_overlap[chr][variant_a]` (:667-678), sets built
                            from the per-read adjacency lists of generate_connectivity_map (:1265-1285, :660-662)
  haplotypic_counts.txt     aReads / bReads are indices into `list(set(reads))` of QNAME strings (:1086, :1106-1115), `variantsBlacklisted` is a set
                            (:1059), the singleton rows follow `set(dict_variant_reads.keys()) - set(all_variants)` (:1181-1224)
  haplotypes.txt            the singleton rows (:1226-1239), same set
Everything else in the files is order-free and comes from the GPU path unchanged.  This module re-orders / re-labels the product's rows by REPLAYING those
set constructions -- the same strings inserted in the same sequence into real `set` objects of this interpreter -- which gives the reference's bytes when
the interpreter hashes strings the way the reference's run did: CPython 3.10 with PYTHONHASHSEED=0 (the golden files were written that way).

Two implementations of the same replay:
  replay_native   (the default behind Config.py_hash_order / --py_hash_order 1) libphz's phz_pyorder_replay: the str hash of a seed-0 CPython 3.10 and
                  its set (probe sequence, growth, difference) restated in C++ (csrc/phz_pyorder.cpp) -- seconds at whole-genome scale, and independent
                  of the interpreter it runs under (any Python version, any hash seed);
  replay          the pure-Python twin with REAL set objects (minutes at that scale; needs CPython 3.10 under PYTHONHASHSEED=0): what the native one is
                  tested against, and selectable with PHZ_PYORDER_PYTHON=1.
"""
from __future__ import annotations

import ctypes as C
import sys
from collections import OrderedDict
from typing import Dict, List

import numpy as np


def check_interpreter():
    if sys.flags.hash_randomization != 0:
        raise SystemExit("     FATAL ERROR: --py_hash_order 1 reproduces the reference's set order only when Python hashes strings deterministically; "
                         "run with PYTHONHASHSEED=0")


def replay(eng, out: Dict[str, str]) -> Dict[str, str]:
    """eng: an Engine after finish() (host arrays of its kept call lines through eng.kept_lines(), QNAME strings in eng.qnames, per-block arrays in
    eng.vcf_blocks); out: its five files as text.  -> the five files in the reference's raw order."""
    check_interpreter()
    cfg = eng.cfg
    nb = len(eng.bam_names)
    chroms = [c for c in eng.all_chroms]
    uid = {c: eng.vs.chroms[c].uid for c in chroms}
    alleles = {c: eng.vs.chroms[c].alleles for c in chroms}
    lines = eng.kept_lines()                 # {chrom: [per BAM: (qid int array, variant int array, class uint8 array) of the kept lines, line order] or None}
    # ---- dict_variant_reads order (rule 2), read_vars (rule 3 + the overwrite of :576-581), haplo_reads[(uid, allele)][bam]
    dvr: "OrderedDict[str, str]" = OrderedDict()           # uid -> chromosome
    read_vars: "OrderedDict[str, OrderedDict[str, list]]" = OrderedDict()
    haplo_reads: Dict[tuple, Dict[int, list]] = {}
    for b in range(nb):
        per_file = []
        for c in chroms:
            L = lines.get(c)
            if L is None or L[b] is None:
                continue
            qid, var, cls = L[b]
            if len(qid) == 0:
                continue                        # a call file without a kept line returns chromosome "" (phaser.py:1299): it does not place this chromosome in read_vars
            names = eng.qnames[c]
            u = uid[c]
            rv: "OrderedDict[str, list]" = OrderedDict()
            for q, v, k in zip(qid.tolist(), var.tolist(), cls.tolist()):
                x = u[v]
                if x not in dvr:
                    dvr[x] = c
                if k < 2:
                    nm = names[q]
                    lst = rv.get(nm)
                    if lst is None:
                        rv[nm] = lst = []
                    lst.append(x)
                    if not (b in cfg.haplo_count_bam_exclude):
                        haplo_reads.setdefault((x, k), {}).setdefault(b, []).append(nm)
            per_file.append((c, rv))
        for c, rv in per_file:
            if c not in read_vars:
                read_vars[c] = OrderedDict()
        for c, rv in per_file:
            tgt = read_vars[c]
            for nm, lst in rv.items():
                tgt[nm] = lst                       # a later BAM replaces the list of a QNAME already seen (:576-581)
    # ---- connectivity map -> sets -> order of the tested pairs (:1265-1285, :660-678)
    pair_order: List[tuple] = []
    tested = set()
    overlap: "OrderedDict[str, OrderedDict[str, object]]" = OrderedDict()
    for c, rv in read_vars.items():
        ov: "OrderedDict[str, list]" = OrderedDict()
        for nm, lst in rv.items():
            for v in lst:
                for o in lst:
                    if o != v:
                        tgt = ov.get(v)
                        if tgt is None:
                            ov[v] = tgt = []
                        tgt.append(o)
        if ov:
            overlap[c] = ov
    for c in overlap:
        for v in overlap[c]:
            overlap[c][v] = set(overlap[c][v])
    for c in overlap:
        for a in overlap[c]:
            for b2 in overlap[c][a]:
                key1 = a + "|" + b2; key2 = b2 + "|" + a
                if key1 not in tested and key2 not in tested:
                    pair_order.append((a, b2))
                    tested.add(key1)
    res = dict(out)
    # ---- variant_connections.txt
    rows = out["variant_connections"].split("\n")
    head, body = rows[0], [r for r in rows[1:] if r]
    by_pair = {}
    for r in body:
        f = r.split("\t")
        by_pair[(f[0], f[1])] = f
    new = [head]
    for a, b2 in pair_order:
        f = by_pair.get((a, b2))
        if f is None:
            f = by_pair[(b2, a)]
            f = [a, b2] + f[2:]
        new.append("\t".join(f))
    assert len(new) - 1 == len(body), "variant_connections: replayed pairs do not match the tested pairs"
    res["variant_connections"] = "\n".join(new) + "\n"
    # ---- blocks in block order (all variants incl. blacklisted), singletons
    blocks = []                 # (chrom, [uids])
    for c, v, _first in eng.vcf_blocks:
        off = 0
        u = uid[c]
        for n in v["size"].tolist():
            blocks.append((c, [u[i] for i in v["var"][off:off + n].tolist()]))
            off += n
    all_variants = [x for _, vs_ in blocks for x in vs_]
    singletons = list(set(dvr.keys()) - set(all_variants))
    black = set()
    for c in chroms:
        cv = eng.vs.chroms[c]
        bl = getattr(cv, "blacklisted", None)
        for i, x in enumerate(uid[c]):
            if (bl is not None and len(bl) == len(cv) and bl[i]) or (c + "_" + str(int(cv.pos[i])) in cfg.haplo_blacklist):
                black.add(x)
    # ---- haplotypic_counts.txt
    rows = out["haplotypic_counts"].split("\n")
    head, body = rows[0], [r.split("\t") for r in rows[1:] if r]
    # block rows come first, in block order, one per BAM that is not excluded and has coverage; singleton rows follow
    k = 0
    new = [head]
    uid_alleles = {}
    for c in chroms:
        for x, al in zip(uid[c], alleles[c]):
            uid_alleles[x] = al
    for c, vs_ in blocks:
        used = [x for x in vs_ if x not in black]
        used_s = ",".join(used)
        bset = set()
        for _h in (0, 1):
            for x in vs_:
                if x in black:
                    bset.add(x)
        bl_s = ",".join(str(x) for x in bset)
        for b in range(nb):
            if b in cfg.haplo_count_bam_exclude:
                continue
            if k >= len(body):
                break
            f = body[k]
            if not (f[0] == c and f[3] == used_s and f[-3] == eng.bam_names[b] and int(f[4]) == len(used)):
                continue                      # this BAM had no coverage of the block: the reference wrote no row (:1118)
            k += 1
            hapA = f[7].split(",") if f[7] != "" else []
            hapB = f[8].split(",") if f[8] != "" else []
            labels = []; id_lists = []
            for h, hx in enumerate((hapA, hapB)):
                set_reads = []; var_reads = []
                for x, allele in zip(used, hx):
                    ai = uid_alleles[x].index(allele)
                    lst = haplo_reads.get((x, ai), {}).get(b, None)
                    if lst is not None:
                        var_reads.append(lst); set_reads += lst
                    else:
                        var_reads.append([])
                order = list(set(set_reads))
                id_lists.append(",".join(order))
                index = {nm: i for i, nm in enumerate(order)}
                labels.append(";".join(",".join(str(index[nm]) for nm in lst) for lst in var_reads))
            f = list(f)
            f[5] = bl_s; f[-2] = labels[0]; f[-1] = labels[1]
            if cfg.output_read_ids == 1:          # the QNAME lists are written in the order of the same sets (:1120-1123)
                f[14] = id_lists[0]; f[15] = id_lists[1]
            new.append("\t".join(f))
    single = body[k:]
    by_var: Dict[str, list] = {}
    bam_index = {nm: i for i, nm in enumerate(eng.bam_names)}
    for f in single:
        if cfg.output_read_ids == 1:              # singleton rows list set(haplo_reads[allele][bam]) (:1196-1204, :1219-1220)
            b = bam_index[f[-3]]
            for k2, col in ((0, 14), (1, 15)):
                lst = haplo_reads.get((f[3], k2), {}).get(b)
                f[col] = ",".join(set(lst)) if lst is not None else ""
        by_var.setdefault(f[3], []).append("\t".join(f))
    n_single = 0
    for x in singletons:
        for r in by_var.get(x, ()):
            new.append(r); n_single += 1
    assert n_single == len(single), "haplotypic_counts: singleton rows do not match the replayed singletons"
    res["haplotypic_counts"] = "\n".join(new) + "\n"
    # ---- haplotypes.txt: block rows stay, singleton rows in the order of the set
    rows = out["haplotypes"].split("\n")
    head, body = rows[0], [r for r in rows[1:] if r]
    nblk = len(blocks)
    blk_rows, single = body[:nblk], body[nblk:]
    names = {}
    for c in chroms:
        cv = eng.vs.chroms[c]
        for i, x in enumerate(uid[c]):
            names[x] = (c, str(int(cv.pos[i])), x if cfg.unique_ids else cv.rsid[i])
    by_key: Dict[tuple, list] = {}
    for r in single:
        f = r.split("\t")
        by_key.setdefault((f[0], f[2], f[5]), []).append(r)
    new = [head] + blk_rows
    n_single = 0
    if cfg.unphased_vars == 1:
        for x in singletons:
            lst = by_key.get(names[x])
            if lst:
                new.append(lst.pop(0)); n_single += 1
    assert n_single == len(single), "haplotypes: singleton rows do not match the replayed singletons"
    res["haplotypes"] = "\n".join(new) + "\n"
    return res



def _names_pool(names):
    """QNAME strings of one chromosome in id order -> (blob bytes, offsets uint32 [n + 1]); `names` is a list of str or already such a pair"""
    if isinstance(names, tuple):
        return names
    n = len(names)
    off = np.zeros(n + 1, dtype=np.uint32)
    if n:
        blob = "".join(names).encode("latin-1")
        off[1:] = np.cumsum(np.fromiter((len(x) for x in names), dtype=np.int64, count=n)).astype(np.uint32)
    else:
        blob = b""
    return blob, off


def replay_native(eng, out: Dict[str, bytes]) -> Dict[str, bytes]:
    """Same contract as replay(), on bytes, through libphz (phz_pyorder_replay)."""
    from . import _lib
    lib = _lib.load()
    cfg = eng.cfg
    chroms = list(eng.all_chroms)
    nb = len(eng.bam_names); nc = len(chroms)
    lines = eng.kept_lines()
    keep = []           # everything the struct points into

    def arr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt); keep.append(a)
        return a.ctypes.data if a.size else 0

    def blob(b):
        b = bytes(b) if not isinstance(b, bytes) else b
        if not b:
            b = b"\0"
        keep.append(b)
        return C.cast(C.c_char_p(b), C.c_void_p).value
    VP = C.c_void_p * max(1, nc)
    uid = VP(); uid_off = VP(); al = VP(); al_off = VP(); rs = VP(); rs_off = VP(); qn = VP(); qn_off = VP(); pos = VP(); black = VP()
    nv = np.zeros(max(1, nc), np.int64); nq = np.zeros(max(1, nc), np.int64)
    for i, c in enumerate(chroms):
        cv = eng.vs.chroms[c]
        P = cv.pools()
        nv[i] = len(cv)
        uid[i] = blob(P["uid"][1]); uid_off[i] = arr(P["uid"][0], np.uint32)
        al[i] = blob(P["allele"][1]); al_off[i] = arr(P["allele"][0], np.uint32)
        rs[i] = blob(P["rsid"][1]); rs_off[i] = arr(P["rsid"][0], np.uint32)
        pos[i] = arr(cv.pos, np.int32)
        b_, o_ = _names_pool(eng.qnames.get(c, []))
        nq[i] = len(o_) - 1
        qn[i] = blob(b_); qn_off[i] = arr(o_, np.uint32)
        bl = getattr(cv, "blacklisted", None)
        m = np.array(bl, dtype=np.uint8) if (bl is not None and len(bl) == len(cv)) else np.zeros(len(cv), np.uint8)
        if cfg.haplo_blacklist:
            m = m | np.fromiter((c + "_" + str(int(p)) in cfg.haplo_blacklist for p in cv.pos), dtype=np.uint8, count=len(cv))
        black[i] = arr(m, np.uint8) if m.any() else 0
    LP = C.c_void_p * max(1, nc * nb)
    lq = LP(); lv = LP(); lc = LP(); nl = np.zeros(max(1, nc * nb), np.int64)
    for i, c in enumerate(chroms):
        L = lines.get(c)
        for b in range(nb):
            t = None if L is None else L[b]
            if t is None:
                continue
            lq[i * nb + b] = arr(t[0], np.int32) or arr(np.zeros(1, np.int32), np.int32); lv[i * nb + b] = arr(t[1], np.int32) or arr(np.zeros(1, np.int32), np.int32)
            lc[i * nb + b] = arr(t[2], np.uint8) or arr(np.zeros(1, np.uint8), np.uint8); nl[i * nb + b] = len(t[0])
    index = {c: i for i, c in enumerate(chroms)}
    bc = []; bo = [0]; bv = []
    for c, v, _first in eng.vcf_blocks:
        off = 0
        var = np.asarray(v["var"])
        for n in np.asarray(v["size"]).tolist():
            bc.append(index[c]); bv.append(var[off:off + n]); off += n; bo.append(bo[-1] + n)
    ex = np.zeros(nb, np.uint8)
    for b in cfg.haplo_count_bam_exclude:
        if 0 <= b < nb:
            ex[b] = 1
    cn = (C.c_char_p * max(1, nc))(*[c.encode() for c in chroms]); bn = (C.c_char_p * max(1, nb))(*[x.encode() for x in eng.bam_names])
    CP = lambda a: C.cast(a, C.POINTER(C.c_void_p))
    I = _lib.phz_pyorder_in(nc, nb, cn, bn, arr(nv, np.int64), CP(pos), CP(uid), CP(uid_off), CP(al), CP(al_off), CP(rs), CP(rs_off), arr(nq, np.int64), CP(qn), CP(qn_off),
                            CP(lq), CP(lv), CP(lc), arr(nl, np.int64), arr(ex, np.uint8) if ex.any() else None, CP(black), len(bc), arr(np.array(bc, np.int32), np.int32),
                            arr(np.array(bo, np.int64), np.int64), arr(np.concatenate(bv) if bv else np.zeros(0, np.int32), np.int32),
                            int(cfg.output_read_ids), int(cfg.unphased_vars), int(cfg.unique_ids))
    texts = [out["variant_connections"], out["haplotypes"], out["haplotypic_counts"]]
    texts = [t if isinstance(t, bytes) else t.encode() for t in texts]
    h = C.c_void_p()
    st = lib.phz_pyorder_replay(C.byref(I), C.cast(C.c_char_p(texts[0]), C.c_void_p), len(texts[0]), C.cast(C.c_char_p(texts[1]), C.c_void_p), len(texts[1]),
                                C.cast(C.c_char_p(texts[2]), C.c_void_p), len(texts[2]), C.byref(h))
    try:
        if st != _lib.PHZ_OK:
            raise _lib.PhzError(st, "raw-byte tier: " + (lib.phz_pyorder_error(h) or b"").decode())
        res = dict(out)
        for which, name in enumerate(("variant_connections", "haplotypes", "haplotypic_counts")):
            p = C.c_void_p(); n = C.c_int64(0)
            lib.phz_pyorder_text(h, which, C.byref(p), C.byref(n))
            res[name] = C.string_at(p, n.value)
        return res
    finally:
        lib.phz_pyorder_free(h)
