"""Raw-byte tier (SURVEY.md 8(a) "T1"): rows and read labels in the order CPython gives the reference's `set` objects.

Three of phASER's five files depend on the iteration order of Python sets of strings (phaser/phaser.py):
  variant_connections.txt   row order and the orientation of every pair: `for variant_b in dict_variant_overlap[chr][variant_a]` (:667-678), sets built
                            from the per-read adjacency lists of generate_connectivity_map (:1265-1285, :660-662)
  haplotypic_counts.txt     aReads / bReads are indices into `list(set(reads))` of QNAME strings (:1086, :1106-1115), `variantsBlacklisted` is a set
                            (:1059), the singleton rows follow `set(dict_variant_reads.keys()) - set(all_variants)` (:1181-1224)
  haplotypes.txt            the singleton rows (:1226-1239), same set
Everything else in the files is order-free and comes from the GPU path unchanged.  This module re-orders / re-labels the product's rows by REPLAYING those
set constructions -- the same strings inserted in the same sequence into real `set` objects of this interpreter -- which gives the reference's bytes when
the interpreter hashes strings the way the reference's run did: CPython 3.10 with PYTHONHASHSEED=0 (the golden files were written that way).  It is a pure
Python pass over every call line (minutes at whole-genome scale): an exactness mode (Config.py_hash_order / --py_hash_order 1), not the fast path.
"""
from __future__ import annotations

import sys
from collections import OrderedDict
from typing import Dict, List


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

