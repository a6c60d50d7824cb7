"""phaser_pop/phaser_expr_matrix.py drop-in: per-sample phaser_gene_ae tables -> gene x sample matrices.

Same arguments (`--gene_ae_dir --features --t --o`) and the same two outputs, `<o>.bed` (aCount|bCount of every sample and gene)
and `<o>.gw_phased.bed` (0|0 where the gene-level count was not genome-wide phased); they are written BGZF-compressed as
`<o>.bed.gz` / `<o>.gw_phased.bed.gz` with the native writer, with tabix indices when the rows are position-sorted.  A plain join on the host -- no GPU work here.

Behaviour kept from the reference (file:line = phaser_pop/phaser_expr_matrix.py):
  * a sample enters only if its gene names, in order, equal column 4 of the features file (:108); otherwise the same error line;
  * the coordinate columns come from the FIRST gene_ae file of the directory listing (:61, :76-82) and every sample column is
    attached BY ROW NUMBER (pandas index alignment, :64-65): if that first file was written with --min_cov and lacks genes,
    the matrix has its rows and the values of the leading genes of the full list -- reproduced, not repaired;
  * directory order: the reference takes os.listdir() as it comes; here the names are sorted unless `order` says otherwise.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List, Tuple


def _read_gene_ae(path: str):
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        lines = [l.rstrip("\n") for l in f if l.strip("\n")]
    cols = lines[0].split("\t")
    rows = [l.split("\t") for l in lines[1:]]
    return cols, rows


def expr_matrix(gene_ae_dir: str, features_path: str, order: str = "sorted") -> Tuple[str, str, List[str]]:
    """-> (text of <o>.bed, text of <o>.gw_phased.bed, log lines)"""
    log = []
    gene_list = []
    for line in open(features_path):
        if not line.strip("\n") or line.startswith("#"):
            continue
        gene_list.append(line.rstrip("\n").split("\t")[3])
    names = [f for f in os.listdir(gene_ae_dir) if ".txt" in f]
    names = sorted(names, reverse=(order == "reversed")) if order in ("sorted", "reversed") else names
    if not names:
        log.append("FATAL ERROR - no files read for input...")
        return "", "", log
    samples = []        # (sample, all values, gw_phased values)
    for fn in names:
        path = os.path.join(gene_ae_dir, fn)
        cols, rows = _read_gene_ae(path)
        if "bam" not in cols or "gw_phased" not in cols:
            continue
        ib = cols.index("bam"); iname = cols.index("name"); ia = cols.index("aCount"); ibc = cols.index("bCount"); ig = cols.index("gw_phased")
        seen = []
        for r in rows:
            if r[ib] not in seen:
                seen.append(r[ib])
        for xs in seen:
            sub = [r for r in rows if r[ib] == xs]
            if [r[iname] for r in sub] == gene_list:
                allv = [r[ia] + "|" + r[ibc] for r in sub]
                gwv = [(r[ia] + "|" + r[ibc]) if int(float(r[ig])) == 1 else "0|0" for r in sub]
                samples.append((xs, allv, gwv))
            else:
                log.append("ERROR - " + path + ":" + xs + " genes are not in correct order...")
    # coordinate columns: one BAM of the first file (:76-82)
    cols, rows = _read_gene_ae(os.path.join(gene_ae_dir, names[0]))
    ib = cols.index("bam")
    one = rows[0][ib] if rows else None
    base = [(i, r) for i, r in enumerate(rows) if r[ib] == one]
    ic = cols.index("contig"); ist = cols.index("start"); isp = cols.index("stop"); iname = cols.index("name")

    def render(which: int) -> str:
        out = ["\t".join(["#contig", "start", "stop", "name"] + [s[0] for s in samples]) + "\n"]
        for i, r in base:                  # i = row number in the first file = the index pandas aligns on
            vals = [(s[which][i] if i < len(s[which]) else "") for s in samples]
            out.append("\t".join([r[ic], r[ist], r[isp], r[iname]] + vals) + "\n")
        return "".join(out)
    return render(1), render(2), log


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gene_ae_dir", required=True); ap.add_argument("--features", required=True)
    ap.add_argument("--t", type=int, default=1); ap.add_argument("--o", required=True)
    args = ap.parse_args(argv)
    print(""); print("##################################################")
    print("        Welcome to phASER-POP v0.1.0 (phaser_amd)")
    print("##################################################"); print("")
    print("#1 Loading gene feature file...")
    print("#2 Loading gene ae files...")
    a, g, log = expr_matrix(args.gene_ae_dir, args.features)
    for l in log:
        print(l)
    if not a:
        return 1
    from . import vcfout
    print("#3 Saving sample matrix (all)...")
    vcfout.write_bgzf(args.o + ".bed.gz", a, args.t)
    print("#4 Saving sample matrix (gw_phased)...")
    vcfout.write_bgzf(args.o + ".gw_phased.bed.gz", g, args.t)
    for path in (args.o + ".bed.gz", args.o + ".gw_phased.bed.gz"):      # bgzip -f ...; tabix -p bed -f ... (:66, :75)
        if not vcfout.tabix_index(path, "bed", args.t):
            print("WARNING - %s is not position-sorted, no tabix index written (tabix refuses such files too)" % path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
