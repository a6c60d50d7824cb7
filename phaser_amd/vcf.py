"""Het-variant loading for the hot path: VCF text -> per-chromosome variant tables.

Mirrors the filter of phaser/phaser.py:396-433 (GT present, no '.', more than one distinct allele,
FILTER contains PASS unless --pass_only 0) and the table of generate_mapping_table :1355-1413
(unique id, SNP-only unless --include_indels), plus the per-variant fields the phasing core derives
from a call line in generate_variant_dict :1418-1462.
"""
from __future__ import annotations

import dataclasses
import gzip
from typing import Dict, List

import numpy as np

_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


@dataclasses.dataclass
class ChromVariants:
    chrom: str
    pos: np.ndarray            # int32, VCF order (must be sorted for the mapper)
    uid: List[str]
    rsid_field: List[str]      # column 3 as written ('.' allowed)
    rsid: List[str]            # '.'/'' replaced by uid (:1451-1455)
    ref: List[str]
    all_alleles: List[List[str]]
    alleles: List[List[str]]   # the individual's alleles in allele-index order (:1430-1435)
    phase: List[List[str]]     # alleles in GT order, or ['-','-'] when unphased (:1437-1443)
    gt: List[str]
    maf_text: List[str]        # str(maf) as written to the table ('None' unless --gw_phase_method 1)
    maf: List[object]          # float, or int 0 when not parseable (:1445-1449)
    ref_len: np.ndarray        # uint8
    a0: np.ndarray             # uint8 base code of alleles[0] (255 when not a single ACGT base)
    a1: np.ndarray
    is_ref: np.ndarray = None      # uint8 [2n]: alleles[k] == REF, index 2*v + k
    phase_idx: np.ndarray = None   # int8 [2n]: position of alleles[k] in the VCF phase, -1 when unphased
    _pools: dict = None

    def pools(self):
        """Separator-joined string pools of the per-variant texts the native row writer prints (phz_rows_format)."""
        if self._pools is None:
            flat = []
            for al in self.alleles:
                flat.append(al[0] if len(al) > 0 else ""); flat.append(al[1] if len(al) > 1 else "")
            self._pools = {"uid": sep_pool(self.uid), "rsid": sep_pool(self.rsid), "allele": sep_pool(flat),
                           "maf": sep_pool([str(x) for x in self.maf]),
                           "maf_val": np.asarray([float(x) for x in self.maf], dtype=np.float64)}
        return self._pools

    def __len__(self):
        return len(self.uid)

    @property
    def is_general(self) -> bool:
        """True when the set is not pure SNPs (some REF longer than one base or some allele not a single ACGT base):
        such sets go through the general mapper (phz_map_reads_general)."""
        return bool((self.ref_len != 1).any() or (self.a0 == 255).any() or (self.a1 == 255).any())

    def allele_pool(self):
        """-> (allele_off uint32 [2n+1], allele_bytes uint8) of the individual's two alleles per variant."""
        off = np.zeros(2 * len(self) + 1, dtype=np.uint32)
        parts = []
        o = 0
        for i, al in enumerate(self.alleles):
            a0 = al[0].encode() if len(al) > 0 else b""
            a1 = al[1].encode() if len(al) > 1 else b""
            off[2 * i] = o; o += len(a0); off[2 * i + 1] = o; o += len(a1)
            parts.append(a0); parts.append(a1)
        off[2 * len(self)] = o
        return off, np.frombuffer(b"".join(parts) + b"\0", dtype=np.uint8).copy()

    def table_rows(self):
        return [[self.chrom, str(int(self.pos[i])), self.uid[i], self.rsid_field[i], ",".join(self.all_alleles[i]),
                 str(int(self.ref_len[i])), self.gt[i], self.maf_text[i]] for i in range(len(self))]


@dataclasses.dataclass
class VariantSet:
    chroms: "Dict[str, ChromVariants]"       # insertion order = VCF order
    het_count: int
    filter_count: int
    indels_excluded: int
    unphased_count: int


def sep_pool(strs):
    """n strings -> (uint32 offsets [n+1], bytes): string i = bytes[off[i] : off[i+1]-1] (one separator byte after each)."""
    if not strs:
        return np.zeros(1, dtype=np.uint32), b"\n"
    b = ("\n".join(strs) + "\n").encode()
    ends = np.flatnonzero(np.frombuffer(b, dtype=np.uint8) == 10)
    if len(ends) != len(strs) or len(b) >= 2 ** 32:
        raise ValueError("string table holds a newline or exceeds 4 GiB")
    off = np.empty(len(strs) + 1, dtype=np.uint32)
    off[0] = 0; off[1:] = ends + 1
    return off, b


def read_text(path: str) -> str:
    if path.endswith(".gz") or path.endswith(".bgz"):
        with gzip.open(path, "rt") as f:
            return f.read()
    with open(path) as f:
        return f.read()


def _load_chunk(task):
    """Filter + table fields for a slice of VCF lines -> ({chrom: column lists}, filter_count, unphased, excluded)."""
    (lines, sample_column, chrom_of_interest, pass_only, include_indels, chr_prefix, id_separator, gw_phase_method, gw_af_field,
     contig_ban) = task
    per: Dict[str, dict] = {}
    filter_count = unphased = excluded = 0
    for line in lines:
        if not line or line[0] == "#":
            continue
        c = line.split("\t")
        chrom0 = c[0]
        for item in contig_ban:
            if item in chrom0:
                raise SystemExit("     FATAL ERROR: Character '%s' must not be present in contig name. Please change id separtor "
                                 "using --id_separator to a character not found in the contig names and try again." % item)
        if chrom_of_interest != "" and chrom_of_interest != chrom0:
            continue
        chrom = chr_prefix + chrom0
        col = per.get(chrom)
        if col is None:
            col = per[chrom] = {k: [] for k in ("pos", "uid", "rsid_field", "rsid", "ref", "all_alleles", "alleles", "phase", "gt",
                                                "maf_text", "maf", "ref_len", "a0", "a1", "r0", "r1", "p0", "p1")}
        fields = c[8].split(":")
        if "GT" not in fields:
            continue
        geno = c[sample_column].split(":")[fields.index("GT")]
        g = list(geno)
        if "." in g:
            continue
        phased = "|" in g
        if phased:
            g.remove("|")
        is_unphased = False
        if "/" in g:
            g.remove("/")
            is_unphased = True
        if len(set(g)) <= 1:
            continue
        if not (pass_only == 0 or "PASS" in c[6].split(";")):
            filter_count += 1
            continue
        unphased += is_unphased
        alts = c[4].split(",")
        every = [c[3]] + alts
        if not (max(len(x) for x in every) == 1 or include_indels == 1):
            excluded += 1
            continue
        uid = chrom + id_separator + c[1] + id_separator + id_separator.join(every)
        maf = None
        if gw_phase_method == 1:
            info = {}
            for item in c[7].split(";"):
                if "=" in item:
                    info[item.split("=")[0]] = item.split("=")[1]
            if gw_af_field in info:
                afs = [float(x) for x in info[gw_af_field].split(",")]
                if len(afs) == len(alts):
                    use = [int(x) - 1 for x in g if x != "." and int(x) != 0]
                    if use:
                        maf = min(min(afs[x], 1 - afs[x]) for x in use)
        # fields the phasing core derives from the table row (generate_variant_dict)
        ind = [every[i] for i in range(len(every)) if str(i) in g]
        ph = [every[int(i)] for i in g] if phased else ["-", "-"]
        mtxt = str(maf)
        try:
            mval = float(mtxt)
        except ValueError:
            mval = 0
        col["pos"].append(int(c[1])); col["ref_len"].append(min(255, len(c[3])))
        col["uid"].append(uid); col["rsid_field"].append(c[2]); col["rsid"].append(c[2] if c[2] not in (".", "") else uid)
        col["ref"].append(c[3]); col["all_alleles"].append(every); col["alleles"].append(ind); col["phase"].append(ph); col["gt"].append(geno)
        col["maf_text"].append(mtxt); col["maf"].append(mval)
        col["a0"].append(_CODE.get(ind[0], 255) if len(ind) > 0 else 255)
        col["a1"].append(_CODE.get(ind[1], 255) if len(ind) > 1 else 255)
        i0 = ind[0] if len(ind) > 0 else ""; i1 = ind[1] if len(ind) > 1 else ""
        col["r0"].append(i0 == c[3]); col["r1"].append(i1 == c[3])
        col["p0"].append(ph.index(i0) if i0 in ph else -1); col["p1"].append(ph.index(i1) if i1 in ph else -1)
    return per, filter_count, unphased, excluded


_US = "\x1f"
_STR_COLS = ("uid", "rsid_field", "rsid", "ref", "gt", "maf_text")
_LIST_COLS = ("all_alleles", "alleles", "phase")
_INT_COLS = ("pos", "ref_len", "a0", "a1", "r0", "r1", "p0", "p1")


_FORK_LINES = None          # the VCF lines, inherited by forked workers (never pickled)


def _load_chunk_compact(task):
    """Worker wrapper: same as _load_chunk on lines [lo, hi) of the inherited text, with the nested Python lists flattened
    into a few big strings / arrays, which cross the process boundary far faster than pickled lists of lists."""
    lo, hi = task[0]
    per, fc, un, ex = _load_chunk((_FORK_LINES[lo:hi],) + tuple(task[1:]))
    out = {}
    for chrom, col in per.items():
        rec = {"n": len(col["uid"]), "maf": col["maf"]}
        for k in _INT_COLS:
            rec[k] = np.asarray(col[k], dtype=np.int64)
        for k in _STR_COLS:
            rec[k] = _US.join(col[k])
        for k in _LIST_COLS:
            rec[k] = _US.join(",".join(x) for x in col[k])
        out[chrom] = rec
    return out, fc, un, ex


def _expand(rec):
    n = rec["n"]
    col = {"maf": rec["maf"]}
    for k in _INT_COLS:
        col[k] = rec[k].tolist()
    for k in _STR_COLS:
        col[k] = rec[k].split(_US) if n else []
    for k in _LIST_COLS:
        col[k] = [x.split(",") if x else [] for x in rec[k].split(_US)] if n else []
    return col


def load_variants(vcf_text: str, sample_column: int = 9, chrom_of_interest: str = "", pass_only: int = 1,
                  include_indels: int = 0, chr_prefix: str = "", id_separator: str = "_", gw_phase_method: int = 0,
                  gw_af_field: str = "AF", contig_ban=("_", ":"), threads: int = 1) -> VariantSet:
    """threads > 1 fans the per-line work out to forked workers (call it before a GPU context exists: forks are cheap then)."""
    lines = vcf_text.split("\n")
    opts = (sample_column, chrom_of_interest, pass_only, include_indels, chr_prefix, id_separator, gw_phase_method, gw_af_field,
            tuple(contig_ban))
    if threads > 1 and len(lines) > 100_000:
        import multiprocessing as mp
        n = min(threads, 64)
        step = (len(lines) + n - 1) // n
        global _FORK_LINES
        _FORK_LINES = lines
        tasks = [((i, min(i + step, len(lines))),) + opts for i in range(0, len(lines), step)]
        with mp.get_context("fork").Pool(n) as pool:
            cparts = pool.map(_load_chunk_compact, tasks, chunksize=1)
        _FORK_LINES = None
        parts = [({chrom: _expand(rec) for chrom, rec in per.items()}, fc, un, ex) for per, fc, un, ex in cparts]
    else:
        parts = [_load_chunk((lines,) + opts)]
    merged: Dict[str, dict] = {}
    filter_count = unphased = excluded = 0
    for per, fc, un, ex in parts:
        filter_count += fc; unphased += un; excluded += ex
        for chrom, col in per.items():
            tgt = merged.get(chrom)
            if tgt is None:
                merged[chrom] = col
            else:
                for k, v in col.items():
                    tgt[k] += v
    out: Dict[str, ChromVariants] = {}
    het = 0
    for chrom, col in merged.items():
        cv = ChromVariants(chrom, np.asarray(col["pos"], dtype=np.int32), col["uid"], col["rsid_field"], col["rsid"], col["ref"],
                           col["all_alleles"], col["alleles"], col["phase"], col["gt"], col["maf_text"], col["maf"],
                           np.asarray(col["ref_len"], dtype=np.uint8), np.asarray(col["a0"], dtype=np.uint8),
                           np.asarray(col["a1"], dtype=np.uint8),
                           np.stack([np.asarray(col["r0"], dtype=np.uint8), np.asarray(col["r1"], dtype=np.uint8)], axis=1).reshape(-1),
                           np.stack([np.asarray(col["p0"], dtype=np.int8), np.asarray(col["p1"], dtype=np.int8)], axis=1).reshape(-1))
        if len(cv.pos) > 1 and bool((np.diff(cv.pos) < 0).any()):
            raise SystemExit("     FATAL ERROR: VCF records of %s are not sorted by position." % chrom)
        het += len(cv.uid)
        cv.pools()                 # string tables of the variant table for the native row writer (built once, with the table)
        out[chrom] = cv
    return VariantSet(out, het, filter_count, excluded, unphased)
