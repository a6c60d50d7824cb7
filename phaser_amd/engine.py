"""The phasing hot path end to end: K_map -> AS cutoff -> K_tally -> binomial test -> components -> block
phasing -> the five text outputs of phaser/phaser.py:process_vcf (write_vcf excluded).

Everything data-parallel runs in libphz.so on the GPU (mapper, AS histogram, per-variant counters, set
construction, variant-pair cells, connected components).  The host keeps what the reference keeps in
Python: the binomial test (scipy, the reference's own third-party arithmetic, phaser.py:1649), ordering
rules (first-appearance orders), per-block phasing and text formatting.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys
from typing import Dict, List, Optional

import numpy as np
import torch
from scipy.stats import binom

from . import _lib
from . import dist as pdist
from .mapper import Calls, Mapper
from . import rows
from .soa import ReadShard
from .vcf import ChromVariants, VariantSet


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _jl(items, sep=","):
    return sep.join(str(x) for x in items)


class Config:
    """The reference's flags that reach the hot path (phaser.py:30-78), same names and defaults."""

    def __init__(self, **kw):
        self.baseq = 10; self.as_q_cutoff = 0.05; self.cc_threshold = 0.01; self.max_block_size = 15
        self.id_separator = "_"; self.unphased_vars = 1; self.gw_phase_method = 0; self.output_read_ids = 0
        self.unique_ids = 0; self.haplo_count_bam_exclude: List[int] = []; self.haplo_blacklist = frozenset()
        self.include_indels = 0
        self.want_vcf = True           # keep per-block info for write_vcf (vcfout.phased_vcf_text)
        self.host_threads = 1          # threads of the native block phasing / row writer (the reference's --threads)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown option " + k)
            setattr(self, k, v)


def percentile_from_hist(h: np.ndarray, q: float) -> float:
    """numpy.percentile(scores, q) (default linear method, numpy/lib/_function_base_impl.py _quantile/_lerp) for the
    multiset scores = {bin - 32768 repeated h[bin] times}, computed from the histogram: same float64 operations on the
    two neighbouring order statistics, without materialising the scores (phaser.py:551)."""
    n = int(h.sum())
    quant = np.true_divide(q, 100)
    virt = (n - 1) * quant
    prev = int(np.floor(virt))
    gamma = np.float64(virt - prev)
    nxt = prev + 1
    if virt >= n - 1:
        prev = nxt = n - 1
    if virt < 0:
        prev = nxt = 0
    csum = np.cumsum(h)
    a = np.int64(int(np.searchsorted(csum, prev, side="right")) - 32768)
    b = np.int64(int(np.searchsorted(csum, nxt, side="right")) - 32768)
    diff = np.subtract(b, a)
    out = np.add(a, diff * gamma)
    if gamma >= 0.5:
        out = np.subtract(b, diff * (1 - gamma))
    return float(out)


class _Shard:
    def __init__(self, calls: Calls, qid, aln, has_as, n_reads):
        self.calls = calls; self.qid = qid; self.aln = aln; self.has_as = has_as; self.n_reads = n_reads
        self.cutoff = 0.0; self.use_cutoff = 0


class Engine:
    def __init__(self, variants: VariantSet, bam_names: List[str], config: Optional[Config] = None, device: int = 0,
                 mapper: Optional[Mapper] = None):
        self.vs = variants
        self.cfg = config or Config()
        self.bam_names = bam_names
        self.mapper = mapper or Mapper(device)
        self.ctx = self.mapper.ctx
        self.lib = self.ctx.lib
        self.all_chroms = list(variants.chroms.keys())        # VCF order, identical on every rank
        self.chrom_list = list(self.all_chroms)               # chromosomes this rank owns (set_owned)
        self.shards: Dict[str, List[Optional[_Shard]]] = {c: [None] * len(bam_names) for c in self.all_chroms}
        self.qnames: Dict[str, List[str]] = {}
        self.n_qid: Dict[str, int] = {c: 0 for c in self.all_chroms}
        self.log: List[str] = []
        self.stats: Dict[str, float] = {}
        self.total_lines = 0

    def set_owned(self, chroms: List[str]):
        """Multi-GPU: restrict this rank to its chromosomes (keeps VCF order)."""
        own = set(chroms)
        self.chrom_list = [c for c in self.all_chroms if c in own]

    # ---------------------------------------------------------------- stage 2: mapping (phaser.py:526-591)
    def add_shard(self, bam_index: int, chrom: str, shard: ReadShard, n_qid: int, qnames: Optional[List[str]] = None):
        """shard.qid must hold QNAME ids that are consistent across the BAMs of this chromosome."""
        cv = self.vs.chroms[chrom]
        vpos = torch.from_numpy(cv.pos)
        if cv.is_general:
            # indel mode (--include_indels 1): classification against the allele strings happens in the general mapper
            aoff, abytes = cv.allele_pool()
            calls, _ = self.mapper.map_general(shard, vpos, torch.from_numpy(cv.ref_len), torch.from_numpy(aoff.astype(np.int32)),
                                               torch.from_numpy(abytes), self.cfg.baseq)
        else:
            calls = self.mapper.map(shard, vpos, self.cfg.baseq, torch.from_numpy(cv.ref_len))
        has_as = shard.has_as
        self.shards[chrom][bam_index] = _Shard(calls, shard.qid.contiguous(), shard.aln_score.contiguous(),
                                               None if has_as is None else has_as.contiguous(), shard.n)
        self.n_qid[chrom] = max(self.n_qid[chrom], n_qid)
        if qnames is not None:
            self.qnames[chrom] = qnames

    def add_shards(self, bam_index: int, items):
        """Several chromosomes of one BAM at once: items = [(chrom, device-resident ReadShard, n_qid, qnames or None)].  SNP-mode
        chromosomes go through ONE batched K_map submission (phz_map_reads_batch); indel-mode ones one by one."""
        batch = [it for it in items if not self.vs.chroms[it[0]].is_general and it[1].device.type == "cuda"]
        rest = [it for it in items if it not in batch]
        if batch:
            calls = self.mapper.map_batch([it[1] for it in batch], [torch.from_numpy(self.vs.chroms[it[0]].pos) for it in batch],
                                          self.cfg.baseq)
            for it, c in zip(batch, calls):
                self.add_mapped(bam_index, it[0], it[1], c, it[2], it[3] if len(it) > 3 else None)
        for it in rest:
            self.add_shard(bam_index, it[0], it[1], it[2], it[3] if len(it) > 3 else None)

    def add_mapped(self, bam_index: int, chrom: str, shard: ReadShard, calls: Calls, n_qid: int, qnames: Optional[List[str]] = None):
        """Attach a shard whose K_map call list already exists."""
        has_as = shard.has_as
        self.shards[chrom][bam_index] = _Shard(calls, shard.qid.contiguous(), shard.aln_score.contiguous(),
                                               None if has_as is None else has_as.contiguous(), shard.n)
        self.n_qid[chrom] = max(self.n_qid[chrom], n_qid)
        if qnames is not None:
            self.qnames[chrom] = qnames

    def _lines(self, sh: _Shard, bam_index: int) -> _lib.phz_lines:
        c = sh.calls
        return _lib.phz_lines(c.n, _p(c.read_idx), _p(c.var_idx), _p(c.code), sh.n_reads, _p(sh.qid), _p(sh.aln), _p(sh.has_as),
                              float(sh.cutoff), int(sh.use_cutoff), bam_index)

    def close_bam(self, bam_index: int):
        """AS quantile cutoff of one BAM over all its chromosomes (phaser.py:545-553)."""
        shards = [self.shards[c][bam_index] for c in self.chrom_list if self.shards[c][bam_index] is not None]
        if self.cfg.as_q_cutoff > 0:
            dev = shards[0].calls.read_idx.device if shards else self.mapper.device
            hist = torch.zeros(_lib.PHZ_AS_BINS, dtype=torch.int64, device=dev)
            space = _lib.PHZ_DEVICE if dev.type == "cuda" else _lib.PHZ_HOST
            for sh in shards:
                if sh.calls.n:
                    if int(sh.aln.abs().max()) >= 32768:
                        raise _lib.PhzError(_lib.PHZ_E_UNSUPPORTED, "AS value outside int16")
                    ln = self._lines(sh, bam_index)
                    self.ctx.check(self.lib.phz_as_histogram(self.ctx.h, C.byref(ln), _p(hist), space))
            pdist.allreduce_sum_(hist)          # the quantile is over ALL chromosomes of this BAM
            h = hist.cpu().numpy()
            if int(h.sum()) > 0:
                cutoff = percentile_from_hist(h, self.cfg.as_q_cutoff * 100)
                self.log.append("          using alignment score cutoff of %d" % cutoff)
                for sh in shards:
                    sh.cutoff = float(cutoff); sh.use_cutoff = 1
            else:
                self.log.append("          no alignment score value found in reads, cannot use cutoff")

    # ---------------------------------------------------------------- stages 3-6
    def _tally_chrom(self, chrom: str):
        cv = self.vs.chroms[chrom]
        nv = len(cv)
        present = [(b, sh) for b, sh in enumerate(self.shards[chrom]) if sh is not None]
        dev = present[0][1].calls.read_idx.device if present else torch.device("cpu")
        space = _lib.PHZ_DEVICE if dev.type == "cuda" else _lib.PHZ_HOST
        arr = (_lib.phz_lines * max(1, len(present)))()
        total = 0
        for i, (b, sh) in enumerate(present):
            arr[i] = self._lines(sh, b)
            total += sh.calls.n
        if cv.is_general:      # codes 5 / 6 carry the class; single-base codes must not be matched through a0 / a1
            a0 = torch.full((nv,), 255, dtype=torch.uint8, device=dev); a1 = torch.full((nv,), 255, dtype=torch.uint8, device=dev)
        else:
            a0 = torch.from_numpy(cv.a0).to(dev); a1 = torch.from_numpy(cv.a1).to(dev)
        var_count = torch.empty(nv * 3, dtype=torch.int32, device=dev); var_first = torch.empty(nv, dtype=torch.int64, device=dev)
        var_distinct = torch.empty(nv * 3, dtype=torch.int32, device=dev); line_cls = torch.empty(max(1, total), dtype=torch.uint8, device=dev)
        var_rank = torch.empty(max(1, nv), dtype=torch.int64, device=dev)
        cap = max(1024, 4 * nv)
        import time as _t
        t0 = _t.perf_counter()
        while True:
            ea = torch.empty(cap, dtype=torch.int32, device=dev); eb = torch.empty(cap, dtype=torch.int32, device=dev)
            cells = torch.empty(cap * 9, dtype=torch.int32, device=dev); linked = torch.empty(cap, dtype=torch.uint8, device=dev)
            out = _lib.phz_tally_out(_p(var_count), _p(var_first), _p(var_distinct), _p(line_cls), cap, _p(ea), _p(eb), _p(cells), _p(linked),
                                     _p(var_rank))
            ne = C.c_int64(0)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            st = self.lib.phz_tally(self.ctx.h, arr, len(present), nv, _p(a0), _p(a1), max(1, self.n_qid[chrom]), C.byref(out),
                                    C.byref(ne), space)
            self.ctx.check(st, allow=(_lib.PHZ_E_CAPACITY,))
            if st == _lib.PHZ_E_CAPACITY:
                cap = int(ne.value) + 16
                continue
            break
        ne = int(ne.value)
        t1 = _t.perf_counter()
        res = {
            "nv": nv, "var_count": var_count.cpu().numpy().reshape(nv, 3), "var_first": var_first.cpu().numpy(),
            "var_distinct": var_distinct.cpu().numpy().reshape(nv, 3), "line_cls": line_cls[:total].cpu().numpy(),
            "ea": ea[:ne].cpu().numpy(), "eb": eb[:ne].cpu().numpy(), "cells": cells[:ne * 9].cpu().numpy().reshape(ne, 9),
            "linked": linked[:ne].cpu().numpy().astype(bool), "dev": dev, "space": space,
            "var_rank": var_rank[:nv].cpu().numpy().view(np.uint64),
        }
        # host copies of the kept call lines (for ordering rules and read lists)
        lv = []; lq = []; lb = []; offs = []
        base = 0
        for b, sh in present:
            offs.append((b, base, sh.calls.n))
            v = sh.calls.var_idx.cpu().numpy()
            lv.append(v); lq.append(sh.qid[sh.calls.read_idx.long()].cpu().numpy()); lb.append(np.full(len(v), b, dtype=np.int32))
            base += sh.calls.n
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
        res["line_var"] = cat(lv, np.int32); res["line_qid"] = cat(lq, np.int32); res["line_bam"] = cat(lb, np.int32)
        res["bam_offsets"] = offs
        self.stats["tally_call_s"] = self.stats.get("tally_call_s", 0.0) + t1 - t0
        self.stats["tally_d2h_s"] = self.stats.get("tally_d2h_s", 0.0) + _t.perf_counter() - t1
        return res

    def tally_all(self):
        """Stage A: K_tally per owned chromosome; returns the two global noise counters of these chromosomes."""
        self.tally = {c: self._tally_chrom(c) for c in self.chrom_list}
        match = mism = 0
        for c in self.chrom_list:
            vc = self.tally[c]["var_count"].astype(np.int64)
            m = vc[:, 0] + vc[:, 1]; mm = vc[:, 2]
            with np.errstate(divide="ignore", invalid="ignore"):
                ok = (m > 0) & ((mm.astype(np.float64) / (mm + m).astype(np.float64)) < 0.05)
            match += int(m[ok].sum()); mism += int(mm[ok].sum())
        return match, mism

    @staticmethod
    def noise_from_counts(match: int, mism: int) -> float:
        """phaser.py:610-632 (global over chromosomes and BAMs)."""
        if match == 0:
            raise SystemExit("     FATAL ERROR: No reads could be matched to variants. Please double check your settings and input files. "
                             "Common reasons for this occurring include: 1) MAPQ or BASEQ set too conservatively 2) BAM and VCF have "
                             "different chromosome names (IE 'chr1' vs '1').")
        return float(mism) / (float(match + mism) * 2)

    def finish(self, binary: bool = False, chunks: bool = False) -> Optional[Dict[str, str]]:
        """Stages 3-6.  With torch.distributed initialised, every rank handles its own chromosomes and rank 0
        returns the assembled files (other ranks return None).  binary=True returns bytes (no decode pass);
        chunks=True returns, per file, the list of buffers in output order (no join pass; write them with writelines)."""
        import time as _t
        t0 = _t.perf_counter()
        match, mism = pdist.allreduce_counts(*self.tally_all())
        noise = self.noise_from_counts(match, mism)
        t1 = _t.perf_counter()
        local = self._fragments(noise)
        t2 = _t.perf_counter()
        frags = pdist.gather_fragments(local)
        self.stats.update({"tally_s": t1 - t0, "fragments_s": t2 - t1})
        self.noise = noise
        if frags is None:
            return None
        t3 = _t.perf_counter()
        chroms = [c for c in self.all_chroms if c in frags]
        out, summary = merge_fragments(frags, chroms, self.cfg, noise, len(self.bam_names))
        # per chromosome with blocks: (name, block arrays of phz_rows_format, blocks before it) -- what write_vcf needs
        self.vcf_blocks = []
        if self.cfg.want_vcf:
            block_index = 0
            for c in chroms:
                if frags[c]["vcf"] is not None:
                    self.vcf_blocks.append((c, frags[c]["vcf"], block_index))
                    block_index += len(frags[c]["vcf"]["size"])
        self.stats["merge_s"] = _t.perf_counter() - t3
        self.log += summary["log"]
        self.phased = summary["phased"]; self.total_lines = summary["lines"]
        if chunks:
            return out
        out = {k: b"".join(v) for k, v in out.items()}
        return out if binary else {k: v.decode() for k, v in out.items()}

    def _fragments(self, noise: float) -> Dict[str, dict]:
        """Stage C for every owned chromosome.  C2 of chromosome i (native, releases the GIL) runs on a helper thread while
        C1 of chromosome i+1 (numpy / scipy / GPU components) runs here."""
        if len(self.chrom_list) <= 1:
            return {c: self.chrom_fragment(c, noise, self.all_chroms.index(c)) for c in self.chrom_list}
        import time as _t
        from concurrent.futures import ThreadPoolExecutor
        local: Dict[str, dict] = {}
        with ThreadPoolExecutor(1) as ex:
            pending = None
            for c in self.chrom_list:
                t0 = _t.perf_counter()
                frag = self.chrom_prepare(c, noise, self.all_chroms.index(c))
                self.stats["prepare_s"] = self.stats.get("prepare_s", 0.0) + _t.perf_counter() - t0
                if pending is not None:
                    pc, pf, fut = pending
                    pf.update(fut.result()); del self._pre[pc]; local[pc] = pf
                pending = (c, frag, ex.submit(rows.format_chrom, self, c, self.cfg.host_threads))
            pc, pf, fut = pending
            pf.update(fut.result()); del self._pre[pc]; local[pc] = pf
        return {c: local[c] for c in self.chrom_list}

    def chrom_fragment(self, c: str, noise: float, chrom_index: int) -> dict:
        """Stage C for one chromosome: C1 = ordering ranks, pair tests, pruning, components (numpy / scipy / GPU);
        C2 = block phasing + row text in native code (rows.format_chrom -> phz_rows_format)."""
        import time as _t
        t0 = _t.perf_counter()
        frag = self.chrom_prepare(c, noise, chrom_index)
        t1 = _t.perf_counter()
        frag.update(rows.format_chrom(self, c, self.cfg.host_threads))
        self.stats["prepare_s"] = self.stats.get("prepare_s", 0.0) + t1 - t0
        self.stats["rows_s"] = self.stats.get("rows_s", 0.0) + _t.perf_counter() - t1
        del self._pre[c]
        return frag

    def chrom_prepare(self, c: str, noise: float, chrom_index: int) -> dict:
        """Stage C1: ordering ranks, pair tests (scipy), pruning, connected components on the GPU, allelic counts.
        Leaves the chromosome's unphased blocks and first-appearance keys in self._pre[c] for stage C2."""
        cfg = self.cfg
        R = self.tally[c]; cv = self.vs.chroms[c]; nv = R["nv"]
        frag = {"chrom": c, "lines": int((R["line_cls"] != 255).sum())}
        if True:
            kept = R["line_cls"] != 255
            cls = R["line_cls"]
            # ---- ordering rule 4: overlap-dict key order, computed by K_tally (k_rank)
            rank = R["var_rank"]
            # ---- test every linked pair (phaser.py:1594-1654)
            sel = np.nonzero(R["linked"])[0]
            ea = R["ea"][sel]; eb = R["eb"][sel]; cells = R["cells"][sel].astype(np.int64)
            swap = rank[eb] < rank[ea]
            va = np.where(swap, eb, ea); vb = np.where(swap, ea, eb)
            cis = cells[:, 0] + cells[:, 4]
            trans = cells[:, 3] + cells[:, 1]
            oth = cells[:, 6] + cells[:, 7] + cells[:, 2] + cells[:, 5] + cells[:, 8]
            sup = np.maximum(cis, trans); tot = cis + trans + oth
            cfgv = np.where(cis > trans, 0, np.where(cis < trans, 1, -1))
            prob = 1 - ((6 * noise) + (10 * math.pow(noise, 2)))
            pv = np.ones(len(sel), dtype=np.float64)
            test = (sup > 0) & ((tot - sup) > 0)
            if test.any():
                pv[test] = binom.cdf(sup[test], tot[test], prob)
            pv[sup == 0] = 0.0
            keep_edge = ~(pv < cfg.cc_threshold)
            # row order of variant_connections is hash order in the reference; we emit sorted by (rank a, rank b)
            eorder = np.lexsort((rank[vb], rank[va]))
            frag["dropped"] = int((~keep_edge).sum())
            # ---- connected components of the surviving graph on the GPU (phaser.py:1861-1882)
            label = self._component_labels(c, ea, eb, keep_edge)
            deg = np.bincount(ea[keep_edge], minlength=nv) + np.bincount(eb[keep_edge], minlength=nv)
            members = np.nonzero(deg > 0)[0]
            P = {"va": va, "vb": vb, "cis": cis, "trans": trans, "sup": sup, "tot": tot, "pv": pv, "eorder": eorder, "ea": ea, "eb": eb,
                 "cfgv": cfgv, "ncomp": 0}
            if len(members):
                lab = label[members]
                o2 = np.lexsort((members, lab))
                lab_s = lab[o2]; mem_s = members[o2]
                starts = np.nonzero(np.r_[True, lab_s[1:] != lab_s[:-1]])[0]
                ends = np.r_[starts[1:], len(lab_s)]
                comp_rank = np.minimum.reduceat(rank[mem_s], starts)
                e_keep = np.nonzero(keep_edge)[0]
                e_lab = label[ea[e_keep]]
                eo = np.argsort(e_lab, kind="stable")
                e_lab_s = e_lab[eo]
                e_starts = np.searchsorted(e_lab_s, lab_s[starts], side="left"); e_ends = np.searchsorted(e_lab_s, lab_s[starts], side="right")
                P.update({"mem_s": mem_s, "starts": starts, "ends": ends, "comp_order": np.argsort(comp_rank, kind="stable"), "e_keep": e_keep,
                          "eo": eo, "e_starts": e_starts, "e_ends": e_ends, "ncomp": len(starts)})
        # ---- first-appearance order keys of this chromosome's variants (rule 2): (BAM of first kept line, chromosome, line)
        vf = R["var_first"]
        seen = np.nonzero(vf >= 0)[0]
        bam_of = np.zeros(len(seen), dtype=np.int64)
        for b_, base, n in R["bam_offsets"]:
            bam_of[(vf[seen] >= base) & (vf[seen] < base + n)] = b_
        ko = np.lexsort((seen, vf[seen], bam_of))
        P.update({"key_bam": bam_of[ko], "key_line": vf[seen][ko], "key_g": seen[ko], "chrom_index": chrom_index})
        if not hasattr(self, "_pre"):
            self._pre = {}
        self._pre[c] = P
        return frag

    def _component_labels(self, c, ea, eb, keep_edge):
        """Connected-component label per variant of the surviving graph, on the GPU (phz_components)."""
        R = self.tally[c]; nv = R["nv"]; dev = R["dev"]
        t_ea = torch.from_numpy(np.ascontiguousarray(ea)).to(dev); t_eb = torch.from_numpy(np.ascontiguousarray(eb)).to(dev)
        t_keep = torch.from_numpy(keep_edge.astype(np.uint8)).to(dev)
        label = torch.empty(max(1, nv), dtype=torch.int32, device=dev)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        self.ctx.check(self.lib.phz_components(self.ctx.h, nv, len(ea), _p(t_ea), _p(t_eb), _p(t_keep), _p(label), R["space"]))
        return label[:nv].cpu().numpy()


HEAD_ASE = ["contig", "start", "stop", "variants", "variantCount", "variantsBlacklisted", "variantCountBlacklisted", "haplotypeA",
            "haplotypeB", "aCount", "bCount", "totalCount", "blockGWPhase", "gwStat", "max_haplo_maf", "bam", "aReads", "bReads"]
HEAD_HAP = ['contig', 'start', 'stop', 'length', 'variants', 'variant_ids', 'variant_alleles', 'reads_hap_a', 'reads_hap_b',
            'reads_total', 'edges_supporting', 'edges_total', 'annotated_phase', 'phase_concordant', 'gw_phase', 'gw_confidence']


def merge_fragments(frags: Dict[str, dict], chrom_list: List[str], cfg: "Config", noise: float, n_bams: int = 1):
    """Stage D (rank 0): order the per-chromosome fragments of the five files in the reference's global order (returns, per
    file, the list of buffers to write one after the other):
    chromosomes in VCF order for connections / blocks; allelic_counts and singleton rows follow the first-appearance
    keys (BAM of the first kept line, chromosome, line), i.e. per first BAM the chromosomes in VCF order.  Works on
    the bytes the row writer produced, so it is also what the multi-GPU gather feeds."""
    cols = list(HEAD_ASE)
    if cfg.output_read_ids == 1:
        cols += ["read_ids_a", "read_ids_b"]
    enc = lambda fields: ("\t".join(fields) + "\n").encode()
    conn = [b"variant_a\tvariant_b\tsupporting_connections\ttotal_connections\tconflicting_configuration_p\tphase_concordant\n"]
    ase = [enc(cols)]
    hap = [enc(HEAD_HAP)]
    cfgf = [enc(['variant_a', 'rsid_a', 'variant_b', 'rsid_b', 'configuration'])]
    allelic = [b"contig\tposition\tvariantID\trefAllele\taltAllele\trefCount\taltCount\ttotalCount\n"]
    dropped = phased = lines = covered = 0
    for c in chrom_list:
        f = frags[c]
        conn.append(f["conn"]); hap.append(f["hap"]); ase.append(f["ase"]); cfgf.append(f["cfg"])
        dropped += f["dropped"]; phased += f["phased"]; lines += f["lines"]; covered += f["allelic_rows"]
    for b in range(n_bams):
        for c in chrom_list:
            f = frags[c]
            s = f["allelic_seg"]; allelic.append(f["allelic"][s[b]:s[b + 1]])
    for key, dst in (("single_ase", ase), ("single_hap", hap)):
        for b in range(n_bams):
            for c in chrom_list:
                f = frags[c]
                s = f[key + "_seg"]; dst.append(f[key][s[b]:s[b + 1]])
    out = {"variant_connections": conn, "allelic_counts": allelic, "haplotypic_counts": ase, "haplotypes": hap, "allele_config": cfgf}
    log = ["     sequencing noise level estimated at %f" % noise,
           "     %d variant connections dropped because of conflicting configurations (threshold = %f)" % (dropped, cfg.cc_threshold),
           "     %d variants covered by at least 1 read" % covered]
    return out, {"log": log, "phased": phased, "lines": lines, "dropped": dropped, "covered": covered}
