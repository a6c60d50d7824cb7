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
from .phase import Block
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
        self.host_threads = 1          # forked workers for block phasing / row formatting (the reference's --threads)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown option " + k)
            setattr(self, k, v)


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
                nz = np.nonzero(h)[0]
                scores = np.repeat(nz.astype(np.int64) - 32768, h[nz])
                cutoff = np.percentile(scores, self.cfg.as_q_cutoff * 100)
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
        cap = max(1024, 4 * nv)
        while True:
            ea = torch.empty(cap, dtype=torch.int32, device=dev); eb = torch.empty(cap, dtype=torch.int32, device=dev)
            cells = torch.empty(cap * 9, dtype=torch.int32, device=dev); linked = torch.empty(cap, dtype=torch.uint8, device=dev)
            out = _lib.phz_tally_out(_p(var_count), _p(var_first), _p(var_distinct), _p(line_cls), cap, _p(ea), _p(eb), _p(cells), _p(linked))
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
        res = {
            "nv": nv, "var_count": var_count.cpu().numpy().reshape(nv, 3), "var_first": var_first.cpu().numpy(),
            "var_distinct": var_distinct.cpu().numpy().reshape(nv, 3), "line_cls": line_cls[:total].cpu().numpy(),
            "ea": ea[:ne].cpu().numpy(), "eb": eb[:ne].cpu().numpy(), "cells": cells[:ne * 9].cpu().numpy().reshape(ne, 9),
            "linked": linked[:ne].cpu().numpy().astype(bool), "dev": dev, "space": space,
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

    def finish(self) -> Optional[Dict[str, str]]:
        """Stages 3-6.  With torch.distributed initialised, every rank handles its own chromosomes and rank 0
        returns the assembled files (other ranks return None)."""
        import time as _t
        t0 = _t.perf_counter()
        match, mism = pdist.allreduce_counts(*self.tally_all())
        noise = self.noise_from_counts(match, mism)
        t1 = _t.perf_counter()
        if self.cfg.host_threads > 1:
            local = self._fragments_parallel(noise)
        else:
            local = {c: self.chrom_fragment(c, noise, self.all_chroms.index(c)) for c in self.chrom_list}
        t2 = _t.perf_counter()
        frags = pdist.gather_fragments(local)
        self.stats.update({"tally_s": t1 - t0, "fragments_s": t2 - t1})
        self.noise = noise
        if frags is None:
            return None
        t3 = _t.perf_counter()
        out, summary = merge_fragments(frags, [c for c in self.all_chroms if c in frags], self.cfg, noise)
        self.stats["merge_s"] = _t.perf_counter() - t3
        self.noise = noise
        self.log += summary["log"]
        self.phased = summary["phased"]; self.total_lines = summary["lines"]; self.vcf_lookup = summary["vcf_lookup"]
        return out

    def _fragments_parallel(self, noise: float) -> Dict[str, dict]:
        """Stage C with the host work fanned out like the reference's parallelize() (phaser.py:2077-2094): C1 (numpy,
        scipy, GPU components) runs here per chromosome; C2 (block phasing + row formatting, pure Python on plain data)
        runs in forked workers over chunks of blocks / singletons.  Workers never touch the GPU."""
        import multiprocessing as mp
        global _FORK_ENGINE
        import time as _t
        t0 = _t.perf_counter()
        frags = {c: self.chrom_prepare(c, noise, self.all_chroms.index(c)) for c in self.chrom_list}
        self.stats["prepare_s"] = _t.perf_counter() - t0
        nblocks = sum(self._pre[c]["ncomp"] for c in self.chrom_list)
        chunk = max(50, min(1500, nblocks // (4 * self.cfg.host_threads) + 1))
        tasks = []
        for c in self.chrom_list:
            P = self._pre[c]
            tasks += [("conn", c, lo, min(lo + 40000, len(P["eorder"]))) for lo in range(0, len(P["eorder"]), 40000)]
            tasks += [("alle", c, lo, min(lo + 40000, len(P["key_g"]))) for lo in range(0, len(P["key_g"]), 40000)]
            tasks += [("blk", c, lo, min(lo + chunk, P["ncomp"])) for lo in range(0, P["ncomp"], chunk)]
        _FORK_ENGINE = self
        ctx = mp.get_context("fork")
        if tasks:
            with ctx.Pool(min(self.cfg.host_threads, len(tasks))) as pool:
                res = pool.map(_fork_task, tasks, chunksize=1)
        else:
            res = []
        phased: Dict[str, set] = {c: set() for c in self.chrom_list}
        for (kind, c, lo, hi), r in zip(tasks, res):
            if kind == "conn":
                frags[c].setdefault("conn", []).append(r)
            elif kind == "alle":
                frags[c].setdefault("allelic", []).extend(r)
            else:
                frags[c].setdefault("blocks", []).append(r[0])
                phased[c].update(r[1])
        self._phased_sets = phased            # must exist before the second fork: singleton workers read it
        stasks = [("sng", c, lo, min(lo + 20000, len(self._pre[c]["key_g"]))) for c in self.chrom_list
                  for lo in range(0, len(self._pre[c]["key_g"]), 20000)]
        if stasks:
            with ctx.Pool(min(self.cfg.host_threads, len(stasks))) as pool:
                sres = pool.map(_fork_task, stasks, chunksize=1)
            for (kind, c, lo, hi), rows in zip(stasks, sres):
                frags[c].setdefault("singles", []).extend(rows)
        for c in self.chrom_list:
            frags[c].setdefault("blocks", []); frags[c].setdefault("singles", []); frags[c].setdefault("conn", []); frags[c].setdefault("allelic", [])
            frags[c]["phased"] = len(phased[c])
        self.stats["rows_pool_s"] = _t.perf_counter() - t0 - self.stats["prepare_s"]
        _FORK_ENGINE = None
        return frags

    def chrom_fragment(self, c: str, noise: float, chrom_index: int) -> dict:
        """Stage C for one chromosome: pair tests, pruning, components (C1), block phasing + output rows (C2), serially."""
        frag = self.chrom_prepare(c, noise, chrom_index)
        P = self._pre[c]
        frag["conn"] = [self._conn_text(c, 0, len(P["eorder"]))]
        frag["allelic"] = self._allelic_rows(c, 0, len(P["key_g"]))
        chunk, phased = self._block_rows(c, 0, P["ncomp"])
        frag["blocks"] = [chunk]
        frag["singles"] = self._single_rows(c, 0, len(P["key_g"]), set(phased))
        frag["phased"] = len(phased)
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
            # ---- ordering rules 3 and 4: read_vars order, overlap-dict key order
            lines = np.nonzero(kept & (cls < 2))[0]
            q = R["line_qid"][lines]; b = R["line_bam"][lines]; v = R["line_var"][lines]
            nq = max(1, self.n_qid[c])
            owner = np.full(nq, -1, dtype=np.int32)
            np.maximum.at(owner, q, b)
            qfirst = np.full(nq, np.iinfo(np.int64).max, dtype=np.int64)
            np.minimum.at(qfirst, q, lines)
            own = b == owner[q]
            lo = lines[own]; qo = q[own]; vo = v[own]
            o = np.lexsort((lo, qfirst[qo]))            # by (first appearance of the QNAME, line)
            qs = qo[o]; vs_ = vo[o]
            rank = np.full(nv, np.iinfo(np.int64).max, dtype=np.int64)
            if len(qs):
                # QNAMEs with >= 2 distinct variants create overlap keys, in list order
                newq = np.r_[True, qs[1:] != qs[:-1]]
                gid = np.cumsum(newq) - 1
                pair_key = gid.astype(np.int64) * (nv + 1) + vs_
                uniq = np.unique(pair_key)
                distinct_per_group = np.bincount((uniq // (nv + 1)).astype(np.int64), minlength=gid[-1] + 1)
                multi = distinct_per_group[gid] >= 2
                idx = np.nonzero(multi)[0]
                uv, first = np.unique(vs_[idx], return_index=True)
                rank[uv] = idx[first]
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
            deg = np.zeros(nv, dtype=np.int64)
            np.add.at(deg, ea[keep_edge], 1); np.add.at(deg, eb[keep_edge], 1)
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
        self._read_lists(R)          # cache the per-variant read lists (shared with forked row workers)
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

    # ---- stage C2 pieces: pure host work on the arrays of self._pre[c] (safe in forked workers)
    def _conn_text(self, c, lo, hi) -> str:
        """variant_connections rows (phaser.py:691-695) for eorder[lo:hi]."""
        P = self._pre[c]; cv = self.vs.chroms[c]
        va, vb, cis, trans, sup, tot, pv = P["va"], P["vb"], P["cis"], P["trans"], P["sup"], P["tot"], P["pv"]
        uid = cv.uid; phase = cv.phase; alle = cv.alleles
        rows = []
        for k in P["eorder"][lo:hi]:
            a = int(va[k]); bb = int(vb[k])
            conc = "."
            if "-" not in phase[a] and "-" not in phase[bb]:
                if cis[k] > trans[k]:
                    conc = 1 if phase[a].index(alle[a][0]) == phase[bb].index(alle[bb][0]) else 0
                elif cis[k] < trans[k]:
                    conc = 1 if phase[a].index(alle[a][1]) == phase[bb].index(alle[bb][0]) else 0
            if sup[k] == 0:
                ptxt = "0"
            elif tot[k] - sup[k] > 0:
                ptxt = str(np.float64(pv[k]))
            else:
                ptxt = "1"
            rows.append("\t".join([uid[a], uid[bb], str(int(sup[k])), str(int(tot[k])), ptxt, str(conc)]) + "\n")
        return "".join(rows)

    def _allelic_rows(self, c, lo, hi):
        """allelic_counts rows (phaser.py:737-749) for the first-appearance keys [lo, hi)."""
        P = self._pre[c]; cv = self.vs.chroms[c]; R = self.tally[c]
        out = []
        ci = P["chrom_index"]
        for kb, kl, g in zip(P["key_bam"][lo:hi].tolist(), P["key_line"][lo:hi].tolist(), P["key_g"][lo:hi].tolist()):
            d = R["var_distinct"][g]
            r0 = int(d[0]); r1 = int(d[1])
            if r0 + r1 > 0:
                out.append(((kb, ci, kl), "\t".join([c, str(int(cv.pos[g])), cv.uid[g], cv.alleles[g][0], cv.alleles[g][1], str(r0), str(r1),
                                                    str(r0 + r1) + "\n"])))
        return out

    def _components(self, c, lo, hi):
        """(members, local edges) of the components ranked [lo, hi) in first-key order (phaser.py:1861-1882)."""
        P = self._pre[c]; cv = self.vs.chroms[c]
        blocks = []
        if P["ncomp"] == 0:
            return blocks
        pos = cv.pos; ea, eb, cfgv = P["ea"], P["eb"], P["cfgv"]
        for ci in P["comp_order"][lo:hi]:
            mem = P["mem_s"][P["starts"][ci]:P["ends"][ci]]
            mem = mem[np.lexsort((mem, pos[mem]))]            # sort_var_ids (:1884): by position, ties by index
            loc = {int(g): i for i, g in enumerate(mem)}
            ek = P["e_keep"][P["eo"][P["e_starts"][ci]:P["e_ends"][ci]]]
            blocks.append((mem, [(loc[int(ea[e])], loc[int(eb[e])], int(cfgv[e])) for e in ek]))
        return blocks

    # ---------------------------------------------------------------- output (phaser.py:832-1243)
    def _read_lists(self, R):
        """Per (variant, class in {0,1}): QNAME ids of kept lines in line order, all BAMs and per BAM."""
        if "by_var" in R:
            return R["by_var"]
        cls = R["line_cls"]
        lines = np.nonzero(cls < 2)[0]
        v = R["line_var"][lines]; k = cls[lines].astype(np.int64)
        key = v.astype(np.int64) * 2 + k
        o = np.argsort(key, kind="stable")
        ks = key[o]
        starts = np.searchsorted(ks, np.arange(R["nv"] * 2), side="left"); ends = np.searchsorted(ks, np.arange(R["nv"] * 2), side="right")
        R["by_var"] = (lines[o], starts, ends)
        return R["by_var"]

    def _block_rows(self, c, comp_lo, comp_hi):
        """Stage C2a for the components ranked [comp_lo, comp_hi): phase them (phaser.py:795-814) and format their rows
        (:865-1172).  Pure host work on plain data -> safe to run in forked workers.  Returns (compact chunk record,
        phased variant indices)."""
        cfg = self.cfg
        blocks_all = self._components(c, comp_lo, comp_hi)
        nb = len(self.bam_names)
        cv = self.vs.chroms[c]
        R = self.tally[c]
        blocks_out = []
        final = []
        for mem, edges in blocks_all:
            blk = Block(len(mem), edges)
            for sub in blk.phase(cfg.max_block_size):
                final.append([(int(mem[i]), a) for i, a in sub])
        # allele-edge lookup for supporting / total edge counts: (a, b) -> cfg for surviving edges
        d: Dict[tuple, int] = {}
        for mem, edges in blocks_all:
            for i, j, k in edges:
                a, b = int(mem[i]), int(mem[j])
                d[(a, b)] = k; d[(b, a)] = k
        in_block = []
        lines_sorted, starts, ends = self._read_lists(R)
        lq = R["line_qid"]; lbam = R["line_bam"]
        for blk in final:
            ase = []; cfgf = []
            blk = sorted(blk, key=lambda t: (int(cv.pos[t[0]]), t[0]))     # sort_var_ids again (:869); already sorted
            variants = [g for g, _ in blk]
            in_block += variants
            ha = "".join(a for _, a in blk)
            hb = "".join(str(int(not int(x))) for x in ha)
            # edges supporting / total (:876-895): ordered allele pairs, halved
            sup = tot = 0
            alle_of = {g: int(a) for g, a in blk}
            for g1 in variants:
                for g2 in variants:
                    if g1 != g2 and (g1, g2) in d:
                        k = d[(g1, g2)]
                        if k >= 0:
                            tot += 1           # exactly one of g2:0 / g2:1 is linked to g1's allele
                            linked_allele = alle_of[g1] if k == 0 else 1 - alle_of[g1]
                            if alle_of[g2] == linked_allele:
                                sup += 1
            sup = sup / 2; tot = tot / 2
            rsids = [cv.rsid[g] for g in variants] if cfg.unique_ids == 0 else [cv.uid[g] for g in variants]
            poss = [int(cv.pos[g]) for g in variants]
            alle = [[], []]; phs = [[], []]; counts = [0, 0]
            for h in (0, 1):
                hx = (ha, hb)[h]
                pool = []
                for i, g in enumerate(variants):
                    k = int(hx[i])
                    a = cv.alleles[g][k]
                    alle[h].append(a)
                    try:
                        phs[h].append(cv.phase[g].index(a))
                    except ValueError:
                        phs[h].append(float("nan"))
                    s_, e_ = starts[g * 2 + k], ends[g * 2 + k]
                    pool.append(lq[lines_sorted[s_:e_]])
                counts[h] = len(np.unique(np.concatenate(pool))) if pool else 0
            usable = [x for x in phs[0] if str(x) != "nan"]
            conc = 1 if len(set(usable)) <= 1 else 0
            pstr = ["".join(str(x).replace("nan", "-") for x in phs[0]), "".join(str(x).replace("nan", "-") for x in phs[1])]
            known = [int(x) for x in phs[0] if x >= 0]
            cor = [phs[0], phs[1]]
            stat = 0.5
            mafs = [cv.maf[g] for g in variants]
            if known:
                ps = set(phs[0])
                if len(ps) == 1:
                    stat = 1
                elif cfg.gw_phase_method == 0:
                    stat = np.mean(known)
                    if stat < 0.5:
                        cor = [[0] * len(variants), [1] * len(variants)]
                    elif stat > 0.5:
                        cor = [[1] * len(variants), [0] * len(variants)]
                    stat = max([stat, 1 - stat])
                elif cfg.gw_phase_method == 1:
                    w = [0, 0]
                    for p_, m_ in zip(phs[0], mafs):
                        if p_ == 0:
                            w[0] += m_
                        elif p_ == 1:
                            w[1] += m_
                    if sum(w) > 0:
                        stat = max(w) / sum(w)
                        if w[0] > w[1]:
                            cor = [[0] * len(variants), [1] * len(variants)]
                        elif w[1] > w[0]:
                            cor = [[1] * len(variants), [0] * len(variants)]
                    else:
                        stat = np.mean(known)
                        if stat < 0.5:
                            cor = [[0] * len(variants), [1] * len(variants)]
                        elif stat > 0.5:
                            cor = [[1] * len(variants), [0] * len(variants)]
                        stat = max([stat, 1 - stat])
            cstr = ["".join(str(x).replace("nan", "-") for x in cor[0]), "".join(str(x).replace("nan", "-") for x in cor[1])]
            hap_row = _jl([c, min(poss), max(poss), max(poss) - min(poss), len(variants), _jl(rsids), _jl(alle[0]) + "|" + _jl(alle[1]),
                           counts[0], counts[1], sum(counts), sup, tot, pstr[0] + "|" + pstr[1], conc, cstr[0] + "|" + cstr[1], stat], "\t") + "\n"
            # haplotypic counts, one row per BAM (:1048-1125)
            for b in range(nb):
                if b in cfg.haplo_count_bam_exclude:
                    continue
                used_alleles = [[], []]; used_vars = []; vreads = [[], []]; upos = []; black = []
                for h in (0, 1):
                    hx = (ha, hb)[h]
                    for i, g in enumerate(variants):
                        upos.append(int(cv.pos[g]))
                        if c + "_" + str(int(cv.pos[g])) not in cfg.haplo_blacklist:
                            k = int(hx[i])
                            if g not in used_vars:
                                used_vars.append(g)
                            used_alleles[h].append(cv.alleles[g][k])
                            s_, e_ = starts[g * 2 + k], ends[g * 2 + k]
                            ln = lines_sorted[s_:e_]
                            vreads[h].append(lq[ln[lbam[ln] == b]])
                        elif g not in black:
                            black.append(g)
                labels = []; ids = []; ns = []
                for h in (0, 1):
                    allq = np.concatenate(vreads[h]) if vreads[h] else np.zeros(0, np.int32)
                    # labels = index into the distinct-read list; we number reads by first appearance (canonical form)
                    uq, first, inv = np.unique(allq, return_index=True, return_inverse=True)
                    order = np.argsort(first, kind="stable")
                    rk = np.empty(len(uq), dtype=np.int64)
                    rk[order] = np.arange(len(uq))
                    txt = list(map(str, rk[inv].tolist()))
                    parts = []; p0 = 0
                    for vr in vreads[h]:
                        parts.append(",".join(txt[p0:p0 + len(vr)])); p0 += len(vr)
                    labels.append(";".join(parts))
                    ns.append(len(uq)); ids.append(uq[order])
                cov = ns[0] + ns[1]
                if cov > 0:
                    gwp = "0/1"
                    if cor[0][0] == 0:
                        gwp = "0|1"
                    elif cor[0][0] == 1:
                        gwp = "1|0"
                    f = [c, min(upos), max(upos), _jl(cv.uid[g] for g in used_vars), len(used_vars), _jl(cv.uid[g] for g in black), len(black),
                         _jl(used_alleles[0]), _jl(used_alleles[1]), ns[0], ns[1], cov, gwp, stat]
                    if cfg.output_read_ids == 1:
                        qn = self.qnames[c]
                        f += [_jl(qn[int(x)] for x in ids[0]), _jl(qn[int(x)] for x in ids[1])]
                    f += [str(max(mafs)), self.bam_names[b], labels[0], labels[1]]
                    ase.append(_jl(f, "\t") + "\n")
            # allele configuration (:1160-1172)
            for ga, aa in zip(variants, alle[0]):
                for gb, ab in zip(variants, alle[1]):
                    if ga != gb:
                        ra = cv.ref[ga] == aa; rb = cv.ref[gb] == ab
                        cfgf.append("\t".join([cv.uid[ga], cv.rsid[ga], cv.uid[gb], cv.rsid[gb], "trans" if ra == rb else "cis"]) + "\n")
            def _gw(x):
                return int(x) if isinstance(x, int) and not isinstance(x, bool) else None
            vinfo = {"uids": [cv.uid[g] for g in variants], "hap": [ha[i] + "|" + hb[i] for i in range(len(variants))],
                     "rsids": [cv.rsid[g] for g in variants], "stat": stat if isinstance(stat, (int, float)) and not isinstance(stat, np.floating) else float(stat),
                     "stat_txt": str(stat), "max_maf_txt": str(max(mafs)),
                     "alleles": [cv.alleles[g] for g in variants], "all_alleles": [cv.all_alleles[g] for g in variants],
                     "gw": [[_gw(cor[0][i]) if int(ha[i]) == 0 else _gw(cor[1][i]), _gw(cor[1][i]) if int(ha[i]) == 0 else _gw(cor[0][i])]
                            for i in range(len(variants))]}
            blocks_out.append({"hap": hap_row, "ase": ase, "cfg": cfgf, "vcf": vinfo})
        # one compact record per call (cheap to ship back from a forked worker): joined text + optional VCF info per block
        chunk = {"hap": "".join(b["hap"] for b in blocks_out), "ase": "".join(r for b in blocks_out for r in b["ase"]),
                 "cfg": "".join(r for b in blocks_out for r in b["cfg"]), "n": len(blocks_out),
                 "vcf": [b["vcf"] for b in blocks_out] if cfg.want_vcf else None}
        return chunk, in_block

    def _single_rows(self, c, lo, hi, in_block):
        """Stage C2b: rows of variants with coverage that ended up in no block (phaser.py:1180-1239), keys [lo, hi)."""
        cfg = self.cfg
        P = self._pre[c]
        ci_ = P["chrom_index"]
        var_keys = [(kb, ci_, kl, g) for kb, kl, g in zip(P["key_bam"][lo:hi].tolist(), P["key_line"][lo:hi].tolist(), P["key_g"][lo:hi].tolist())]
        nb = len(self.bam_names)
        cv = self.vs.chroms[c]
        R = self.tally[c]
        lines_sorted, starts, ends = self._read_lists(R)
        lq = R["line_qid"]; lbam = R["line_bam"]
        singles = []
        single_bam_fast = nb == 1 and cfg.output_read_ids != 1 and not cfg.haplo_count_bam_exclude
        if cfg.unphased_vars == 1:
            for kb, kc, kl, g in var_keys:
                vc = R["var_count"][g]
                if int(vc[0]) + int(vc[1]) == 0 or g in in_block:       # removed at :769-774, or phased
                    continue
                ph = cv.phase[g]
                rows_a = []
                if c + "_" + str(int(cv.pos[g])) not in cfg.haplo_blacklist:
                    for b in range(nb):
                        if b in cfg.haplo_count_bam_exclude:
                            continue
                        if single_bam_fast:
                            # one BAM: distinct QNAMEs per (variant, allele) were counted on the GPU (var_distinct)
                            n0, n1 = int(R["var_distinct"][g][0]), int(R["var_distinct"][g][1])
                            per_allele = None
                        else:
                            per_allele = []
                            for k in (0, 1):
                                ln = lines_sorted[starts[g * 2 + k]:ends[g * 2 + k]]
                                per_allele.append(np.unique(lq[ln[lbam[ln] == b]]))
                            n0, n1 = len(per_allele[0]), len(per_allele[1])
                        cov = n0 + n1
                        if cov > 0:
                            ps = (str(ph.index(cv.alleles[g][0])) + "|" + str(ph.index(cv.alleles[g][1]))) if "-" not in ph else "0/1"
                            f = [c, str(int(cv.pos[g])), str(int(cv.pos[g])), cv.uid[g], "1", "", "0", cv.alleles[g][0], cv.alleles[g][1],
                                 str(n0), str(n1), str(cov), ps, "1"]
                            if cfg.output_read_ids == 1:
                                qn = self.qnames[c]
                                f += [_jl(qn[int(x)] for x in per_allele[0]), _jl(qn[int(x)] for x in per_allele[1])]
                            f += [str(cv.maf[g]), self.bam_names[b], "", ""]
                            rows_a.append("\t".join(f) + "\n")
                dd = R["var_distinct"][g]
                ps = (str(ph.index(cv.alleles[g][0])) + "|" + str(ph.index(cv.alleles[g][1]))) if "-" not in ph else "-|-"
                name = cv.rsid[g] if cfg.unique_ids == 0 else cv.uid[g]
                hrow = "\t".join([c, str(int(cv.pos[g]) - 1), str(int(cv.pos[g])), "1", "1", name,
                                  cv.alleles[g][0] + "|" + cv.alleles[g][1], str(int(dd[0])), str(int(dd[1])), str(int(dd[0]) + int(dd[1])),
                                  "0", "0", ps, str(float("nan")), ps, str(float("nan"))]) + "\n"
                singles.append(((kb, kc, kl), "".join(rows_a), hrow))
        return singles


_FORK_ENGINE = None


def _fork_task(task):
    kind, c, lo, hi = task
    e = _FORK_ENGINE
    if kind == "conn":
        return e._conn_text(c, lo, hi)
    if kind == "alle":
        return e._allelic_rows(c, lo, hi)
    if kind == "blk":
        return e._block_rows(c, lo, hi)
    return e._single_rows(c, lo, hi, e._phased_sets[c])


HEAD_ASE = ["contig", "start", "stop", "variants", "variantCount", "variantsBlacklisted", "variantCountBlacklisted", "haplotypeA",
            "haplotypeB", "aCount", "bCount", "totalCount", "blockGWPhase", "gwStat", "max_haplo_maf", "bam", "aReads", "bReads"]
HEAD_HAP = ['contig', 'start', 'stop', 'length', 'variants', 'variant_ids', 'variant_alleles', 'reads_hap_a', 'reads_hap_b',
            'reads_total', 'edges_supporting', 'edges_total', 'annotated_phase', 'phase_concordant', 'gw_phase', 'gw_confidence']


def merge_fragments(frags: Dict[str, dict], chrom_list: List[str], cfg: "Config", noise: float):
    """Stage D (rank 0): assemble the five files from per-chromosome fragments in the reference's global order:
    chromosomes in VCF order for connections / blocks, first-appearance keys (BAM, chromosome, line) for
    allelic_counts and singletons.  Pure Python on plain data, so it is what the multi-GPU gather feeds."""
    cols = list(HEAD_ASE)
    if cfg.output_read_ids == 1:
        cols += ["read_ids_a", "read_ids_b"]
    conn = ["variant_a\tvariant_b\tsupporting_connections\ttotal_connections\tconflicting_configuration_p\tphase_concordant\n"]
    ase = ["\t".join(cols) + "\n"]
    hap = ["\t".join(HEAD_HAP) + "\n"]
    cfgf = ["\t".join(['variant_a', 'rsid_a', 'variant_b', 'rsid_b', 'configuration']) + "\n"]
    allelic = []
    singles = []
    dropped = phased = lines = 0
    lookup = {}
    block_index = 0
    for c in chrom_list:
        f = frags[c]
        conn += f["conn"]
        dropped += f["dropped"]; phased += f["phased"]; lines += f["lines"]
        allelic += [(tuple(k), r) for k, r in f["allelic"]]
        for ch in f["blocks"]:
            hap.append(ch["hap"]); ase.append(ch["ase"]); cfgf.append(ch["cfg"])
            if ch.get("vcf") is not None:
                for v in ch["vcf"]:
                    block_index += 1
                    for i, uid in enumerate(v["uids"]):
                        lookup[uid] = (v, i, block_index)
            else:
                block_index += ch["n"]
        singles += [(tuple(k), ra, rh) for k, ra, rh in f["singles"]]
    allelic.sort(key=lambda t: t[0])
    singles.sort(key=lambda t: t[0])
    for _, ra, rh in singles:
        ase.append(ra)
    for _, ra, rh in singles:
        hap.append(rh)
    out = {"variant_connections": "".join(conn),
           "allelic_counts": "contig\tposition\tvariantID\trefAllele\taltAllele\trefCount\taltCount\ttotalCount\n" + "".join(r for _, r in allelic),
           "haplotypic_counts": "".join(ase), "haplotypes": "".join(hap), "allele_config": "".join(cfgf)}
    log = ["     sequencing noise level estimated at %f" % noise,
           "     %d variant connections dropped because of conflicting configurations (threshold = %f)" % (dropped, cfg.cc_threshold),
           "     %d variants covered by at least 1 read" % len(allelic)]
    return out, {"log": log, "phased": phased, "lines": lines, "dropped": dropped, "covered": len(allelic), "vcf_lookup": lookup}
