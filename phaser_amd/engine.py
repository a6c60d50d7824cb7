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
from .soa import ReadShard, as16_plane
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
        self.device_rows = True        # stages T7-O2 on the GPU (phz_rowsdev_*); False: the host twin (phz_rows_format_multi), which also takes a pass the device stage refuses with PHZ_E_UNSUPPORTED (label text beyond 4 GiB)
        self.fetch_text = True         # copy the finished row text to (page-locked) host memory; False leaves it in HBM (bench)
        self.py_hash_order = 0         # 1: rows / read labels in the order CPython 3.10 gives the reference's sets (pyorder.py; needs PYTHONHASHSEED=0)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown option " + k)
            setattr(self, k, v)


def percentile_from_hist(h: np.ndarray, q: float) -> float:
    """numpy.percentile(scores, q) (default linear method, numpy/lib/_function_base_impl.py _quantile/_lerp) for the
    multiset scores = {bin - 32768 repeated h[bin] times}, computed from the histogram: same float64 operations on the
    two neighbouring order statistics, without materialising the scores (phaser.py:551)."""
    # alignment scores live in a narrow band of the 64Ki bins: find the rows of 256 bins that hold counts, work on that band only
    rows = np.flatnonzero(h.reshape(256, 256).any(axis=1)) if h.size == 65536 else np.zeros(0, np.int64)
    if h.size == 65536 and len(rows) == 0:
        return None                          # no alignment score at all
    first_bin = int(rows[0]) * 256 if len(rows) else 0
    h = h[first_bin:(int(rows[-1]) + 1) * 256] if len(rows) else h
    return percentile_from_band(h, first_bin, q)


def percentile_from_sparse(bins: np.ndarray, counts: np.ndarray, q: float) -> float:
    """The same from the occupied bins alone (phz_as_histogram_sparse: ascending bin numbers and their counts)."""
    if len(bins) == 0:
        return None
    first_bin = int(bins[0])
    h = np.zeros(int(bins[-1]) - first_bin + 1, dtype=np.int64)
    h[bins.astype(np.int64) - first_bin] = counts
    return percentile_from_band(h, first_bin, q)


def percentile_from_band(h: np.ndarray, first_bin: int, q: float) -> float:
    """h[i] = number of scores equal to first_bin + i - 32768."""
    n = int(h.sum())
    if n == 0:
        return None
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
    a = np.int64(int(np.searchsorted(csum, prev, side="right")) + first_bin - 32768)
    b = np.int64(int(np.searchsorted(csum, nxt, side="right")) + first_bin - 32768)
    diff = np.subtract(b, a)
    out = np.add(a, diff * gamma)
    if gamma >= 0.5:
        out = np.subtract(b, diff * (1 - gamma))
    return float(out)


def binom_cdf_dedup(k: np.ndarray, n: np.ndarray, p: float) -> np.ndarray:
    """scipy.stats.binom.cdf(k, n, p) (the reference's call, phaser.py:1649) evaluated once per DISTINCT (k, n): read counts are
    small integers, so millions of pairs share a few thousand distinct arguments.  Same function, same scalar arguments, same
    bits -- only fewer evaluations (165 ns each)."""
    if len(k) == 0:
        return np.zeros(0, dtype=np.float64)
    w = int(n.max()) + 1
    if w * w <= (1 << 24):
        key = n * w + k
        used = np.zeros(w * w, dtype=bool)
        used[key] = True
        uniq = np.flatnonzero(used)
        lut = np.empty(w * w, dtype=np.float64)
        lut[uniq] = binom.cdf(uniq % w, uniq // w, p)
        return lut[key]
    key = n.astype(np.int64) * w + k
    uniq, inv = np.unique(key, return_inverse=True)
    return binom.cdf(uniq % w, uniq // w, p)[inv]


class _Shard:
    def __init__(self, calls: Calls, qid, aln, has_as, n_reads, as16=None):
        self.calls = calls; self.qid = qid; self.aln = aln; self.has_as = has_as; self.n_reads = n_reads
        self.as16 = as16               # the AS column as one 2-byte plane (soa.as16_plane): what the tally kernels gather
        self.cutoff = 0.0; self.use_cutoff = 0
        self.cut_dev = None            # the BAM's cutoff block on the device (phz_as_cutoff_enqueue), when the percentile never visited the host
        self.as_absmax = None


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
        self._log: List[str] = []
        self._pending_cut = []         # (log index, device block, shards) of AS cutoffs still on the device (close_bam -> resolve_cutoffs)
        self._lazy_cut = []            # (BAM, shards with lines, device block, shards, log index) of percentiles not yet enqueued (close_bam -> _issue_cutoffs)
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
                                               None if has_as is None else has_as.contiguous(), shard.n, as16_plane(shard, self.ctx if getattr(self, "lib", None) is not None else None))
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
                                          self.cfg.baseq, aux=False)          # the phasing stage reads (record, variant, code) only
            for it, c in zip(batch, calls):
                self.add_mapped(bam_index, it[0], it[1], c, it[2], it[3] if len(it) > 3 else None)
        for it in rest:
            self.add_shard(bam_index, it[0], it[1], it[2], it[3] if len(it) > 3 else None)

    def add_mapped(self, bam_index: int, chrom: str, shard: ReadShard, calls: Calls, n_qid: int, qnames: Optional[List[str]] = None):
        """Attach a shard whose K_map call list already exists."""
        has_as = shard.has_as
        self.shards[chrom][bam_index] = _Shard(calls, shard.qid.contiguous(), shard.aln_score.contiguous(),
                                               None if has_as is None else has_as.contiguous(), shard.n, as16_plane(shard, self.ctx if getattr(self, "lib", None) is not None else None))
        self.n_qid[chrom] = max(self.n_qid[chrom], n_qid)
        if qnames is not None:
            self.qnames[chrom] = qnames

    def _lines(self, sh: _Shard, bam_index: int, var_base: int = 0, qid_base: int = 0) -> _lib.phz_lines:
        # the pointer fields never change for a shard: built once (eight data_ptr() calls and a ctypes struct: ~20 us, x 22 shards x 2 stages
        # per pass), the scalars refreshed
        ln = getattr(sh, "_ln", None)
        if ln is None or sh._ln_n != sh.calls.n:
            c = sh.calls
            cached = c.__dict__.get("_phz_ln") if hasattr(c, "__dict__") else None      # the same call list handed to a new Engine (every pass of the bench)
            if cached is not None and cached[1] is c.read_idx and cached[2] is sh.qid and cached[3] is sh.aln and cached[4] is sh.has_as and cached[5] == c.n \
                    and cached[6] is sh.as16:
                ln = cached[0]
            else:
                ln = _lib.phz_lines(c.n, _p(c.read_idx), _p(c.var_idx), _p(c.code), sh.n_reads, _p(sh.qid), _p(sh.aln), _p(sh.has_as), 0.0, 0, 0, 0, 0, _p(sh.as16))
                if hasattr(c, "__dict__"):
                    c.__dict__["_phz_ln"] = (ln, c.read_idx, sh.qid, sh.aln, sh.has_as, c.n, sh.as16)
            sh._ln = ln; sh._ln_n = c.n
        ln.as_cutoff = float(sh.cutoff); ln.use_cutoff = int(sh.use_cutoff); ln.bam_index = bam_index; ln.var_base = var_base; ln.qid_base = qid_base
        ln.as_cutoff_dev = _p(sh.cut_dev) if sh.cut_dev is not None else None
        return ln

    def close_bam(self, bam_index: int):
        """AS quantile cutoff of one BAM over all its chromosomes (phaser.py:545-553)."""
        shards = [self.shards[c][bam_index] for c in self.chrom_list if self.shards[c][bam_index] is not None]
        if self.cfg.as_q_cutoff > 0:
            dev = shards[0].calls.read_idx.device if shards else self.mapper.device
            hb = self.mapper.__dict__.get("_as_hist")            # one device histogram (and its page-locked host copy) per mapper, not per pass
            if hb is None or hb[0].device != dev:
                hb = (torch.zeros(_lib.PHZ_AS_BINS, dtype=torch.int64, device=dev),
                      torch.zeros(_lib.PHZ_AS_BINS, dtype=torch.int64, pin_memory=(dev.type == "cuda")))
                try:
                    self.mapper._as_hist = hb
                except Exception:
                    pass
            hist = hb[0]
            live = [sh for sh in shards if sh.calls.n]
            cutoff = None; done = False
            if live and dev.type == "cuda" and not pdist.collectives_live():
                # one rank: nothing to all-reduce, so the histogram stays on the device and only its occupied bins come back (a few hundred
                # bytes instead of 512 KB; one call, one host wait)
                # (histogram -> occupied bins -> numpy.percentile's formula, all inside phz_as_cutoff; percentile_from_sparse is its Python twin)
                arr = (_lib.phz_lines * len(live))(*[self._lines(sh, bam_index) for sh in live])
                torch.cuda.current_stream(dev).synchronize()
                if os.environ.get("PHZ_AS_CUTOFF_HOST") != "1":
                    # ... and the percentile itself stays there too (phz_as_cutoff_enqueue: no host wait between the histogram and the tally; the kernels of the
                    # tally read the cutoff from the block).  The value reaches the log when somebody asks for it (resolve_cutoffs: the CLI after every BAM,
                    # finish() at the latest)
                    blk = torch.empty(4, dtype=torch.float64, device=dev)          # this Engine's own (another Engine on the same mapper may still hold a pending block)
                    for sh in shards:
                        sh.cut_dev = blk; sh.use_cutoff = 1; sh.cutoff = 0.0
                    # ... and it is ISSUED right in front of the tally that reads it (_tally_genome -> _issue_cutoffs), or when somebody asks for the value: kernels
                    # enqueued here would finish long before the host has put the tally call together, and the GPU would wait for it
                    self._lazy_cut.append((bam_index, live, blk, shards, len(self._log)))
                    self._log.append(None)          # placeholder of the BAM's log line
                    return
                val = C.c_double(0.0); found = C.c_int32(0)
                st = self.ctx.check(self.lib.phz_as_cutoff(self.ctx.h, arr, len(live), float(self.cfg.as_q_cutoff * 100), C.byref(val), C.byref(found)),
                                    allow=(_lib.PHZ_E_CAPACITY,))
                if st == 0:
                    cutoff = float(val.value) if found.value else None
                    done = True
            if not done:
                cutoff = self._as_cutoff_dense(hb, live, dev, bam_index)
            if cutoff is not None:
                self._log.append("          using alignment score cutoff of %d" % cutoff)
                for sh in shards:
                    sh.cutoff = float(cutoff); sh.use_cutoff = 1
            else:
                self._log.append("          no alignment score value found in reads, cannot use cutoff")

    @property
    def log(self) -> List[str]:
        """The log lines of the run (the reference's stdout lines of the stages this class covers).  Reading them brings in whatever is still on the device
        (the AS cutoffs of close_bam): a caller that streams passes and never looks at the log pays no host wait for them."""
        if self._pending_cut or self._lazy_cut:
            self.resolve_cutoffs()
        return self._log

    @log.setter
    def log(self, value):
        self._log = value

    def _issue_cutoffs(self):
        """Enqueue the device-side AS percentiles close_bam left for later (one small submission per BAM on the ctx stream, no host wait)."""
        if not self._lazy_cut:
            return
        lazy, self._lazy_cut = self._lazy_cut, []
        for bam_index, live, blk, shards, at in lazy:
            arr = (_lib.phz_lines * len(live))(*[self._lines(sh, bam_index) for sh in live])
            self.ctx.check(self.lib.phz_as_cutoff_enqueue(self.ctx.h, arr, len(live), float(self.cfg.as_q_cutoff * 100), _p(blk)))
            self._pending_cut.append((at, blk, shards))

    def resolve_cutoffs(self):
        """The AS cutoffs that were computed on the device without a host wait (close_bam): read them back (one small copy per BAM), write the BAM's log line
        (phaser.py:552-553) where it belongs, refuse an input whose AS values do not fit int16."""
        self._issue_cutoffs()
        if not self._pending_cut:
            return
        pend, self._pending_cut = self._pending_cut, []
        self.ctx.check(self.lib.phz_ctx_sync(self.ctx.h))
        for at, blk, shards in pend:
            v = blk.cpu().numpy()
            if v[2] != 0:
                raise _lib.PhzError(_lib.PHZ_E_UNSUPPORTED, "AS value outside int16")
            if v[1] != 0:
                self._log[at] = "          using alignment score cutoff of %d" % float(v[0])
                for sh in shards:
                    sh.cutoff = float(v[0])
            else:
                self._log[at] = "          no alignment score value found in reads, cannot use cutoff"
                for sh in shards:
                    sh.use_cutoff = 0; sh.cut_dev = None

    def _as_cutoff_dense(self, hb, live, dev, bam_index):
        """The dense 64 Ki-bin histogram (all-reduced over the ranks) -> cutoff."""
        hist = hb[0]
        hist.zero_()
        if live and dev.type == "cuda":         # every shard of the BAM in one submission (it also refuses AS values outside int16)
            arr = (_lib.phz_lines * len(live))(*[self._lines(sh, bam_index) for sh in live])
            torch.cuda.synchronize(dev)
            self.ctx.check(self.lib.phz_as_histogram_batch(self.ctx.h, arr, len(live), _p(hist)))
        else:
            for sh in live:
                if sh.as_absmax is None:
                    sh.as_absmax = int(sh.aln.abs().max())
                if sh.as_absmax >= 32768:
                    raise _lib.PhzError(_lib.PHZ_E_UNSUPPORTED, "AS value outside int16")
                ln = self._lines(sh, bam_index)
                self.ctx.check(self.lib.phz_as_histogram(self.ctx.h, C.byref(ln), _p(hist), _lib.PHZ_HOST))
        pdist.allreduce_sum_(hist)          # the quantile is over ALL chromosomes of this BAM
        if dev.type == "cuda":
            hb[1].copy_(hist, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            h = hb[1].numpy()
        else:
            h = hist.numpy()
        return percentile_from_hist(h, self.cfg.as_q_cutoff * 100)

    # ---------------------------------------------------------------- stages 3-6
    def _bases(self):
        """Joint index spaces of this rank's chromosomes (VCF order): first variant / first QNAME id of each."""
        vb = {}; qb = {}; v = q = 0
        for c in self.chrom_list:
            vb[c] = v; qb[c] = q
            v += len(self.vs.chroms[c]); q += max(1, self.n_qid[c])
        return vb, qb, v, q

    def _pinned(self, name: str, count: int, dtype):
        """numpy array over page-locked host memory, kept per name and grown on demand (D2H at full PCIe rate).  The pool belongs to
        THIS Engine (rowsdev.PinnedPool): the arrays a pass hands out are overwritten by this Engine's next pass only, never by another
        Engine of the process; a dead Engine's buffers are reused by the next one, so a stream of Engines page-locks once."""
        from . import rowsdev
        dt = np.dtype(dtype)
        need = max(1, count) * dt.itemsize
        return rowsdev.pool_of(self).get("tally_" + name, need).view(dt)[:count]

    def _tally_genome(self) -> dict:
        """K_tally over every (chromosome, BAM) shard of this rank in ONE submission; results fetched into pinned host arrays."""
        import time as _t
        nb = len(self.bam_names)
        vb, qb, NV, NQ = self._bases()
        lines = []; line_base = {}; total = 0
        dev = None
        for c in self.chrom_list:
            for b, sh in enumerate(self.shards[c]):
                if sh is None:
                    continue
                dev = sh.calls.read_idx.device
                lines.append(self._lines(sh, b, vb[c], qb[c]))
                line_base[(c, b)] = (total, sh.calls.n)
                total += sh.calls.n
        if dev is None:
            dev = torch.device("cpu")
        space = _lib.PHZ_DEVICE if dev.type == "cuda" else _lib.PHZ_HOST
        t0 = _t.perf_counter()
        key = (str(dev), tuple(self.chrom_list))
        cached = self.vs.__dict__.get("_allele_codes")
        if cached is None or cached[0] != key:                  # allele base codes of the joint variant space: constant for the variant set
            a0 = np.concatenate([np.full(len(self.vs.chroms[c]), 255, np.uint8) if self.vs.chroms[c].is_general else self.vs.chroms[c].a0
                                 for c in self.chrom_list]) if self.chrom_list else np.zeros(0, np.uint8)
            a1 = np.concatenate([np.full(len(self.vs.chroms[c]), 255, np.uint8) if self.vs.chroms[c].is_general else self.vs.chroms[c].a1
                                 for c in self.chrom_list]) if self.chrom_list else np.zeros(0, np.uint8)
            a0 = np.ascontiguousarray(a0, dtype=np.uint8); a1 = np.ascontiguousarray(a1, dtype=np.uint8)
            if space == _lib.PHZ_DEVICE:
                a0 = torch.from_numpy(a0).to(dev); a1 = torch.from_numpy(a1).to(dev)
                torch.cuda.synchronize(dev)
            cached = self.vs.__dict__["_allele_codes"] = (key, a0, a1)
        if space == _lib.PHZ_DEVICE:
            pa0, pa1 = _p(cached[1]), _p(cached[2])
        else:
            pa0, pa1 = C.c_void_p(cached[1].ctypes.data), C.c_void_p(cached[2].ctypes.data)
        arr = (_lib.phz_lines * max(1, len(lines)))(*lines)
        sz = _lib.phz_tally_sizes()
        self._issue_cutoffs()          # (the percentiles of close_bam: on the stream right in front of the tally that reads them)
        pair_stage = None
        if space == _lib.PHZ_DEVICE and self.cfg.device_rows and os.environ.get("PHZ_TALLY_PAIRS_FUSED", "1") == "1":
            # the device row stage follows: its first stage (distinct read-count pairs + the p-value-independent sorts) is issued by the same native call as the tally
            from . import rowsdev
            T, keys, n_slots = rowsdev.pair_stage_inputs(self)
            pst = C.c_int32(0)
            self.ctx.check(self.lib.phz_tally_pairs(self.ctx.h, arr, len(lines), NV, pa0, pa1, NQ, nb, C.byref(sz), space, T.h, C.c_void_p(keys.ctypes.data), C.byref(pst)))
            pair_stage = (keys, n_slots, int(pst.value), T)
        else:
            self.ctx.check(self.lib.phz_tally(self.ctx.h, arr, len(lines), NV, pa0, pa1, NQ, nb, C.byref(sz), space))
        t1 = _t.perf_counter()
        G = {"nv": NV, "nb": nb, "var_base": vb, "line_base": line_base, "n_lines": int(sz.n_lines), "n_kept": int(sz.n_kept),
             "n_edges": int(sz.n_edges), "n_read_list": int(sz.n_read_list), "noise": (int(sz.noise_match), int(sz.noise_mismatch)),
             "resident": True,          # the results are still in HBM: the device row stage and phz_components use them in place
             "fetched": False}
        if pair_stage is not None:
            G["pair_stage"] = pair_stage
        self.stats["tally_call_s"] = self.stats.get("tally_call_s", 0.0) + t1 - t0
        return G

    def _fetch_tally(self):
        """Copy the resident tally results into pinned host arrays (what the HOST row stage and the tools read)."""
        import time as _t
        G = self.G
        if G.get("fetched", True):
            return
        t1 = _t.perf_counter()
        NV = G["nv"]; nb = G["nb"]; ne = G["n_edges"]; nrl = G["n_read_list"]
        G.update({"var_count": self._pinned("var_count", NV * 3, np.int32), "var_first": self._pinned("var_first", NV, np.int64),
                  "var_distinct": self._pinned("var_distinct", NV * 3, np.int32), "var_rank": self._pinned("var_rank", NV, np.uint64),
                  "ea": self._pinned("ea", ne, np.int32), "eb": self._pinned("eb", ne, np.int32), "cto": self._pinned("cto", ne * 3, np.int32),
                  "linked": self._pinned("linked", ne, np.uint8), "stats": self._pinned("stats", ne * 5, np.int32),
                  "rl_start": self._pinned("rl_start", NV * 2 * nb + 1, np.uint32), "rl_qid": self._pinned("rl_qid", nrl, np.int32)})
        vp = lambda a: C.c_void_p(a.ctypes.data) if a.size else None
        out = _lib.phz_tally_out(vp(G["var_count"]), vp(G["var_first"]), vp(G["var_distinct"]), vp(G["var_rank"]), None, vp(G["ea"]), vp(G["eb"]),
                                 None, vp(G["linked"]), vp(G["cto"]), vp(G["rl_start"]), vp(G["rl_qid"]), vp(G["stats"]))
        self.ctx.check(self.lib.phz_tally_fetch(self.ctx.h, C.byref(out), _lib.PHZ_HOST))
        G["var_count"] = G["var_count"].reshape(NV, 3); G["var_distinct"] = G["var_distinct"].reshape(NV, 3); G["cto"] = G["cto"].reshape(ne, 3)
        G["stats"] = G["stats"].reshape(5, ne)       # planes: same-configuration, opposite, supporting, total, chosen configuration
        G["fetched"] = True
        self.stats["tally_d2h_s"] = self.stats.get("tally_d2h_s", 0.0) + _t.perf_counter() - t1

    def chrom_view(self, c: str) -> dict:
        """One chromosome's part of the tally results, local variant indices (views; for tests and tools).  The views sit in this
        Engine's page-locked buffers: valid until THIS Engine's next pass (no other Engine touches them); copy what must outlive it."""
        self._fetch_tally()
        G = self.G; v0 = G["var_base"][c]; nv = len(self.vs.chroms[c])
        lo = int(np.searchsorted(G["ea"], v0, side="left")); hi = int(np.searchsorted(G["ea"], v0 + nv, side="left"))
        vc = G["var_count"][v0:v0 + nv]
        return {"nv": nv, "var_count": vc, "var_distinct": G["var_distinct"][v0:v0 + nv], "var_first": G["var_first"][v0:v0 + nv],
                "var_rank": G["var_rank"][v0:v0 + nv], "ea": G["ea"][lo:hi] - v0, "eb": G["eb"][lo:hi] - v0, "cto": G["cto"][lo:hi],
                "linked": G["linked"][lo:hi].astype(bool), "kept": int(vc.sum())}

    def kept_lines(self) -> dict:
        """Host arrays of the kept call lines, per chromosome and BAM in line order: (QNAME id, variant index, class 0 ref / 1 alt / 2 other) -- what
        pyorder.replay walks.  The classes come from the resident tally (phz_tally_fetch line_cls)."""
        G = self.G
        n = G["n_lines"]
        cls_all = np.zeros(max(1, n), dtype=np.uint8)
        o = _lib.phz_tally_out(None, None, None, None, C.c_void_p(cls_all.ctypes.data), None, None, None, None, None, None, None, None)
        self.ctx.check(self.lib.phz_tally_fetch(self.ctx.h, C.byref(o), _lib.PHZ_HOST))
        out = {}
        for c in self.chrom_list:
            per = []
            for b, sh in enumerate(self.shards[c]):
                if sh is None or (c, b) not in G["line_base"]:
                    per.append(None); continue
                base, m = G["line_base"][(c, b)]
                cls = cls_all[base:base + m]
                keep = cls != 255
                ri = sh.calls.read_idx.cpu().numpy().astype(np.int64)
                qid = sh.qid.cpu().numpy()[ri]
                per.append((qid[keep], sh.calls.var_idx.cpu().numpy()[keep], cls[keep]))
            out[c] = per
        return out

    def tally_all(self):
        """Stage A: K_tally over this rank's chromosomes; returns the two global noise counters of these chromosomes."""
        self.G = self._tally_genome()
        return self.G["noise"]            # k_noise: the two counters of phaser.py:610-632 over these chromosomes' variants

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
        chunks=True returns, per file, the list of buffers in output order (no join pass): write them with dist.write_chunks (with
        several ranks some of them are byte ranges of the other ranks' spool files) and call dist.cleanup_spool() on EVERY rank
        afterwards.  The chunks are views of this Engine's own page-locked buffers (rowsdev.PinnedPool): write or copy them before
        THIS Engine runs its next pass; passes of other Engines in the process never overwrite them."""
        import time as _t
        if self.cfg.py_hash_order and pdist.world()[1] > 1:
            # refused on EVERY rank before the first collective (a rank-0-only refusal after the gather left the others in a barrier)
            raise _lib.PhzError(_lib.PHZ_E_UNSUPPORTED, "py_hash_order needs all chromosomes on one rank")
        t0 = _t.perf_counter()
        match, mism = pdist.allreduce_counts(*self.tally_all())
        noise = self.noise_from_counts(match, mism)
        t1 = _t.perf_counter()
        local = self._fragments(noise)
        t2 = _t.perf_counter()
        frags = pdist.gather_fragments(local, getattr(self, "spool_dir", None), self.all_chroms)
        self.stats.update({"tally_s": t1 - t0, "fragments_s": t2 - t1})
        self.noise = noise
        if frags is None:
            if not chunks:
                pdist.cleanup_spool()       # waits (barrier) until rank 0 has read the spooled row text
            return None
        t3 = _t.perf_counter()
        chroms = [c for c in self.all_chroms if c in frags]
        blocks_order = block_chrom_order(frags, chroms)
        out, summary = merge_fragments(frags, chroms, self.cfg, noise, len(self.bam_names), blocks_order)
        # per chromosome with blocks: (name, block arrays of phz_rows_format, blocks before it) -- what write_vcf needs
        self.vcf_blocks = []
        if self.cfg.want_vcf or self.cfg.py_hash_order:
            block_index = 0
            for c in blocks_order:                 # block numbers (PI of write_vcf, phaser.py:863-867) follow the order of the block files
                if frags[c]["vcf"] is not None:
                    self.vcf_blocks.append((c, frags[c]["vcf"], block_index))
                    block_index += len(frags[c]["vcf"]["size"])
        self.stats["merge_s"] = _t.perf_counter() - t3
        self._log += summary["log"]
        self.phased = summary["phased"]; self.total_lines = summary["lines"]
        if self.cfg.py_hash_order:
            # raw-byte tier: replay the reference's set constructions over the same strings (an exactness mode: pure Python over every call line)
            from . import pyorder
            import os as _os
            raw = {k: b"".join(pdist.as_bytes(x) for x in v) for k, v in out.items()}
            if _os.environ.get("PHZ_PYORDER_PYTHON") == "1":          # the pure-Python twin with real set objects (CPython 3.10 under PYTHONHASHSEED=0 only)
                text = pyorder.replay(self, {k: v.decode() for k, v in raw.items()})
                raw = {k: v.encode() for k, v in text.items()}
            else:                                                       # libphz's restatement of the str hash and the set: any interpreter, seconds at genome scale
                raw = pyorder.replay_native(self, raw)
            if chunks:
                return {k: [v] for k, v in raw.items()}
            return raw if binary else {k: v.decode() for k, v in raw.items()}
        if chunks:
            return out
        out = {k: b"".join(pdist.as_bytes(x) for x in v) for k, v in out.items()}
        pdist.cleanup_spool()
        return out if binary else {k: v.decode() for k, v in out.items()}

    def _fragments(self, noise: float) -> Dict[str, dict]:
        """Stage C for every owned chromosome: C1 = pair tests, pruning, components, ordering keys (numpy / scipy / GPU, the
        heavy parts once over all chromosomes); C2 = block phasing + row text, all chromosomes through ONE native thread pool."""
        import time as _t
        from . import rowsdev
        G = self.G
        if self.cfg.device_rows and G.get("resident") and getattr(self, "lib", None) is not None and rowsdev.supported(self.cfg):
            try:
                t0 = _t.perf_counter()
                frags = rowsdev.run(self, noise, fetch_text=self.cfg.fetch_text)
                self.stats["rows_device_s"] = self.stats.get("rows_device_s", 0.0) + _t.perf_counter() - t0
                self.rows_path = "device"
                return frags
            except _lib.PhzError as e:
                if e.status != _lib.PHZ_E_UNSUPPORTED:
                    raise
                self.rows_fallback = str(e)          # the host stage takes the pass (a limit of the device stage, named in the message)
        self.rows_path = "host"
        self._fetch_tally()
        t0 = _t.perf_counter()
        frags = self._prepare(noise)
        t1 = _t.perf_counter()
        done = rows.format_chroms(self, self.chrom_list, self.cfg.host_threads)
        for c in self.chrom_list:
            frags[c].update(done[c])
        self._pre = {}
        self.stats["prepare_s"] = self.stats.get("prepare_s", 0.0) + t1 - t0
        self.stats["rows_s"] = self.stats.get("rows_s", 0.0) + _t.perf_counter() - t1
        return {c: frags[c] for c in self.chrom_list}

    def _prepare(self, noise: float) -> Dict[str, dict]:
        """Stage C1.  Genome-wide: the nine cells -> supporting / total counts, the binomial test (scipy, the reference's own
        call at phaser.py:1649), pruning, connected components on the GPU.  Per chromosome (slices of those arrays, local
        variant indices): row orders, component lists, first-appearance keys -> self._pre[c] for the row writer."""
        import time as _t
        cfg = self.cfg
        G = self.G
        NV = G["nv"]
        tp0 = _t.perf_counter()
        # ---- test every linked pair (phaser.py:1594-1654)
        # the three sums per pair (same configuration / opposite / other) come from the device (k_edge_final)
        linked = G["linked"].view(bool)
        st = G["stats"]
        if linked.all():
            sel = np.arange(len(linked)); ea_g = G["ea"]; eb_g = G["eb"]
            cis, trans, sup, tot, cfgv = st[0], st[1], st[2], st[3], st[4]
        else:
            sel = np.nonzero(linked)[0]
            ea_g = G["ea"][sel]; eb_g = G["eb"][sel]
            cis, trans, sup, tot, cfgv = (np.ascontiguousarray(st[k][sel]) for k in range(5))
        prob = 1 - ((6 * noise) + (10 * math.pow(noise, 2)))
        pv = np.ones(len(sel), dtype=np.float64)
        tp1 = _t.perf_counter()
        test = (sup > 0) & ((tot - sup) > 0)
        if test.any():
            pv[test] = binom_cdf_dedup(sup[test], tot[test], prob)
        pv[sup == 0] = 0.0
        keep_edge = ~(pv < cfg.cc_threshold)
        # ---- connected components of the surviving graph on the GPU (phaser.py:1861-1882)
        tp2 = _t.perf_counter()
        keep_all = np.zeros(len(G["ea"]), dtype=np.uint8)
        keep_all[sel[keep_edge]] = 1
        label_all = self._component_labels(keep_all)
        tp3 = _t.perf_counter()
        for k_, v_ in (("prep_cells_s", tp1 - tp0), ("prep_binom_s", tp2 - tp1), ("prep_components_s", tp3 - tp2)):
            self.stats[k_] = self.stats.get(k_, 0.0) + v_
        frags: Dict[str, dict] = {}
        self._pre = {}
        vb = G["var_base"]
        keep_u8 = keep_edge.astype(np.uint8)
        self._label_all = label_all
        bounds = np.searchsorted(ea_g, [vb[c] for c in self.chrom_list] + [NV], side="left")
        for ci, c in enumerate(self.chrom_list):
            nv = len(self.vs.chroms[c]); v0 = vb[c]
            lo = int(bounds[ci]); hi = int(bounds[ci + 1])
            vc = G["var_count"][v0:v0 + nv]
            frags[c] = {"chrom": c, "lines": int(vc.sum()), "dropped": int(hi - lo - int(keep_u8[lo:hi].sum()))}
            # the ordering stage (row orders, component lists, first-appearance keys) runs inside the native row writer on these
            # slices (phz_rows_in.raw)
            self._pre[c] = {"v0": v0, "nv": nv, "ea": ea_g[lo:hi], "eb": eb_g[lo:hi], "cis": cis[lo:hi], "trans": trans[lo:hi],
                            "sup": sup[lo:hi], "tot": tot[lo:hi], "cfgv": cfgv[lo:hi], "pv": pv[lo:hi], "keep": keep_u8[lo:hi],
                            "chrom_index": self.all_chroms.index(c)}
        return frags

    def _component_labels(self, keep_all):
        """Connected-component label (smallest member) per variant of the surviving graph, on the GPU (phz_components), using the
        edge list K_tally left in HBM."""
        G = self.G; NV = G["nv"]
        label = self._pinned("label", NV, np.int32)
        if NV == 0:
            return label
        st = self.lib.phz_components(self.ctx.h, NV, len(keep_all), None, None, C.c_void_p(keep_all.ctypes.data) if len(keep_all) else None,
                                     C.c_void_p(label.ctypes.data), _lib.PHZ_HOST)
        self.ctx.check(st)
        return label


HEAD_ASE = ["contig", "start", "stop", "variants", "variantCount", "variantsBlacklisted", "variantCountBlacklisted", "haplotypeA",
            "haplotypeB", "aCount", "bCount", "totalCount", "blockGWPhase", "gwStat", "max_haplo_maf", "bam", "aReads", "bReads"]
HEAD_HAP = ['contig', 'start', 'stop', 'length', 'variants', 'variant_ids', 'variant_alleles', 'reads_hap_a', 'reads_hap_b',
            'reads_total', 'edges_supporting', 'edges_total', 'annotated_phase', 'phase_concordant', 'gw_phase', 'gw_confidence']


def block_chrom_order(frags: Dict[str, dict], chrom_list: List[str]) -> List[str]:
    """The chromosomes in the order the reference lists them in variant_connections / haplotypes / haplotypic_counts / allele_config: `read_vars` -- whose key
    order becomes that of dict_variant_overlap (phaser.py:646-650) and of the blocks (:784) -- is keyed by the chromosome `process_mapping_result` returns, which is
    "" for a call file WITHOUT a kept line (:1299), and new keys are appended BAM after BAM (:573-574).  So a chromosome takes its place with the first BAM that
    has a kept line on it, chromosomes of one BAM in VCF order (= chrom_list order); one with no kept line anywhere has no rows.  With one BAM, or whenever the
    first BAM covers every chromosome, this is the VCF order (found by tools/stress_parity.py in round 5: 11 sparse BAMs x 3 chromosomes)."""
    key = {c: ((frags[c].get("first_bam1", 0) or (1 << 30)), i) for i, c in enumerate(chrom_list)}
    return sorted(chrom_list, key=lambda c: key[c])


def merge_fragments(frags: Dict[str, dict], chrom_list: List[str], cfg: "Config", noise: float, n_bams: int = 1, blocks_order: List[str] = None):
    """Stage D (rank 0): order the per-chromosome fragments of the five files in the reference's global order (returns, per
    file, the list of buffers to write one after the other):
    connections / blocks by chromosome in block_chrom_order (the VCF order unless the first BAM misses a chromosome); allelic_counts and singleton rows
    follow the first-appearance keys (BAM of the first kept line, chromosome, line), i.e. per first BAM the chromosomes in VCF order.  Works on
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
    for c in (blocks_order if blocks_order is not None else block_chrom_order(frags, chrom_list)):
        f = frags[c]
        conn += f["conn"]; hap += f["hap"]; ase += f["ase"]; cfgf += f["cfg"]
        dropped += f["dropped"]; phased += f["phased"]; lines += f["lines"]; covered += f["allelic_rows"]
    for key, dst in (("allelic", allelic), ("single_ase", ase), ("single_hap", hap)):
        for b in range(n_bams):
            for c in chrom_list:
                f = frags[c]
                dst += [p for p, pb in zip(f[key], f[key + "_bam"]) if pb == b]
    out = {"variant_connections": conn, "allelic_counts": allelic, "haplotypic_counts": ase, "haplotypes": hap, "allele_config": cfgf}
    log = ["     sequencing noise level estimated at %f" % noise,
           "     %d variant connections dropped because of conflicting configurations (threshold = %f)" % (dropped, cfg.cc_threshold),
           "     %d variants covered by at least 1 read" % covered]
    return out, {"log": log, "phased": phased, "lines": lines, "dropped": dropped, "covered": covered}
