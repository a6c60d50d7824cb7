"""Glue for the device row stage (phz_rowsdev_* in libphz.so, phaser_amd/csrc/phz_rowsdev.hip): stages T7-O2 of the phasing path on the
GPU -- pair-test bookkeeping, pruning, components, ordering, block phasing (phase_v3), haplotype read sets and the text of the five
output files (phaser/phaser.py:686-726, :1861-1882, :2107-2324, :691-695, :737-749, :865-1239) -- on the results phz_tally left in HBM.

The only arithmetic kept on the host is the reference's own third-party call: scipy.stats.binom.cdf (phaser.py:1649), evaluated once per
DISTINCT (supporting, total) read-count pair the device reports (a few thousand per genome), with the same scalar arguments.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import sys

import numpy as np
import torch

from . import _lib
from .vcf import sep_pool


def _vp(a):
    if a is None:
        return None
    return C.c_void_p(a.ctypes.data) if isinstance(a, np.ndarray) else C.cast(C.c_char_p(a), C.c_void_p)


class Tables:
    """The per-variant tables of one rank's chromosomes in HBM (uploaded once per variant set and chromosome list)."""

    def __init__(self, ctx, vs, chrom_list: List[str], cfg):
        self.ctx = ctx; self.lib = ctx.lib
        self.chrom_list = list(chrom_list)
        cvs = [vs.chroms[c] for c in chrom_list]
        n = [len(cv) for cv in cvs]
        self.nv = int(sum(n))
        v0 = np.zeros(len(cvs) + 1, dtype=np.int64)
        np.cumsum(n, out=v0[1:])
        self.chrom_v0 = v0

        def joint(name, per_item=1):
            offs = []; parts = []; base = 0
            for cv in cvs:
                off, b = cv.pools()[name]
                offs.append(off[:-1].astype(np.int64) + base)
                parts.append(bytes(b)); base += len(b)
            if base >= 2 ** 32:
                raise _lib.PhzError(_lib.PHZ_E_UNSUPPORTED, "string tables beyond 4 GiB")
            offs.append(np.array([base], dtype=np.int64))
            return np.ascontiguousarray(np.concatenate(offs).astype(np.uint32)), b"".join(parts) or b"\n"

        cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs).astype(dt)) if xs else np.zeros(0, dt)
        keep = self._keep = []
        uid = joint("uid"); rsid = joint("rsid"); alle = joint("allele"); maf = joint("maf")
        cn = sep_pool(list(chrom_list))
        pos = cat([cv.pos for cv in cvs], np.int32); mafv = cat([cv.pools()["maf_val"] for cv in cvs], np.float64)
        is_ref = cat([cv.is_ref for cv in cvs], np.uint8); phase = cat([cv.phase_idx for cv in cvs], np.int8)
        bl = None
        marks = []
        for c, cv in zip(chrom_list, cvs):
            m = cv.blacklisted if getattr(cv, "blacklisted", None) is not None and len(cv.blacklisted) == len(cv) else np.zeros(len(cv), np.uint8)
            if cfg.haplo_blacklist:
                named = np.fromiter((c + "_" + str(int(p)) in cfg.haplo_blacklist for p in cv.pos), dtype=np.uint8, count=len(cv))
                m = m | named
            marks.append(m)
        if marks and any(m.any() for m in marks):
            bl = cat(marks, np.uint8)
        keep += [uid, rsid, alle, maf, cn, pos, mafv, is_ref, phase, bl, v0]
        t = _lib.phz_rowsdev_tables(self.nv, len(cvs), _vp(v0), _vp(cn[0]), _vp(cn[1]), _vp(pos), _vp(uid[0]), _vp(uid[1]), _vp(rsid[0]), _vp(rsid[1]),
                                    _vp(alle[0]), _vp(alle[1]), _vp(maf[0]), _vp(maf[1]), _vp(mafv), _vp(is_ref), _vp(phase), _vp(bl))
        h = C.c_void_p()
        ctx.check(self.lib.phz_rowsdev_create(ctx.h, C.byref(t), C.byref(h)))
        self.h = h
        self._keep = []            # the library holds its own copies

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.phz_rowsdev_destroy(self.h)
                self.h = None
        except Exception:
            pass


_TABLES_LOCK = __import__("threading").Lock()


def tables_for(eng, ctx=None) -> Tables:
    """Cached on the variant set: one upload per (device, chromosome list, blacklist).  The tables are plain device memory of the
    ctx's GPU -- any phz_ctx of that device may run the row stage on them -- so a helper thread uploads them through a ctx of its
    own (`ctx`) while the Engine's ctx is busy elsewhere; the upload has completed when the constructor returns."""
    with _TABLES_LOCK:
        cache = eng.vs.__dict__.setdefault("_rowsdev_tables", {})
        key = (eng.ctx.device, tuple(eng.chrom_list), frozenset(eng.cfg.haplo_blacklist))
        t = cache.get(key)
        if t is None:
            cache.clear()              # one resident copy per variant set is enough (a new chromosome list replaces the old tables)
            t = cache[key] = Tables(ctx or eng.ctx, eng.vs, eng.chrom_list, eng.cfg)
        return t


class PinnedPool:
    """Page-locked host buffers of ONE owner (an Engine), kept per name and grown on demand: D2H at the full PCIe rate and no page-locking
    cost per pass.  Two live owners never share memory: what an Engine hands out (the G[...] arrays, chrom_view() views, the text chunks of
    finish()) is overwritten only by THAT Engine's next pass.  Buffers return to a process-wide free list when their owner goes away, so a
    process that creates Engines one after the other (a sample stream) still page-locks once -- EXCEPT buffers somebody outside still holds
    views of (`out = Engine(...).finish(chunks=True)` keeps the chunks after the Engine is collected): those are left to their holders and
    freed with the last view, never recycled under them.  Every view handed out is a slice of the buffer's root ndarray, so the root's
    reference count says whether views are alive."""

    _free: List[np.ndarray] = []            # root arrays of buffers given up by dead owners, reusable by the next pool
    _arena = {"buf": None, "used": 0, "owner": None}

    def __init__(self):
        self._bufs: Dict[str, np.ndarray] = {}

    @staticmethod
    def _new_root(nbytes: int) -> np.ndarray:
        return torch.empty(max(1, nbytes), dtype=torch.uint8, pin_memory=torch.cuda.is_available()).numpy()      # the ndarray's base keeps the tensor alive

    # What sys.getrefcount reports for an array beyond the references its caller holds, seen from ONE Python frame below the caller (the
    # call's own argument slots): CPython 3.10 keeps a reference on the caller's value stack during the call, 3.11+ does not, so the number is
    # MEASURED at import on an array with no views and checked on one with a view (`_calibrate`); None = the interpreter did not behave as
    # expected and nothing is ever recycled (buffers are then left to the garbage collector: slower, never wrong).
    _overhead: Optional[int] = None

    @staticmethod
    def _raw_refs(root: np.ndarray) -> int:
        return sys.getrefcount(root)

    @staticmethod
    def _calibrate() -> Optional[int]:
        r = np.empty(16, dtype=np.uint8)
        c0 = PinnedPool._raw_refs(r)                 # the caller holds exactly one reference (the local)
        v = r[:8]
        c1 = PinnedPool._raw_refs(r)                 # ... and a view holds one more (its .base)
        mv = memoryview(v)[2:4]                      # the form finish(chunks=True) hands out: a memoryview slice over a view of the root
        c2 = PinnedPool._raw_refs(r)
        del v
        c3 = PinnedPool._raw_refs(r)                 # the memoryview alone keeps the view, hence the root, referenced
        del mv
        c4 = PinnedPool._raw_refs(r)
        if c1 == c0 + 1 and c2 == c1 and c3 == c1 and c4 == c0 and c0 >= 2:
            return c0 - 1
        return None

    @staticmethod
    def _unreferenced(root: np.ndarray, held: int) -> bool:
        """No view of `root` is alive outside the `held` references the caller knows of.  Every view handed out (ndarray slices, memoryview
        slices over them) keeps one reference on the root, so the root's reference count beyond `held` + the calibrated call overhead says so.
        Uncalibrated interpreter: never (the buffer is not recycled)."""
        oh = PinnedPool._overhead
        return oh is not None and sys.getrefcount(root) <= held + oh

    def get(self, name: str, nbytes: int) -> np.ndarray:
        """uint8 view of this owner's buffer `name` (>= nbytes).  Text buffers ('rows_*') are carved from the arena prepare_arena()
        page-locked ahead of time, as long as no other live owner holds it and they fit."""
        a = PinnedPool._arena
        b = a["buf"]
        if b is not None and name.startswith("rows_") and name not in self._bufs and a["owner"] in (None, id(self)):
            lo = (a["used"] + 4095) & ~4095
            if lo + nbytes <= b.size:
                a["owner"] = id(self)
                a["used"] = lo + nbytes
                return b[lo:lo + nbytes]
        t = self._bufs.get(name)
        if t is None or t.size < nbytes:
            want = max(1, nbytes + nbytes // 8 + 4096)
            t = None
            free = PinnedPool._free
            fit = [i for i, f in enumerate(free) if f.size >= nbytes]
            if fit:
                t = free.pop(min(fit, key=lambda i: free[i].size))
            if t is None:
                t = self._new_root(want)
            old = self._bufs.pop(name, None)
            if old is not None and self._unreferenced(old, 1):
                free.append(old)
            del old
            self._bufs[name] = t
        return t[:nbytes]

    def new_pass(self):
        """The owner's previous text buffers inside the arena are given up (as its per-name buffers always are)."""
        a = PinnedPool._arena
        if a["owner"] == id(self):
            a["used"] = 0

    def release(self):
        """Owner gone: its buffers may serve the next pool and the arena is free again -- unless views of them are still alive outside."""
        names = list(self._bufs)
        for name in names:
            root = self._bufs.pop(name)
            if self._unreferenced(root, 1):
                PinnedPool._free.append(root)
            del root
        del PinnedPool._free[:-32]                # bounded: a long stream of Engines keeps the 32 newest buffers
        a = PinnedPool._arena
        if a["owner"] == id(self):
            a["owner"] = None; a["used"] = 0
            b = a["buf"]
            if b is not None and not self._unreferenced(b, 2):       # text chunks carved from the arena outlive the Engine: the arena is theirs now
                a["buf"] = None


PinnedPool._overhead = PinnedPool._calibrate()


def prepare_arena(nbytes: int):
    """Page-lock one host region for the row text of the coming pass ahead of time (the CLI does it on a helper thread while the BAM is
    decoded: page-locking costs ~0.1 s per GB).  The text buffers of the first Engine that asks are carved from it as long as they fit."""
    if not torch.cuda.is_available():
        return
    a = PinnedPool._arena
    if a["owner"] is not None:
        return                                   # in use by a live Engine: its views stay valid
    b = a["buf"]
    if b is None or b.size < nbytes:
        a["buf"] = PinnedPool._new_root(nbytes)
    a["used"] = 0


def pool_of(eng) -> PinnedPool:
    """The Engine's own pool (created on first use, released when the Engine is collected)."""
    p = eng.__dict__.get("_pinned_pool")
    if p is None:
        import weakref
        p = eng.__dict__["_pinned_pool"] = PinnedPool()
        weakref.finalize(eng, p.release)
    return p


_BINOM_DIRECT = None


def binom_cdf(k: np.ndarray, n: np.ndarray, p: float) -> np.ndarray:
    """scipy.stats.binom.cdf(k, n, p) -- the reference's call (phaser.py:1649) -- for integer arrays 0 <= k <= n.  The public method spends
    most of its time on argument handling; what it computes for these arguments is 1.0 where k >= n and clip(_binom_cdf(k, n, p), 0, 1)
    elsewhere (scipy/stats/_distn_infrastructure.py rv_discrete.cdf, _discrete_distns.py binom_gen._cdf).  The ufunc is called directly ONLY
    after it has reproduced the public method bit for bit on a probe grid in this process; any surprise (another scipy layout, a different
    result) leaves the public method in charge."""
    global _BINOM_DIRECT
    from scipy.stats import binom
    if _BINOM_DIRECT is None:
        _BINOM_DIRECT = False
        try:
            import scipy.special._ufuncs as scu
            f = scu._binom_cdf
            kk, nn = np.meshgrid(np.arange(0, 70, dtype=np.int64), np.arange(1, 70, dtype=np.int64))
            m = kk <= nn
            kk = np.concatenate([kk[m], np.array([150, 300, 999, 1000, 4000, 0])]); nn = np.concatenate([nn[m], np.array([300, 300, 1000, 1000, 5000, 100000])])
            ok = True
            for q in (0.9934, 0.97, 0.999999, 0.5):
                direct = np.where(kk >= nn, 1.0, np.clip(f(kk.astype(np.float64), nn, q), 0, 1))
                ok = ok and bool(np.array_equal(direct, binom.cdf(kk, nn, q)))
            if ok:
                _BINOM_DIRECT = f
        except Exception:
            _BINOM_DIRECT = False
    if _BINOM_DIRECT is False or not (0.0 <= p <= 1.0) or (len(k) and (k.min() < 0 or int(n.min()) < 0)):
        return binom.cdf(k, n, p)
    return np.where(k >= n, 1.0, np.clip(_BINOM_DIRECT(k if k.dtype == np.float64 else k.astype(np.float64), n, p), 0, 1))


def supported(cfg) -> bool:
    return True          # every option of the reference is formatted on the device (round 5: --gw_phase_method 1; round 6: --output_read_ids 1)


def pair_stage_inputs(eng):
    """(tables, page-locked key buffer, slots) of stage 1 for this Engine -- what Engine._tally_genome hands to phz_tally_pairs so that the tally and the first
    kernels of the row stage are issued by ONE native call."""
    import os as _os
    T = tables_for(eng)
    if _os.environ.get("PHZ_ROWS_PAIR_SLOTS") and not getattr(T, "_pair_slots_forced", False):          # tests: start from a tiny table so that the growth path runs
        eng.ctx.check(eng.lib.phz_rowsdev_set_pair_slots(T.h, int(_os.environ["PHZ_ROWS_PAIR_SLOTS"]))); T._pair_slots_forced = True
    n_slots = int(eng.lib.phz_rowsdev_pair_slots(T.h))
    keys = pool_of(eng).get("pair_keys", n_slots * 8).view(np.uint64)
    return T, keys, n_slots


def run(eng, noise: float, fetch_text: bool = True) -> Dict[str, dict]:
    """-> {chrom: fragment} in the format Engine._fragments returns (row text per file as buffers over page-locked host memory, in the
    reference's order), or raises PhzError(PHZ_E_UNSUPPORTED) when the host stage has to take the pass."""
    import time as _t
    cfg = eng.cfg; ctx = eng.ctx; lib = eng.lib
    G = eng.G
    nb = G["nb"]
    t0 = _t.perf_counter()
    T = tables_for(eng)
    t1 = _t.perf_counter()
    # ---- stage 1: distinct (total, supporting) pairs -> scipy -> value + repr text per slot (phaser.py:1645-1652)
    import os as _os
    if _os.environ.get("PHZ_ROWS_PAIR_SLOTS") and not getattr(T, "_pair_slots_forced", False):          # tests: start from a tiny table so that the growth path runs
        ctx.check(lib.phz_rowsdev_set_pair_slots(T.h, int(_os.environ["PHZ_ROWS_PAIR_SLOTS"]))); T._pair_slots_forced = True
    fused = G.pop("pair_stage", None)          # (keys, slots, status) when phz_tally_pairs already ran stage 1 behind the tally
    while True:
        n_slots = int(lib.phz_rowsdev_pair_slots(T.h))
        if fused is not None and fused[1] == n_slots and fused[3] is T:
            keys = fused[0]
            st_ = ctx.check(fused[2], allow=(_lib.PHZ_E_CAPACITY,))
            fused = None
        else:
            fused = None
            keys = pool_of(eng).get("pair_keys", n_slots * 8).view(np.uint64)
            sh0 = sorted(((base, base + n, b) for (c, b), (base, n) in G["line_base"].items()))          # the tally's shards: the first stage then prepares the first-appearance keys too
            lo0 = np.array([x[0] for x in sh0], dtype=np.int64); hi0 = np.array([x[1] for x in sh0], dtype=np.int64); sb0 = np.array([x[2] for x in sh0], dtype=np.int32)
            ctx.check(lib.phz_rowsdev_set_shards(T.h, len(sh0), _vp(lo0), _vp(hi0), _vp(sb0)))
            st_ = ctx.check(lib.phz_rowsdev_pair_keys(ctx.h, T.h, _vp(keys)), allow=(_lib.PHZ_E_CAPACITY,))
        if st_ != _lib.PHZ_E_CAPACITY:
            break
        # very deep coverage: more distinct (supporting, total) pairs than the table holds -- quadruple it (the handle keeps the size) and redo the stage
        ctx.check(lib.phz_rowsdev_set_pair_slots(T.h, n_slots * 4))
        eng.stats["rowsdev_n_pair_table_growths"] = eng.stats.get("rowsdev_n_pair_table_growths", 0) + 1
    t1a = _t.perf_counter()
    # the occupied slots as the argument arrays of the binomial call, laid out natively (five numpy passes over the table were 0.09 ms of every pass)
    sc = eng.__dict__.get("_pair_scratch")
    if sc is None or len(sc[0]) < n_slots:
        sc = eng.__dict__["_pair_scratch"] = (np.empty(n_slots, np.uint32), np.empty(n_slots, np.float64), np.empty(n_slots, np.int64))
    n_used = int(lib.phz_pair_slots_used(_vp(keys), n_slots, _vp(sc[0]), _vp(sc[1]), _vp(sc[2])))
    if n_used < 0:
        raise _lib.PhzError(_lib.PHZ_E_ARG, "phz_pair_slots_used")
    used = sc[0][:n_used]; sup = sc[1][:n_used]; tot = sc[2][:n_used]
    prob = 1 - ((6 * noise) + (10 * math.pow(noise, 2)))
    t1b = _t.perf_counter()
    pv = binom_cdf(sup, tot, prob) if n_used else np.zeros(0, dtype=np.float64)
    t1c = _t.perf_counter()
    # values and text by slot (float.__repr__ of the value: what the reference's str(p) writes, phaser.py:693) -- laid out natively
    slot_pv = np.empty(n_slots, dtype=np.float64)
    txt_off = np.empty(n_slots + 1, dtype=np.uint32)
    txt = np.empty(n_slots + 40 * len(used) + 64, dtype=np.uint8)
    nbytes_txt = lib.phz_pair_slot_text(_vp(used), _vp(np.ascontiguousarray(pv, dtype=np.float64)), len(used), n_slots, _vp(slot_pv), _vp(txt_off),
                                        _vp(txt), txt.size)
    if nbytes_txt < 0:
        raise _lib.PhzError(_lib.PHZ_E_ARG, "phz_pair_slot_text")
    t2 = _t.perf_counter()
    # ---- stage 2
    # the options record of the stage: the same for every pass over the same shards (a sample stream, the bench): kept with the variant tables, rebuilt when anything
    # in it changes (0.05 ms of Python per pass, inside the window in which the GPU waits for the p-values)
    sh = sorted(((base, base + n, b) for (c, b), (base, n) in G["line_base"].items()))
    okey = (tuple(eng.bam_names), tuple(sh), tuple(cfg.haplo_count_bam_exclude or ()), int(cfg.unique_ids), int(cfg.gw_phase_method), int(cfg.output_read_ids),
            int(cfg.unphased_vars), int(cfg.max_block_size), 1 if (cfg.want_vcf or cfg.py_hash_order) else 0, float(cfg.cc_threshold))
    oc = T.__dict__.get("_opts_cache")
    if oc is not None and oc[0] == okey:
        o = oc[1]
    else:
        bam_off, bam_txt = sep_pool(list(eng.bam_names))
        ex = None
        if cfg.haplo_count_bam_exclude:
            ex = np.zeros(nb, dtype=np.uint8)
            for b in cfg.haplo_count_bam_exclude:
                if 0 <= b < nb:
                    ex[b] = 1
        lo = np.array([x[0] for x in sh], dtype=np.int64); hi = np.array([x[1] for x in sh], dtype=np.int64); sb = np.array([x[2] for x in sh], dtype=np.int32)
        o = _lib.phz_rowsdev_opts(nb, _vp(bam_off), _vp(bam_txt), _vp(ex), len(sh), _vp(lo), _vp(hi), _vp(sb), int(cfg.unique_ids), int(cfg.gw_phase_method),
                                  int(cfg.output_read_ids), int(cfg.unphased_vars), int(cfg.max_block_size), 1 if (cfg.want_vcf or cfg.py_hash_order) else 0, float(cfg.cc_threshold))
        T.__dict__["_opts_cache"] = (okey, o, (bam_off, bam_txt, ex, lo, hi, sb))          # (the arrays the record points at live with it)
    if cfg.output_read_ids == 1:
        # the QNAME strings behind the template ids (phaser.py:1120-1123, :1196-1204): template ids are per chromosome, the pool lists the chromosomes' names one after the other
        names = []; qbase = np.zeros(len(eng.chrom_list) + 1, dtype=np.int64)
        for ci, c in enumerate(eng.chrom_list):
            names.extend(eng.qnames.get(c) or [])          # (a chromosome without a read in any BAM has no QNAME table)
            qbase[ci + 1] = len(names)
        q_off, q_txt = sep_pool(names)
        o.qname_off = _vp(q_off); o.qname = _vp(q_txt); o.qname_base = _vp(qbase)
        _keep_q = (q_off, q_txt, qbase)
    R = _lib.phz_rowsdev_result()
    # copy-as-written: a page-locked region sized by what the last pass over these tables wrote (+ 1/8); the run copies every finished text there on a second stream while
    # the remaining writers run.  The first pass over a variant set (size unknown), or a pass that outgrew the guess, takes the texts afterwards as before.
    pool = pool_of(eng)
    pool.new_pass()                             # THIS Engine's previous text buffers are given up; another Engine's are never touched
    arena = None
    guess = int(T.__dict__.get("_text_total", 0))
    if fetch_text and guess > 0 and _os.environ.get("PHZ_ROWS_COPY_AS_WRITTEN", "1") == "1":
        arena = pool.get("rows_all", guess + guess // 8 + 8 * 4096)
        o.host_text = _vp(arena); o.host_text_cap = int(arena.size)
    else:
        o.host_text = None; o.host_text_cap = 0
    t2b = _t.perf_counter()
    try:
        ctx.check(lib.phz_rowsdev_run(ctx.h, T.h, C.byref(o), _vp(slot_pv), _vp(txt_off), _vp(txt), C.byref(R)))
    finally:
        o.host_text = None; o.host_text_cap = 0
        if cfg.output_read_ids == 1:
            o.qname_off = None; o.qname = None; o.qname_base = None          # (the record is kept; the pool of this pass is not)
    t3 = _t.perf_counter()
    nch = len(eng.chrom_list)
    frags: Dict[str, dict] = {c: {"chrom": c, "lines": 0, "dropped": 0, "phased": 0, "allelic_rows": 0, "n_blocks": 0, "vcf": None} for c in eng.chrom_list}
    for name in _lib.PHZ_TXT_NAMES:
        for c in eng.chrom_list:
            frags[c][name] = []
            if name in ("allelic", "single_ase", "single_hap"):
                frags[c][name + "_bam"] = []
    total_bytes = 0
    for f, name in enumerate(_lib.PHZ_TXT_NAMES):
        nbytes = int(R.bytes[f]); total_bytes += nbytes
        if not fetch_text:
            continue
        hoff = int(R.host_off[f])
        if arena is not None and hoff >= 0:
            buf = arena[hoff:hoff + nbytes]          # copied by the run itself, beside its last kernels
        else:
            buf = pool.get("rows_" + name, nbytes)
            ctx.check(lib.phz_rowsdev_fetch_text(ctx.h, T.h, f, _vp(buf) if nbytes else None, nbytes))
        mv = memoryview(buf)
        nseg = nch if f < 4 else nb * nch
        so = [int(R.seg_off[f][i]) for i in range(nseg + 1)]
        if f < 4:
            for ci, c in enumerate(eng.chrom_list):
                if so[ci + 1] > so[ci]:
                    frags[c][name].append(mv[so[ci]:so[ci + 1]])
        else:
            for b in range(nb):
                for ci, c in enumerate(eng.chrom_list):
                    i = b * nch + ci
                    if so[i + 1] > so[i]:
                        frags[c][name].append(mv[so[i]:so[i + 1]]); frags[c][name + "_bam"].append(b)
    for ci, c in enumerate(eng.chrom_list):
        frags[c]["first_bam1"] = int(R.chrom_first_bam[ci]) + 1          # 0: no kept line on this chromosome (place in the block order: engine.block_chrom_order)
    if eng.chrom_list:
        f0 = frags[eng.chrom_list[0]]            # the counts only ever enter sums over chromosomes
        f0["lines"] = int(G["n_kept"]); f0["dropped"] = int(R.dropped); f0["phased"] = int(R.phased); f0["allelic_rows"] = int(R.allelic_rows)
    if (cfg.want_vcf or cfg.py_hash_order) and int(R.n_blocks) > 0:
        nbk = int(R.n_blocks); nvr = int(R.n_blk_vars)
        size = np.empty(nbk, np.int32); var = np.empty(nvr, np.int32); hap = np.empty(nvr, np.uint8); cor = np.empty(2 * nvr, np.int8)
        stat = np.empty(nbk, np.float64); stat_int = np.empty(nbk, np.uint8); maxmaf = np.empty(nbk, np.int32)
        ctx.check(lib.phz_rowsdev_fetch_blocks(ctx.h, T.h, _vp(size), _vp(var), _vp(hap), _vp(cor), _vp(stat), _vp(stat_int), _vp(maxmaf)))
        b0 = v0 = 0
        for ci, c in enumerate(eng.chrom_list):
            kb = int(R.chrom_blocks[ci]); kv = int(R.chrom_blk_vars[ci])
            frags[c]["n_blocks"] = kb
            if kb:
                frags[c]["vcf"] = {"size": size[b0:b0 + kb], "var": var[v0:v0 + kv], "hap": hap[v0:v0 + kv], "cor": cor[2 * v0:2 * (v0 + kv)],
                                   "stat": stat[b0:b0 + kb], "stat_int": stat_int[b0:b0 + kb], "maxmaf": maxmaf[b0:b0 + kb]}
            b0 += kb; v0 += kv
    t4 = _t.perf_counter()
    st = eng.stats
    for k, v in (("rowsdev_tables_s", t1 - t0), ("rowsdev_pairs_s", t2 - t1), ("rowsdev_run_s", t3 - t2), ("rowsdev_fetch_s", t4 - t3),
                 ("rowsdev_pairs_keys_call_s", t1a - t1), ("rowsdev_pairs_numpy_s", t1b - t1a), ("rowsdev_pairs_scipy_s", t1c - t1b), ("rowsdev_pairs_text_s", t2 - t1c),
                 ("rowsdev_run_glue_s", t2b - t2)):
        st[k] = st.get(k, 0.0) + v
    st["rowsdev_gpu_ms"] = st.get("rowsdev_gpu_ms", 0.0) + float(R.gpu_ms)
    st["rowsdev_text_bytes"] = float(total_bytes)
    T.__dict__["_text_total"] = sum(((int(R.bytes[f]) + 4095) & ~4095) for f in range(len(_lib.PHZ_TXT_NAMES)))
    for k in ("n_components", "n_complex", "n_exceptions", "n_big_segments", "n_blocks"):
        st["rowsdev_" + k] = float(getattr(R, k))
    return frags
