"""Host wrapper around phz_map_reads (K_map): shard in, call list out.  GPU only, no fallback."""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Optional

import torch

from . import _lib
from . import soa
from .soa import ReadShard


@dataclasses.dataclass
class Calls:
    """Allele calls in mapper order (record, segment, variant position) -- see include/phz.h."""
    read_idx: torch.Tensor   # int32
    var_idx: torch.Tensor    # int32
    code: torch.Tensor       # uint8: 0..3 = A,C,G,T; 4 = other text
    aux0: Optional[torch.Tensor] = None      # int32 bits of uint32; None for a call list made without the text planes (aux=False)
    aux1: Optional[torch.Tensor] = None

    @property
    def n(self) -> int:
        return int(self.read_idx.numel())

    def cpu(self) -> "Calls":
        return Calls(*(None if t is None else t.cpu() for t in (self.read_idx, self.var_idx, self.code, self.aux0, self.aux1)))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


@dataclasses.dataclass
class TextPool:
    """Read offsets of the characters of every code-4 call of a general-mapper run (host tensors)."""
    call_off: torch.Tensor    # int32 [n_calls + 1]
    roff: torch.Tensor        # int32


class Mapper:
    def __init__(self, device: int = 0, ctx: Optional[_lib.Context] = None):
        self.ctx = ctx or _lib.Context(device)
        self.device = torch.device("cuda", self.ctx.device)

    def load_variants(self, slot: int, vpos, ref_len=None) -> "_lib.phz_variants":
        """phz_load_variants: one chromosome's het-variant table (host numpy arrays or tensors) made resident in the ctx under `slot`;
        the returned record holds device pointers for map(..., resident=...) over every BAM's shard of the chromosome."""
        import numpy as np
        vp = vpos.cpu().numpy() if isinstance(vpos, torch.Tensor) else np.asarray(vpos)
        vp = np.ascontiguousarray(vp, dtype=np.int32)
        rl = np.ones(len(vp), dtype=np.uint8) if ref_len is None else np.ascontiguousarray(ref_len.cpu().numpy() if isinstance(ref_len, torch.Tensor) else ref_len, dtype=np.uint8)
        out = _lib.phz_variants()
        self.ctx.check(self.ctx.lib.phz_load_variants(self.ctx.h, int(slot), C.c_void_p(vp.ctypes.data), C.c_void_p(rl.ctypes.data), len(vp), _lib.PHZ_HOST, C.byref(out)))
        return out

    def map(self, shard: ReadShard, vpos: Optional[torch.Tensor], baseq: int, ref_len: Optional[torch.Tensor] = None,
            cap: Optional[int] = None, resident=None) -> Calls:
        """Runs K_map.  Host tensors are staged through HBM by the library; cuda tensors are used in place.  resident: a table loaded by
        load_variants (device shards only) instead of vpos."""
        on_gpu = shard.device.type == "cuda"
        space = _lib.PHZ_DEVICE if on_gpu else _lib.PHZ_HOST
        out_dev = shard.device
        if resident is not None and not on_gpu:
            raise _lib.PhzError(_lib.PHZ_E_ARG, "a resident variant table serves device-resident shards")
        if resident is None:
            vpos = vpos.to(out_dev).to(torch.int32).contiguous()
        if ref_len is not None:
            ref_len = ref_len.to(torch.uint8).contiguous()
            if bool((ref_len != 1).any()):
                raise _lib.PhzError(_lib.PHZ_E_UNSUPPORTED, "indel variants (ref_len != 1) are not supported by K_map yet")
        r = _lib.phz_reads(shard.n, int(shard.cigar.numel()), int(shard.seq2.numel()), _ptr(shard.pos),
                           _ptr(shard.cigar_off), _ptr(shard.cigar), _ptr(shard.seq_off), _ptr(shard.seq2),
                           _ptr(shard.qual), _ptr(soa.bq_plane(shard)) if on_gpu else None)
        v = resident if resident is not None else _lib.phz_variants(int(vpos.numel()), _ptr(vpos), None)
        if cap is None:
            cap = shard.n // 2 + 4096
        if on_gpu:
            torch.cuda.synchronize(out_dev)     # inputs were produced on torch's stream; K_map runs on the ctx stream
        while True:
            bufs = [torch.empty(cap, dtype=torch.int32, device=out_dev), torch.empty(cap, dtype=torch.int32, device=out_dev),
                    torch.empty(cap, dtype=torch.uint8, device=out_dev), torch.empty(cap, dtype=torch.int32, device=out_dev),
                    torch.empty(cap, dtype=torch.int32, device=out_dev)]
            c = _lib.phz_calls(cap, *[_ptr(b) for b in bufs])
            n = C.c_int64(0)
            st = self.ctx.lib.phz_map_reads(self.ctx.h, C.byref(r), C.byref(v), int(baseq), C.byref(c), C.byref(n), space)
            self.ctx.check(st, allow=(_lib.PHZ_E_CAPACITY,))
            if st == _lib.PHZ_E_CAPACITY:
                cap = int(n.value) + 16
                continue
            m = int(n.value)
            return Calls(*[b[:m] for b in bufs])

    def prepare_batch(self, shards, vposs, baseq: int, caps, aux: bool = True):
        """ctypes argument arrays + output buffers for phz_map_reads_batch over device-resident shards (built once, reusable
        for repeated passes).  -> (call(), bufs, n_out) where call() submits the whole batch and returns the status.
        aux=False: (record, variant, code) only -- what the phasing stage reads; the planes behind the mapper's allele TEXT are not written."""
        n = len(shards)
        R = (_lib.phz_reads * n)(); V = (_lib.phz_variants * n)(); O = (_lib.phz_calls * n)(); N = (C.c_int64 * n)()
        bufs = []
        keep = []
        for i, (sh, vp, cap) in enumerate(zip(shards, vposs, caps)):
            if sh.device.type != "cuda":
                raise _lib.PhzError(_lib.PHZ_E_ARG, "map_batch needs shards resident in HBM")
            vp = vp.to(sh.device).to(torch.int32).contiguous(); keep.append(vp)
            R[i] = _lib.phz_reads(sh.n, int(sh.cigar.numel()), int(sh.seq2.numel()), _ptr(sh.pos), _ptr(sh.cigar_off), _ptr(sh.cigar),
                                  _ptr(sh.seq_off), _ptr(sh.seq2), _ptr(sh.qual), _ptr(soa.bq_plane(sh)))
            V[i] = _lib.phz_variants(int(vp.numel()), _ptr(vp), None)
            b = [torch.empty(cap, dtype=torch.int32, device=sh.device), torch.empty(cap, dtype=torch.int32, device=sh.device),
                 torch.empty(cap, dtype=torch.uint8, device=sh.device)]
            b += [torch.empty(cap, dtype=torch.int32, device=sh.device), torch.empty(cap, dtype=torch.int32, device=sh.device)] if aux else [None, None]
            bufs.append(b)
            O[i] = _lib.phz_calls(cap, *[_ptr(t) for t in b])
        h = self.ctx.h; fn = self.ctx.lib.phz_map_reads_batch; bq = int(baseq)

        def call():
            return fn(h, n, R, V, bq, O, N)
        call.keep = (R, V, O, N, keep, bufs)
        return call, bufs, N

    def map_batch(self, shards, vposs, baseq: int, aux: bool = True):
        """K_map over several device-resident (chromosome, BAM) shards in one submission (phz_map_reads_batch: the reference's
        pool.map over chromosomes, phaser.py:533).  -> list of Calls."""
        if not shards:
            return []
        caps = [sh.n // 2 + 4096 for sh in shards]
        torch.cuda.synchronize(shards[0].device)
        while True:
            call, bufs, N = self.prepare_batch(shards, vposs, baseq, caps, aux)
            st = call()
            self.ctx.check(st, allow=(_lib.PHZ_E_CAPACITY,))
            if st == _lib.PHZ_E_CAPACITY:
                caps = [max(c, int(N[i]) + 16) for i, c in enumerate(caps)]
                continue
            return [Calls(*[None if t is None else t[:int(N[i])] for t in bufs[i]]) for i in range(len(shards))]

    def map_general(self, shard: ReadShard, vpos: torch.Tensor, ref_len: torch.Tensor, allele_off: torch.Tensor,
                    allele_bytes: torch.Tensor, baseq: int, want_text: bool = False):
        """K_map_general (indel mode).  Returns Calls (codes 5/6 = allele 0/1) and, when want_text, a TextPool."""
        on_gpu = shard.device.type == "cuda"
        space = _lib.PHZ_DEVICE if on_gpu else _lib.PHZ_HOST
        dev = shard.device
        vpos = vpos.to(dev).to(torch.int32).contiguous(); ref_len = ref_len.to(dev).to(torch.uint8).contiguous()
        allele_off = allele_off.to(dev).to(torch.int32).contiguous(); allele_bytes = allele_bytes.to(dev).to(torch.uint8).contiguous()
        r = _lib.phz_reads(shard.n, int(shard.cigar.numel()), int(shard.seq2.numel()), _ptr(shard.pos), _ptr(shard.cigar_off),
                           _ptr(shard.cigar), _ptr(shard.seq_off), _ptr(shard.seq2), _ptr(shard.qual))
        v = _lib.phz_variants_general(int(vpos.numel()), _ptr(vpos), _ptr(ref_len), _ptr(allele_off), _ptr(allele_bytes),
                                      int(allele_bytes.numel()))
        cap = shard.n // 2 + 4096
        tcap = 4096
        if on_gpu:
            torch.cuda.synchronize(dev)
        while True:
            bufs = [torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
                    torch.empty(cap, dtype=torch.uint8, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
                    torch.empty(cap, dtype=torch.int32, device=dev)]
            toff = torch.zeros(cap + 1, dtype=torch.int32, device=dev) if want_text else None
            troff = torch.zeros(tcap, dtype=torch.int32, device=dev) if want_text else None
            c = _lib.phz_calls(cap, *[_ptr(b) for b in bufs])
            n = C.c_int64(0); nt = C.c_int64(0)
            st = self.ctx.lib.phz_map_reads_general(self.ctx.h, C.byref(r), C.byref(v), int(baseq), C.byref(c), C.byref(n),
                                                    _ptr(toff), _ptr(troff), tcap, C.byref(nt), space)
            self.ctx.check(st, allow=(_lib.PHZ_E_CAPACITY,))
            if st == _lib.PHZ_E_CAPACITY:
                cap = max(cap, int(n.value) + 16); tcap = max(tcap, int(nt.value) + 16)
                continue
            m = int(n.value)
            calls = Calls(*[b[:m] for b in bufs])
            pool = TextPool(toff[:m + 1].cpu(), troff[:int(nt.value)].cpu()) if want_text else None
            return calls, pool
