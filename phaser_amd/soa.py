"""Structure-of-arrays read shards for the HIP mapper (layout documented in include/phz.h).

Two packers feed the same layout:
  * pack_fixed()  -- vectorised torch packer for fixed-length reads already held as arrays
                     (synthetic generator; runs on the GPU for the bench so shards are born in HBM)
  * pack_sam()    -- general packer for SAM text records (variable length, odd records normalised
                     so the kernel never has to clamp; see "normalisation" below)
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .synth import QUAL_NONACGT, SUB_IUPAC, SUB_N

OP_CODE = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
OP_G = 9   # genome advance without bases (packer-only op, see pack_sam)


@dataclasses.dataclass
class ReadShard:
    """One (chromosome, BAM) shard.  Arrays are torch tensors (cpu or cuda) with int32/uint8 dtype;
    int32 tensors carry the uint32 fields bit-for-bit (all values are < 2**31)."""
    pos: torch.Tensor
    cigar_off: torch.Tensor
    cigar: torch.Tensor
    seq_off: torch.Tensor
    seq2: torch.Tensor
    qual: torch.Tensor
    # per-read fields consumed after the mapper (tally / writers), same order as pos
    qid: Optional[torch.Tensor] = None         # int32 template (QNAME) id
    aln_score: Optional[torch.Tensor] = None   # int32 AS:i value
    has_as: Optional[torch.Tensor] = None      # uint8, 0 when the record carries no AS tag
    iupac: Optional[Dict[Tuple[int, int], str]] = None   # (read, offset) -> original character

    @property
    def n(self) -> int:
        return int(self.pos.numel())

    @property
    def device(self):
        return self.pos.device

    def nbytes_map_inputs(self) -> int:
        """Bytes of the arrays K_map may touch (algorithmic input bytes of one pass)."""
        return sum(int(t.numel()) * t.element_size() for t in
                   (self.pos, self.cigar_off, self.cigar, self.seq_off, self.seq2, self.qual))

    def slice(self, lo: int, hi: int) -> "ReadShard":
        """Records [lo, hi) as a shard of their own (views; offsets stay absolute, which the mapper accepts because it only
        ever indexes cigar / seq2 / qual through cigar_off / seq_off)."""
        f = lambda t, a, b: None if t is None else t[a:b]
        return ReadShard(self.pos[lo:hi], self.cigar_off[lo:hi + 1], self.cigar, self.seq_off[lo:hi + 1], self.seq2, self.qual,
                         f(self.qid, lo, hi), f(self.aln_score, lo, hi), f(self.has_as, lo, hi), self.iupac)

    def to(self, device) -> "ReadShard":
        f = lambda t: None if t is None else t.to(device)
        return ReadShard(f(self.pos), f(self.cigar_off), f(self.cigar), f(self.seq_off), f(self.seq2), f(self.qual),
                         f(self.qid), f(self.aln_score), f(self.has_as), self.iupac)


AS16_NONE = -32768          # phz.h PHZ_AS16_NONE / PHZ_AS16_RANGE
AS16_RANGE = 32767


def as16_plane(shard: "ReadShard", ctx=None) -> Optional[torch.Tensor]:
    """The shard's AS column as ONE 2-byte plane with the has-AS flag folded in (SURVEY.md 8(a) M1 `as:int16`; phz_lines.read_as16): what the
    AS histogram and the per-line pass of K_tally gather per call line -- 2 bytes on one memory line instead of 4 + 1 on two.  AS16_NONE = no
    AS tag; a value outside [-32766, 32766] becomes +-AS16_RANGE and is refused by the histogram like any value outside int16.  Built once per
    shard (elementwise pass on the shard's device) and kept on it."""
    a = shard.aln_score
    if a is None:
        return None
    c = shard.__dict__.get("_as16")
    if c is not None and c[0] is a and c[1] is shard.has_as:
        return c[2]
    if ctx is not None and a.device.type == "cuda" and a.dtype == torch.int32 and a.is_contiguous() and (shard.has_as is None or shard.has_as.is_contiguous()):
        # libphz's own one-line kernel on the ctx stream (the consumers' stream): torch's elementwise kernels would do, but the first of them in a
        # fresh process pays ~90 ms of lazy code-object loading -- the CLI's "H2D + K_map" stage went from 0.02 to 0.11 s on that alone
        import ctypes as C
        x = torch.empty(a.numel(), dtype=torch.int16, device=a.device)
        torch.cuda.current_stream(a.device).synchronize()          # the column may have been produced on torch's stream
        ctx.check(ctx.lib.phz_as_plane(ctx.h, C.c_void_p(a.data_ptr()), None if shard.has_as is None else C.c_void_p(shard.has_as.data_ptr()), a.numel(),
                                         C.c_void_p(x.data_ptr())))
    else:
        x = a.clamp(-AS16_RANGE, AS16_RANGE).to(torch.int16)
        if shard.has_as is not None:
            x = x.masked_fill(shard.has_as == 0, AS16_NONE)
        x = x.contiguous()
    shard.__dict__["_as16"] = (a, shard.has_as, x)
    return x


def bq_plane(shard: "ReadShard") -> Optional[torch.Tensor]:
    """ONE byte per base holding both things an allele call needs (phz_reads.bq): bits 7:6 the 2-bit base, bits 5:0 min(phred, 62); 63 = escape -- a non-ACGT
    base (quality byte with the 0x80 flag) or a phred above 62 -- for which K_map goes back to seq2 / qual.  K_map's bytes under a het SNP then come from one
    memory line instead of two.  MEASURED (tools/ab_kmap_oneplane.py, profiles/r06/ab_kmap_oneplane_production.txt, identical call lists): the production
    instantiation does not get faster (1.0940 -> 1.0928 ms on the whole-genome sample: the kernel is latency-, not line-bound), only the slower profiling
    instantiation does (-3.7 %).  So the plane is OFF by default -- it would cost 6 GB per sample for nothing -- and PHZ_MAP_ONE_PLANE=1 turns it on (built
    once per device-resident shard by an elementwise pass, kept on the shard's quality tensor)."""
    import os
    if shard.qual is None or shard.qual.device.type != "cuda" or os.environ.get("PHZ_MAP_ONE_PLANE") != "1":
        return None
    q = shard.qual
    c = getattr(q, "_phz_bq", None)           # kept ON the quality tensor: slices of a shard (ReadShard.slice) share it, and so the plane
    if c is not None:
        return c
    n = q.numel()
    out = torch.empty(n, dtype=torch.uint8, device=q.device)
    step = 1 << 28                            # 256 M bases at a time: the temporaries stay under 2 GB whatever the shard's size
    shifts = torch.arange(4, device=q.device, dtype=torch.uint8) * 2
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        qq = q[lo:hi]
        base = ((shard.seq2[lo // 4:(hi + 3) // 4].unsqueeze(1) >> shifts) & 3).reshape(-1)[:hi - lo]
        v = (base << 6) | torch.clamp(qq & 0x7F, max=62)
        out[lo:hi] = torch.where(((qq & 0x80) != 0) | ((qq & 0x7F) > 62), torch.full_like(qq, 63), v)
    q._phz_bq = out
    return out


def pack_fixed(pos: torch.Tensor, cigar_off: torch.Tensor, cigar: torch.Tensor, seq: torch.Tensor,
               qual: torch.Tensor, qid=None, aln_score=None) -> ReadShard:
    """seq: uint8 [n, L] base codes 0..3, 4 = N;  qual: uint8 [n, L] phred."""
    n, L = seq.shape
    dev = seq.device
    check_sorted(pos)
    Lp = (L + 3) // 4 * 4
    isn = seq > 3
    code = torch.where(isn, torch.zeros_like(seq), seq)           # SUB_N == 0
    q = torch.where(isn, qual | QUAL_NONACGT, qual)
    if Lp != L:
        pad = torch.zeros(n, Lp - L, dtype=torch.uint8, device=dev)
        code = torch.cat([code, pad], 1)
        q = torch.cat([q, pad], 1)
    c4 = code.view(n, Lp // 4, 4)
    seq2 = (c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6)).contiguous()
    seq_off = (torch.arange(n + 1, device=dev, dtype=torch.int64) * (Lp // 4)).to(torch.int32)
    return ReadShard(pos.to(torch.int32).contiguous(), cigar_off.to(torch.int32).contiguous(),
                     cigar.to(torch.int32).contiguous(), seq_off, seq2.reshape(-1), q.reshape(-1).contiguous(),
                     None if qid is None else qid.to(torch.int32), None if aln_score is None else aln_score.to(torch.int32),
                     None if aln_score is None else torch.ones(n, dtype=torch.uint8, device=dev))


def check_sorted(pos):
    """The mapper is a merge join over coordinate-sorted records (read_variant_map.py:98-114; the reference gets them from an
    indexed BAM region query): every packer refuses a shard with an inversion, the kernels do not re-check."""
    if len(pos) > 1:
        bad = bool((pos[1:] < pos[:-1]).any())
        if bad:
            raise ValueError("records are not coordinate-sorted")


def pack_readbatch(rb) -> ReadShard:
    """Pack a synth.ReadBatch (already samtools-filtered) on whatever device it lives."""
    return pack_fixed(rb.pos, rb.cigar_off, rb.cigar, rb.seq, rb.qual, rb.qid, rb.aln_score)


# --------------------------------------------------------------------------------------------------
_BASE_LUT = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _BASE_LUT[ord(_c)] = _i


def parse_cigar(cigar: str) -> List[Tuple[int, int]]:
    """CIGAR text -> [(op_code or -1 for characters the reference ignores, length)]
    (read_variant_map.py:191-231 builds the number from digit characters the same way)."""
    out = []
    num = 0
    for ch in cigar:
        o = ord(ch)
        if 48 <= o <= 57:
            num = num * 10 + (o - 48)
        else:
            out.append((OP_CODE.get(ch, -1), num))
            num = 0
    return out


def pack_sam(records: List[Tuple[int, str, str, str]]) -> ReadShard:
    """records: (pos, cigar_text, seq_text, qual_text) per SAM line, coordinate-sorted.

    Normalisation (keeps the kernel free of string-clamping logic while matching the reference on odd
    records): the reference zips SEQ with QUAL (read_variant_map.py:179) and slices with Python
    clamping (:200, :220), so only nb = min(len(SEQ), len(QUAL)) bases exist.  An M/=/X op that runs
    past nb is rewritten as M(avail) + G(rest) where G (op 9) only advances the genome cursor; an I op
    keeps its (possibly zero) available length so that the "later insertion overwrites" rule survives.
    """
    n = len(records)
    pos = np.zeros(n, dtype=np.int32)
    cigar_off = np.zeros(n + 1, dtype=np.int64)
    seq_off = np.zeros(n + 1, dtype=np.int64)
    cig: List[int] = []
    seq_chunks = []
    qual_chunks = []
    iupac: Dict[Tuple[int, int], str] = {}
    for r, (p, cg, sq, ql) in enumerate(records):
        pos[r] = p
        nb = min(len(sq), len(ql))
        read_pos = 0
        for op, ln in parse_cigar(cg):
            if op in (0, 7, 8):
                avail = max(0, min(read_pos + ln, nb) - min(read_pos, nb))
                if avail == ln:
                    cig.append((ln << 4) | op)
                else:
                    if avail:
                        cig.append((avail << 4) | op)
                    cig.append(((ln - avail) << 4) | OP_G)
                read_pos += ln
            elif op == 1:
                avail = max(0, min(read_pos + ln, nb) - min(read_pos, nb))
                cig.append((avail << 4) | 1)
                read_pos += ln
            elif op == 4:
                cig.append((ln << 4) | 4)
                read_pos += ln
            elif op in (2, 3):
                cig.append((ln << 4) | op)
            # H, P and unknown characters have no effect in the reference (:227-229)
        cigar_off[r + 1] = len(cig)
        nbp = (nb + 3) // 4 * 4
        sb = np.frombuffer(sq[:nb].encode("latin-1"), dtype=np.uint8)
        qb = np.frombuffer(ql[:nb].encode("latin-1"), dtype=np.uint8).astype(np.int16) - 33
        qb = np.clip(qb, 0, 127).astype(np.uint8)
        code = _BASE_LUT[sb]
        odd = code == 255
        if odd.any():
            sub = np.where((sb == ord("N")) | (sb == ord("D")), SUB_N, SUB_IUPAC).astype(np.uint8)
            for j in np.nonzero(odd)[0]:
                if sub[j] == SUB_IUPAC:
                    iupac[(r, int(j))] = chr(sb[j])
            code = np.where(odd, sub, code)
            qb = np.where(odd, qb | QUAL_NONACGT, qb).astype(np.uint8)
        cp = np.zeros(nbp, dtype=np.uint8); cp[:nb] = code
        qp = np.zeros(nbp, dtype=np.uint8); qp[:nb] = qb
        c4 = cp.reshape(-1, 4)
        seq_chunks.append((c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8))
        qual_chunks.append(qp)
        seq_off[r + 1] = seq_off[r] + nbp // 4
    seq2 = np.concatenate(seq_chunks) if seq_chunks else np.zeros(0, np.uint8)
    qual = np.concatenate(qual_chunks) if qual_chunks else np.zeros(0, np.uint8)
    check_sorted(pos)
    if seq_off[-1] >= 2 ** 31 or cigar_off[-1] >= 2 ** 31:
        raise ValueError("shard too large for 32-bit offsets; split it")
    t = torch.from_numpy
    return ReadShard(t(pos), t(cigar_off.astype(np.int32)), t(np.asarray(cig, dtype=np.int64).astype(np.int32)),
                     t(seq_off.astype(np.int32)), t(seq2), t(qual), iupac=iupac)
