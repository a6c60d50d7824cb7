"""BGZF / BAM reading and writing without htslib (none is installed here or on the GPU box).

Reader = what phASER gets from `samtools view -h BAM 'chr': | samtools view -Sh [-F 0x400] [-f 2] -q MAPQ`
(phaser/phaser.py:1346, :505-513): records of one reference sequence filtered by duplicate flag, proper-pair
flag and MAPQ.  The `-L bed` restriction is an optimisation only (the mapper emits nothing for reads that
touch no het site) and is not applied.  Two readers with identical results: shards_from_bam (pure Python,
reference implementation for the tests) and shards_from_bam_native (C++ in libphz.so: multi-threaded
inflate, packer, QNAME interning -- SURVEY.md 8(f) next-1; ~60x the Python reader).
"""
from __future__ import annotations

import gzip
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import soa
from .samio import QnameInterner

_SEQ_NT16 = "=ACMGRSVTWYHKDBN"
_CIG = "MIDNSHP=X"


def read_bam_bytes(path: str) -> bytes:
    with gzip.open(path, "rb") as f:          # BGZF is a multi-member gzip file
        return f.read()


def parse_header(buf: bytes):
    if buf[:4] != b"BAM\x01":
        raise ValueError("not a BAM file")
    l_text, = struct.unpack_from("<i", buf, 4)
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", buf, off); off += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", buf, off); off += 4
        name = buf[off:off + l_name - 1].decode(); off += l_name
        l_ref, = struct.unpack_from("<i", buf, off); off += 4
        refs.append((name, l_ref))
    return refs, off


def iter_records(buf: bytes, off: int) -> Iterator[tuple]:
    n = len(buf)
    while off + 4 <= n:
        bs, = struct.unpack_from("<i", buf, off)
        s = off + 4
        ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, _nref, _npos, tlen = struct.unpack_from("<iiBBHHHiiii", buf, s)
        p = s + 32
        qname = buf[p:p + l_rn - 1].decode(); p += l_rn
        cigar = np.frombuffer(buf, dtype="<u4", count=n_cig, offset=p); p += 4 * n_cig
        seq = buf[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2
        qual = buf[p:p + l_seq]; p += l_seq
        aux = buf[p:s + bs]
        yield ref_id, pos, mapq, flag, tlen, qname, cigar, l_seq, seq, qual, aux
        off = s + bs


_AUX_SIZE = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4, "A": 1}
_AUX_FMT = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}


def aux_AS(aux: bytes):
    """Value of the last AS tag (the mapper keeps the last AS: column, read_variant_map.py:56-59)."""
    p = 0; n = len(aux); val = None
    while p + 3 <= n:
        tag = aux[p:p + 2]; t = chr(aux[p + 2]); p += 3
        if t in _AUX_SIZE:
            if tag == b"AS" and t in _AUX_FMT:
                val = struct.unpack_from(_AUX_FMT[t], aux, p)[0]
            p += _AUX_SIZE[t]
        elif t in "ZH":
            e = aux.index(b"\0", p); p = e + 1
        elif t == "B":
            st = chr(aux[p]); cnt, = struct.unpack_from("<i", aux, p + 1)
            p += 5 + cnt * _AUX_SIZE[st]
        else:
            break
    return val


def shards_from_bam(path: str, interners: Dict[str, QnameInterner], mapq: int, remove_dups: bool, paired_end: bool,
                    isize_cutoff: float = 0.0, chroms=None) -> Dict[str, soa.ReadShard]:
    """-> {reference name: ReadShard with qid / aln_score / has_as} for the records samtools would pass on."""
    buf = read_bam_bytes(path)
    refs, off = parse_header(buf)
    by: Dict[str, list] = {}
    for ref_id, pos, mq, flag, tlen, qname, cigar, l_seq, seq, qual, aux in iter_records(buf, off):
        if ref_id < 0:
            continue
        chrom = refs[ref_id][0]
        if chroms is not None and chrom not in chroms:
            continue
        if mq < mapq or (remove_dups and (flag & 0x400)) or (paired_end and not (flag & 0x2)):
            continue
        if not (isize_cutoff == 0 or abs(tlen) <= isize_cutoff):
            continue
        cg = "".join("%d%s" % (int(c) >> 4, _CIG[int(c) & 15] if (int(c) & 15) < 9 else "?") for c in cigar) if len(cigar) else "*"
        if l_seq:
            s = "".join(_SEQ_NT16[b >> 4] + _SEQ_NT16[b & 15] for b in seq)[:l_seq]
            q = "*" if qual[0] == 0xFF else bytes(x + 33 for x in qual).decode("latin-1")
        else:
            s = "*"; q = "*"
        by.setdefault(chrom, []).append((qname, pos + 1, cg, s, q, aux_AS(aux)))
    out = {}
    for chrom, recs in by.items():
        it = interners.setdefault(chrom, QnameInterner())
        sh = soa.pack_sam([(r[1], r[2], r[3], r[4]) for r in recs])
        sh.qid = torch.tensor([it(r[0]) for r in recs], dtype=torch.int32)
        sh.aln_score = torch.tensor([0 if r[5] is None else r[5] for r in recs], dtype=torch.int32)
        sh.has_as = torch.tensor([0 if r[5] is None else 1 for r in recs], dtype=torch.uint8)
        out[chrom] = sh
    return out


# ----------------------------------------------------------------------------------------- writer (tests)
_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def write_bam(path: str, refs: List[Tuple[str, int]], records: List[dict]):
    """records: dicts with ref_id, pos (1-based), mapq, flag, tlen, qname, cigar [(op_code, len)], seq (text), qual (phred list),
    tags {"AS": int, ...}."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs)))
    for name, ln in refs:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    code = {c: i for i, c in enumerate(_SEQ_NT16)}
    for r in records:
        qn = r["qname"].encode() + b"\0"
        seq = r["seq"]; l_seq = len(seq)
        nib = [code.get(c, 15) for c in seq] + [0]
        sb = bytes((nib[i] << 4) | nib[i + 1] for i in range(0, l_seq, 2))
        qb = bytes(r["qual"]) if r.get("qual") is not None else b"\xff" * l_seq
        cg = b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in r["cigar"])
        aux = b""
        for k, v in r.get("tags", {}).items():
            aux += k.encode() + b"i" + struct.pack("<i", v)
        body = struct.pack("<iiBBHHHiiii", r["ref_id"], r["pos"] - 1, len(qn), r["mapq"], 4680, len(r["cigar"]), r["flag"], l_seq,
                           r["ref_id"], max(0, r.get("mate_pos", r["pos"]) - 1), r["tlen"]) + qn + cg + sb + qb + aux
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        for i in range(0, len(out), 60000):
            f.write(_bgzf_block(bytes(out[i:i + 60000])))
        f.write(_EOF)


def readbatch_to_bam(path: str, rbs, refs: List[Tuple[str, int]]):
    """Write synth.ReadBatch objects (one per chromosome, unfiltered) as one coordinate-sorted BAM."""
    from . import synth
    names = [r[0] for r in refs]
    recs = []
    for rb in rbs:
        rid = names.index(rb.chrom)
        seq = rb.seq.cpu().numpy(); qual = rb.qual.cpu().numpy()
        lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
        for i in range(len(rb)):
            a, b = int(rb.cigar_off[i]), int(rb.cigar_off[i + 1])
            cg = [(int(c) & 15, int(c) >> 4) for c in rb.cigar[a:b].tolist()]
            recs.append({"ref_id": rid, "pos": int(rb.pos[i]), "mapq": int(rb.mapq[i]), "flag": int(rb.flag[i]), "tlen": int(rb.tlen[i]),
                         "qname": rb.qname(i), "cigar": cg, "seq": lut[seq[i]].tobytes().decode(), "qual": qual[i].tolist(),
                         "tags": {"NH": 1, "AS": int(rb.aln_score[i])}})
    write_bam(path, refs, recs)


def readbatch_to_bam_native(path: str, rbs, refs: List[Tuple[str, int]], threads: int = 0):
    """Same file content as readbatch_to_bam (the inflated BAM stream is byte-identical), written by the native encoder in
    libphz.so (phz_bam_write): what the at-scale runs use, there being no samtools in the image."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    names = [r[0] for r in refs]
    arr = (_lib.phz_read_batch * max(1, len(rbs)))()
    keep = []

    def T(t, dt):
        a = np.ascontiguousarray(t.cpu().numpy().astype(dt, copy=False)); keep.append(a)
        return C.c_void_p(a.ctypes.data)
    for k, rb in enumerate(rbs):
        x = arr[k]
        x.n = len(rb); x.ref_id = names.index(rb.chrom); x.L = rb.L
        x.pos = T(rb.pos, np.int32); x.flag = T(rb.flag, np.int32); x.mapq = T(rb.mapq, np.int32); x.tlen = T(rb.tlen, np.int32)
        x.aln_score = T(rb.aln_score, np.int32); x.qid = T(rb.qid, np.int32); x.cigar_off = T(rb.cigar_off, np.int64)
        x.cigar = T(rb.cigar, np.uint32); x.seq = T(rb.seq, np.uint8); x.qual = T(rb.qual, np.uint8)
        x.qname_prefix = rb.qname_prefix.encode()
    nm = (C.c_char_p * len(refs))(*[r[0].encode() for r in refs])
    ln = np.asarray([r[1] for r in refs], dtype=np.int32)
    st = lib.phz_bam_write(path.encode(), len(refs), nm, C.c_void_p(ln.ctypes.data), arr, len(rbs), int(threads))
    if st != _lib.PHZ_OK:
        raise _lib.PhzError(st, "phz_bam_write failed")


# ----------------------------------------------------------------------------------------- native reader (libphz.so)
class NativeInterner:
    """QNAME -> id map of one chromosome; same first-appearance numbering as samio.QnameInterner, continued across BAMs.

    Two homes: the C++ table (phz_interner) that the host decoders fill, and a DEVICE store (name bytes + offsets in id order) that
    the GPU path fills with phz_intern_device.  While only the GPU path is used the C++ table stays empty -- it is synchronised from
    the store when something host-side needs it (names for --output_read_ids, a BAM that goes through the host decoder), and from
    then on the store is dropped and the chromosome stays on the host table."""

    def __init__(self):
        import ctypes as C
        from . import _lib
        self.lib = _lib.load()
        h = C.c_void_p()
        self.lib.phz_interner_create(C.byref(h))
        self._h = h
        self._store = None             # {"blob": uint8 cuda, "off": int32 cuda [n + 1], "n": ids, "bytes": blob bytes in use}

    def _host_size(self) -> int:
        return int(self.lib.phz_interner_size(self._h))

    def _sync(self):
        """names [host size, store size) of the device store enter the C++ table (ids come out the same: inserted in id order)"""
        st = self._store
        if st is None or st["n"] <= self._host_size():
            return
        import ctypes as C
        k0 = self._host_size(); n = st["n"]
        off = st["off"][k0:n + 1].cpu().numpy().astype(np.uint32)
        b0 = int(off[0])
        blob = st["blob"][b0:int(off[-1])].cpu().numpy()
        off = (off - np.uint32(b0)).astype(np.uint32)
        ids = np.zeros(n - k0, dtype=np.int32)
        self.lib.phz_intern(self._h, C.c_void_p(blob.ctypes.data), C.c_void_p(off.ctypes.data), n - k0, C.c_void_p(ids.ctypes.data))
        assert int(ids[-1]) == n - 1 and self._host_size() == n

    def intern_device(self, ctx, qnames, qname_off, n: int):
        """ids of a device-resident shard (qnames uint8 [cuda], qname_off int32 [n + 1, cuda]); None when this chromosome's names
        already live on the host table only."""
        import ctypes as C
        st = self._store
        if st is None and self._host_size() > 0:
            return None
        dev = qnames.device
        n_old = st["n"] if st else 0
        qid = torch.empty(max(1, n), dtype=torch.int32, device=dev); first = torch.empty(max(1, n), dtype=torch.int32, device=dev)
        nn = C.c_int64(0)
        P = lambda t: C.c_void_p(t.data_ptr())
        rc = self.lib.phz_intern_device(ctx.h, P(qnames), P(qname_off), n, P(st["blob"]) if st else None, P(st["off"]) if st else None, n_old,
                                        P(qid), P(first), C.byref(nn))
        ctx.check(rc)
        m = int(nn.value)
        if m:
            base = st["bytes"] if st else 0
            new_off = torch.empty(m + 1, dtype=torch.int32, device=dev)
            tot = C.c_int64(0)
            ctx.check(self.lib.phz_names_append_device(ctx.h, P(qnames), P(qname_off), P(first), m, base, P(new_off), None, C.byref(tot)))
            total = int(tot.value)
            if st and st["blob"].numel() >= total:
                blob = st["blob"]
            else:                                   # grow geometrically: a sample's BAMs arrive one after the other
                blob = torch.empty(max(total, int(1.5 * (st["blob"].numel() if st else 0))), dtype=torch.uint8, device=dev)
                if st:
                    blob[:base] = st["blob"][:base]
            ctx.check(self.lib.phz_names_append_device(ctx.h, P(qnames), P(qname_off), P(first), m, base, P(new_off), P(blob), C.byref(tot)))
            off = torch.cat([st["off"][:n_old], new_off]) if st else new_off
            self._store = {"blob": blob, "off": off, "n": n_old + m, "bytes": total}
        elif st is None:
            self._store = {"blob": torch.empty(1, dtype=torch.uint8, device=dev), "off": torch.zeros(1, dtype=torch.int32, device=dev), "n": 0, "bytes": 0}
        return qid[:n]

    @property
    def h(self):
        """the C++ table, for host-side interning: brought up to date, and from now on the only home of this chromosome's names"""
        self._sync()
        self._store = None
        return self._h

    def __len__(self):
        return max(self._host_size(), self._store["n"] if self._store else 0)

    def pool(self):
        """-> (blob bytes, offsets uint32 [n + 1]) of the names in id order, without building a Python string per name (the native raw-byte tier reads this:
        pyorder.replay_native; 40 M names of a whole-genome sample as a list of str cost seconds and gigabytes)"""
        import ctypes as C
        self._sync()
        n = self._host_size()
        off = np.zeros(n + 1, dtype=np.uint32)
        cap = 1 << 20
        while True:
            blob = np.zeros(cap, dtype=np.uint8)
            st = self.lib.phz_interner_names(self._h, C.c_void_p(blob.ctypes.data), cap, C.c_void_p(off.ctypes.data))
            if st == 0:
                break
            cap = int(off[n]) + 16
        return blob[:int(off[n])].tobytes(), off

    @property
    def names(self) -> List[str]:
        import ctypes as C
        self._sync()
        n = self._host_size()
        off = np.zeros(n + 1, dtype=np.uint32)
        cap = 1 << 20
        while True:
            blob = np.zeros(cap, dtype=np.uint8)
            st = self.lib.phz_interner_names(self._h, C.c_void_p(blob.ctypes.data), cap, C.c_void_p(off.ctypes.data))
            if st == 0:
                break
            cap = int(off[n]) + 16
        raw = blob.tobytes()
        return [raw[int(off[i]):int(off[i + 1])].decode() for i in range(n)]

    def __del__(self):
        try:
            self.lib.phz_interner_destroy(self._h)
        except Exception:
            pass


def shards_from_bam_native(path: str, interners: Dict[str, "NativeInterner"], mapq: int, remove_dups: bool, paired_end: bool,
                           isize_cutoff: float = 0.0, chroms=None, threads: int = 0) -> Dict[str, soa.ReadShard]:
    """Same result as shards_from_bam, produced by the C++ decoder/packer in libphz.so (multi-threaded inflate)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    import time as _t0m
    _topen = _t0m.perf_counter()
    h = C.c_void_p()
    if chroms is not None:
        # only the BGZF members holding these chromosomes are inflated (a rank of a multi-GPU run reads its own share of the file)
        names = [str(c).encode() for c in chroms]
        arr = (C.c_char_p * max(1, len(names)))(*names) if names else (C.c_char_p * 1)()
        st = lib.phz_bam_open_refs(path.encode(), threads, arr, len(names), None, 0, C.byref(h))
    else:
        st = lib.phz_bam_open(path.encode(), threads, C.byref(h))
    if st != 0:
        raise _lib.PhzError(st, "cannot read BAM " + path)
    import os as _os, sys as _sys, time as _t
    _prof = _os.environ.get("PHZ_TIMING"); _t0 = _t.perf_counter(); _tin = 0.0

    class _Owner:                      # closes the decoder when the last shard array built on its memory is gone
        def __init__(self, lib, h):
            self.lib = lib; self.h = h

        def __del__(self):
            try:
                self.lib.phz_bam_close(self.h)
            except Exception:
                pass
    owner = _Owner(lib, h)
    n_ref = lib.phz_bam_n_ref(h)
    names = [lib.phz_bam_ref_name(h, i).decode() for i in range(n_ref)]
    mask = np.array([1 if (chroms is None or nm in chroms) else 0 for nm in names], dtype=np.uint8)
    ns = C.c_int(0)
    st = lib.phz_bam_decode(h, C.c_void_p(mask.ctypes.data), int(mapq), 0x2 if paired_end else 0, 0x400 if remove_dups else 0,
                            float(isize_cutoff), threads, C.byref(ns))
    if st == _lib.PHZ_E_UNSUPPORTED:
        raise _lib.PhzError(st, "BAM is not coordinate-sorted (the mapper is a merge join over sorted reads)")
    if st != 0:
        raise _lib.PhzError(st, "BAM decode failed")
    out = {}
    _t1 = _t.perf_counter()
    for i in range(ns.value):
        hs = _lib.phz_host_shard()
        lib.phz_bam_shard(h, i, C.byref(hs))
        n = hs.n_reads

        def arr(ptr, count, dt):       # zero-copy: the tensors live in the decoder's buffers (kept alive through `owner`)
            ct = {torch.int32: C.c_int32, torch.uint8: C.c_uint8}[dt]
            return torch.from_numpy(_lib.native_view(ptr, count, ct, owner))
        chrom = hs.ref_name.decode()
        sh = soa.ReadShard(arr(hs.pos, n, torch.int32), arr(hs.cigar_off, n + 1, torch.int32), arr(hs.cigar, hs.n_ops, torch.int32),
                           arr(hs.seq_off, n + 1, torch.int32), arr(hs.seq2, hs.n_seq_bytes, torch.uint8),
                           arr(hs.qual, hs.n_seq_bytes * 4, torch.uint8))
        it = interners.setdefault(chrom, NativeInterner())
        qid = np.zeros(n, dtype=np.int32)
        _ti = _t.perf_counter()
        lib.phz_intern(it.h, hs.qnames, hs.qname_off, n, C.c_void_p(qid.ctypes.data))
        _tin += _t.perf_counter() - _ti
        sh.qid = torch.from_numpy(qid)
        sh.aln_score = arr(hs.aln_score, n, torch.int32)
        sh.has_as = arr(hs.has_as, n, torch.uint8)
        out[chrom] = sh
    if _prof:
        _sys.stderr.write("[phz timing]   bam: open+inflate %.2f s, decode+filter+pack %.2f s, shard views %.2f s, qname interning %.2f s\n"
                          % (_t0 - _topen, _t1 - _t0, _t.perf_counter() - _t1 - _tin, _tin))
    return out


def shards_from_bam_device(ctx, path: str, interners: Dict[str, "NativeInterner"], mapq: int, remove_dups: bool, paired_end: bool,
                           isize_cutoff: float = 0.0, chroms=None, device="cuda:0", threads: int = 0, _depth: int = 0) -> Optional[Dict[str, soa.ReadShard]]:
    """The same shards as shards_from_bam_native, decoded ON THE GPU (phz_bamdev_*: K_inflate, record hop, k_pack) and left in HBM.
    QNAME interning stays on the host (the interner persists across BAMs): the name bytes are the only part that travels back.
    Returns None when the file needs the host path (the library says PHZ_E_UNSUPPORTED; the reason goes to stderr under PHZ_TIMING)."""
    import ctypes as C, os as _os, sys as _sys, time as _t
    from . import _lib
    lib = _lib.load()
    _prof = _os.environ.get("PHZ_TIMING"); t0 = _t.perf_counter()
    f = _lib.phz_bam_filters(int(mapq), 0x2 if paired_end else 0, 0x400 if remove_dups else 0, float(isize_cutoff))
    h = C.c_void_p()
    if chroms is not None:
        names = [str(c).encode() for c in chroms]
        arr = (C.c_char_p * max(1, len(names)))(*names) if names else (C.c_char_p * 1)()
        st = lib.phz_bamdev_open(ctx.h, path.encode(), arr, len(names), C.byref(f), C.byref(h))
    else:
        st = lib.phz_bamdev_open(ctx.h, path.encode(), None, 0, C.byref(f), C.byref(h))
    if st == _lib.PHZ_E_UNSUPPORTED and b"32-bit offsets" in (lib.phz_last_error(ctx.h) or b"") and _depth < 6:
        # a very deep BAM: one call is limited to 2^32 bytes of names / base groups -- decode the chromosomes in two halves of
        # about equal compressed size (the member ranges of each half are all that is copied and inflated)
        w = bam_ref_weights(path, threads)
        names_all = [c for c in (chroms if chroms is not None else list(w)) if w.get(str(c), 0) > 0 or chroms is not None]
        if len(names_all) > 1:
            order = sorted(names_all, key=lambda c: -w.get(str(c), 0))
            halves = ([], []); tot = [0, 0]
            for c in order:
                k = 0 if tot[0] <= tot[1] else 1
                halves[k].append(c); tot[k] += w.get(str(c), 0)
            merged = {}
            for part in halves:
                sub = shards_from_bam_device(ctx, path, interners, mapq, remove_dups, paired_end, isize_cutoff, chroms=part, device=device,
                                             threads=threads, _depth=_depth + 1)
                if sub is None:
                    return None
                merged.update(sub)
            return merged
    if st in (_lib.PHZ_E_UNSUPPORTED, _lib.PHZ_E_ARG, _lib.PHZ_E_NOMEM):
        # declined, not enough HBM for the inflated stream, or a file the plan cannot read / a record chain that breaks: the host decoder is the one that reports on files
        # (same messages as before the device path existed)
        if _prof:
            _sys.stderr.write("[phz timing]   bam: device path declined (status %d: %s), using the host decoder\n" % (st, lib.phz_last_error(ctx.h).decode()))
        return None
    if st != 0:
        raise _lib.PhzError(st, "cannot read BAM %s: %s" % (path, lib.phz_last_error(ctx.h).decode()))
    t1 = _t.perf_counter()
    try:
        n_ref = lib.phz_bamdev_n_ref(h)
        tab = (_lib.phz_dev_shard * max(1, n_ref))()
        out = {}; held = {}
        dev = torch.device(device)
        for i in range(n_ref):
            sz = _lib.phz_bamdev_sizes()
            lib.phz_bamdev_sizes_of(h, i, C.byref(sz))
            n = int(sz.n_reads)
            if n == 0:
                continue
            chrom = lib.phz_bamdev_ref_name(h, i).decode()
            e = lambda count, dt: torch.empty(max(1, int(count)), dtype=dt, device=dev)
            t = {"pos": e(n, torch.int32), "cigar_off": e(n + 1, torch.int32), "cigar": e(sz.n_ops, torch.int32), "seq_off": e(n + 1, torch.int32),
                 "seq2": e(sz.n_seq_bytes, torch.uint8), "qual": e(sz.n_seq_bytes * 4, torch.uint8), "aln_score": e(n, torch.int32),
                 "has_as": e(n, torch.uint8), "qname_off": e(n + 1, torch.int32), "qnames": e(sz.n_qname_bytes, torch.uint8)}
            for k, v in t.items():
                setattr(tab[i], k, v.data_ptr())
            held[chrom] = (t, n, sz)
        st = lib.phz_bamdev_pack(h, tab, n_ref)
        if st != 0:
            raise _lib.PhzError(st, "device BAM pack failed: " + lib.phz_last_error(ctx.h).decode())
    finally:
        lib.phz_bamdev_close(h)
    t2 = _t.perf_counter(); tin = 0.0
    for chrom, (t, n, sz) in held.items():
        sh = soa.ReadShard(t["pos"][:n], t["cigar_off"][:n + 1], t["cigar"][:int(sz.n_ops)], t["seq_off"][:n + 1], t["seq2"][:int(sz.n_seq_bytes)],
                           t["qual"][:int(sz.n_seq_bytes) * 4])
        it = interners.setdefault(chrom, NativeInterner())
        ti = _t.perf_counter()
        qid_d = it.intern_device(ctx, t["qnames"], t["qname_off"], n)
        if qid_d is not None:
            sh.qid = qid_d
        else:                                   # this chromosome's names are on the host table (an earlier BAM went through the host decoder)
            qn = t["qnames"][:int(sz.n_qname_bytes)].cpu().numpy(); qo = t["qname_off"][:n + 1].cpu().numpy()
            qid = np.zeros(n, dtype=np.int32)
            lib.phz_intern(it.h, C.c_void_p(qn.ctypes.data), C.c_void_p(qo.ctypes.data), n, C.c_void_p(qid.ctypes.data))
            sh.qid = torch.from_numpy(qid).to(dev)
        tin += _t.perf_counter() - ti
        sh.aln_score = t["aln_score"][:n]; sh.has_as = t["has_as"][:n]
        out[chrom] = sh
    if _prof:
        _sys.stderr.write("[phz timing]   bam (device): plan + H2D + inflate + hop %.2f s, allocate + pack %.2f s, other %.2f s, qname ids %.2f s\n"
                          % (t1 - t0, t2 - t1, _t.perf_counter() - t2 - tin, tin))
    return out


def bam_ref_weights(path: str, threads: int = 0) -> Dict[str, int]:
    """-> {reference name: compressed bytes its records occupy in the BAM}: a proxy of the record count per chromosome that costs a
    few dozen member inflations (binary search over the BGZF member table), used as LPT weights before anything is decoded."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    cap = 1 << 16
    w = np.zeros(cap, dtype=np.int64)
    none = (C.c_char_p * 1)()
    h = C.c_void_p()
    st = lib.phz_bam_open_refs(path.encode(), threads, none, 0, C.c_void_p(w.ctypes.data), cap, C.byref(h))
    if st != 0:
        raise _lib.PhzError(st, "cannot read BAM " + path)
    try:
        n = lib.phz_bam_n_ref(h)
        return {lib.phz_bam_ref_name(h, i).decode(): int(w[i]) for i in range(min(n, cap))}
    finally:
        lib.phz_bam_close(h)
