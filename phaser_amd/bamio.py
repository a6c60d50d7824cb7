"""BGZF / BAM reading and writing without htslib (none is installed here or on the GPU box).

Reader = what phASER gets from `samtools view -h BAM 'chr': | samtools view -Sh [-F 0x400] [-f 2] -q MAPQ`
(phaser/phaser.py:1346, :505-513): records of one reference sequence filtered by duplicate flag, proper-pair
flag and MAPQ.  The `-L bed` restriction is an optimisation only (the mapper emits nothing for reads that
touch no het site) and is not applied.  This is the functional Python path; a native multi-threaded
inflate + packer is the "next-1" row of SURVEY.md 8(f).
"""
from __future__ import annotations

import gzip
import struct
import zlib
from typing import Dict, Iterator, List, Tuple

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
