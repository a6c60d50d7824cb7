"""Drop-in for phASER's mapper module: same function, arguments, stdin/stdout contract and output file.

Mirrors phaser/read_variant_map.py:3 `do_read_variant_map(variant_table, baseq, o, splice,
isize_cutoff)`: SAM text (with @SQ header lines) on stdin, the variant table TSV written by
generate_mapping_table (phaser/phaser.py:1402-1404), one output line per (record, variant) hit:
    qname, unique_id, rsid, allele, AS, genotype, maf          (read_variant_map.py:117)
The computation itself runs in the HIP kernel K_map through libphz.so; the host only parses text,
packs the structure of arrays, and prints.  No GPU or no library => this raises (no CPU path).
"""
from __future__ import annotations

import sys
from typing import List

import numpy as np
import torch

from . import soa
from .mapper import Mapper

_BASES = "ACGT"


class VariantTable:
    """Rows of the mapper's variant table (read_variant_map.py:126-135)."""

    def __init__(self, path: str):
        self.chr: List[str] = []; self.pos: List[int] = []; self.id: List[str] = []; self.rsid: List[str] = []
        self.alleles: List[str] = []; self.ref_len: List[int] = []; self.gt: List[str] = []; self.maf: List[str] = []
        with open(path) as f:
            for line in f:
                c = line.rstrip().split("\t")
                if len(c) < 8:
                    continue
                self.chr.append(c[0]); self.pos.append(int(c[1])); self.id.append(c[2]); self.rsid.append(c[3])
                self.alleles.append(c[4]); self.ref_len.append(int(c[5])); self.gt.append(c[6]); self.maf.append(c[7])


def _individual_alleles(alleles_field: str, gt: str):
    """The individual's alleles in allele-index order (generate_variant_dict, phaser.py:1430-1435)."""
    every = alleles_field.split(",")
    g = list(gt)
    return [every[i] for i in range(len(every)) if str(i) in g]


def _allele_text(code, aux0, aux1, seq, qual, baseq):
    """Exact allele text of one call.  Plain single-base calls come straight from `code`; for the rare
    composite ones (inserted bases spliced after the SNP, IUPAC symbols) the kernel reports which read
    offsets make up the text and the host only copies the characters."""
    if code < 4:
        return _BASES[code]
    def ch(x):
        return seq[x] if (ord(qual[x]) - 33) >= baseq else "N"
    s = ""
    if aux0 != 0xFFFFFFFF:
        s += ch(aux0)
    ilen = aux1 & 0xFFF
    ioff = aux1 >> 12
    for x in range(ioff, ioff + ilen):
        s += ch(x)
    return s.replace("D", "")


def _native_sam(data: bytes, isize_cutoff: float, threads: int):
    """phz_sam_parse: header contigs + one packed shard per chromosome (input order).  -> (handle owner, contigs, [(chrom, shard)])"""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    h = C.c_void_p()
    st = lib.phz_sam_parse(C.cast(C.c_char_p(data), C.c_void_p), len(data), float(isize_cutoff), int(threads), C.byref(h))
    if st != _lib.PHZ_OK:
        msg = (lib.phz_sam_error(h) or b"").decode()
        lib.phz_sam_free(h)
        raise _lib.PhzError(st, msg or "phz_sam_parse failed")

    class _Owner:
        def __init__(self):
            self.h = h; self.data = data          # the records point into `data`

        def __del__(self):
            try:
                lib.phz_sam_free(self.h)
            except Exception:
                pass
    owner = _Owner()
    contigs = [lib.phz_sam_contig(h, i).decode() for i in range(lib.phz_sam_n_contigs(h))]
    shards = []
    for i in range(lib.phz_sam_n_shards(h)):
        hs = _lib.phz_host_shard()
        lib.phz_sam_shard(h, i, C.byref(hs))
        n = hs.n_reads

        def arr(ptr, count, dt):
            ct = {torch.int32: C.c_int32, torch.uint8: C.c_uint8}[dt]
            return torch.from_numpy(_lib.native_view(ptr, count, ct, owner))
        sh = soa.ReadShard(arr(hs.pos, n, torch.int32), arr(hs.cigar_off, n + 1, torch.int32), arr(hs.cigar, hs.n_ops, torch.int32),
                           arr(hs.seq_off, n + 1, torch.int32), arr(hs.seq2, hs.n_seq_bytes, torch.uint8),
                           arr(hs.qual, hs.n_seq_bytes * 4, torch.uint8))
        shards.append((hs.ref_name.decode(), sh))
    return owner, contigs, shards


def _do_native(table, baseq, o, isize_cutoff, mapper, threads, data):
    """SNP-mode fast path: native SAM parse / pack, K_map, native TSV formatting."""
    import ctypes as C
    from . import _lib
    from .vcf import sep_pool
    lib = _lib.load()
    owner, contigs, shards = _native_sam(data, float(isize_cutoff), threads)
    if not lib.phz_sam_stream_order(owner.h):
        # a record (kept or dropped by the isize filter) steps backwards, or a chromosome comes back: the reference's result is no longer
        # the stateless rule; do_read_variant_map's Python path follows its forward-only variant buffer
        raise _lib.PhzError(_lib.PHZ_E_UNSUPPORTED, "SAM stream is not one coordinate-sorted run per chromosome")
    tchroms = []
    for c in table.chr:
        if not tchroms or tchroms[-1] != c:
            tchroms.append(c)
    for rc, _ in shards:
        for vc in tchroms:
            if vc != rc and vc not in contigs:
                print("Error, VCF and BAM contigs do not match VCF = %s BAM = %s" % (vc, rc))
                sys.exit(1)
    tchr = np.asarray(table.chr, dtype=object)
    with open(o, "wb") as out:
        for si, (chrom, shard) in enumerate(shards):
            vsel = np.nonzero(tchr == chrom)[0]
            if len(vsel) == 0 or shard.n == 0:
                continue
            vpos = torch.tensor([table.pos[i] for i in vsel], dtype=torch.int32)
            calls = mapper.map(shard, vpos, int(baseq)).cpu()
            if calls.n == 0:
                continue
            pools = [sep_pool([col[i] for i in vsel]) for col in (table.id, table.rsid, table.gt, table.maf)]
            arrs = [np.ascontiguousarray(t.numpy()) for t in (calls.read_idx, calls.var_idx, calls.code, calls.aux0, calls.aux1)]
            p = C.c_void_p(); n = C.c_int64(0)
            args = [C.c_void_p(a.ctypes.data) for a in arrs]
            pool_args = []
            for off, b in pools:
                pool_args += [C.c_void_p(off.ctypes.data), C.cast(C.c_char_p(b), C.c_void_p)]
            st = lib.phz_sam_calls_tsv(owner.h, si, calls.n, *args, int(baseq), *pool_args, int(threads), C.byref(p), C.byref(n))
            if st != _lib.PHZ_OK:
                raise _lib.PhzError(st, "phz_sam_calls_tsv failed")
            try:
                out.write(C.string_at(p, n.value))
            finally:
                lib.phz_buf_free(p)


def _segment_spans(cigar: str, n_bases: int):
    """(start, length of the pseudo read) of every N-split segment, as split_read builds them (read_variant_map.py:191-232): M / = / X take
    what the read still has (a slice clamps at the end of the string), D adds placeholders, N closes the segment."""
    spans = []
    num = 0; rpos = 0; gpos = 0; start = 0; plen = 0
    for c in cigar:
        if "0" <= c <= "9":
            num = num * 10 + ord(c) - 48
            continue
        if c in "MX=":
            plen += max(0, min(num, n_bases - rpos)); rpos += num; gpos += num
        elif c == "N":
            spans.append((start, plen)); gpos += num; start = gpos; plen = 0
        elif c == "D":
            plen += num; gpos += num
        elif c in "IS":
            rpos += num
        num = 0
    spans.append((start, plen))
    return spans


def _buffer_floors(events, recs, vpos):
    """The reference's streaming variant buffer on a chromosome whose records are NOT in coordinate order.  The buffer is the index range
    [b_lo, L) of the chromosome's (position-sorted) variants: L counts the variants consumed so far -- skipped because they lie behind the
    current record (read_variant_map.py:88-93) or appended up to the end of a segment (:106-112); the variant stream never rewinds -- and
    b_lo the consumed ones pruned for lying behind SOME earlier record (:37-50, done for every record of the stream, also one the isize
    filter then drops).  A record sees only what is inside the buffer (:114), so its calls are the stateless rule's calls on variants
    >= the b_lo of its moment.  -> b_lo per kept record (0 everywhere on a sorted stream).  events = (POS, kept index | -1 | -2) in stream order:
    -1 = dropped by the isize filter BEFORE the skip step (:51), -2 = passes it, runs the skip step (:88-93: the variants behind it are consumed and
    never buffered) and then gets no alignment from split_read (:170, an N in its CIGAR while --splice is not 1)."""
    import bisect
    floors = [0] * len(recs)
    b_lo = 0; L = 0; nv = len(vpos)
    for pos, k in events:
        lb = bisect.bisect_left(vpos, pos)
        b_lo = max(b_lo, min(lb, L))
        if k == -1:
            continue
        if L < lb:
            L = lb; b_lo = lb
        if k < 0:                   # -2: a record split_read returns nothing for (an N in the CIGAR with --splice != 1): the skip above has run, nothing is appended
            continue
        floors[k] = b_lo
        rec = recs[k]
        for start, plen in _segment_spans(rec[2], min(len(rec[3]), len(rec[4]))):
            L = max(L, bisect.bisect_right(vpos, pos + start + plen))
    return floors


def do_read_variant_map(variant_table, baseq, o, splice, isize_cutoff, _mapper=None, threads: int = 0):
    table = VariantTable(variant_table)
    snp_only = all(rl == 1 for rl in table.ref_len) and all(len(a) == 1 and a in _BASES for al in
                                                            (_individual_alleles(x, g) for x, g in zip(table.alleles, table.gt)) for a in al)
    stream = sys.stdin
    if splice == 1 and snp_only:
        # the whole stream through native code: parse + pack (host threads), K_map, TSV formatting.  Streams the native parser declines
        # (records of a chromosome out of coordinate order) go on below, where the reference's forward-only variant buffer is followed
        from . import _lib
        import io
        data = sys.stdin.buffer.read() if hasattr(sys.stdin, "buffer") else sys.stdin.read().encode()
        try:
            return _do_native(table, baseq, o, isize_cutoff, _mapper or Mapper(), threads, data)
        except _lib.PhzError as e:
            if getattr(e, "status", None) != _lib.PHZ_E_UNSUPPORTED:
                raise
            stream = io.TextIOWrapper(io.BytesIO(data))
    contigs: List[str] = []
    # records grouped per chromosome in input order
    chrom_order: List[str] = []
    by_chrom = {}
    events = {}                 # per chromosome: (POS, index among the kept records or -1) of EVERY record, in stream order
    last_chrom = None
    read_counter = 0
    for line in stream:
        cols = line.rstrip().split("\t")
        if cols[0][0:3] == "@SQ":
            contigs.append(cols[1].split(":")[1])
        elif cols[0][0:1] != "@":
            read_counter += 1
            template_length = abs(int(cols[8]))
            if cols[2] != last_chrom:
                if cols[2] in events:
                    # The reference cannot follow such a stream either: its variant stream never rewinds (read_variant_map.py:88-93) and
                    # identify_allele (:236) compares positions only, so the returning chromosome's records would be matched against
                    # whatever chromosome's variants sit in the buffer.  phaser.py:1346 hands the mapper one chromosome per run.
                    print("Error, the records of %s are not contiguous in the SAM stream (a chromosome comes back after another one); "
                          "sort the input by coordinate" % cols[2])
                    sys.exit(1)
                last_chrom = cols[2]
                events[last_chrom] = []
            if not (isize_cutoff == 0 or template_length <= isize_cutoff):
                events[last_chrom].append((int(cols[3]), -1))       # dropped, but the reference prunes its variant buffer before it drops it (:37-51)
                continue
            if not (splice == 1 or "N" not in cols[5]):
                events[last_chrom].append((int(cols[3]), -2))       # the reference still consumes the variants behind this record before split_read drops it
                continue
            alignment_score = ""
            for i in range(11, len(cols)):
                if cols[i].startswith("AS:"):
                    alignment_score = str(int(cols[i].split(":")[2]))
            chrom = cols[2]
            if chrom not in by_chrom:
                by_chrom[chrom] = []
                chrom_order.append(chrom)
            events[chrom].append((int(cols[3]), len(by_chrom[chrom])))
            by_chrom[chrom].append((cols[0], int(cols[3]), cols[5], cols[9], cols[10], alignment_score))

    # VCF / BAM contig check (read_variant_map.py:66-71)
    tchroms = []
    for c in table.chr:
        if not tchroms or tchroms[-1] != c:
            tchroms.append(c)
    for rc in chrom_order:
        for vc in tchroms:
            if vc != rc and vc not in contigs:
                print("Error, VCF and BAM contigs do not match VCF = %s BAM = %s" % (vc, rc))
                sys.exit(1)

    mapper = _mapper or Mapper()
    tchr = np.asarray(table.chr, dtype=object)
    with open(o, "w") as out:
        for chrom in chrom_order:
            recs = by_chrom[chrom]
            vsel = np.nonzero(tchr == chrom)[0]
            if len(vsel) == 0 or not recs:
                continue
            vpos = torch.tensor([table.pos[i] for i in vsel], dtype=torch.int32)
            ref_len = torch.tensor([table.ref_len[i] for i in vsel], dtype=torch.uint8)
            # The kernels want a chromosome's records in coordinate order.  A stream that is out of order is mapped in sorted order, its
            # lines are put back into stream order, and the calls the reference's forward-only variant buffer would not have made are
            # dropped (_buffer_floors: a record that steps backwards misses the variants the buffer has already let go)
            order = None; floors = None
            ev = events[chrom]
            if any(ev[i][0] < ev[i - 1][0] for i in range(1, len(ev))):
                floors = _buffer_floors(ev, recs, [table.pos[i] for i in vsel])
                order = sorted(range(len(recs)), key=lambda i: recs[i][1])
                floors = [floors[i] for i in order]
                recs = [recs[i] for i in order]
            shard = soa.pack_sam([(r[1], r[2], r[3], r[4]) for r in recs])
            ind = [_individual_alleles(table.alleles[i], table.gt[i]) for i in vsel]
            general = bool((ref_len != 1).any()) or any(len(a) != 1 or a not in _BASES for al in ind for a in al)
            lines = []; line_rec = []
            if general:
                # indel mode: the kernel classifies against the allele strings (codes 5 / 6) and reports, for any other
                # text, which read offsets compose it
                off = [0]; blob = b""
                for al in ind:
                    for a in (al + ["", ""])[:2]:
                        blob += a.encode(); off.append(len(blob))
                calls, pool = mapper.map_general(shard, vpos, ref_len, torch.tensor(off, dtype=torch.int32),
                                                 torch.tensor(list(blob + b"\0"), dtype=torch.uint8), int(baseq), want_text=True)
                calls = calls.cpu()
                ri = calls.read_idx.tolist(); vi = calls.var_idx.tolist(); cd = calls.code.tolist()
                toff = pool.call_off.tolist(); tro = pool.roff.tolist()
                for k in range(len(ri)):
                    if floors is not None and vi[k] < floors[ri[k]]:
                        continue
                    rec = recs[ri[k]]; v = int(vsel[vi[k]])
                    line_rec.append(ri[k])
                    if cd[k] == 5 or cd[k] == 6:
                        allele = ind[vi[k]][cd[k] - 5]
                    elif cd[k] < 4:
                        allele = _BASES[cd[k]]
                    else:
                        allele = "".join((rec[3][x] if (ord(rec[4][x]) - 33) >= baseq else "N") for x in tro[toff[k]:toff[k + 1]]).replace("D", "")
                    lines.append("\t".join([rec[0], table.id[v], table.rsid[v], allele, rec[5], table.gt[v], table.maf[v]]))
            else:
                calls = mapper.map(shard, vpos, int(baseq), ref_len).cpu()
                ri = calls.read_idx.tolist(); vi = calls.var_idx.tolist(); cd = calls.code.tolist()
                a0 = (calls.aux0.to(torch.int64) & 0xFFFFFFFF).tolist(); a1 = (calls.aux1.to(torch.int64) & 0xFFFFFFFF).tolist()
                for k in range(len(ri)):
                    if floors is not None and vi[k] < floors[ri[k]]:
                        continue
                    rec = recs[ri[k]]; v = int(vsel[vi[k]])
                    line_rec.append(ri[k])
                    allele = _allele_text(cd[k], a0[k], a1[k], rec[3], rec[4], baseq)
                    lines.append("\t".join([rec[0], table.id[v], table.rsid[v], allele, rec[5], table.gt[v], table.maf[v]]))
            if lines and order is not None:
                back = sorted(range(len(lines)), key=lambda k: order[line_rec[k]])          # stable: the calls of a record keep their order
                lines = [lines[k] for k in back]
            if lines:
                out.write("\n".join(lines) + "\n")
