"""Seeded synthetic VCF / read generator for the phASER hot path (own code, no reference parts).

Shapes follow SURVEY.md section 8(d): fixed-length paired reads, CIGAR mix
68 % plain M, 22 % one N, 4 % two N, 2 % I, 2 % D, 2 % soft clip; Illumina-binned
qualities; 0.2 % base error; MAPQ / duplicate / proper-pair flag mixes; AS:i tag.

Everything is produced as structure-of-arrays torch tensors so the same code
runs on the CPU (golden fixtures, tests) and directly in HBM on cuda:0 (bench).
Text renderings (SAM / VCF / variant table) are only for small inputs.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np
import torch

# BAM CIGAR op codes (SAM spec 4.2): M I D N S H P = X
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)
CIGAR_CHARS = "MIDNSHP=X"
BASES = "ACGT"

# qual byte layout used by the SoA: low 7 bits = phred, bit 7 set
# when the base is not one of ACGT; the 2-bit seq code then carries the subtype.
QUAL_NONACGT = 0x80
SUB_N = 0        # behaves like 'N': never yields a call
SUB_IUPAC = 1    # any other character: yields a call that matches no allele


def ref_base(g: torch.Tensor) -> torch.Tensor:
    """Stateless pseudo-random reference base (0..3) at 1-based genome position g."""
    x = (g.to(torch.int64) * 2654435761) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 2246822519) & 0xFFFFFFFF
    x = x ^ (x >> 13)
    return (x & 3).to(torch.uint8)


@dataclasses.dataclass
class Variants:
    chrom: str
    pos: torch.Tensor        # int32 [n] sorted, unique, 1-based
    ref: torch.Tensor        # uint8 [n] base code
    alt: torch.Tensor        # uint8 [n] base code
    gt: List[str]            # "0|1", "1|0" or "0/1"
    hap_alt: torch.Tensor    # uint8 [n]: which haplotype (0/1) carries ALT
    rsid: List[str]
    ref_text: Optional[List[str]] = None   # set when some variants are indels (REF / ALT longer than one base)
    alt_text: Optional[List[str]] = None

    def __len__(self):
        return int(self.pos.numel())


def make_variants(chrom: str, start: int, end: int, n_snps: int, seed: int,
                  n_genes: Optional[int] = None, unphased_frac: float = 0.05,
                  dot_id_frac: float = 0.1, indel_frac: float = 0.0):
    """Het SNPs clustered into 'genes'; returns (Variants, gene_start, gene_end, gene_weight)."""
    g = torch.Generator().manual_seed(seed)
    if n_genes is None:
        n_genes = max(1, n_snps // 10)
    span = torch.exp(torch.rand(n_genes, generator=g) * (np.log(20000.0) - np.log(1000.0)) + np.log(1000.0)).to(torch.int64)
    gstart = torch.randint(start, max(start + 1, end - 20001), (n_genes,), generator=g, dtype=torch.int64)
    gend = gstart + span
    weight = torch.exp(torch.randn(n_genes, generator=g))
    # SNP positions: uniform inside genes, genes drawn uniformly; oversample then unique
    which = torch.randint(0, n_genes, (n_snps * 2 + 16,), generator=g)
    off = (torch.rand(n_snps * 2 + 16, generator=g) * span[which].double()).to(torch.int64)
    pos = torch.unique(gstart[which] + off)
    perm = torch.randperm(pos.numel(), generator=g)[:n_snps]
    pos = torch.sort(pos[perm]).values
    n = pos.numel()
    ref = ref_base(pos)
    alt = ((ref.to(torch.int64) + 1 + torch.randint(0, 3, (n,), generator=g)) % 4).to(torch.uint8)
    hap_alt = torch.randint(0, 2, (n,), generator=g).to(torch.uint8)
    unph = torch.rand(n, generator=g) < unphased_frac
    dot = torch.rand(n, generator=g) < dot_id_frac
    gt = []
    rsid = []
    unph_l = unph.tolist(); hap_l = hap_alt.tolist(); dot_l = dot.tolist()
    for i in range(n):
        if unph_l[i]:
            gt.append("0/1")
        else:
            gt.append("1|0" if hap_l[i] == 0 else "0|1")
        rsid.append("." if dot_l[i] else "rs%d" % (1000 + i))
    v = Variants(chrom, pos.to(torch.int32), ref, alt, gt, hap_alt, rsid)
    if indel_frac > 0:
        # a share of the sites become deletions (REF = 2-4 reference bases, ALT = its first base) or insertions
        # (REF = one base, ALT = that base + 1-3 bases); reads are NOT edited for them (parity fixtures, not biology)
        kind = torch.rand(n, generator=g)
        extra = torch.randint(1, 4, (n,), generator=g)
        ins_bases = torch.randint(0, 4, (n, 3), generator=g)
        v.ref_text = []; v.alt_text = []
        for i in range(n):
            p0 = int(pos[i])
            if float(kind[i]) < indel_frac / 2:
                k = int(extra[i])
                v.ref_text.append("".join(BASES[int(ref_base(torch.tensor([p0 + d]))[0])] for d in range(k + 1)))
                v.alt_text.append(BASES[int(ref[i])])
            elif float(kind[i]) < indel_frac:
                k = int(extra[i])
                v.ref_text.append(BASES[int(ref[i])])
                v.alt_text.append(BASES[int(ref[i])] + "".join(BASES[int(b)] for b in ins_bases[i, :k]))
            else:
                v.ref_text.append(BASES[int(ref[i])]); v.alt_text.append(BASES[int(alt[i])])
    return v, gstart, gend, weight


@dataclasses.dataclass
class ReadBatch:
    """Coordinate-sorted records of ONE chromosome as structure-of-arrays (pre-filter)."""
    chrom: str
    L: int
    pos: torch.Tensor        # int32 [n] 1-based leftmost aligned position (SAM POS)
    flag: torch.Tensor       # int32 [n]
    mapq: torch.Tensor       # uint8 [n]
    tlen: torch.Tensor       # int32 [n]
    aln_score: torch.Tensor  # int32 [n]  (AS:i)
    qid: torch.Tensor        # int32 [n]  template id; QNAME = prefix + str(qid)
    cigar_off: torch.Tensor  # int64 [n+1]
    cigar: torch.Tensor      # uint32-as-int64 [n_ops]  len<<4 | op
    seq: torch.Tensor        # uint8 [n, L] base code 0..3; 4 = N
    qual: torch.Tensor       # uint8 [n, L] phred
    qname_prefix: str = "s0.b0.r"

    def __len__(self):
        return int(self.pos.numel())

    def select(self, keep: torch.Tensor) -> "ReadBatch":
        idx = torch.nonzero(keep).flatten()
        counts = (self.cigar_off[1:] - self.cigar_off[:-1])[idx]
        new_off = torch.zeros(idx.numel() + 1, dtype=torch.int64, device=self.pos.device)
        new_off[1:] = torch.cumsum(counts, 0)
        # gather ragged cigar
        rep = torch.repeat_interleave(torch.arange(idx.numel(), device=idx.device), counts)
        within = torch.arange(int(new_off[-1]), device=idx.device) - new_off[:-1][rep]
        src = self.cigar_off[:-1][idx][rep] + within
        return ReadBatch(self.chrom, self.L, self.pos[idx], self.flag[idx], self.mapq[idx], self.tlen[idx],
                         self.aln_score[idx], self.qid[idx], new_off, self.cigar[src], self.seq[idx], self.qual[idx],
                         self.qname_prefix)

    def qname(self, i: int) -> str:
        return self.qname_prefix + str(int(self.qid[i]))


def samtools_keep(rb: ReadBatch, mapq: int, remove_dups: bool = True, paired_end: bool = True) -> torch.Tensor:
    """Filter semantics of the reference's samtools pipeline (phaser.py:1346, :505-513):
    -q MAPQ, -F 0x400 when remove_dups, -f 2 when paired_end."""
    keep = rb.mapq.to(torch.int32) >= mapq
    if remove_dups:
        keep &= (rb.flag & 0x400) == 0
    if paired_end:
        keep &= (rb.flag & 0x2) != 0
    return keep


@dataclasses.dataclass
class ReadPlan:
    """Per-read primitive fields, coordinate-sorted; the expensive per-base content is filled per chunk."""
    chrom: str
    L: int
    pos: torch.Tensor      # int64
    flag: torch.Tensor
    mapq: torch.Tensor
    tlen: torch.Tensor
    qid: torch.Tensor
    hap: torch.Tensor
    seed: int

    def __len__(self):
        return int(self.pos.numel())


def make_read_plan(v: Variants, gstart, gend, weight, n_pairs: int, seed: int, L: int = 76,
                   device: str = "cpu", all_pass: bool = False) -> ReadPlan:
    """Place read pairs on genes (log-normal expression), draw flags/MAPQ, sort by position.
    all_pass=True makes every record pass the samtools filters (bench: the filter is upstream of the path)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda *shape: torch.rand(*shape, generator=g, device=dev)
    prob = (weight / weight.sum()).to(dev)          # normalised on the host: a device-side sum may differ in the last bit run to run
    gstart = gstart.to(dev); gend = gend.to(dev); weight = weight.to(dev)
    n = 2 * n_pairs
    gene = torch.multinomial(prob, n_pairs, replacement=True, generator=g)
    frag_start = gstart[gene] - L + (rnd(n_pairs).double() * (gend[gene] - gstart[gene] + L).double()).to(torch.int64)
    frag_start = torch.clamp(frag_start, min=1)
    tl = torch.clamp((torch.randn(n_pairs, generator=g, device=dev) * 60 + 250).to(torch.int64), min=L)
    hap = torch.randint(0, 2, (n_pairs,), generator=g, device=dev, dtype=torch.int64)
    pos = torch.stack([frag_start, frag_start + tl - L], 1).reshape(n)
    tlen = torch.stack([tl, -tl], 1).reshape(n)
    pair = torch.arange(n_pairs, device=dev).repeat_interleave(2)
    proper = (rnd(n_pairs) < 0.97).repeat_interleave(2)
    dup = (rnd(n_pairs) < 0.15).repeat_interleave(2)
    mate2 = (torch.arange(n, device=dev) & 1) == 1
    flag = torch.where(mate2, torch.full((n,), 0x1 | 0x10 | 0x80, device=dev), torch.full((n,), 0x1 | 0x20 | 0x40, device=dev))
    mq_r = rnd(n)
    lowq = torch.tensor([0, 1, 3], device=dev)[torch.randint(0, 3, (n,), generator=g, device=dev)]
    mapq = torch.where(mq_r < 0.9, torch.full((n,), 255, device=dev), lowq)
    if all_pass:
        flag = flag | 0x2
        mapq = torch.full((n,), 255, device=dev)
    else:
        flag = flag | torch.where(proper, 0x2, 0) | torch.where(dup, 0x400, 0)
    order = torch.sort(pos, stable=True).indices
    return ReadPlan(v.chrom, L, pos[order], flag[order].to(torch.int32), mapq[order].to(torch.uint8),
                    tlen[order].to(torch.int32), pair[order].to(torch.int32), hap.repeat_interleave(2)[order], seed)


def fill_reads(plan: ReadPlan, lo: int, hi: int, v: Variants, n_rate: float = 0.0005, err_rate: float = 0.002,
               qname_prefix: str = "s0.b0.r") -> ReadBatch:
    """Generate alignment templates, bases, qualities and AS for plan records [lo, hi)."""
    dev = plan.pos.device
    L = plan.L
    g = torch.Generator(device=dev).manual_seed(plan.seed * 1000003 + lo)
    rnd = lambda *shape: torch.rand(*shape, generator=g, device=dev)
    rint = lambda a, b, shape: torch.randint(a, b, shape, generator=g, device=dev, dtype=torch.int64)
    snp_pos = v.pos.to(dev).to(torch.int64)
    pos = plan.pos[lo:hi]
    hap_r = plan.hap[lo:hi]
    n = hi - lo

    # ---- alignment templates
    t = rnd(n)
    typ = torch.zeros(n, dtype=torch.int64, device=dev)
    for thr in [0.68, 0.90, 0.94, 0.96, 0.98]:
        typ += (t >= thr).to(torch.int64)
    lead = torch.zeros(n, dtype=torch.int64, device=dev); trail = torch.zeros_like(lead)
    blk = torch.zeros(n, 3, dtype=torch.int64, device=dev)
    gtype = torch.full((n, 2), -1, dtype=torch.int64, device=dev)
    glen = torch.zeros(n, 2, dtype=torch.int64, device=dev)

    def intron(m):
        return torch.exp(rnd(m).double() * (np.log(50000.0) - np.log(50.0)) + np.log(50.0)).to(torch.int64)

    blk[:, 0] = L
    m = typ == 1
    s = 10 + (rnd(n) * (L - 20)).to(torch.int64)
    blk[m, 0] = s[m]; blk[m, 1] = L - s[m]; gtype[m, 0] = OP_N; glen[m, 0] = intron(n)[m]
    m = typ == 2
    s1 = 10 + (rnd(n) * (L - 40)).to(torch.int64)
    s2 = s1 + 10 + (rnd(n) * (L - 20 - s1).clamp(min=1)).to(torch.int64)
    s2 = torch.minimum(s2, torch.full_like(s2, L - 10))
    blk[m, 0] = s1[m]; blk[m, 1] = (s2 - s1)[m]; blk[m, 2] = L - s2[m]
    gtype[m, 0] = OP_N; gtype[m, 1] = OP_N; glen[m, 0] = intron(n)[m]; glen[m, 1] = intron(n)[m]
    m = typ == 3
    k = 1 + rint(0, 3, (n,)); a = 5 + (rnd(n) * (L - 10 - k)).to(torch.int64)
    blk[m, 0] = a[m]; blk[m, 1] = (L - k - a)[m]; gtype[m, 0] = OP_I; glen[m, 0] = k[m]
    m = typ == 4
    k = 1 + rint(0, 5, (n,)); a = 5 + (rnd(n) * (L - 10)).to(torch.int64)
    blk[m, 0] = a[m]; blk[m, 1] = (L - a)[m]; gtype[m, 0] = OP_D; glen[m, 0] = k[m]
    m = typ == 5
    k = 1 + rint(0, 10, (n,)); side = rnd(n) < 0.5
    lead[m & side] = k[m & side]; trail[m & ~side] = k[m & ~side]; blk[m, 0] = (L - k)[m]

    # ---- per-base genome coordinate (-1 where the base is not aligned)
    j = torch.arange(L, device=dev).unsqueeze(0)
    r = j - lead.unsqueeze(1)
    G = torch.full((n, L), -1, dtype=torch.int64, device=dev)
    rs = torch.zeros(n, dtype=torch.int64, device=dev)   # read offset (after lead clip) of current block
    go = torch.zeros(n, dtype=torch.int64, device=dev)   # genome offset of current block
    for b in range(3):
        inb = (r >= rs.unsqueeze(1)) & (r < (rs + blk[:, b]).unsqueeze(1))
        G = torch.where(inb, pos.unsqueeze(1) + go.unsqueeze(1) + (r - rs.unsqueeze(1)), G)
        if b < 2:
            is_i = gtype[:, b] == OP_I
            has = gtype[:, b] >= 0
            rs = rs + blk[:, b] + torch.where(is_i, glen[:, b], torch.zeros_like(rs))
            go = go + blk[:, b] + torch.where(has & ~is_i, glen[:, b], torch.zeros_like(go))

    aligned = G >= 0
    base = ref_base(torch.clamp(G, min=0)).to(torch.int64)
    if snp_pos.numel() > 0:
        idx = torch.searchsorted(snp_pos, torch.clamp(G, min=0)).clamp(max=snp_pos.numel() - 1)
        hit = aligned & (snp_pos[idx] == G)
        carries_alt = v.hap_alt.to(dev).to(torch.int64)[idx] == hap_r.unsqueeze(1)
        base = torch.where(hit & carries_alt, v.alt.to(dev).to(torch.int64)[idx], base)
    rand_base = rint(0, 4, (n, L))
    base = torch.where(aligned, base, rand_base)
    err = rnd(n, L) < err_rate
    base = torch.where(err, (base + 1 + rint(0, 3, (n, L))) % 4, base)
    isn = rnd(n, L) < n_rate
    seq = torch.where(isn, torch.full_like(base, 4), base).to(torch.uint8)
    qb = rnd(n, L)
    qual = torch.full((n, L), 37, dtype=torch.uint8, device=dev)
    qual[qb < 0.20] = 25
    qual[qb < 0.05] = 11
    qual[qb < 0.02] = 2
    mism = (err & aligned).sum(1)
    gaps = ((gtype == OP_I) | (gtype == OP_D)).sum(1)
    aln = (2 * L - 2 * mism - gaps).to(torch.int32)

    # ---- packed cigar via 7 slots: S M g M g M S
    slot_op = torch.stack([torch.full_like(lead, OP_S), torch.full_like(lead, OP_M), gtype[:, 0].clamp(min=0),
                           torch.full_like(lead, OP_M), gtype[:, 1].clamp(min=0), torch.full_like(lead, OP_M),
                           torch.full_like(lead, OP_S)], 1)
    slot_len = torch.stack([lead, blk[:, 0], glen[:, 0], blk[:, 1], glen[:, 1], blk[:, 2], trail], 1)
    valid = torch.stack([lead > 0, blk[:, 0] > 0, gtype[:, 0] >= 0, blk[:, 1] > 0, gtype[:, 1] >= 0, blk[:, 2] > 0,
                         trail > 0], 1)
    counts = valid.sum(1)
    cigar_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    cigar_off[1:] = torch.cumsum(counts, 0)
    cigar = ((slot_len << 4) | slot_op)[valid]
    return ReadBatch(plan.chrom, L, pos.to(torch.int32), plan.flag[lo:hi], plan.mapq[lo:hi], plan.tlen[lo:hi], aln,
                     plan.qid[lo:hi], cigar_off, cigar, seq, qual, qname_prefix)


def make_reads(v: Variants, gstart, gend, weight, n_pairs: int, seed: int, L: int = 76,
               device: str = "cpu", qname_prefix: str = "s0.b0.r", n_rate: float = 0.0005,
               err_rate: float = 0.002) -> ReadBatch:
    """All records of one BAM x chromosome (pre-filter), coordinate-sorted."""
    plan = make_read_plan(v, gstart, gend, weight, n_pairs, seed, L, device)
    return fill_reads(plan, 0, len(plan), v, n_rate, err_rate, qname_prefix)


# --------------------------------------------------------------------------- text renderings (small inputs)

def cigar_string(rb: ReadBatch, i: int) -> str:
    a, b = int(rb.cigar_off[i]), int(rb.cigar_off[i + 1])
    if a == b:
        return "*"
    return "".join("%d%s" % (int(c) >> 4, CIGAR_CHARS[int(c) & 15]) for c in rb.cigar[a:b].tolist())


def sam_lines(rb: ReadBatch, contigs: List[tuple]) -> List[str]:
    """SAM text exactly as the mapper expects on stdin (header @SQ lines + records)."""
    out = ["@HD\tVN:1.6\tSO:coordinate"]
    for name, ln in contigs:
        out.append("@SQ\tSN:%s\tLN:%d" % (name, ln))
    seq = rb.seq.cpu().numpy(); qual = rb.qual.cpu().numpy()
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    pos = rb.pos.tolist(); flag = rb.flag.tolist(); mapq = rb.mapq.tolist(); tlen = rb.tlen.tolist()
    asc = rb.aln_score.tolist(); qid = rb.qid.tolist()
    for i in range(len(rb)):
        s = lut[seq[i]].tobytes().decode()
        q = (qual[i] + 33).astype(np.uint8).tobytes().decode()
        mate_pos = pos[i] + tlen[i] - rb.L if tlen[i] > 0 else pos[i] + tlen[i] + rb.L
        out.append("\t".join([rb.qname_prefix + str(qid[i]), str(flag[i]), rb.chrom, str(pos[i]), str(mapq[i]),
                              cigar_string(rb, i), "=", str(max(1, mate_pos)), str(tlen[i]), s, q,
                              "NH:i:1", "HI:i:1", "AS:i:%d" % asc[i], "nM:i:0"]))
    return out


def vcf_lines(vs: List[Variants], sample: str = "S1") -> List[str]:
    """Plain-text single-sample VCF (10 columns) as parse_sample's shell filter leaves it."""
    out = ["##fileformat=VCFv4.2", "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">",
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample]
    for v in vs:
        pos = v.pos.tolist(); ref = v.ref.tolist(); alt = v.alt.tolist()
        for i in range(len(v)):
            rt = BASES[ref[i]] if v.ref_text is None else v.ref_text[i]
            at = BASES[alt[i]] if v.alt_text is None else v.alt_text[i]
            out.append("\t".join([v.chrom, str(pos[i]), v.rsid[i], rt, at, "100", "PASS",
                                  "AF=0.%d" % (1 + (pos[i] % 49)), "GT", v.gt[i]]))
    return out
