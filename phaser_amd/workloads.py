"""Synthetic workloads of BASELINE.json's configs, generated directly in HBM (SURVEY.md 8(d) shapes)."""
from __future__ import annotations

import torch

from . import soa, synth

CHR1_LEN = 248_956_422

CONFIGS = {
    # name: (chrom, length, het SNPs, records)
    "C1": ("chr22", 20_000_000, 1_000, 100_000),
    "C2": ("chr1", CHR1_LEN, 40_000, 50_000_000),
}


# hg38 autosome lengths, chr1..chr22
HG38_AUTOSOMES = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
                  133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468]


def genome_plan(total_records: int = 80_000_000, total_snps: int = 1_500_000, seed: int = 777, scale: float = 1.0):
    """BASELINE.json configs[2]: one GTEx-shape sample over the autosomes, het SNPs and records proportional to
    chromosome length.  -> [(chrom, length, n_snps, n_records, seed)] in VCF order; the same plan on every rank."""
    total_len = sum(HG38_AUTOSOMES)
    return [("chr%d" % (i + 1), ln, max(1, int(total_snps * scale * ln / total_len)), max(2, int(total_records * scale * ln / total_len)), seed + i)
            for i, ln in enumerate(HG38_AUTOSOMES)]


def make_shard(chrom: str, length: int, n_snps: int, n_records: int, seed: int, device: str,
               chunk: int = 2_000_000, keep_sample: int = 0, read_seed=None):
    """One (chromosome, BAM) shard with every record passing the upstream samtools filters.
    Returns (variants, ReadShard on `device`, sample ReadBatch on the CPU holding the first keep_sample records).
    read_seed: seed of the reads alone (several BAMs of one sample share `seed`, i.e. the variants, and differ in read_seed)."""
    v, gs, ge, w = synth.make_variants(chrom, 1, length, n_snps, seed, n_genes=max(1, n_snps // 10))
    plan = synth.make_read_plan(v, gs, ge, w, (n_records + 1) // 2, seed + 1 if read_seed is None else read_seed, device=device, all_pass=True)
    n = len(plan)
    parts = []
    sample = None
    cig_base = 0
    seq_base = 0
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        rb = synth.fill_reads(plan, lo, hi, v)
        if lo < keep_sample:
            m = min(keep_sample, hi) - lo
            keep = torch.zeros(hi - lo, dtype=torch.bool, device=rb.pos.device); keep[:m] = True
            s = rb.select(keep)
            s = synth.ReadBatch(s.chrom, s.L, s.pos.cpu(), s.flag.cpu(), s.mapq.cpu(), s.tlen.cpu(), s.aln_score.cpu(),
                                s.qid.cpu(), s.cigar_off.cpu(), s.cigar.cpu(), s.seq.cpu(), s.qual.cpu(), s.qname_prefix)
            if sample is None:
                sample = s
            else:       # append (host side)
                sample = synth.ReadBatch(s.chrom, s.L, torch.cat([sample.pos, s.pos]), torch.cat([sample.flag, s.flag]),
                                         torch.cat([sample.mapq, s.mapq]), torch.cat([sample.tlen, s.tlen]),
                                         torch.cat([sample.aln_score, s.aln_score]), torch.cat([sample.qid, s.qid]),
                                         torch.cat([sample.cigar_off, s.cigar_off[1:] + sample.cigar_off[-1]]),
                                         torch.cat([sample.cigar, s.cigar]), torch.cat([sample.seq, s.seq]),
                                         torch.cat([sample.qual, s.qual]), s.qname_prefix)
        sh = soa.pack_readbatch(rb)
        parts.append((sh, cig_base, seq_base))
        cig_base += int(sh.cigar.numel()); seq_base += int(sh.seq2.numel())
        del rb
    if cig_base >= 2 ** 31 or seq_base >= 2 ** 31:
        raise ValueError("shard exceeds 32-bit offsets")
    cat = torch.cat
    # QNAME ids as an interner hands them out: numbered by first appearance in coordinate order (mates share the id)
    q = cat([p[0].qid for p in parts]).to(torch.int64)
    nq = int(q.max()) + 1
    first = torch.full((nq,), q.numel(), dtype=torch.int64, device=q.device)
    first.scatter_reduce_(0, q, torch.arange(q.numel(), device=q.device), "amin")
    rank = torch.empty(nq, dtype=torch.int64, device=q.device)
    rank[torch.argsort(first)] = torch.arange(nq, device=q.device)
    qid_all = rank[q].to(torch.int32)
    shard = soa.ReadShard(
        cat([p[0].pos for p in parts]),
        cat([p[0].cigar_off[:-1] + p[1] for p in parts] + [torch.tensor([cig_base], dtype=torch.int32, device=parts[0][0].pos.device)]),
        cat([p[0].cigar for p in parts]),
        cat([p[0].seq_off[:-1] + p[2] for p in parts] + [torch.tensor([seq_base], dtype=torch.int32, device=parts[0][0].pos.device)]),
        cat([p[0].seq2 for p in parts]), cat([p[0].qual for p in parts]),
        qid_all, cat([p[0].aln_score for p in parts]), cat([p[0].has_as for p in parts]))
    return v, shard, sample
