#!/usr/bin/env python3
"""GPU-box end-to-end run of the drop-in CLI at BASELINE.json configs[2] shape FROM FILES: a synthetic coordinate-sorted BAM
(unfiltered: duplicates, improper pairs, low MAPQ present) and a bgzipped VCF are written to /tmp first (native writers,
not timed), then `python -m phaser_amd.phaser` runs on them exactly as a user would: BGZF inflate + BAM decode + filters +
QNAME interning on the host, H2D, K_map, AS cutoff, K_tally, phasing, the five files (+ the phased VCF with --write_vcf 1).
usage: tools/run_cli_scale.py [scale=1.0] [threads=32] [write_vcf=0] [n_bams=1]   (n_bams > 1: the multi-tissue shape of configs[3])"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from phaser_amd import bamio, synth, vcfout
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
        114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 32
write_vcf = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n_bams = int(sys.argv[4]) if len(sys.argv) > 4 else 1
total_len = sum(HG38)
t0 = time.perf_counter()
refs = [("chr%d" % (i + 1), ln) for i, ln in enumerate(HG38)]
vsets = []
nrec = 0
bams = []
vcfgz = "/tmp/cli_scale.vcf.gz"
t_gen = t_write = 0.0
for bi in range(n_bams):
    tg = time.perf_counter()
    batches = []
    for i, ln in enumerate(HG38):
        chrom = "chr%d" % (i + 1)
        n_snps = int(1_500_000 * scale * ln / total_len); n_pairs = int(40_000_000 * scale * ln / total_len)
        v, gs, ge, w = synth.make_variants(chrom, 1, ln, n_snps, 777 + i, n_genes=max(1, n_snps // 10))
        plan = synth.make_read_plan(v, gs, ge, w, n_pairs, 1777 + i + 1000 * bi, device="cuda")
        for lo in range(0, len(plan), 2_000_000):
            # later BAMs reuse part of the first BAM's QNAME space on purpose (same template names across tissues do occur)
            rb = synth.fill_reads(plan, lo, min(len(plan), lo + 2_000_000), v, qname_prefix="s0.b%d.%d." % (min(bi, 1), i))
            batches.append(synth.ReadBatch(rb.chrom, rb.L, rb.pos.cpu(), rb.flag.cpu(), rb.mapq.cpu(), rb.tlen.cpu(), rb.aln_score.cpu(), rb.qid.cpu(),
                                           rb.cigar_off.cpu(), rb.cigar.cpu(), rb.seq.cpu(), rb.qual.cpu(), rb.qname_prefix))
            nrec += len(rb)
        if bi == 0:
            vsets.append(v)
        del plan
    torch.cuda.synchronize(); t_gen += time.perf_counter() - tg
    tw = time.perf_counter()
    bam = "/tmp/cli_scale.%d.bam" % bi if n_bams > 1 else "/tmp/cli_scale.bam"
    bamio.readbatch_to_bam_native(bam, batches, refs, threads)
    bams.append(bam)
    del batches
    t_write += time.perf_counter() - tw
tw = time.perf_counter()
vcfout.write_bgzf(vcfgz, "\n".join(synth.vcf_lines(vsets)) + "\n", threads, index="vcf")       # the reference asks for a tabix-indexed VCF (phaser.py:31)
t_write += time.perf_counter() - tw
print("inputs: %d records in %d BAM(s), %d het SNPs | generate %.1fs | write BAM (%.2f GB) + VCF.gz (%.1f MB) %.1fs" %
      (nrec, n_bams, sum(len(v) for v in vsets), t_gen, sum(os.path.getsize(b) for b in bams) / 1e9, os.path.getsize(vcfgz) / 1e6, t_write), flush=True)
bam = ",".join(bams)
torch.cuda.empty_cache()
from phaser_amd import phaser
t3 = time.perf_counter()
rc = phaser.main(["--vcf", vcfgz, "--bam", bam, "--sample", "S1", "--mapq", ",".join(["255"] * n_bams), "--baseq", "10", "--paired_end", "1", "--o", "/tmp/cli_scale_out",
                  "--threads", str(threads), "--write_vcf", str(write_vcf)])
t4 = time.perf_counter()
sizes = {n: os.path.getsize("/tmp/cli_scale_out.%s.txt" % n) for n in ("allelic_counts", "variant_connections", "haplotypes", "haplotypic_counts", "allele_config")}
print("CLI rc=%d wall %.1fs for %d BAM records (%.0f records/s end to end from files) | outputs %s" % (rc, t4 - t3, nrec, nrec / (t4 - t3), sizes))
if os.environ.get("PHZ_CLI_SWEEP_CHUNK_MB"):        # BAM-stage experiment: the same command again with other H2D chunk sizes (warm process: compare the bam lines only)
    for mb in os.environ["PHZ_CLI_SWEEP_CHUNK_MB"].split(","):
        os.environ["PHZ_BAM_CHUNK_MB"] = mb
        print("=== PHZ_BAM_CHUNK_MB=%s" % mb, flush=True)
        sys.stderr.write("=== PHZ_BAM_CHUNK_MB=%s\n" % mb); sys.stderr.flush()
        phaser.main(["--vcf", vcfgz, "--bam", bam, "--sample", "S1", "--mapq", ",".join(["255"] * n_bams), "--baseq", "10", "--paired_end", "1", "--o", "/tmp/cli_scale_out",
                     "--threads", str(threads), "--write_vcf", str(write_vcf)])
if os.environ.get("PHZ_CLI_SUBPROCESS"):            # the same command in FRESH processes (what a user runs): the stage timers of each run, PHZ_TIMING=1
    import subprocess
    for k in range(int(os.environ["PHZ_CLI_SUBPROCESS"])):
        env = dict(os.environ, PHZ_TIMING="1")
        t5 = time.perf_counter()
        pr = subprocess.run([sys.executable, "-m", "phaser_amd.phaser", "--vcf", vcfgz, "--bam", bam, "--sample", "S1", "--mapq", ",".join(["255"] * n_bams), "--baseq", "10",
                             "--paired_end", "1", "--o", "/tmp/cli_scale_out", "--threads", str(threads), "--write_vcf", str(write_vcf)], cwd=REPO, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        keep = [l for l in pr.stdout.split("\n") if l.startswith("[phz timing]") and ("bam device:" in l or " vcf" in l or "total" in l or not l.startswith("[phz timing]  "))]
        print("=== fresh process %d: rc %d, process wall %.2f s (interpreter + imports included)\n%s" % (k, pr.returncode, time.perf_counter() - t5, "\n".join(keep)), flush=True)
if os.environ.get("PHZ_CLI_RERUN_ENV"):            # the same command again in this (warm) process under other switches, e.g. "PHZ_BAM_CRC=0;PHZ_BAM_CRC=1": compare the bam lines
    for setting in os.environ["PHZ_CLI_RERUN_ENV"].split(";"):
        k, v_ = setting.split("=", 1)
        os.environ[k] = v_
        sys.stderr.write("=== %s\n" % setting); sys.stderr.flush()
        phaser.main(["--vcf", vcfgz, "--bam", bam, "--sample", "S1", "--mapq", ",".join(["255"] * n_bams), "--baseq", "10", "--paired_end", "1", "--o", "/tmp/cli_scale_out",
                     "--threads", str(threads), "--write_vcf", str(write_vcf)])
