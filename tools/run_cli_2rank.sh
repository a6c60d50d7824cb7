#!/bin/bash
# Runs ON THE GPU BOX after tools/run_cli_scale.py left /tmp/cli_scale.bam + /tmp/cli_scale.vcf.gz: the same CLI command as 2 ranks
# sharing the box's one GPU (PHZ_DIST_BACKEND=gloo carries the collectives): every rank opens only its chromosomes' BGZF members,
# chromosomes are LPT-assigned by BAM byte spans, row text goes through spool files.  Compares the outputs with the 1-rank run.
set -u
R=$PWD
export PHZ_DIST_BACKEND=gloo PHZ_TIMING=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29731 -m phaser_amd.phaser \
  --vcf /tmp/cli_scale.vcf.gz --bam /tmp/cli_scale.bam --sample S1 --mapq 255 --baseq 10 --paired_end 1 --o /tmp/cli_2rank_out --threads 16 --write_vcf 0 2>&1 | grep -v "^$" | tail -25
for f in allelic_counts variant_connections haplotypes haplotypic_counts allele_config; do
  a=$(md5sum < /tmp/cli_scale_out.$f.txt); b=$(md5sum < /tmp/cli_2rank_out.$f.txt)
  if [ "$a" == "$b" ]; then echo "$f: identical to the 1-rank run"; else echo "$f: DIFFERS"; fi
done
