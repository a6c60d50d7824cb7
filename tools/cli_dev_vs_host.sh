#!/bin/bash
# Runs ON THE GPU BOX after tools/run_cli_scale.py left /tmp/cli_scale.bam and /tmp/cli_scale.vcf.gz: the CLI once with the device BAM
# path and once with PHZ_BAM_HOST=1; the five files must be byte-identical.
set -u
cd $(dirname $0)/..
for mode in dev host; do
  if [ $mode = host ]; then export PHZ_BAM_HOST=1; else unset PHZ_BAM_HOST; fi
  t0=$(date +%s%N)
  python -m phaser_amd.phaser --vcf /tmp/cli_scale.vcf.gz --bam /tmp/cli_scale.bam --sample S1 --mapq 255 --baseq 10 --paired_end 1 \
      --o /tmp/cmp_$mode --threads 32 --write_vcf 0 > /tmp/cmp_$mode.log 2>&1
  rc=$?; t1=$(date +%s%N)
  echo "$mode: $(( (t1 - t0) / 1000000 )) ms wall (process start to exit), rc=$rc; $(grep -c . /tmp/cmp_$mode.log) log lines"
done
for f in allelic_counts variant_connections haplotypes haplotypic_counts allele_config; do
  if cmp -s /tmp/cmp_dev.$f.txt /tmp/cmp_host.$f.txt; then echo "$f identical ($(stat -c %s /tmp/cmp_dev.$f.txt) bytes)"; else echo "$f DIFFERS"; fi
done
