#!/bin/bash
# Runs ON THE GPU BOX: are the CLI's outputs a function of its input files alone?  Generates the inputs twice and runs the CLI on the
# same files three times, with device memory filled with a byte pattern by a throw-away process in between (a kernel that reads
# memory it never wrote shows up as a changed hash).  usage: tools/cli_determinism.sh [scale=0.25]
set -u
S=${1:-0.25}
hashes() { for n in allelic_counts variant_connections haplotypes haplotypic_counts allele_config; do sha256sum /tmp/cli_scale_out.$n.txt | cut -c1-16; done | tr '\n' ' '; zcat /tmp/cli_scale_out.vcf.gz | sha256sum | cut -c1-16; }
fill() { python -c "
import torch
free, total = torch.cuda.mem_get_info()
x = torch.empty(int(free * 0.9), dtype=torch.uint8, device='cuda'); x.fill_($1); torch.cuda.synchronize()
print('filled %.0f GB of device memory with byte $1' % (x.numel() / 1e9))"; }
rerun() { python -c "
import sys; sys.path.insert(0, '.')
from phaser_amd import phaser
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    rc = phaser.main(['--vcf', '/tmp/cli_scale.vcf.gz', '--bam', '/tmp/cli_scale.bam', '--sample', 'S1', '--mapq', '255', '--baseq', '10', '--paired_end', '1', '--o', '/tmp/cli_scale_out', '--threads', '64', '--write_vcf', '1'])
print('rc', rc)"; }
python tools/run_cli_scale.py $S 64 1 1 2>&1 | grep "^inputs\|CLI rc"
echo "gen A  bam $(sha256sum /tmp/cli_scale.bam | cut -c1-16)  outputs $(hashes)"
fill 165
rerun; echo "rerun on A after 0xA5 fill: outputs $(hashes)"
fill 255
rerun; echo "rerun on A after 0xFF fill: outputs $(hashes)"
rerun; echo "rerun on A, no fill:        outputs $(hashes)"
fill 90
python tools/run_cli_scale.py $S 64 1 1 2>&1 | grep "^inputs\|CLI rc"
echo "gen B  bam $(sha256sum /tmp/cli_scale.bam | cut -c1-16)  outputs $(hashes)"
