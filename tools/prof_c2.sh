#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of a bench run that keeps the configs[1] entry (chr1, 50M records, 40k het SNPs: deep read lists) and a
# small main workload; copies the stats to gpurun_out/<tag>_c2_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-c2}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pc2
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc2 -o p -- python $R/bench.py --no-cpu --no-bam --records 2000000 --snps 40000 --steps 2 --warmup 1 --phasing-passes 2 > /tmp/pc2.log 2>&1
cp "$(find /tmp/pc2 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/${tag}_c2_kernel_stats.csv
tail -c 400 /tmp/pc2.log
