#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of a phasing-only bench run; copies the kernel stats to gpurun_out/<tag>_kernel_stats.csv
# usage: tools/prof_phasing.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-phasing}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $R/bench.py --no-cpu --no-bam --no-c2 --steps 2 --warmup 1 > /tmp/pp.log 2>&1
mkdir -p $R/gpurun_out
f=$(find /tmp/pp -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
tail -c 600 /tmp/pp.log
