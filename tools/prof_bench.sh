#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the default bench command; prints our kernels and copies the stats file.
# usage: tools/prof_bench.sh <name>   -> gpurun_out/<name>_kernel_stats.csv + gpurun_out/<name>.json
set -u
R=$PWD; mkdir -p $R/gpurun_out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python $R/bench.py > /tmp/pb.log 2>&1
grep "^{" /tmp/pb.log | tail -1 > $R/gpurun_out/$1.json
f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/$1_kernel_stats.csv
python $R/tools/kstats.py $f | grep -v rocprim
