#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of phasing passes at one share of configs[2] -> gaps between kernels of the last pass (tools/pass_gaps.py)
# usage: tools/prof_pass_gaps.sh <tag> <share>
set -u
R=$PWD; TAG=$1; SH=${2:-0.125}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pg
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o p -- python $R/tools/pass_sweep.py --shares $SH --passes 4 > $OUT/gaps_$SH.log 2>&1
f=$(find /tmp/pg -name "*kernel_trace.csv" | head -1)
python $R/tools/pass_gaps.py $f 1 > $OUT/pass_gaps_$SH.txt 2>&1
cat $OUT/pass_gaps_$SH.txt
