#!/bin/bash
# Runs ON THE GPU BOX: the GPU test suite, the default bench line, and a rocprofv3 kernel-stats pass of the same bench command.
# usage: tools/gpu_round.sh <tag> [pytest-args...]      -> gpurun_out/<tag>/{pytest.log,bench.json,bench.err,kernel_stats.csv,...}
set -u
R=$PWD; TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc=$?"
grep "^{" $OUT/bench.out | tail -1 > $OUT/bench.json
tail -3 $OUT/bench.err
cd /tmp; rm -rf /tmp/pb
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python $R/bench.py --no-cpu --no-c2 > /tmp/pb.log 2>&1
grep "^{" /tmp/pb.log | tail -1 > $OUT/bench_rocprof.json
f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/kernel_stats.csv && python $R/tools/kstats.py $f | head -40
