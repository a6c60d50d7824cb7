#!/bin/bash
# Runs ON THE GPU BOX: tools/final_round.sh without the PMC passes (for a round end where phz_map.hip / phz_tally.hip / phz_rowsdev.hip did not change since the
# committed passes: bench.py gates them by source hash).  usage: tools/final_round_nopmc.sh <tag> [stress_seed]
set -u
R=$PWD; TAG=$1; SEED=${2:-5100}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc=$?"; grep "^{" $OUT/bench.out | tail -1 > $OUT/bench.json; tail -2 $OUT/bench.err
cd /tmp; rm -rf /tmp/pb
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python $R/bench.py --no-cpu --no-c2 --no-bam > /tmp/pb.log 2>&1
grep "^{" /tmp/pb.log | tail -1 > $OUT/bench_rocprof.json
f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
cd $R
timeout 900 python tools/stress_parity.py 40 $SEED > $OUT/stress_parity.txt 2>&1; tail -2 $OUT/stress_parity.txt
timeout 600 python tools/fuzz_product_mapper.py 300 $SEED > $OUT/fuzz_product_mapper.txt 2>&1; tail -2 $OUT/fuzz_product_mapper.txt
