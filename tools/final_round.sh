#!/bin/bash
# Runs ON THE GPU BOX: the evidence set of a round -- GPU suite, PMC passes of K_map and of the phasing pass, the default bench line with the rocprofv3 kernel stats of
# the same command, fresh-seed stress / fuzz of the product against the oracles.  usage: tools/final_round.sh <tag> [stress_seed=5100] [fuzz_seed=51]
set -u
R=$PWD; TAG=$1; SSEED=${2:-5100}; FSEED=${3:-51}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
tools/prof_pmc_c3.sh $TAG > $OUT/pmc_kmap.log 2>&1; grep "k_map" $OUT/pmc_kmap.log | cut -c1-300
tools/prof_pmc_tally.sh $TAG > $OUT/pmc_tally.log 2>&1; tail -3 $OUT/pmc_tally.log
cd $R
timeout 900 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc=$?"; grep "^{" $OUT/bench.out | tail -1 > $OUT/bench.json; tail -2 $OUT/bench.err
cd /tmp; rm -rf /tmp/pb
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python $R/bench.py --no-cpu --no-c2 > /tmp/pb.log 2>&1
grep "^{" /tmp/pb.log | tail -1 > $OUT/bench_rocprof.json
f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
cd $R
timeout 900 python tools/stress_parity.py 40 $SSEED > $OUT/stress_parity.txt 2>&1; tail -3 $OUT/stress_parity.txt
timeout 600 python tools/fuzz_product_mapper.py 300 $FSEED > $OUT/fuzz_product_mapper.txt 2>&1; tail -2 $OUT/fuzz_product_mapper.txt
tools/prof_pass_gaps.sh $TAG 1.0 > /dev/null 2>&1; head -12 $OUT/pass_gaps_1.0.txt
timeout 600 python tools/pass_sweep.py --c2 > $OUT/pass_sweep.txt 2>&1; tail -3 $OUT/pass_sweep.txt | cut -c1-200
