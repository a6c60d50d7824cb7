#!/bin/bash
# Runs ON THE GPU BOX: calibration of rocprofv3's memory-side counters on gfx950 against access patterns with a known byte count
# (tools/membench.py / phz_membench): FETCH_SIZE, WRITE_SIZE and -- where this rocprofv3 lists them -- the raw TCC_EA0 request counters
# they derive from, each in its own --pmc pass (kernel trace only).  Writes gpurun_out/<tag>/calib/{counters.txt,plain.jsonl,<set>.csv,
# calibration.json}; the table is kept under profiles/ and read by bench.py (roofline.traffic).
# usage: tools/prof_calib.sh <tag>
set -u
R=$PWD; OUT=$R/gpurun_out/$1/calib; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null > $OUT/counters_all.txt
grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z0-9_]*REQ[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*\|TCP_TCC_[A-Z0-9_]*" $OUT/counters_all.txt | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
python $R/tools/membench.py 30 3 > $OUT/plain.jsonl 2> $OUT/plain.err; echo "plain rc=$?"; cat $OUT/plain.jsonl | cut -c1-220
run() { name=$1; shift
  rm -rf /tmp/cal_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/cal_$name -o p -- python $R/tools/membench.py 30 2 > /tmp/cal_$name.log 2>&1
  f=$(find /tmp/cal_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/$name.csv; grep "k_mb_" $f >> $OUT/$name.csv; echo "$name: $(wc -l < $OUT/$name.csv) rows"; else echo "no counter file for $name"; tail -3 /tmp/cal_$name.log; fi
}
have() { grep -q "Counter_Name.*:.$1\$" $OUT/counters_all.txt; }
run fetch FETCH_SIZE
run write WRITE_SIZE
set1=""; for c in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum; do have $c && set1="$set1 $c"; done
[ -n "$set1" ] && run ea_rd $set1
set2=""; for c in TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUBBLE_sum; do have $c && set2="$set2 $c"; done
[ -n "$set2" ] && run ea_wr $set2
set3=""; for c in TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum; do have $c && set3="$set3 $c"; done
[ -n "$set3" ] && run tcc $set3
python $R/tools/calib_table.py $OUT > $OUT/calibration.txt; cat $OUT/calibration.txt
