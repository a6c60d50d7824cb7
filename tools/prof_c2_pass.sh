#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of phasing passes over the configs[1] shard alone (chr1, 50 M records, 40,000 het SNPs), then a
# PMC pass (FETCH_SIZE / WRITE_SIZE per kernel) of the same command.    usage: tools/prof_c2_pass.sh <tag> [passes]
set -u
R=$PWD; TAG=$1; P=${2:-12}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pc2
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc2 -o p -- python $R/tools/pass_sweep.py --only-c2 --passes $((P - 1)) > $OUT/c2_pass_rocprof.log 2>&1
f=$(find /tmp/pc2 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/c2_pass_kernel_stats.csv && python $R/tools/kstats_pass.py $f $P 15 > $OUT/c2_pass_kernels.txt
tail -3 $OUT/c2_pass_rocprof.log; head -40 $OUT/c2_pass_kernels.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc2p
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pc2p -o p -- python $R/tools/pass_sweep.py --only-c2 --passes 1 > /tmp/pc2p.log 2>&1
  f=$(find /tmp/pc2p -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $c > $OUT/c2_pass_pmc_$c.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if "at::native" in k or "rocclr" in k or "k_map" in k or "k_compact" in k or "k_tile_window" in k:
        continue
    acc[k[:60]] += float(r["Counter_Value"]); n[k[:60]] += 1
print("# %s in KiB summed over 2 passes (one sizing, one reported); FETCH_SIZE x2 on gfx950" % sys.argv[2])
for k in sorted(acc, key=lambda k: -acc[k])[:40]:
    print("%-62s disp %4d  %12.1f KiB" % (k, n[k], acc[k]))
print("total KiB", sum(acc.values()))
PY
done
head -12 $OUT/c2_pass_pmc_FETCH_SIZE.txt
