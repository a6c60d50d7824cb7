#!/bin/bash
# Runs ON THE GPU BOX: K_map_general on the configs[1] shard (het SNPs handed over as allele strings): wall time per pass, the
# size of the work list, and the rocprofv3 kernel times of the same command.  usage: tools/prof_general.sh <tag>
set -u
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
PHZ_GEN_DBG=1 timeout 600 python $R/tools/kmap_general_time.py 2>&1 | grep "K_map_general" | tail -2 > $OUT/kmap_general.txt
rm -rf /tmp/pg
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o p -- python $R/tools/kmap_general_time.py > /tmp/pg.log 2>&1
f=$(find /tmp/pg -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python $R/tools/kstats.py $f | grep -v rocprim >> $OUT/kmap_general.txt
cat $OUT/kmap_general.txt
# instruction counters of the same command (separate --pmc pass, kernel trace only)
rm -rf /tmp/pgc
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pgc -o p -- python $R/tools/kmap_general_time.py > /tmp/pgc.log 2>&1
f=$(find /tmp/pgc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" >> $OUT/kmap_general.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if "gen" in n: agg[n[:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    w = d["SQ_WAVES"][-1]
    print("pmc %-34s waves %8.0f  per wave: %s" % (k, w, " ".join("%s=%.0f" % (c.replace("SQ_", ""), v[-1] / w) for c, v in sorted(d.items()) if c != "SQ_WAVES")))
PY
tail -6 $OUT/kmap_general.txt
