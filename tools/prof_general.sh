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
