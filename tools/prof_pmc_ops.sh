#!/bin/bash
# Runs ON THE GPU BOX: instruction / wait counters of the mapper kernels on the configs[1] shard (20M records to keep it short).
# usage: tools/prof_pmc_ops.sh <outdir-under-gpurun_out> [env assignments...]
set -u
R=$PWD; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for kv in "$@"; do export "$kv"; done
run() { name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python $R/tools/kmap_prof_driver.py 20000000 1 > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/$name.csv; grep "k_map" $f >> $OUT/$name.csv; else echo "no counter file for $name"; tail -5 /tmp/pmc_$name.log; fi
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
python - $OUT <<'PY'
import csv, sys, collections, os
for name in ("sq1", "sq2"):
    f = os.path.join(sys.argv[1], name + ".csv")
    if not os.path.exists(f): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(name, k, " ".join("%s=%.3g" % (c.replace("SQ_", ""), v[-1]) for c, v in sorted(d.items())))
PY
