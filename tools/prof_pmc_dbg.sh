#!/bin/bash
# Runs ON THE GPU BOX: instruction counters of k_map under the PHZ_MAP_DBG ablation switches (which phases cost what).
set -u
R=$PWD; export TMPDIR=/tmp; cd /tmp
for dbg in 0 8 16 17 2; do
  rm -rf /tmp/pd
  PHZ_MAP_DBG=$dbg timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d /tmp/pd -o p -- python $R/tools/kmap_prof_driver.py 50000000 1 > /tmp/pd.log 2>&1
  f=$(find /tmp/pd -name "*counter_collection.csv" | head -1)
  python - "$f" "$dbg" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_map" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = agg["SQ_WAVES"][-1]
print("dbg=%s" % sys.argv[2], " ".join("%s/wave=%.0f" % (k.replace("SQ_INSTS_", ""), v[-1] / w) for k, v in sorted(agg.items()) if k != "SQ_WAVES"))
PY
done
