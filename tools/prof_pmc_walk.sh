#!/bin/bash
# Runs ON THE GPU BOX: K_map time and per-wave instruction counters under PHZ_MAP_DBG switches (usage: prof_pmc_walk.sh "0,8,64,128,32")
set -u
R=$PWD; export TMPDIR=/tmp; cd /tmp
L=${1:-0,8,64,128,32}
python $R/tools/kmap_phases.py $L 3 10
rm -rf /tmp/pw
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pw -o p -- python $R/tools/kmap_phases.py $L 0 1 > /tmp/pw.log 2>&1
f=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
python - "$f" "$L" <<'PY'
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "k_map" in r["Kernel_Name"]:
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
dbgs = sys.argv[2].split(",")
for d, i in zip(dbgs, ids[-len(dbgs):]):
    c = rows[i]; w = c["SQ_WAVES"]
    print("dbg=%-4s" % d, " ".join("%s/wave=%.0f" % (k.replace("SQ_", ""), v / w) for k, v in sorted(c.items()) if k != "SQ_WAVES"), "waves=%d" % w)
PY
