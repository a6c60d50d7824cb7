#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace (per dispatch: grid, start, end) of one phasing pass; writes gpurun_out/<tag>_trace_<kernel>.txt with the
# dispatches of the kernels whose name contains $2, in stream order.  usage: tools/prof_trace.sh <tag> <kernel substring>
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; pat=$2
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ptr
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptr -o p -- python $R/bench.py --no-cpu --no-bam --no-c2 --steps 2 --warmup 1 --phasing-passes 1 > /tmp/ptr.log 2>&1
f=$(find /tmp/ptr -name '*kernel_trace.csv' | head -1)
python - "$f" "$pat" > $R/gpurun_out/${tag}_trace_${pat}.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    print(r["Kernel_Name"][:60].replace("(anonymous namespace)::", ""), "grid", r.get("Grid_Size_X", r.get("Grid_Size", "?")), "wg", r.get("Workgroup_Size_X", "?"), "us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
tail -45 $R/gpurun_out/${tag}_trace_${pat}.txt
