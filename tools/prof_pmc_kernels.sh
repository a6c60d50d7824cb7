#!/bin/bash
# Runs ON THE GPU BOX: instruction / wait counters per kernel of the phasing pass (one rocprofv3 --pmc pass, kernel trace only).
# usage: tools/prof_pmc_kernels.sh <tag> [kernel-name substrings...]   -> gpurun_out/<tag>_pmc_kernels.txt
set -u
R=$PWD; tag=$1; shift
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pmck
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmck -o p -- python $R/bench.py --no-cpu --no-bam --no-c2 --steps 2 --warmup 1 --phasing-passes 1 > /tmp/pmck.log 2>&1
f=$(find /tmp/pmck -name "*counter_collection.csv" | head -1)
mkdir -p $R/gpurun_out
python - "$f" "$@" > $R/gpurun_out/${tag}_pmc_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if "at::native" in k or "rocclr" in k or (want and not any(w in k for w in want)):
        continue
    k = k[:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]
print("%-50s %5s " % ("kernel (per dispatch)", "disp") + " ".join("%12s" % x[3:] for x in names))
for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"]):
    d = max(1, n[k])
    print("%-50s %5d " % (k, d) + " ".join("%12.4g" % (acc[k][x] / d) for x in names))
PY
tail -40 $R/gpurun_out/${tag}_pmc_kernels.txt | cut -c1-170
