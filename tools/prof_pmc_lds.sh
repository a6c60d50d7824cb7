#!/bin/bash
# Runs ON THE GPU BOX: LDS counters per kernel of one phasing pass (rocprofv3 --pmc, kernel trace only) -> gpurun_out/<tag>_pmc_lds.txt
set -u
R=$PWD; tag=$1
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pmcl
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmcl -o p -- python $R/bench.py --no-cpu --no-bam --no-c2 --steps 2 --warmup 1 --phasing-passes 1 > /tmp/pmcl.log 2>&1
f=$(find /tmp/pmcl -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then tail -5 /tmp/pmcl.log; exit 1; fi
python - "$f" > $R/gpurun_out/${tag}_pmc_lds.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if "at::native" in k or "rocclr" in k or "rocprim" in k: continue
    k = k[:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
    if r["Counter_Name"] == names[0]: n[k] += 1
print("%-46s %5s " % ("kernel (per dispatch)", "disp") + " ".join("%18s" % x[3:] for x in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_LDS_BANK_CONFLICT", 0)):
    d = max(1, n[k])
    print("%-46s %5d " % (k, d) + " ".join("%18.4g" % (acc[k][x] / d) for x in names))
PY
head -24 $R/gpurun_out/${tag}_pmc_lds.txt | cut -c1-200
