#!/bin/bash
# Runs ON THE GPU BOX: instruction / wait counters of K_inflate over a whole-genome BAM (/tmp/cli_scale.bam, written by tools/run_cli_scale.py first).
# usage: tools/prof_pmc_inflate.sh <outdir-under-gpurun_out>
set -u
R=$PWD; OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
[ -f /tmp/cli_scale.bam ] || (cd $R && timeout 600 python tools/run_cli_scale.py > /tmp/gen.log 2>&1)
run() { name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python $R/tools/inflate_check.py /tmp/cli_scale.bam > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/$name.csv; grep "k_inflate" $f >> $OUT/$name.csv; else echo "no counter file for $name"; tail -5 /tmp/pmc_$name.log; fi
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
run mem TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
python - $OUT <<'PY'
import csv, sys, collections, os
for name in ("sq1", "sq2", "mem"):
    f = os.path.join(sys.argv[1], name + ".csv")
    if not os.path.exists(f): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(name, k, " ".join("%s=%.4g" % (c.replace("SQ_", ""), v[-1]) for c, v in sorted(d.items())))
PY
