#!/bin/bash
# Runs ON THE GPU BOX: PMC passes for k_map (each counter group in its own run, kernel-trace only).
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out>
set -u
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
run() { # name counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python $R/tools/kmap_prof_driver.py 50000000 2 > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/$name.csv; grep "k_map" $f >> $OUT/$name.csv; else echo "no counter file for $name"; tail -5 /tmp/pmc_$name.log; fi
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
ls -la $OUT
