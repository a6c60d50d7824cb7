#!/bin/bash
# Runs ON THE GPU BOX: HBM traffic of k_map on the bench workload (configs[2], whole genome), FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (kernel-trace only), plus the instruction / wait counters.  Writes gpurun_out/<tag>/pmc_kmap_c3/ with a
# meta.json that names the kernel source the numbers belong to (bench.py reports them only while that hash matches).
set -u
R=$PWD; OUT=$R/gpurun_out/$1/pmc_kmap_c3; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- python $R/bench.py --no-cpu --no-phasing --no-c2 --no-bam --steps 3 --warmup 1 > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/$name.csv; grep "k_map\|k_compact\|k_tile_window" $f >> $OUT/$name.csv; else echo "no counter file for $name"; tail -5 /tmp/pmc_$name.log; fi
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
# the raw memory-side request counters FETCH_SIZE / WRITE_SIZE derive from, where this rocprofv3 lists them: request counts by size let
# bench.py price K_map's gathers and streams apart (tools/prof_calib.sh calibrates both on known byte counts)
rocprofv3 -L 2>/dev/null > /tmp/ea_names.txt
have() { grep -q "Counter_Name.*:.$1\$" /tmp/ea_names.txt; }
s1=""; for c in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum; do have $c && s1="$s1 $c"; done
[ -n "$s1" ] && run ea_rd $s1
s2=""; for c in TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUBBLE_sum; do have $c && s2="$s2 $c"; done
[ -n "$s2" ] && run ea_wr $s2
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
python - $OUT $R <<'PY'
import csv, sys, os, json, hashlib, collections
out, repo = sys.argv[1], sys.argv[2]
sha = hashlib.sha256(open(os.path.join(repo, "phaser_amd/csrc/phz_map.hip"), "rb").read()).hexdigest()[:16]
json.dump({"workload": "configs[2]", "kernel_source_sha16": sha, "command": "python bench.py --no-cpu --no-phasing --no-c2 --no-bam --steps 3 --warmup 1",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch; FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md)"}, open(os.path.join(out, "meta.json"), "w"), indent=1)
for name in ("fetch", "write", "ea_rd", "ea_wr", "sq1", "sq2"):
    f = os.path.join(out, name + ".csv")
    if not os.path.exists(f): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(name, k, " ".join("%s=%.4g(n=%d)" % (c.replace("SQ_", ""), sum(v) / len(v), len(v)) for c, v in sorted(d.items())))
PY
