#!/bin/bash
# Runs ON THE GPU BOX: HBM traffic of the K_tally family over one phasing pass of the bench workload (configs[2]): FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes (kernel trace only) of a bench run with two phasing passes.  Keeps, per counter, the
# dispatches from the first k_as_hist on (everything before is workload generation and the mapper steps) whose kernels belong to
# phz_tally.hip or are the rocPRIM sorts / scans / selects it calls.  Writes gpurun_out/<tag>/pmc_ktally_c3/ with a meta.json naming
# the kernel source; bench.py reports the numbers only while that hash matches.   usage: tools/prof_pmc_tally.sh <tag>
set -u
R=$PWD; OUT=$R/gpurun_out/$1/pmc_ktally_c3; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift
  rm -rf /tmp/pmct_$name
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmct_$name -o p -- python $R/bench.py --no-cpu --no-bam --no-c2 --steps 2 --warmup 1 --phasing-passes 2 > /tmp/pmct_$name.log 2>&1
  f=$(find /tmp/pmct_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" $OUT/$name.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
start = next((i for i, r in enumerate(rows) if "k_as_hist" in r["Kernel_Name"]), None)
keep = []
if start is not None:
    fam = ("k_as_hist", "k_line", "k_qsort", "k_keys", "k_rank", "k_pair", "k_distinct", "k_uf", "k_edge", "k_unique", "k_noise", "k_scan", "k_tab", "k_item", "k_read", "rocprim")
    keep = [r for r in rows[start:] if any(k in r["Kernel_Name"] for k in fam) and "k_map" not in r["Kernel_Name"] and "k_compact" not in r["Kernel_Name"]]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
w.writeheader()
for r in keep: w.writerow({k: r[k] for k in w.fieldnames})
print(len(rows), "dispatch rows,", len(keep), "kept")
PY
  else echo "no counter file for $name"; tail -5 /tmp/pmct_$name.log; fi
}
run fetch FETCH_SIZE
run write WRITE_SIZE
python - $OUT $R <<'PY'
import csv, sys, os, json, hashlib
out, repo = sys.argv[1], sys.argv[2]
sha = hashlib.sha256(open(os.path.join(repo, "phaser_amd/csrc/phz_tally.hip"), "rb").read()).hexdigest()[:16]
tot = {}
passes = 0
for name in ("fetch", "write"):
    rows = list(csv.DictReader(open(os.path.join(out, name + ".csv"))))
    tot[name] = sum(float(r["Counter_Value"]) for r in rows)
    passes = max(passes, sum(1 for r in rows if "k_as_hist" in r["Kernel_Name"]))
json.dump({"workload": "configs[2]", "kernel_source_sha16": sha, "passes": passes,
           "command": "python bench.py --no-cpu --no-bam --no-c2 --steps 2 --warmup 1 --phasing-passes 2",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB summed over the K_tally family of all passes; FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md)"},
          open(os.path.join(out, "meta.json"), "w"), indent=1)
print("passes", passes, "fetch KiB", tot["fetch"], "write KiB", tot["write"], "-> GB per pass", (2 * tot["fetch"] + tot["write"]) * 1024 / max(1, passes) / 1e9)
PY
