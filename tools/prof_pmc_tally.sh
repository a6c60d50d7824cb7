#!/bin/bash
# Runs ON THE GPU BOX: HBM traffic of the phasing pass of the bench workload (configs[2]), split into the K_tally family (phz_tally.hip) and the
# device row stage (phz_rowsdev.hip): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (kernel trace only) of a bench run with
# phasing passes.  Dispatches are attributed by their place in the stream: k_as_hist .. k_edge_final = tally, k_pair_keys .. the last row /
# label / vcf kernel = rows (the shared scan / sort kernels of phz_sort.h go to the family whose section they run in).  Writes
# gpurun_out/<tag>/pmc_ktally_c3/ with a meta.json naming the kernel sources; bench.py reports the numbers only while those hashes match.
# usage: tools/prof_pmc_tally.sh <tag>
set -u
R=$PWD; OUT=$R/gpurun_out/$1/pmc_ktally_c3; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
run() { name=$1; shift
  rm -rf /tmp/pmct_$name
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmct_$name -o p -- python $R/bench.py --no-cpu --no-bam --no-c2 --steps 2 --warmup 1 --phasing-passes 1 > /tmp/pmct_$name.log 2>&1
  f=$(find /tmp/pmct_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" $OUT/$name.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
keep = []
fam = None
for r in rows:
    k = r["Kernel_Name"]
    if "k_as_hist" in k or "k_line" in k:
        fam = "tally"
    elif "k_pair_keys" in k:
        fam = "rows"
    elif "k_map" in k or "k_compact" in k or "k_tile_window" in k or k.startswith("at::") or "at::native" in k:
        fam = None
    if fam and not ("at::native" in k or "rocclr" in k):
        keep.append((fam, r))
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=["Family", "Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
w.writeheader()
for fam, r in keep:
    w.writerow({"Family": fam, "Dispatch_Id": r["Dispatch_Id"], "Kernel_Name": r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60], "Counter_Name": r["Counter_Name"],
                "Counter_Value": r["Counter_Value"]})
print(len(rows), "dispatch rows,", len(keep), "kept")
PY
  else echo "no counter file for $name"; tail -5 /tmp/pmct_$name.log; fi
}
run fetch FETCH_SIZE
run write WRITE_SIZE
python - $OUT $R <<'PY'
import csv, sys, os, json, hashlib
out, repo = sys.argv[1], sys.argv[2]
sha = lambda f: hashlib.sha256(open(os.path.join(repo, f), "rb").read()).hexdigest()[:16]
tot = {"tally": {}, "rows": {}}
passes = 0
for name in ("fetch", "write"):
    rows = list(csv.DictReader(open(os.path.join(out, name + ".csv"))))
    for fam in tot:
        tot[fam][name] = sum(float(r["Counter_Value"]) for r in rows if r["Family"] == fam)
    passes = max(passes, sum(1 for r in rows if r["Kernel_Name"].startswith("k_line")))
json.dump({"workload": "configs[2]", "kernel_source_sha16": sha("phaser_amd/csrc/phz_tally.hip"), "rows_source_sha16": sha("phaser_amd/csrc/phz_rowsdev.hip"), "passes": passes,
           "kib": tot, "command": "python bench.py --no-cpu --no-bam --no-c2 --steps 2 --warmup 1 --phasing-passes 1",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB summed over the family's dispatches of all passes; FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md)"},
          open(os.path.join(out, "meta.json"), "w"), indent=1)
for fam in tot:
    print(fam, "passes", passes, "fetch KiB", tot[fam]["fetch"], "write KiB", tot[fam]["write"], "-> GB per pass", (2 * tot[fam]["fetch"] + tot[fam]["write"]) * 1024 / max(1, passes) / 1e9)
PY
