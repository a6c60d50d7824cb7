"""Prints per-pass times of our kernels from a rocprofv3 kernel_stats.csv of tools/prof_phasing.sh (12 phasing passes per run)."""
import csv
import sys

f = sys.argv[1]; passes = int(sys.argv[2]) if len(sys.argv) > 2 else 12; floor = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    if 'anonymous namespace)::k_' not in r['Name'] or 'at::native' in r['Name'] or 'k_ub_' in r['Name']:
        continue
    n = r['Name'].split('(anonymous namespace)::')[1][:50]
    c = int(r['Calls'])
    if c % passes == 0:
        per = int(r['TotalDurationNs']) / passes / 1e3
        tot += per
        if per > floor:
            print("%-52s %4d/pass %9.1f us" % (n, c // passes, per))
print("sum of our kernels per pass: %.1f us" % tot)
