#!/usr/bin/env python3
"""Print every kernel of a rocprofv3 kernel_stats.csv sorted by total time (GPU-box helper): name, calls, average, total."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-90s calls %6s avg %10.1f us  total %9.2f ms  %5.1f%%" % (n[:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6,
                                                                      100 * float(r["TotalDurationNs"]) / tot))
