#!/usr/bin/env python3
"""Where a phasing pass's wall time goes between kernels: reads a rocprofv3 --kernel-trace CSV of tools/pass_sweep.py, cuts it into passes at every
k_as_hist dispatch and prints, for the LAST pass: kernels, busy time, idle time, the largest gaps (host waits) with the kernels around them, and the
number of gaps by size class.      usage: tools/pass_gaps.py <kernel_trace.csv> [pass index from the end, default 1]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")) for r in rows]
ks.sort()
starts = [i for i, k in enumerate(ks) if k[2].startswith("k_as_hist")]
if len(starts) < back + 1:
    sys.exit("not enough passes in the trace")
lo = starts[-back - 1] if back + 1 <= len(starts) else 0
lo = starts[-back]
hi = len(ks)
if back > 1:
    hi = starts[-back + 1]
# cut at the last of our kernels (the next pass's set-up or torch kernels may follow)
p = [k for k in ks[lo:hi] if "at::native" not in k[2] and "rocclr" not in k[2].lower()]
span = p[-1][1] - p[0][0]
busy = sum(e - s for s, e, _ in p)
gaps = [(p[i + 1][0] - p[i][1], p[i][2][:40], p[i + 1][2][:40]) for i in range(len(p) - 1)]
print("pass: %d kernels, first start -> last end %.1f us, kernels busy %.1f us, between kernels %.1f us" % (len(p), span / 1e3, busy / 1e3, (span - busy) / 1e3))
cls = [(0, 3), (3, 6), (6, 10), (10, 20), (20, 50), (50, 1e9)]
for a, b in cls:
    g = [x[0] for x in gaps if a * 1e3 <= x[0] < b * 1e3]
    print("  gaps %4.0f-%-6.0f us: %4d, %8.1f us together" % (a, min(b, 9999), len(g), sum(g) / 1e3))
print("largest gaps:")
for g in sorted(gaps, reverse=True)[:14]:
    print("  %7.1f us after %-40s before %s" % (g[0] / 1e3, g[1], g[2]))
short = sorted(p, key=lambda k: k[1] - k[0])
print("kernels shorter than 5 us: %d (%.1f us together); 5-20 us: %d" % (sum(1 for k in p if k[1] - k[0] < 5000), sum(k[1] - k[0] for k in p if k[1] - k[0] < 5000) / 1e3,
                                                                   sum(1 for k in p if 5000 <= k[1] - k[0] < 20000)))
