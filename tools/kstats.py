#!/usr/bin/env python3
"""Print the K_map / K_tally family rows of a rocprofv3 kernel_stats.csv (GPU-box helper)."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_map", "k_tile", "k_chunk", "k_compact", "k_line", "k_keys", "k_rank", "k_pair", "k_distinct", "k_as_hist", "k_uf",
                            "k_edge", "k_unique", "k_gen", "k_scan", "k_inflate", "k_pack", "k_hop", "k_intern", "rocprim")):
        print("%-70s calls %5s avg %10.1f us" % (n.replace("(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
