#!/usr/bin/env python3
"""Turns the --pmc passes of tools/prof_calib.sh into calibration.json: per access pattern the bytes its lanes read / wrote, the requests it issued,
and what every collected counter reported per launch -- hence the factor FETCH_SIZE / WRITE_SIZE must be multiplied by for that pattern.
usage: tools/calib_table.py <dir with plain.jsonl and *.csv>"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
pat = {}
for line in open(os.path.join(d, "plain.jsonl")):
    if line.startswith("{"):
        r = json.loads(line)
        pat[r["kernel"]] = r
counters = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        k = k[5:] if k.startswith("void ") else k
        k = k.split("(")[0].strip()
        counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"unit_note": "FETCH_SIZE / WRITE_SIZE are KiB per launch as rocprofv3 prints them; *_bytes = x 1024; factor = bytes the pattern must move / bytes the counter reports",
       "patterns": {}}
for k, p in pat.items():
    c = {n: sum(v) / len(v) for n, v in counters.get(k, {}).items()}
    e = dict(p); e["counters_per_launch"] = c
    is_gather = p["unit"] > 0
    is_write = "write" in p["pattern"]
    if "FETCH_SIZE" in c and not is_write:
        fb = c["FETCH_SIZE"] * 1024
        e["fetch_bytes_reported"] = fb
        if is_gather:
            e["fetch_bytes_reported_per_request"] = fb / p["requests"]
            # the array's first-32-bytes-of-every-unit are all touched: at a 32-byte DRAM granule the memory side must deliver 32 B per request
            e["factor_if_32B_granule"] = 32.0 * p["requests"] / fb if fb else None
        else:
            e["factor"] = p["array_bytes"] / fb if fb else None
    if "WRITE_SIZE" in c and is_write:
        wb = c["WRITE_SIZE"] * 1024
        e["write_bytes_reported"] = wb
        e["factor"] = p["array_bytes"] / wb if wb else None
    out["patterns"][p["pattern"]] = e
json.dump(out, open(os.path.join(d, "calibration.json"), "w"), indent=1)
for n, e in out["patterns"].items():
    print("%-16s %-22s lanes %.3e B  req %.3e  %.3f ms  %s" % (n, e["kernel"], e["lane_bytes"], e["requests"], e["seconds"] * 1e3,
          "  ".join("%s=%.4g" % (a, b) for a, b in sorted(e.items()) if a.startswith(("fetch_", "write_", "factor")) and b is not None)))
    if e["counters_per_launch"]:
        print("                 " + "  ".join("%s=%.5g" % kv for kv in sorted(e["counters_per_launch"].items())))
