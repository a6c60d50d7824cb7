#!/usr/bin/env python3
"""Static instruction counts per loop of one kernel in a hipcc -save-temps .s file (which loops carry the VALU / SALU weight).
usage: isa_loops.py <file.s> <mangled-kernel-name-substring>"""
import re, sys
text = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and sys.argv[2] in l and re.match(r"^_Z\w+:", l))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
lines = text[start:end + 1]
lab = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: lab[m.group(1)] = i
cnt = lambda body, pat: sum(1 for x in body if re.match(pat, x))
print("kernel lines", len(lines), "valu", cnt(lines, r"^\s*v_"), "salu", cnt(lines, r"^\s*s_"), "ds", cnt(lines, r"^\s*ds_"))
seen = {}
for i, l in enumerate(lines):
    m = re.match(r"^\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        seen[m.group(1)] = max(seen.get(m.group(1), 0), i)
for n, b in sorted(seen.items(), key=lambda x: lab[x[0]]):
    a = lab[n]; body = lines[a:b + 1]
    print("%-10s %5d..%5d  valu %4d salu %4d ds %3d vmem %3d" % (n, a, b, cnt(body, r"^\s*v_"), cnt(body, r"^\s*s_"), cnt(body, r"^\s*ds_"), cnt(body, r"^\s*(global|buffer|flat)_")))
