#!/bin/bash
# Runs ON THE GPU BOX: A/B of environment switches on the default bench workload (mapper step + phasing pass, no CPU legs).
# usage: tools/ab_env.sh <tag> "VAR=val ..." ["VAR=val ..." ...]     -> gpurun_out/<tag>/ab.txt (one line per setting)
R=$PWD; TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/ab.txt
for setting in "" "$@"; do
  env $setting timeout 600 python bench.py --no-cpu --no-c2 --no-bam --steps 100 > /tmp/ab.out 2> /tmp/ab.err
  python - "$setting" <<'P' >> $OUT/ab.txt
import json, sys
line = [l for l in open('/tmp/ab.out') if l.startswith('{')]
if not line:
    print("%-40s FAILED" % sys.argv[1]); sys.exit(0)
d = json.loads(line[-1]); p = d.get('phasing', {})
print("%-40s step %.4f ms  k_map %.4f ms  phasing pass %.3f ms  tally kernels %.3f ms  rows gpu %.3f ms" % (
    sys.argv[1] or "(default)", d['ms_per_step'], d['roofline']['kernel_ms_avg'], 1e3 * p.get('seconds_per_pass', 0),
    p.get('roofline', {}).get('kernel_ms_sum_over_ranks', 0), p.get('gpu_ms_per_pass_max_rank', 0)))
P
done
cat $OUT/ab.txt
