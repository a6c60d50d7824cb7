#!/bin/bash
# Runs ON THE GPU BOX: K_inflate A/B (default build + the variants under phaser_amd/variants/libphz_hot*.so) on a whole-genome BAM, verified against zlib
# (the ablation variants -DPHZ_INF_ABL=1|2|3 give wrong output on purpose: only their times count).
mkdir -p gpurun_out/inf2
[ -f /tmp/cli_scale.bam ] || timeout 500 python tools/run_cli_scale.py > gpurun_out/inf2/cli.log 2>&1
for v in "" $(ls phaser_amd/variants/ 2>/dev/null | grep "^libphz_hot.*so$" | sed 's/libphz_//; s/.so//'); do
  if [ -n "$v" ]; then export PHZ_LIB_PATH=phaser_amd/variants/libphz_$v.so; fi
  echo "== ${v:-default}"; timeout 600 python tools/inflate_check.py /tmp/cli_scale.bam 2>&1 | grep "cli_scale.bam\|verified\|differ" | sed -n '2,4p'
done
