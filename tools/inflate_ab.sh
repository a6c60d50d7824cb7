#!/bin/bash
# Runs ON THE GPU BOX: K_inflate A/B (default build + the variants under phaser_amd/variants/libphz_hot*.so) on a whole-genome BAM, verified against zlib.
mkdir -p gpurun_out/inf2
timeout 400 python -m pytest tests/test_gpu_bamdev.py -x -q > gpurun_out/inf2/pytest.log 2>&1; tail -2 gpurun_out/inf2/pytest.log
PHZ_TIMING=1 timeout 500 python tools/run_cli_scale.py > gpurun_out/inf2/cli.log 2>&1; grep -n "H2D + K_inflate\|\] total\|bam decode" gpurun_out/inf2/cli.log
for v in "" $(ls phaser_amd/variants/ 2>/dev/null | grep "^libphz_hot.*so$" | sed 's/libphz_//; s/.so//'); do
  if [ -n "$v" ]; then export PHZ_LIB_PATH=phaser_amd/variants/libphz_$v.so; fi
  echo "== ${v:-default}"; timeout 600 python tools/inflate_check.py 2>&1 | grep "cli_scale.bam\|inf_text\|verified\|differ" | sed -n '2,3p;5,8p'
done
