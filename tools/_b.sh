timeout 600 python -m pytest tests/test_gpu_mapper.py -m gpu -x -q 2>&1 | tail -1
for i in 1 2; do timeout 300 python bench.py --no-cpu --no-phasing --no-bam 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['secondary']['ms_per_step'], d['secondary']['kernel_ms_avg'])"; done
