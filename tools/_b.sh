timeout 600 python -m pytest tests/test_gpu_mapper.py tests/test_gpu_pipeline.py -m gpu -x -q -k "indel or general" 2>&1 | tail -1 >&2; timeout 300 python tools/fuzz_product_mapper.py 600 3 2>&1 | tail -1
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/pb
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python $R/tools/kmap_general_time.py > /tmp/pb.log 2>&1
grep K_map_general /tmp/pb.log
f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1); python $R/tools/kstats.py $f | grep -v rocprim | grep "gen\|scan\|k_map"
