cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p4
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -o p -- python /root/repo/tools/run_cli_scale.py 0.25 32 0 4 > /tmp/p4.log 2>&1
f=$(find /tmp/p4 -name '*kernel_stats.csv' | head -1)
cp "$f" /root/repo/gpurun_out/cli4_kernel_stats.csv
grep "finish:" /tmp/p4.log | cut -c1-200
