mkdir -p gpurun_out/r05l
for rep in 1 2; do
for mode in os three; do
  if [ $mode = os ]; then export PHZ_SORT_ONE_LAUNCH=1; else unset PHZ_SORT_ONE_LAUNCH; fi
  python bench.py --no-cpu --no-c2 --no-bam --phasing-passes 11 --steps 100 > gpurun_out/r05l/ab_${mode}_$rep.out 2> gpurun_out/r05l/ab_${mode}_$rep.err
  python -c "
import json; d=json.loads([l for l in open('gpurun_out/r05l/ab_${mode}_$rep.out') if l.startswith('{')][-1]); p=d['phasing']; print('$mode $rep', 'step %.4f ms kernel %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg']), 'pass %.3f ms'%(p['seconds_per_pass']*1e3), 'gpu %.3f'%p['gpu_ms_per_pass_max_rank'], sorted(round(p['phased_variants']/x*1e3,3) for x in p['passes']))"
done; done
unset PHZ_SORT_ONE_LAUNCH
(timeout 600 python -m pytest tests/test_gpu_mapper.py -x -q -k "overflow or random_vs_oracle or dense_windows" 2>&1 | tail -4)
PHZ_TIMING=1 timeout 600 python bench.py --from-files --steps 2 --warmup 1 > gpurun_out/r05l/ff_timing.out 2> gpurun_out/r05l/ff_timing.err; grep "phz timing" gpurun_out/r05l/ff_timing.err | tail -24 | cut -c1-160
