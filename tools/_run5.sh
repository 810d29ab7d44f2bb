mkdir -p gpurun_out/r02
FFQ_STREAM_PROF=1 timeout 600 python tools/stream_rate.py > gpurun_out/r02/stream_rate_prof.txt 2>&1
grep -v "^\[ffq stream\] [0-9]* fills" gpurun_out/r02/stream_rate_prof.txt | tail -12; grep "fills" gpurun_out/r02/stream_rate_prof.txt | awk 'NR%4==0' | tail -8
for i in 1 2 3; do
  for v in nopipe pipe; do
    if [ $v = nopipe ]; then export FFQ_NO_PIPE=1; else unset FFQ_NO_PIPE; fi
    python bench.py --workload single-1g --no-cpu-baseline --no-others 2>gpurun_out/r02/err_$v.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['path_roofline']
print('$v', 'value %.1f step %.4f ms spread %s index %.4f path frac %.4f' % (d['value'], d['ms_per_step'], d['ms_per_step_spread'], p['ms_index'], p['frac']))"
  done
done
tail -3 gpurun_out/r02/err_pipe.txt
