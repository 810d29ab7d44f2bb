#!/bin/bash
# interleaved bench runs of library A (gpurun_ab/libffq_hip_A.so) and B (in-tree) on this box
R=$(cd "$(dirname "$0")/.." && pwd)
wl=${1:-single-1g}; reps=${2:-3}
for i in $(seq $reps); do
  for v in A B; do
    if [ $v = A ]; then export FFQ_HIP_LIB=$R/gpurun_ab/libffq_hip_A.so; else unset FFQ_HIP_LIB; fi
    python $R/bench.py --workload $wl --no-cpu-baseline --no-others 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d.get('path_roofline',{})
print('$v', '$wl', 'value %.1f  step %.4f ms  index %.4f  chain %.4f  decode %.4f' % (d['value'], d['ms_per_step'], p.get('ms_index',0), p.get('ms_chain',0), p.get('ms_decode',0)))"
  done
done
