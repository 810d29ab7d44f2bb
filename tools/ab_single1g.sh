#!/bin/bash
# Same-box A/B of single-1g: step, index kernel and the REMAINDER (step - index kernel) of library A
# (gpurun_ab/libffq_hip_A.so, built by tools/ab_build.sh <ref>) against B (the in-tree build), alternating.
#   tools/ab_single1g.sh [alternations] [extra bench flags, e.g. "--time-every 1"]
R=$(cd "$(dirname "$0")/.." && pwd)
reps=${1:-5}; extra=${2:-}
for i in $(seq $reps); do
  for v in A B; do
    if [ $v = A ]; then export FFQ_HIP_LIB=$R/gpurun_ab/libffq_hip_A.so; else unset FFQ_HIP_LIB; fi
    python $R/bench.py --workload single-1g --no-cpu-baseline --no-others $extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; pr=d.get('hbm_read_probe') or {}
print('$v rep $i  step %.4f ms (%s)  index %.4f ms (%s)  remainder %.1f us  value %.1f GB/s  probe %s' % (d['ms_per_step'], d['ms_per_step_spread'], r['avg_launch_ms'], r['launch_ms_spread'], (d['ms_per_step'] - r['avg_launch_ms']) * 1e3, d['value'], pr.get('value')))"
  done
done
