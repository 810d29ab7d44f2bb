#!/bin/bash
# Same-box A/B of two whole TREES (library + Python + bench.py) on single-1g: A = gpurun_ab/r04tree (git archive of the
# round-4 end commit, built in place), B = this tree.  Alternating; prints step, index kernel, remainder.
#   tools/ab_tree.sh [alternations] [extra bench flags]
R=$(cd "$(dirname "$0")/.." && pwd)
reps=${1:-5}; extra=${2:-}
for i in $(seq $reps); do
  for v in A B; do
    if [ $v = A ]; then T=$R/gpurun_ab/r04tree; else T=$R; fi
    (cd $T && python bench.py --workload single-1g --no-cpu-baseline --no-others $extra 2>/dev/null | tail -1) | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; pr=d.get('hbm_read_probe') or {}
print('tree $v rep $i  step %.4f ms (%s)  index %.4f ms  remainder %.1f us  value %.1f GB/s  probe %s' % (d['ms_per_step'], d.get('ms_per_step_spread'), r['avg_launch_ms'], (d['ms_per_step'] - r['avg_launch_ms']) * 1e3, d['value'], pr.get('value')))"
  done
done
