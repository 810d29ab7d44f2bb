#!/bin/bash
# same-box A/B of one environment switch of the library over bench.py workloads:
#   tools/ab_env.sh VAR "workloads" [reps]     (A = VAR unset, B = VAR=1)
R=$(cd "$(dirname "$0")/.." && pwd)
var=$1; wls=${2:-single-1g wrapped-10g}; reps=${3:-2}
for rep in $(seq $reps); do
for wl in $wls; do
  for v in A B; do
    if [ $v = A ]; then unset $var; else export $var=1; fi
    python $R/bench.py --workload $wl --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', '$v', 'ms_per_step', d['ms_per_step'], 'index_ms', d['roofline']['avg_launch_ms'], d['roofline']['kernel'], d['roofline']['launch_ms_spread'])
"
  done
done
done
