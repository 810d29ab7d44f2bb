#!/bin/bash
# same-box A/B of the index kernel: one tile per workgroup (G=0) against persistent workgroups (FFQ_SCAN_PERSIST=G),
# with the experiment switches of k_scan_lines_p (FFQ_SCAN_PEXP: 31 no entry stores, 32 no second barrier, 33 both;
# results are wrong with those, only the index kernel's time means anything)
R=$(cd "$(dirname "$0")/.." && pwd)
wls=${@:-single-1g wrapped-10g}
for rep in 1 2; do
for wl in $wls; do
  for cfg in 0:0 2048:0 2048:31 2048:32 2048:33 1024:33; do
    g=${cfg%%:*}; x=${cfg##*:}
    if [ $g = 0 ]; then unset FFQ_SCAN_PERSIST; else export FFQ_SCAN_PERSIST=$g; fi
    export FFQ_SCAN_PEXP=$x
    python $R/bench.py --workload $wl --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', 'G=$g exp=$x', 'ms_per_step', d['ms_per_step'], 'index_ms', d['roofline']['avg_launch_ms'], d['roofline']['launch_ms_spread'])
"
  done
done
done
