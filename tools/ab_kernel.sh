#!/bin/bash
# same-box A/B of ONE kernel's rocprof average: tools/ab_kernel.sh <workload> <kernel-substring> [reps]
# (A = gpurun_ab/libffq_hip_A.so built by tools/ab_build.sh <ref>, B = the in-tree build)
R=$(cd "$(dirname "$0")/.." && pwd)
wl=${1:-wrapped-10g}; kn=${2:-k_chain_wave}; reps=${3:-2}
cd /tmp && export TMPDIR=/tmp
for i in $(seq $reps); do
  for v in A B; do
    if [ $v = A ]; then export FFQ_HIP_LIB=$R/gpurun_ab/libffq_hip_A.so; else unset FFQ_HIP_LIB; fi
    rm -rf /tmp/abk
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o p -- python $R/bench.py --workload $wl --no-cpu-baseline --no-others > /dev/null 2>&1
    python - "$v" "$kn" <<'PY'
import csv, sys
for r in csv.DictReader(open("/tmp/abk/p_kernel_stats.csv")):
    if sys.argv[2] in r["Name"] or "k_scan_lines" in r["Name"]:
        print(sys.argv[1], r["Name"].split("(")[0][-40:], "calls", r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
  done
done
