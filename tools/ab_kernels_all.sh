#!/bin/bash
# same-box A/B of EVERY kernel's rocprof average over one bench.py workload: tools/ab_kernels_all.sh <workload> [reps]
# (A = gpurun_ab/libffq_hip_A.so built by tools/ab_build.sh <ref>, B = the in-tree build)
R=$(cd "$(dirname "$0")/.." && pwd)
wl=${1:-single-1g}; reps=${2:-2}
cd /tmp && export TMPDIR=/tmp
for i in $(seq $reps); do
  for v in A B; do
    if [ $v = A ]; then export FFQ_HIP_LIB=$R/gpurun_ab/libffq_hip_A.so; else unset FFQ_HIP_LIB; fi
    rm -rf /tmp/abk
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o p -- python $R/bench.py --workload $wl --no-cpu-baseline --no-others > /dev/null 2>&1
    python - "$v" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open("/tmp/abk/p_kernel_stats.csv")) if "ffq::" in r["Name"] and "synth" not in r["Name"] and "probe" not in r["Name"]]
print(sys.argv[1], "  ".join("%s %.1f" % (r["Name"].split("(")[0].replace("void ", "").replace("ffq::", "")[:24], float(r["AverageNs"]) / 1e3) for r in rows))
PY
  done
done
