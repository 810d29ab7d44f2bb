#!/bin/bash
# per-kernel rocprof averages of a python script: tools/kstats_py.sh script.py [args]
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksp -o p -- python "$R/$1" "${@:2}" > /dev/null 2>&1
python - <<'PY'
import csv
rows = [r for r in csv.DictReader(open("/tmp/ksp/p_kernel_stats.csv")) if "ffq::" in r["Name"] and "synth" not in r["Name"]]
for r in rows: print("%-44s calls %4s avg %8.1f us" % (r["Name"].split("(")[0].replace("void ", "").replace("ffq::", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
