#!/bin/bash
# per-kernel average times of tools/run_scan.py under the current environment
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o p -- python $R/tools/run_scan.py ${1:-1073741824} 6 ${2:-single} ${3:-} > /dev/null 2>&1
python - <<PY
import csv
tot = 0
for r in csv.DictReader(open("/tmp/kst/p_kernel_stats.csv")):
    n = r["Name"]
    if "ffq::k_" in n and "synth" not in n:
        print("   %-44s calls %3s avg_us %8.1f" % (n.split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3)); tot += float(r["AverageNs"]) / 1e3
print("   sum %.1f us" % tot)
PY
