#!/bin/bash
export FFQ_USE_PROBE_BUILD=1   # the ablation switches exist only in libffq_probe.so
# per-kernel time of the scan pipeline under each ablation level of k_chain_wave
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 3 4; do
  FFQ_ABLATE=$a rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$a -o p -- python $R/tools/run_scan.py 1073741824 6 > /dev/null 2>&1
  echo "ablate=$a"
  python - <<PY
import csv
for r in csv.DictReader(open("/tmp/abl_$a/p_kernel_stats.csv")):
    n = r["Name"]
    if "ffq::k_" in n and "synth" not in n:
        print("   %-44s calls %3s avg_us %8.1f" % (n.split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
