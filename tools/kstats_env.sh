#!/bin/bash
# per-kernel rocprof averages of one bench.py workload under an environment setting: tools/kstats_env.sh <workload> [VAR=val ...]
R=$(cd "$(dirname "$0")/.." && pwd)
wl=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kse
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kse -o p -- python $R/bench.py --workload $wl --no-cpu-baseline --no-others > /dev/null 2>&1
python - "$*" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open("/tmp/kse/p_kernel_stats.csv")) if "ffq::" in r["Name"] and "synth" not in r["Name"] and "probe" not in r["Name"]]
print("[%s]" % sys.argv[1], "  ".join("%s %sx%.1f" % (r["Name"].split("(")[0].replace("void ", "").replace("ffq::", "")[:28], r["Calls"], float(r["AverageNs"]) / 1e3) for r in rows))
PY
