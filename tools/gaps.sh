#!/bin/bash
# kernel timeline of the last steps of a bench run: durations and gaps between kernels
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp; rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o p -- python $R/bench.py --workload ${1:-single-1g} --steps 12 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv
rows = [r for r in csv.DictReader(open("/tmp/gp/p_kernel_trace.csv")) if "ffq::k_" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-32:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-34s start %8.1f dur %7.1f us  gap_from_prev_end %6.1f  stream/queue %s" % (r["Kernel_Name"].split("(")[0][:34], (s - t0) / 1e3, (e - s) / 1e3, ((s - prev_end) / 1e3) if prev_end else 0.0, r.get("Queue_Id", "?")))
    prev_end = e
PY
