#!/bin/bash
# instruction mix per kernel: tools/pmc_insts.sh [bytes] [single|wrapped] [decode]
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmci; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pmci -o p -- python $R/tools/run_scan.py ${1:-1073741824} 3 ${2:-single} ${3:-} > /dev/null 2>&1
python - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("/tmp/pmci/p_counter_collection.csv")):
    n = r["Kernel_Name"].split("(")[0]
    if "ffq::k_" in n and "synth" not in n:
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in agg.items():
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print("%-40s waves %9.0f | per wave: VALU %7.0f SALU %7.0f LDS %6.0f VMEM_RD %5.0f VMEM_WR %5.0f" % (
        n[:40], w, *[sum(d[k]) / len(d[k]) / max(w, 1) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")]))
PY
