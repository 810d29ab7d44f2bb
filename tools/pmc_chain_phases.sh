#!/bin/bash
export FFQ_USE_PROBE_BUILD=1   # the ablation switches exist only in libffq_probe.so
# VALU / SALU / LDS instructions per wave of k_chain_wave cut short after each phase (FFQ_ABLATE 1..4, 0 = whole)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for a in 1 2 3 4 0; do
  rm -rf /tmp/pmcc_$a
  FFQ_ABLATE=$a rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcc_$a -o p -- python $R/tools/run_scan.py ${1:-1073741824} 3 ${2:-wrapped} > /dev/null 2>&1
  python - $a <<'PY'
import csv, collections, sys
a = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("/tmp/pmcc_%s/p_counter_collection.csv" % a)):
    n = r["Kernel_Name"].split("(")[0]
    if "k_chain_wave" in n:
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in agg.items():
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print("ablate %s %-30s waves %9.0f | per wave: VALU %7.0f SALU %7.0f LDS %6.0f VMEM_RD %5.0f VMEM_WR %5.0f" % (
        a, n[-30:], w, *[sum(d[k]) / len(d[k]) / max(w, 1) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")]))
PY
done
