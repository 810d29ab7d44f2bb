#!/bin/bash
# Kernel stats + HBM counters of the single pass on LONG four-line reads: tools/profile_long4.sh <tag> [L] [bytes]
#   -> gpurun_out/profiles/<tag>_long4-<L>/ {sweep.txt, rocprofv3_kernel_stats.csv, pmc_fetch_write.json}
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; L=${2:-3000}; bytes=${3:-4294967296}
out=$R/gpurun_out/profiles/${tag}_long4-$L; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $R/tools/shape_sweep_decode.py $bytes $L > $out/sweep.txt 2>&1
rm -rf /tmp/pl_ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_ks -o p -- python $R/tools/shape_sweep_decode.py $bytes $L > /dev/null 2>&1
cp /tmp/pl_ks/p_kernel_stats.csv $out/rocprofv3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pl_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pl_$c -o p -- python $R/tools/shape_sweep_decode.py $bytes $L > /dev/null 2>&1
done
python - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        rows = csv.DictReader(open("/tmp/pl_%s/p_counter_collection.csv" % c))
    except OSError:
        continue
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        if "ffq::k_" not in name:
            continue
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for name, d in agg.items():
    res[name] = {"%s_KiB_avg_per_launch" % k: sum(v) / len(v) for k, v in d.items()}
    res[name]["launches"] = max(len(v) for v in d.values())
json.dump(res, open(out + "/pmc_fetch_write.json", "w"), indent=1)
PY
git -C $R rev-parse HEAD > $out/COMMIT 2>/dev/null || cp $R/.commit $out/COMMIT 2>/dev/null
echo "profile written to $out"
