#!/bin/bash
# Profile evidence of one workload: tools/profile_round.sh <tag> <workload>
#   -> gpurun_out/profiles/<tag>_<workload>/ {bench.json, rocprofv3_kernel_stats.csv,
#      bench_under_rocprof.json, pmc_fetch_write.json}; copy what is to be judged into profiles/.
# Counters are collected in their own passes (--kernel-trace + --pmc only), one counter per pass.
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; wl=${2:-single-1g}; extra=${3:-}        # extra: e.g. --single-pass (the directory then ends in -singlepass)
sfx=""; [ "$extra" = "--single-pass" ] && sfx="-singlepass"
out=$R/gpurun_out/profiles/${tag}_${wl}${sfx}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --workload $wl --no-others --no-cpu-baseline $extra > $out/bench.json 2> $out/bench.err
rm -rf /tmp/pr_ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_ks -o p -- \
    python $R/bench.py --workload $wl --no-cpu-baseline --no-others $extra > $out/bench_under_rocprof.json 2> /dev/null
cp /tmp/pr_ks/p_kernel_stats.csv $out/rocprofv3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pr_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pr_$c -o p -- \
        python $R/bench.py --workload $wl --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-others $extra > /dev/null 2>&1
done
python - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        rows = csv.DictReader(open("/tmp/pr_%s/p_counter_collection.csv" % c))
    except OSError:
        continue
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        if "ffq::k_" not in name or "synth" in name:
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
