set -x
mkdir -p gpurun_out/r02
python tools/mall_probe.py > gpurun_out/r02/mall_probe.txt 2>&1
for wl in single-1g decode-10g wrapped-10g; do python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02/base_$wl.json; done
tail -30 gpurun_out/r02/mall_probe.txt
python - <<'PY'
import json
for wl in ("single-1g","decode-10g","wrapped-10g"):
    d=json.load(open("gpurun_out/r02/base_%s.json"%wl)); print(wl, d["value"], d["ms_per_step"], d["path_roofline"])
PY
