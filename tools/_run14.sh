mkdir -p gpurun_out/r02
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for wl in single-1g decode-10g wrapped-10g; do bash tools/profile_round.sh r02_a $wl > /dev/null 2>&1; done
ls gpurun_out/profiles/
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/profiles/r02_a_*/bench.json")):
    d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["path_roofline"]["frac"])
PY
head -12 gpurun_out/profiles/r02_a_single-1g/rocprofv3_kernel_stats.csv
