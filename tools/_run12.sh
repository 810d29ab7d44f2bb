mkdir -p gpurun_out/r02
timeout 600 python tools/lookback_probe.py 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err; tail -3 gpurun_out/r02/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02/bench_default.json"))
print("single-1g", d["value"], d["ms_per_step"], d["ms_per_step_spread"], d["roofline"]["frac"], d["path_roofline"]["frac"])
print("host_inclusive", d["host_inclusive"]["value"], d["host_inclusive"]["stream_fd"])
for k,v in d["other_workloads"].items(): print(k, v["value"], v["ms_per_step"], v["ms_per_step_spread"], v["roofline"]["kernel"], v["roofline"]["frac"], v["path_roofline"]["frac"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
PY
