mkdir -p gpurun_out/r02
timeout 600 python tools/lookback_probe.py 2>&1 | tail -5
timeout 300 python tools/prof_chain.py $((1<<30)) wrapped 2>&1 | tail -8
timeout 900 python tools/shape_sweep_wrapped.py $((512<<20)) 1000,3000,5000,20000,60000 2>&1 | tail -6
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
