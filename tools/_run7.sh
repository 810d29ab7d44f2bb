mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_index.py -x -q -m gpu 2>&1 | tail -5
FFQ_STREAM_PROF=1 timeout 600 python tools/stream_rate.py 2>&1 | grep -v "^\[ffq stream\] [0-9]* fills" | tail -9
FFQ_STREAM_PROF=1 timeout 600 python tools/stream_rate.py 2>&1 | grep "fills" | awk 'NR%4==0' | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config_size" 2>&1 | tail -5
