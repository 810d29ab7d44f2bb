mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_index.py -x -q -m gpu --durations=5 2>&1 | tail -12
timeout 900 python -m pytest tests/test_stream.py -x -q -m gpu 2>&1 | tail -3
FFQ_STREAM_PROF=1 timeout 600 python tools/stream_rate.py > gpurun_out/r02/stream_rate_prof3.txt 2>&1
grep -v "^\[ffq stream\] [0-9]* fills" gpurun_out/r02/stream_rate_prof3.txt | tail -9;  grep "fills" gpurun_out/r02/stream_rate_prof3.txt | awk 'NR%4==0' | tail -8
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ranked or golden_files or edge_corpus or fuzz_corpus or kilobase or too_small" 2>&1 | tail -8
timeout 900 python tools/shape_sweep_wrapped.py 2>&1 | tail -9
