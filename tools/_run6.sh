mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_stream.py tests/test_index.py -x -q -m gpu 2>&1 | tail -5
FFQ_STREAM_PROF=1 timeout 600 python tools/stream_rate.py > gpurun_out/r02/stream_rate_prof2.txt 2>&1
grep -v "^\[ffq stream\] [0-9]* fills" gpurun_out/r02/stream_rate_prof2.txt | tail -12; grep "fills" gpurun_out/r02/stream_rate_prof2.txt | awk 'NR%4==0' | tail -8
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
