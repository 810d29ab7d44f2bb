FFQ_STREAM_PROF=1 timeout 600 python tools/stream_rate.py > gpurun_out/r02/stream_rate_prof.txt 2>&1
cat gpurun_out/r02/stream_rate_prof.txt
