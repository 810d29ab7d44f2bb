mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_stream.py tests/test_index.py tests/test_fasta_open.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r02/t_stream.txt
cat gpurun_out/r02/t_stream.txt
timeout 600 python tools/stream_rate.py > gpurun_out/r02/stream_rate.txt 2>&1
cat gpurun_out/r02/stream_rate.txt
