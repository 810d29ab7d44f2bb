mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_sharded.py tests/test_gpu_parity.py::test_table_cut tests/test_abi.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r02/t_sharded.txt
cat gpurun_out/r02/t_sharded.txt
