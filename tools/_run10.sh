timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_wrapped" --tb=short 2>&1 | tail -30
