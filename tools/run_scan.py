"""Small driver for profiling: N scans of a synthetic buffer resident in HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 30)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = hip.Context(0)
n = nbytes // 322
buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device='cuda')
ctx.synth_single(buf.data_ptr(), 0, n, 42)
table = torch.empty((n + 64, 6), dtype=torch.int64, device='cuda')
ctx.reserve(n * 322)
for i in range(reps):
    rc, res = ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 64)
    print("index %.1f us chain %.1f us total %.1f us path %d n %d" % (res.ms_index * 1e3, res.ms_chain * 1e3, res.ms_total * 1e3, res.path, res.n_records), flush=True)
