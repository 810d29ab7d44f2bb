"""Small driver for profiling: N scans of a synthetic buffer resident in HBM.
usage: run_scan.py [bytes] [reps] [single|wrapped] [decode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
from fastqandfurious_amd.sharded import SyntheticShard
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 30)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "single"
decode = len(sys.argv) > 4 and sys.argv[4] in ("decode", "singlepass")
single_pass = len(sys.argv) > 4 and sys.argv[4] == "singlepass"
ctx = hip.Context(0)
sh = SyntheticShard(ctx, kind, nbytes, 0, 1, torch.device("cuda:0"))
n = sh.ext_scanned_bytes
cap = sh.max_records
table = torch.empty((cap, 6), dtype=torch.int64, device='cuda')
qual = qoff = None
if decode:
    qual = torch.empty(max(n // 2 + 4096, ((n + 16383) >> 14) * hip.SEG_STRIDE), dtype=torch.int8, device='cuda')
    qoff = torch.empty(cap + 1, dtype=torch.int64, device='cuda')
ctx.reserve(n)
for i in range(reps):
    rc, res = ctx.scan_device(sh.ext.data_ptr(), n, table.data_ptr(), cap,
                              flags=(hip.F_DECODE_QUAL if decode else 0) | (hip.F_SINGLE_PASS if single_pass else 0),
                              d_qual=qual.data_ptr() if decode else None, qual_cap=qual.numel() if decode else 0,
                              d_qoff=qoff.data_ptr() if decode else None)
    print("index %.1f us chain %.1f us decode %.1f us total %.1f us path %d retries %d n %d" % (res.ms_index * 1e3, res.ms_chain * 1e3, res.ms_decode * 1e3, res.ms_total * 1e3, res.path, res.retries, res.n_records), flush=True)
