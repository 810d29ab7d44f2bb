"""Per-phase cycle sums of k_chain_wave (FFQ_PROF): prof_chain.py [bytes] [single|wrapped]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()          # the instrumented build (libffq_probe.so): probes and ablation switches live there
from fastqandfurious_amd.sharded import SyntheticShard
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 30)
kind = sys.argv[2] if len(sys.argv) > 2 else "single"
ctx = hip.Context(0)
sh = SyntheticShard(ctx, kind, nbytes, 0, 1, torch.device("cuda:0"))
n, cap = sh.ext_scanned_bytes, sh.max_records
table = torch.empty((cap, 6), dtype=torch.int64, device='cuda')
ctx.reserve(n)
os.environ['FFQ_PROF'] = '1'
os.environ['FFQ_NO_FAST4'] = '1'
for i in range(3):
    rc, res = ctx.scan_device(sh.ext.data_ptr(), n, table.data_ptr(), cap)
    print("index %.1f us chain %.1f us path %d n %d" % (res.ms_index * 1e3, res.ms_chain * 1e3, res.path, res.n_records), flush=True)
