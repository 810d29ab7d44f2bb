"""k_chain_wave cut short after each phase (FFQ_ABLATE): ablate_chain.py [bytes] [single|wrapped]
1: window in LDS, 2: nodes numbered, 3: scanner calls, 4: membership, 0: everything"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
for abl in (0, 1, 2, 3, 4, 0):
    os.environ['FFQ_ABLATE'] = str(abl)
    ms = []
    for i in range(8):
        rc, res = ctx.scan_device(sh.ext.data_ptr(), n, table.data_ptr(), cap)
        ms.append((res.ms_index, res.ms_chain))
    ms = np.array(ms[2:])
    print("ablate", abl, "index %.1f us chain-total %.1f us" % (ms[:, 0].mean() * 1e3, ms[:, 1].mean() * 1e3), "path", res.path, flush=True)
