"""Interleaved within-process A/B of k_scan_lines ablation levels (FFQ_K1_ABLATE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()          # the instrumented build (libffq_probe.so): probes and ablation switches live there
levels = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6,7".split(","))]
nbytes = int(float(sys.argv[2])) if len(sys.argv) > 2 else (1 << 30)
ctx = hip.Context(0)
n = nbytes // 322
buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device='cuda')
ctx.synth_single(buf.data_ptr(), 0, n, 42)
table = torch.empty((n + 64, 6), dtype=torch.int64, device='cuda')
ctx.reserve(n * 322)
os.environ['FFQ_ABLATE'] = '1'
res = {a: [] for a in levels}
probe = []
for rnd in range(12):
    for a in levels:
        os.environ['FFQ_K1_ABLATE'] = str(a)
        rc, r = ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 64)
        if rnd >= 2:
            res[a].append(r.ms_index * 1e3)
    probe.append(ctx.read_probe(buf.data_ptr(), n * 322, 0, 3) * 1e3)
print("read probe: min %.1f med %.1f us" % (min(probe), float(np.median(probe))))
for a in levels:
    print("k1 ablate %d: min %.1f  med %.1f us" % (a, min(res[a]), float(np.median(res[a]))))
