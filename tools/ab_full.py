"""A/B: index kernel on a buffer that is a whole number of tiles (unchecked loads) vs one that is not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
ctx = hip.Context(0)
n = (1 << 30) // 322 + 8
buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device='cuda')
ctx.synth_single(buf.data_ptr(), 0, n, 42)
table = torch.empty((n + 64, 6), dtype=torch.int64, device='cuda')
ctx.reserve(n * 322)
res = {}
for rnd in range(10):
    for nb in ((1 << 30), (1 << 30) - 302):
        rc, r = ctx.scan_device(buf.data_ptr(), nb, table.data_ptr(), n + 64)
        if rnd >= 2: res.setdefault(nb, []).append(r.ms_index * 1e3)
for nb, v in res.items():
    print("n_bytes %d (%s): index min %.1f med %.1f us" % (nb, "whole tiles" if nb % 16384 == 0 else "ragged", min(v), float(np.median(v))))
