"""k_scan_lines / whole-scan time vs input size, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()          # the instrumented build (libffq_probe.so): probes and ablation switches live there
ctx = hip.Context(0)
nmax = (10 << 30) // 322
buf = torch.empty(nmax * 322 + 64, dtype=torch.uint8, device='cuda')
ctx.synth_single(buf.data_ptr(), 0, nmax, 42)
table = torch.empty((nmax + 64, 6), dtype=torch.int64, device='cuda')
ctx.reserve(nmax * 322)
for gib in (0.125, 0.25, 0.5, 1, 2, 4, 10):
    n = int(gib * (1 << 30)) // 322
    idx, tot = [], []
    for i in range(8):
        rc, r = ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 64)
        if i >= 2:
            idx.append(r.ms_index * 1e3); tot.append(r.ms_total * 1e3)
    pr = ctx.read_probe(buf.data_ptr(), n * 322, 0, 5) * 1e3
    gb = n * 322 / 1e9
    print("%6.3f GiB: probe %8.1f us (%.2f TB/s) | index min %8.1f med %8.1f us (%.2f TB/s) | total med %8.1f us (%.2f TB/s file) path %d"
          % (gib, pr, gb / pr * 1e3 / 1e3, min(idx), float(np.median(idx)), gb / float(np.median(idx)) * 1e3 / 1e3, float(np.median(tot)), gb / float(np.median(tot)) * 1e3 / 1e3, r.path), flush=True)
