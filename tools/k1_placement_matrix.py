"""Index kernel time for every pair (input allocation, index allocation) in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()          # the instrumented build (libffq_probe.so): probes and ablation switches live there
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 32)
n = nbytes // 322
c0 = hip.Context(0)
bufs, ctxs = [], []
for i in range(4):
    b = torch.empty(n * 322 + 64, dtype=torch.uint8, device='cuda')
    c0.synth_single(b.data_ptr(), 0, n, 42)
    bufs.append(b)
    c = hip.Context(0); c.reserve(n * 322); ctxs.append(c)
for rnd in range(2):
    for bi, b in enumerate(bufs):
        print("round", rnd, "input", bi, " ".join("%.1f" % (c.read_probe(b.data_ptr(), n * 322, 3, 5) * 1e3) for c in ctxs),
              "us (one number per index allocation); read probe %.1f" % (c0.read_probe(b.data_ptr(), n * 322, 0, 5) * 1e3), flush=True)
