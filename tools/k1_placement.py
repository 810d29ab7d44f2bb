"""Does the index kernel's time depend on where its output lands?  One input buffer, several
contexts (each with its own index allocation, some padded apart by dummy allocations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()          # the instrumented build (libffq_probe.so): probes and ablation switches live there
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 30)
n = nbytes // 322
c0 = hip.Context(0)
buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device='cuda')
print('[input] %#x' % buf.data_ptr(), file=sys.stderr)
c0.synth_single(buf.data_ptr(), 0, n, 42)
pads = []
ctxs = []
for i in range(8):
    c = hip.Context(0)
    c.reserve(n * 322)
    ctxs.append(c)
    pads.append(torch.empty((i * 2 + 1) * 1234567, dtype=torch.uint8, device='cuda'))
for rnd in range(3):
    print("round", rnd, " ".join("%.1f" % (c.read_probe(buf.data_ptr(), n * 322, 3, 6) * 1e3) for c in ctxs), "us (one number per context)", flush=True)
