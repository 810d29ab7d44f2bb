"""What a decoupled look-back over the tiles costs inside the index kernel (ffq_read_probe mode 7)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 30)
ctx = hip.Context(0)
n = nbytes // 322
buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device='cuda')
ctx.synth_single(buf.data_ptr(), 0, n, 42)
ctx.reserve(n * 322)
for rnd in range(4):
    k3 = ctx.read_probe(buf.data_ptr(), n * 322, 3, 10)
    k7 = ctx.read_probe(buf.data_ptr(), n * 322, 7, 10)
    k8 = ctx.read_probe(buf.data_ptr(), n * 322, 8, 10)
    nt = ctx.read_probe(buf.data_ptr(), n * 322, 6, 10)
    print("round %d: index kernel alone %.1f us, with the look-back %.1f us (non-temporal read probe %.1f us); with non-temporal entry stores %.1f us" % (rnd, k3 * 1e3, k7 * 1e3, nt * 1e3, k8 * 1e3), flush=True)
