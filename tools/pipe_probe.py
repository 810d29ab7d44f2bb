"""Persistent streaming loop with a lagged two-level prefix (k_pipe_probe, ffq_read_probe modes 200 + lag):
lag 0 = the loop alone; the prefixes are checked against the line index's tile counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()          # the instrumented build (libffq_probe.so): probes and ablation switches live there
nbytes = int(float(sys.argv[1])) if len(sys.argv) > 1 else (1 << 30)
ctx = hip.Context(0)
n = nbytes // 322
buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device='cuda')
ctx.synth_single(buf.data_ptr(), 0, n, 42)
ctx.reserve(n * 322)
for rnd in range(3):
    k3 = ctx.read_probe(buf.data_ptr(), n * 322, 3, 10)
    nt = ctx.read_probe(buf.data_ptr(), n * 322, 6, 10)
    print("round %d: index kernel alone %.1f us, pure non-temporal read %.1f us" % (rnd, k3 * 1e3, nt * 1e3), flush=True)
    for big in (0, 10):
        for lag in (0, 1, 2, 3, 4, 6):
            t = ctx.read_probe(buf.data_ptr(), n * 322, 200 + big + lag, 10)
            print("   persistent loop, %s per CU, %s: %.1f us" % ("two workgroups (72 KiB of LDS)" if big else "four workgroups (36 KiB of LDS)",
                  "no prefix" if lag == 0 else "prefix resolved %d iterations later" % lag, t * 1e3), flush=True)
