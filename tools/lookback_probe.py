"""What a decoupled look-back over the tiles costs inside the index kernel (ffq_read_probe).
mode 7: a window of 64 descriptors per round trip (one per lane); modes 100 + K + 16 * barrier + 32 * variant:
64 * K descriptors per round trip; variant 1 stores only, 2 stores + one window load without waiting,
3 every 4th tile only (super-tile traffic), 4 long sleeps between polls, 6 / 7 variant 2 with plain cached /
sc0 loads.  Rounds and retries per look-back come out on stderr.  The first line is the box's health:
the plain index kernel and the pure non-temporal read."""
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
for rnd in range(2):
    k3 = ctx.read_probe(buf.data_ptr(), n * 322, 3, 10)
    nt = ctx.read_probe(buf.data_ptr(), n * 322, 6, 10)
    print("round %d: index kernel alone %.1f us, pure non-temporal read %.1f us" % (rnd, k3 * 1e3, nt * 1e3), flush=True)
    t7 = ctx.read_probe(buf.data_ptr(), n * 322, 7, 10)
    print("   mode 7 (window 64, moving): %.1f us" % (t7 * 1e3), flush=True)
    for variant in (0, 1, 2, 3, 4, 6, 7):
        for K in (1, 2, 4) + ((8,) if variant == 0 else ()):
            for bar in ((0, 16) if variant == 0 and K > 1 else (0,)):
                mode = 100 + K + bar + 32 * variant
                t = ctx.read_probe(buf.data_ptr(), n * 322, mode, 10)
                print("   variant %d window %3d%s: %.1f us" % (variant, 64 * K, ", barrier" if bar else "", t * 1e3), flush=True)
                sys.stderr.flush()
