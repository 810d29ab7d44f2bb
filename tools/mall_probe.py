"""Does the 256 MiB Infinity Cache serve a second read?  (a) a pure read of X MiB repeated,
plain and non-temporal loads; (b) the real decode pipeline on X MiB buffers: per-GiB time of
the index kernel, the row/offset kernels and the decode kernel as a function of X."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()          # the instrumented build (libffq_probe.so): probes and ablation switches live there
ctx = hip.Context(0)
MIB = 1 << 20
big = torch.empty(2048 * MIB + 64, dtype=torch.uint8, device='cuda')
nmax = (2048 * MIB) // 322
ctx.synth_single(big.data_ptr(), 0, nmax, 42)
print("== pure re-read of X MiB (20 launches back to back)")
for x in (16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2048):
    nb = x * MIB
    a = ctx.read_probe(big.data_ptr(), nb, 0, 20)
    nt = ctx.read_probe(big.data_ptr(), nb, 6, 20)
    g = ctx.read_probe(big.data_ptr(), nb, 1, 20)
    print("  %5d MiB: plain %.2f TB/s   non-temporal %.2f TB/s   grid-stride %.2f TB/s" % (x, nb / a / 1e9, nb / nt / 1e9, nb / g / 1e9), flush=True)
print("== decode pipeline on X MiB (per GiB of input: index / rows+offsets / decode ms)")
for x in (32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 2048):
    n = (x * MIB) // 322
    nb = n * 322
    table = torch.empty((n + 64, 6), dtype=torch.int64, device='cuda')
    qual = torch.empty(nb // 2 + 64, dtype=torch.int8, device='cuda')
    qoff = torch.empty(n + 65, dtype=torch.int64, device='cuda')
    ctx.reserve(nb)
    mi, mc, md = [], [], []
    for i in range(12):
        rc, res = ctx.scan_device(big.data_ptr(), nb, table.data_ptr(), n + 64, flags=hip.F_DECODE_QUAL,
                                  d_qual=qual.data_ptr(), qual_cap=qual.numel(), d_qoff=qoff.data_ptr())
        assert res.path == 3 and res.n_records == n
        if i >= 4:
            mi.append(res.ms_index); mc.append(res.ms_chain); md.append(res.ms_decode)
    s = 1024.0 / x
    print("  %5d MiB: index %.3f  rows+offsets %.3f  decode %.3f  ms/GiB   (decode: %.2f TB/s of algorithmic bytes)"
          % (x, min(mi) * s, min(mc) * s, min(md) * s, (2 * 150 + 16) * n / min(md) / 1e9), flush=True)
    del table, qual, qoff
