"""The index kernel lands in one of two modes per process (0.157 / 0.162 ms per GiB, profiles/r06_probes/ab_single1g_r04_vs_head.txt).
Does the mode follow the ADDRESS of the input buffer?  One process: S-single 1 GiB generated at several offsets of one arena and
into several separate allocations; median ms_index of 40 scans each (FFQ_F_POLL_RESULT off: events around the kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd  # noqa: F401
from fastqandfurious_amd import hip
ctx = hip.Context(0)
n_rec = (1 << 30) // 322
n = n_rec * 322
table = torch.empty((n_rec + 64, 6), dtype=torch.int64, device="cuda")
ctx.reserve(n)


def rate(ptr):
    ctx.synth_single(ptr, 0, n_rec, seed=42)
    ctx.sync()
    ms = []
    for _ in range(40):
        rc, res = ctx.scan_device(ptr, n, table.data_ptr(), table.shape[0])
        ms.append(res.ms_index)
    ms.sort()
    return ms[len(ms) // 2], ms[2], ms[-3]


arena = torch.empty(n + (96 << 20), dtype=torch.uint8, device="cuda")
for rep in range(2):
    for off in (0, 16 << 10, 64 << 10, 1 << 20, 2 << 20, 6 << 20, 32 << 20, 64 << 20):
        m = rate(arena.data_ptr() + off)
        print("arena %#x + %9d: index median %.4f ms  (p5 %.4f p95 %.4f)" % (arena.data_ptr(), off, *m), flush=True)
bufs = []
for i in range(6):
    b = torch.empty(n + 4096 * (i + 1), dtype=torch.uint8, device="cuda")
    bufs.append(b)
    m = rate(b.data_ptr())
    print("allocation %d at %#x: index median %.4f ms  (p5 %.4f p95 %.4f)" % (i, b.data_ptr(), *m), flush=True)
m = rate(bufs[0].data_ptr())
print("allocation 0 again: %.4f" % m[0])
