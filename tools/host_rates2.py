"""ffq_scan_host rate from pageable memory (with and without the decode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip, synth
ctx = hip.Context(0)
for mib in (16, 64, 256, 1024):
    n = (mib << 20) // 322
    data = synth.single(0, n, seed=42)
    for flags in (0, hip.F_DECODE_QUAL):
        ctx.scan_host(data, flags=flags)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); out = ctx.scan_host(data, flags=flags); best = min(best, time.perf_counter() - t0)
        assert int(out[1].n_records) == n
        print("%5d MiB%s: %.1f GB/s" % (mib, " +decode" if flags else "", data.size / best / 1e9), flush=True)
