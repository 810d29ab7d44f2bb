import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip, synth
hip.use_probe_build()          # the instrumented build (libffq_probe.so): probes and ablation switches live there
os.environ['FFQ_DEBUG'] = '1'
ctx = hip.Context(0)
data, start = synth.wrapped(0, 2000, seed=43)
table, res = ctx.scan_host(data)
print("path", res.path, "n", res.n_records)
tiles = (start // 16384)
for g in range(1, 6):
    lo = g * 8 * 16384
    i = np.searchsorted(start, lo)
    print("group", g, "first record start >= own_lo:", start[i], "(nl coord = start)", " prev:", start[i-1])
