"""Streaming-read ceiling vs k_scan_lines, interleaved in one process."""
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
table = torch.empty((n + 64, 6), dtype=torch.int64, device='cuda')
ctx.reserve(n * 322)
for rnd in range(4):
    a = ctx.read_probe(buf.data_ptr(), n * 322, 0, 10)
    b = ctx.read_probe(buf.data_ptr(), n * 322, 1, 10)
    nt = ctx.read_probe(buf.data_ptr(), n * 322, 6, 10)
    k = ctx.read_probe(buf.data_ptr(), n * 322, 2, 10)
    k3 = ctx.read_probe(buf.data_ptr(), n * 322, 3, 10)
    k4 = ctx.read_probe(buf.data_ptr(), n * 322, 4, 10)
    idx = []
    for i in range(5):
        rc, res = ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 64)
        idx.append(res.ms_index)
    idx2 = []
    for i in range(5):
        rc, res = ctx.scan_device(buf.data_ptr(), (n * 322) >> 14 << 14, table.data_ptr(), n + 64)
        idx2.append(res.ms_index)
    gb = n * 322 / 1e9
    print("round %d: read probe tile/block %.1f us (%.2f TB/s)  grid-stride %.1f us (%.2f TB/s)  tile/block non-temporal %.1f us (%.2f TB/s)  k_scan_lines in a step %.1f (mean %.1f) us, alone back to back %.1f us, alone between events %.1f us (ragged variant %.1f us)"
          % (rnd, a * 1e3, gb / a, b * 1e3, gb / b, nt * 1e3, gb / nt, min(idx) * 1e3, sum(idx) / len(idx) * 1e3, k * 1e3, k3 * 1e3, k4 * 1e3), flush=True)
    print("   k_scan_lines of the 5 scans after the probes:", " ".join("%.1f" % (x * 1e3) for x in idx),
          "| whole tiles only:", " ".join("%.1f" % (x * 1e3) for x in idx2))
