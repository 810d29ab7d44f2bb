"""Scan rate across read lengths (fixed-length four-line records, 256 MiB each): perf cliffs?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
ctx = hip.Context(0)
rng = np.random.default_rng(0)
for L in (20, 36, 50, 100, 151, 250, 1000, 10000, 100000, 1000000):
    hdr = b"@SRR000001.%d 1:N:0:1\n"
    rec_len = len(hdr % 1234567) + 2 * L + 4
    n = max(4, (256 << 20) // rec_len)
    # one template record repeated (content does not matter for the rate), unique-ish headers
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
    qual = rng.choice(np.frombuffer(bytes(range(35, 74)), dtype=np.uint8), size=L).tobytes()
    rec = (hdr % 1234567) + seq + b"\n+\n" + qual + b"\n"
    data = np.frombuffer(rec * n, dtype=np.uint8)
    d = torch.from_numpy(data.copy()).cuda()
    cap = n + 64
    table = torch.empty((cap, 6), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()        # (the context's stream does not wait for torch's: include/ffq.h)
    ctx.reserve(d.numel())
    ms = []
    for i in range(5):
        rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), cap)
        ms.append(res.ms_total)
    assert int(res.n_records) == n, (res.n_records, n)
    print("L %7d: %8d records, path %d, index %.3f ms chain %.3f ms -> %.2f TB/s" % (L, n, res.path, res.ms_index, res.ms_chain, d.numel() / (min(ms) * 1e-3) / 1e12), flush=True)
    del d, table
