"""Scan rate across read lengths for records wrapped at 80 columns (WRAP=n: at n columns) (general chain kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
ctx = hip.Context(0)
rng = np.random.default_rng(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else (64 << 20)
W = int(os.environ.get("WRAP", "80"))
# sizes above 64 MiB: a 64 MiB block of distinct records, repeated on the device (the host generator makes ~10 MB/s)
block = min(size, 64 << 20)
reps = max(1, size // block)
Ls = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (100, 300, 1000, 3000, 5000, 20000, 60000)
for L in Ls:
    # every record different (a periodic stream keeps a false chain alive for ever)
    n = max(3, block // (2 * L + 2 * (L // W) + 40))
    qa = np.frombuffer(bytes(range(35, 74)), dtype=np.uint8)
    parts = []
    for i in range(n):
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
        qual = rng.choice(qa, size=L).tobytes()
        w = lambda b: b"\n".join(b[k:k + W] for k in range(0, L, W))
        parts.append(b"@SRR000001.%d 1:N:0:1\n" % i + w(seq) + b"\n+\n" + w(qual) + b"\n")
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    d = torch.from_numpy(data.copy()).cuda()
    if reps > 1:
        d = d.repeat(reps)
        n *= reps
    cap = n + 64
    table = torch.empty((cap, 6), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()        # (the context's stream does not wait for torch's: include/ffq.h)
    ctx.reserve(d.numel()); ctx.forget()
    ms = []
    for i in range(4):
        rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), cap)
        ms.append(res.ms_total)
    assert int(res.n_records) == n, (res.n_records, n)
    print("L %7d: %8d records, path %d retries %d, index %.3f ms chain %.3f ms -> %.3f TB/s" % (L, n, res.path, res.retries, res.ms_index, res.ms_chain, d.numel() / (min(ms) * 1e-3) / 1e12), flush=True)
    del d, table
