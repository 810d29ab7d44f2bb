"""Rate of the dense configuration (path 2): short records wrapped at 80 columns, 1 GiB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
ctx = hip.Context(0)
rng = np.random.default_rng(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 100
qa = np.frombuffer(bytes(range(35, 74)), dtype=np.uint8)
parts = []
for i in range(40000):
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
    qual = rng.choice(qa, size=L).tobytes()
    w = lambda b: b"\n".join(b[k:k + 80] for k in range(0, L, 80))
    parts.append(b"@SRR000001.%d 1:N:0:1\n" % i + w(seq) + b"\n+\n" + w(qual) + b"\n")
chunk = b"".join(parts)
reps = (1 << 30) // len(chunk)
data = np.frombuffer(chunk * reps, dtype=np.uint8)
n = 40000 * reps
d = torch.from_numpy(data.copy()).cuda()
table = torch.empty((n + 64, 6), dtype=torch.int64, device="cuda")
ctx.reserve(d.numel())
for i in range(4):
    rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64)
    print("L %d: path %d retries %d index %.3f ms chain %.3f ms -> %.2f TB/s (%d records)" % (L, res.path, res.retries, res.ms_index, res.ms_chain, d.numel() / (res.ms_total * 1e-3) / 1e12, res.n_records), flush=True)
