import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()          # the instrumented build (libffq_probe.so): probes and ablation switches live there
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
size = 4 << 20
rng = np.random.default_rng(0)
n = max(3, size // (2 * L + 2 * (L // 80) + 40))
qa = np.frombuffer(bytes(range(35, 74)), dtype=np.uint8)
parts = []
for i in range(n):
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
    qual = rng.choice(qa, size=L).tobytes()
    w = lambda b: b"\n".join(b[k:k + 80] for k in range(0, L, 80))
    parts.append(b"@SRR000001.%d 1:N:0:1\n" % i + w(seq) + b"\n+\n" + w(qual) + b"\n")
data = np.frombuffer(b"".join(parts), dtype=np.uint8)
ctx = hip.Context(0)
d = torch.from_numpy(data.copy()).cuda()
table = torch.empty((n + 64, 6), dtype=torch.int64, device="cuda")
os.environ["FFQ_DEBUG"] = "1"
os.environ["FFQ_NO_FAST4"] = "1"
rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64)
print("path", res.path, "n", res.n_records, "of", n)
