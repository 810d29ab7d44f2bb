import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
hip.use_probe_build()
ctx = hip.Context(0)
for seed in (0, 1, 2):
  for L in (1000, 1500):
    rng = np.random.default_rng(seed)
    block = 64 << 20
    n = block // (2 * L + 2 * (L // 80) + 40)
    qa = np.frombuffer(bytes(range(35, 74)), dtype=np.uint8)
    parts = []
    for i in range(n):
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
        qual = rng.choice(qa, size=L).tobytes()
        w = lambda b: b"\n".join(b[k:k + 80] for k in range(0, L, 80))
        parts.append(b"@SRR000001.%d 1:N:0:1\n" % i + w(seq) + b"\n+\n" + w(qual) + b"\n")
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    d = torch.from_numpy(data.copy()).cuda()
    table = torch.empty((n + 64, 6), dtype=torch.int64, device="cuda")
    ctx.reserve(d.numel()); ctx.forget()
    for i in range(3):
        if i == 2: os.environ["FFQ_DEBUG"] = "1"
        rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64)
        os.environ.pop("FFQ_DEBUG", None)
    print("seed", seed, "L", L, "n", n, "path", res.path, "retries", res.retries, flush=True)
