"""Rate of the last tier (wave walker, FFQ_F_FORCE_SERIAL) on short and on very long records."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip, synth
ctx = hip.Context(0)
rng = np.random.default_rng(0)
def run(name, data, n):
    d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    table = torch.empty((n + 64, 6), dtype=torch.int64, device="cuda")
    rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64, flags=hip.F_FORCE_SERIAL)
    rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64, flags=hip.F_FORCE_SERIAL)
    print("%-28s %8d records path %d: %.2f ms -> %.2f GB/s, %.2f us/record" % (name, res.n_records, res.path, res.ms_total, d.numel() / res.ms_total / 1e6, res.ms_total * 1e3 / max(1, res.n_records)))
run("S-single 16 MiB", synth.single(0, 52000, seed=42).tobytes(), 52000)
L = 60000
qa = np.frombuffer(bytes(range(35, 74)), dtype=np.uint8)
parts = []
for i in range(140):
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
    qual = rng.choice(qa, size=L).tobytes()
    w = lambda b: b"\n".join(b[k:k + 80] for k in range(0, L, 80))
    parts.append(b"@r%d\n" % i + w(seq) + b"\n+\n" + w(qual) + b"\n")
run("60 kb wrapped at 80, 16 MiB", b"".join(parts), 140)
