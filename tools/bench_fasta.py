"""Device FASTA scan rate: 1 GiB of 60-column FASTA resident in HBM (entries of ~10 kb and of ~200 b)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
ctx = hip.Context(0)
rng = np.random.default_rng(0)
for seqlen in (10000, 200):
    lines = seqlen // 60
    body = b"\n".join(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=60).tobytes() for _ in range(lines)) + b"\n"
    rec = b">contig_000001 len=%d\n" % seqlen + body
    n = (1 << 30) // len(rec)
    data = np.frombuffer(b"\n" + rec * n, dtype=np.uint8)
    d = torch.from_numpy(data.copy()).cuda()
    table = torch.empty((n + 64, 6), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()        # (the context's stream does not wait for torch's: include/ffq.h)
    ctx.reserve(d.numel())
    ms = []
    for i in range(6):
        rc, res = ctx.scan_fasta_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64)
        ms.append((res.ms_index, res.ms_chain, res.ms_total))
    assert int(res.n_records) == n - 1, (res.n_records, n)
    mi = min(m[2] for m in ms)
    print("FASTA %5d-base entries: %8d entries, index %.3f ms + rows %.3f ms -> %.2f TB/s, %.1f M entries/s"
          % (seqlen, n, ms[-1][0], ms[-1][1], d.numel() / (mi * 1e-3) / 1e12, n / (mi * 1e-3) / 1e6), flush=True)
    del d, table
