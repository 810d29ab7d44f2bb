"""What the FIRST scan of a context costs on each shape of input (the tiers it tries and leaves on the way to the one
that takes it) against the steady state: wall clock of scan_device, 1 GiB, after ctx.forget().
tools/first_scan.py [bytes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
ctx = hip.Context(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30)
rng = np.random.default_rng(0)
qa = np.frombuffer(bytes(range(35, 74)), dtype=np.uint8)


def block_of(L, wrap):
    parts, tot, i = [], 0, 0
    while tot < (32 << 20):
        n = int(rng.integers(L // 2, L * 3 // 2 + 1))
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tobytes()
        qual = rng.choice(qa, size=n).tobytes()
        if wrap:
            seq = b"\n".join(seq[k:k + wrap] for k in range(0, n, wrap))
            qual = b"\n".join(qual[k:k + wrap] for k in range(0, n, wrap))
        r = b"@read%d len=%d\n" % (i, n) + seq + b"\n+\n" + qual + b"\n"
        parts.append(r); tot += len(r); i += 1
    return np.frombuffer(b"".join(parts), dtype=np.uint8), i


for wrap in (0, 80):
    for L in (150, 1000, 10000, 100000):
        block, i = block_of(L, wrap)
        reps = max(1, size // block.size)
        d = torch.from_numpy(block.copy()).cuda().repeat(reps)
        n = i * reps
        cap = n + 64
        table = torch.empty((cap, 6), dtype=torch.int64, device="cuda")
        qual = torch.empty(((d.numel() + 16383) >> 14) * 16384, dtype=torch.int8, device="cuda")
        qoff = torch.empty(cap + 1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.reserve(d.numel())
        out = []
        for name, flags in (("scan", 0), ("decode", hip.F_DECODE_QUAL), ("single pass", hip.F_DECODE_QUAL | hip.F_SINGLE_PASS)):
            ctx.forget()
            ts = []
            for it in range(5):
                t0 = time.perf_counter()
                rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), cap, flags=flags, d_qual=qual.data_ptr(), qual_cap=qual.numel(), d_qoff=qoff.data_ptr())
                ts.append((time.perf_counter() - t0) * 1e3)
                assert rc == 0 and int(res.n_records) == n, (rc, int(res.n_records), n)
                if it == 0: first = (res.path, res.retries)
            out.append("%s: first %.2f ms (path %d, retries %d), then %.2f (path %d)" % (name, ts[0], first[0], first[1], min(ts[1:]), res.path))
        print("%s L ~%6d: %s" % ("wrapped  " if wrap else "four-line", L, " | ".join(out)), flush=True)
        del d, table, qual, qoff
