"""Four-line records across read lengths, with and without the Phred decode (packed / single pass): where the decode's
single pass (csrc/ffq_fused.h) changes layout (segments: lines up to 512 bytes behind a tile's end; in place: any).
tools/shape_sweep_decode.py [bytes] [L,L,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqandfurious_amd
from fastqandfurious_amd import hip
ctx = hip.Context(0)
rng = np.random.default_rng(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30)
LS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (150, 250, 400, 600, 1000, 3000, 10000, 30000)
for L in LS:
    # variable lengths around L (0.5 L .. 1.5 L), a 32 MiB block of distinct records repeated on the device
    parts, tot, i = [], 0, 0
    while tot < (32 << 20):
        n = int(rng.integers(L // 2, L * 3 // 2 + 1))
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tobytes()
        qual = rng.choice(np.frombuffer(bytes(range(35, 74)), dtype=np.uint8), size=n).tobytes()
        r = b"@read%d len=%d\n" % (i, n) + seq + b"\n+\n" + qual + b"\n"
        parts.append(r); tot += len(r); i += 1
    block = np.frombuffer(b"".join(parts), dtype=np.uint8)
    reps = max(1, size // block.size)
    d = torch.from_numpy(block.copy()).cuda().repeat(reps)
    n = i * reps
    cap = n + 64
    table = torch.empty((cap, 6), dtype=torch.int64, device="cuda")
    qual = torch.empty(((d.numel() + 16383) >> 14) * 16384, dtype=torch.int8, device="cuda")
    qoff = torch.empty(cap + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()        # (the context's stream does not wait for torch's: include/ffq.h)
    ctx.reserve(d.numel()); ctx.forget()
    out = []
    for name, flags in (("scan", 0), ("decode packed", hip.F_DECODE_QUAL), ("decode single pass", hip.F_DECODE_QUAL | hip.F_SINGLE_PASS)):
        ms = []
        for _ in range(4):
            rc, res = ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), cap, flags=flags, d_qual=qual.data_ptr(), qual_cap=qual.numel(), d_qoff=qoff.data_ptr())
            if rc != 0 or int(res.n_records) != n:
                print('  FAIL', name, 'rc', rc, hip.last_error() if hasattr(hip,'last_error') else '', 'n', int(res.n_records), 'want', n, 'path', res.path, 'end_state', res.end_state, 'qual', int(res.n_qual_bytes), qual.numel(), flush=True)
                break
            ms.append(res.ms_total)
        if len(ms) == 4: out.append("%s path %d %.2f TB/s" % (name, res.path, d.numel() / (min(ms) * 1e-3) / 1e12))
    print("L ~%6d (%8d records): %s" % (L, n, " | ".join(out)), flush=True)
    del d, table, qual, qoff
