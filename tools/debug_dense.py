"""Debug: dense four-line input through the fast path (FFQ_DEBUG=1 FFQ_USE_PROBE_BUILD=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip
from oracle import ffq_oracle as oracle
ctx = hip.Context(0)
for seed in (1, 3):
    rng = np.random.default_rng(300 + seed)
    qch = np.frombuffer(b"@+I5@", dtype=np.uint8)
    parts = []
    for i in range(90000):
        L = int(rng.integers(1, 14))
        q = rng.choice(qch, size=L).tobytes()
        parts.append(b"@%d\n" % i + b"ACGTN"[:1] * L + b"\n+\n" + q + b"\n")
    data = b"".join(parts)
    for cut in (0, 5):
        d = data[:len(data) - cut]
        want, end, status, off = oracle.scan(d)
        table, res = ctx.scan_host(d, table_cap=len(want) + 8)
        print("seed", seed, "cut", cut, "path", res.path, "n", res.n_records, len(want), "end", res.end_state, end, "tail", d[-30:], flush=True)
        if len(want):
            print("   last rows want", want[-1], "irr candidates near", flush=True)
