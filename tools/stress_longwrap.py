"""Differential run for LONG wrapped records (round 6: the general kernels with three look-ahead tiles, ffq_chain.h LA):
seeded streams of reads of 2 ... 40 kbp wrapped at 50 ... 120 columns, quality lines that begin with '@' and '+' planted,
random truncations; every scan's rows / end state / last status / offset against the oracle.  The context is NOT told to
forget between seeds: it moves between its tiers (one look-ahead tile, three, list ranking) as the inputs change.
   tools/stress_longwrap.py [seeds]        FFQ_STRESS_SEED0=n: first seed"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import fastqandfurious_amd  # noqa: F401
from fastqandfurious_amd import hip
from oracle import ffq_oracle as oracle

ctx = hip.default_context(0)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SEED0 = int(os.environ.get("FFQ_STRESS_SEED0", "0"))
QA = np.frombuffer(bytes(range(33, 74)), dtype=np.uint8)
bad, paths = 0, {}
for seed in range(SEED0, SEED0 + nseeds):
    rng = np.random.default_rng(90000 + seed)
    lo, hi = [(2000, 6000), (6000, 12000), (10000, 24000), (20000, 40000), (300, 40000)][seed % 5]
    W = int(rng.integers(50, 121))
    total = int(rng.integers(1 << 20, 6 << 20))
    parts, size, i = [], 0, 0
    while size < total:
        L = int(rng.integers(lo, hi + 1))
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L)
        qual = rng.choice(QA, size=L)
        if seed % 3 == 0:                       # hostile: quality lines that begin with '@' / '+' (one line in five)
            starts = np.arange(0, L, W)
            pick = starts[rng.random(starts.size) < 0.2]
            qual[pick] = rng.choice(np.frombuffer(b"@+", dtype=np.uint8), size=pick.size)
        w = lambda b: b"\n".join(b[k:k + W] for k in range(0, L, W))
        rec = b"@r%d len=%d\n" % (i, L) + w(seq.tobytes()) + b"\n+" + (b"r%d" % i if seed & 1 else b"") + b"\n" + w(qual.tobytes()) + b"\n"
        parts.append(rec); size += len(rec); i += 1
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    cut = int(rng.integers(0, 3))
    if cut == 1:
        data = data[:-int(rng.integers(1, 5000))]
    for kw, extra in ((dict(), 0), (dict(eof=False), 0), (dict(offset=int(data.size) // 3), hip.F_FORCE_GENERAL), (dict(), hip.F_FORCE_GENERAL)):
        want, end, status, off = oracle.scan(data, **kw)
        table, res = ctx.scan_host(data, flags=extra, **kw)
        ok = (table.shape == want.shape and (table == want).all() and int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off)
        paths[(seed % 5, int(res.path), int(res.retries))] = paths.get((seed % 5, int(res.path), int(res.retries)), 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, kw, extra, "path", res.path, "n", len(want), int(res.n_records), "end", end, int(res.end_state), flush=True)
print("seeds", nseeds, "scans", 4 * nseeds, "mismatches", bad, "(kind, path, retries) -> count", dict(sorted(paths.items())))
