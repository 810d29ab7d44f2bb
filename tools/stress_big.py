"""Fewer, larger hostile inputs (tens of MB: hundreds of groups, repair passes, dense regions,
every tier) against the oracle, with the decode.  tools/stress_big.py [seeds]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip
from oracle import ffq_oracle as oracle
import test_gpu_parity as T
ctx = hip.default_context(0)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
paths = {}
for seed in range(nseeds):
    rng = np.random.default_rng(70000 + int(os.environ.get("FFQ_STRESS_SEED0", "0")) + seed)
    kind = seed % 4
    if kind == 0:
        data = T._mess(rng, 120000, fatal=False)
    elif kind == 1:
        data = T.mutate(rng, T.random_records(rng, 60000, 100, 160), 6)
    elif kind == 2:
        data = T.mutate(rng, T.random_records(rng, 40000, 50, 300, wrap=80), 6)
    else:
        data = T._with_dense_regions(rng, "start middle end", "wrapped") * 3
    for kw in (dict(), dict(eof=False, offset=len(data) // 3)):
        ctx.forget()
        want, end, status, off = oracle.scan(data, **kw)
        table, res, qual, qoff = ctx.scan_host(data, flags=hip.F_DECODE_QUAL, table_cap=len(want) + 8, **kw)
        wq, wqoff = oracle.decode_quals(data, want)
        ok = (table.shape == want.shape and (table == want).all() and int(res.end_state) == end and
              int(res.last_status) == status and int(res.end_offset) == off and (qoff == wqoff).all() and (qual == wq).all())
        paths[(kind, int(res.path))] = paths.get((kind, int(res.path)), 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "kind", kind, kw, "path", res.path, "n", len(want), int(res.n_records), flush=True)
print("seeds", nseeds, "mismatches", bad, "MB per input ~", len(data) >> 20, "paths", dict(sorted(paths.items())))
