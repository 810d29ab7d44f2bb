"""One-off stress: seeded hostile inputs, GPU scan (+decode) vs oracle.  tools/stress_diff.py [seeds]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip
from oracle import ffq_oracle as oracle
import test_gpu_parity as T
def same_quals(res, want, qual, qoff, wq, wqoff):
    """packed stream, or -- res.path 6, FFQ_F_SINGLE_PASS -- segmented: record i = qual[qoff[i] : qoff[i] + pos5 - pos4]"""
    if int(res.path) != 6 and not (int(res.path) & 8):        # (6: the fast path's single pass; | 8: the general path's, every byte in place)
        return qoff.shape == wqoff.shape and (qoff == wqoff).all() and qual.shape == wq.shape and (qual == wq).all()
    n = len(want)
    if qoff.shape[0] != n + 1:
        return False
    if n == 0:
        return True
    lens = want[:, 5] - want[:, 4]
    if not (qoff[1:n] >= qoff[:n - 1] + lens[:n - 1]).all() or int(qoff[n]) != int(qoff[n - 1] + lens[n - 1]):
        return False
    idx = np.repeat(qoff[:n] - wqoff[:n], lens) + np.arange(wq.size)
    return bool((qual[idx] == wq).all())


ROOM = int(os.environ.get("FFQ_STRESS_QUAL_ROOM", "0")) or None      # 16384: room for the in-place layouts (with FFQ_STRESS_FLAGS=16)
EXTRA = int(os.environ.get("FFQ_STRESS_FLAGS", "0"))      # e.g. 16 = FFQ_F_SINGLE_PASS: the same inputs through the single-pass kernel
ctx = hip.default_context(0)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
paths = {}
SEED0 = int(os.environ.get("FFQ_STRESS_SEED0", "0"))      # first seed (new inputs: a run with another offset)
for seed in range(SEED0, SEED0 + nseeds):
    rng = np.random.default_rng(5000 + seed)
    kind = seed % 5
    if kind == 0:
        data = T._mess(rng, 4000, fatal=False)
    elif kind == 1:
        data = T._mess(rng, 4000, fatal=True)
    elif kind == 2:
        data = T.random_records(rng, 600, 200, 4000, wrap=int(rng.integers(30, 100)))
    elif kind == 3:
        data = T.random_records(rng, 3000, 1, 300, wrap=0, repeat_hdr=bool(seed & 1))
    else:
        data = T.random_records(rng, 2000, 50, 300, wrap=80)
    cut = int(rng.integers(0, 50))
    if cut:
        data = data[:-cut]
    for kw, extra in ((dict(), 0), (dict(eof=False), 0), (dict(offset=len(data) // 3), 0), (dict(), hip.F_FORCE_SERIAL),
                      (dict(eof=False, offset=7), hip.F_FORCE_SERIAL)):
        want, end, status, off = oracle.scan(data, **kw)
        table, res, qual, qoff = ctx.scan_host(data, flags=hip.F_DECODE_QUAL | extra | EXTRA, qual_room=ROOM, **kw)
        wq, wqoff = oracle.decode_quals(data, want)
        ok = (table.shape == want.shape and (table == want).all() and int(res.end_state) == end and
              int(res.last_status) == status and int(res.end_offset) == off and same_quals(res, want, qual, qoff, wq, wqoff))
        paths[(kind, int(res.path))] = paths.get((kind, int(res.path)), 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "kind", kind, kw, "path", res.path, "n", len(want), int(res.n_records), flush=True)
print("seeds", nseeds, "mismatches", bad, "paths (kind, path) -> count", dict(sorted(paths.items())))
