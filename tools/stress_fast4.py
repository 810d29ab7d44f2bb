"""Stress of the four-line fast path: regular 4-line records with a few random edits (bytes
turned into newlines / '@' / '+', bytes deleted or inserted, records wrapped, CRLF, a cut end),
GPU scan (+decode) vs oracle.  The fast path must either reproduce the oracle or decline.
tools/stress_fast4.py [seeds] [wrapped]   (wrapped: the same edits on records folded at 60-100
columns -- the general path's kernels, repairs included)"""
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
    if int(res.path) != 6 and not (int(res.path) & 8):        # (| 8: the general path's one-pass decode, every byte in place)
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


EXTRA = int(os.environ.get("FFQ_STRESS_FLAGS", "0"))      # e.g. 16 = FFQ_F_SINGLE_PASS: the same inputs through the single-pass kernel


def main():
    ctx = hip.default_context(0)
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    wrapped = len(sys.argv) > 2 and sys.argv[2] == "wrapped"
    bad = 0
    paths = {}
    for seed in range(nseeds):
        rng = np.random.default_rng(9000 + int(os.environ.get("FFQ_STRESS_SEED0", "0")) + seed)
        lo, hi = ((100, 160), (20, 60), (250, 400), (1, 30), (1000, 3000))[seed % 5]
        nrec = int(rng.integers(200, 6000)) if hi < 1000 else int(rng.integers(50, 600))
        data = T.random_records(rng, nrec, lo, hi, wrap=int(rng.integers(60, 101)) if wrapped else 0,
                                repeat_hdr=bool(seed & 1))
        nedits = (0, 1, 2, 5, 20)[(seed // 5) % 5]
        data = T.mutate(rng, data, nedits)
        if seed % 3 == 0:
            data = data[:len(data) - int(rng.integers(1, 300))]
        ctx.forget()
        for kw in (dict(), dict(eof=False), dict(offset=len(data) // 3), dict(sentinel=False, offset=5)):
            want, end, status, off = oracle.scan(data, **kw)
            table, res, qual, qoff = ctx.scan_host(data, flags=hip.F_DECODE_QUAL | EXTRA, **kw)
            wq, wqoff = oracle.decode_quals(data, want)
            ok = (table.shape == want.shape and (table == want).all() and int(res.end_state) == end and
                  int(res.last_status) == status and int(res.end_offset) == off and same_quals(res, want, qual, qoff, wq, wqoff))
            key = (nedits, int(res.path))
            paths[key] = paths.get(key, 0) + 1
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, kw, "edits", nedits, "path", res.path, "n", len(want), int(res.n_records),
                      flush=True)
            ctx.forget()
    print("seeds", nseeds, "mismatches", bad, "paths (edits, path) -> count", dict(sorted(paths.items())))


if __name__ == "__main__":
    main()
