"""Differential stress of the DENSE tiers (tiles over their slot: k_rows4's DENSE instantiation, k_dense_walk):
seeded hostile short-line input -- tiny records with '@' / '+' heavy qualities, blank stretches, wrapped tiny
records, random edits, regular 100-300 base blocks in between -- GPU scan (+ decode) against the oracle, through
the fast path (where it stands) and with it switched off, at eof and not, from an offset, from a fresh context and
from one that remembers.     tools/stress_dense.py [seeds]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip
from oracle import ffq_oracle as oracle


import test_gpu_parity as T
dense_mess = T.dense_mess


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(os.environ.get("FFQ_STRESS_SEED0", "0"))
    bad, paths = 0, {}
    keep = hip.Context(0)
    for seed in range(seed0, seed0 + nseeds):
        rng = np.random.default_rng(77000 + seed)
        data = dense_mess(rng, int(rng.integers(100000, 1500000)), hostile=(seed % 3 != 0))
        cut = int(rng.integers(0, 40)) if seed % 2 else 0
        if cut:
            data = data[:-cut]
        fresh = hip.Context(0)
        for kw, fl, ctx in ((dict(), 0, fresh), (dict(), hip.F_FORCE_GENERAL, fresh), (dict(eof=False), 0, keep),
                            (dict(offset=len(data) // 3), hip.F_FORCE_GENERAL, keep), (dict(offset=len(data) // 2, eof=False), 0, fresh)):
            want, end, status, off = oracle.scan(data, **kw)
            table, res, qual, qoff = ctx.scan_host(data, flags=hip.F_DECODE_QUAL | fl, table_cap=len(want) + 8, **kw)
            wq, wqoff = oracle.decode_quals(data, want)
            ok = (table.shape == want.shape and (table == want).all() and int(res.end_state) == end and
                  int(res.last_status) == status and int(res.end_offset) == off and
                  qoff.shape == wqoff.shape and (qoff == wqoff).all() and qual.shape == wq.shape and (qual == wq).all())
            paths[(fl, int(res.path))] = paths.get((fl, int(res.path)), 0) + 1
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, kw, "flags", fl, "path", res.path, "n", len(want), int(res.n_records),
                      "end", end, int(res.end_state), "status", status, int(res.last_status), flush=True)
        fresh.close() if hasattr(fresh, "close") else None
    print("seeds", nseeds, "mismatches", bad, "(flags, path) -> count", dict(sorted(paths.items())))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
