"""Stress of the single pass with the in-place layout (FFQ_F_SINGLE_PASS, qual_room = INPLACE_STRIDE): four-line records
with LONG lines -- reads of hundreds to tens of thousands of bases, quality lines that begin with '@' / '+' or consist of
base letters only (false evidence for the long-line guess of k_scan_ident: the pass must be refused, not wrong), a few
random edits, cut ends -- GPU scan + decode vs oracle.   tools/stress_long4.py [seeds]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip
from oracle import ffq_oracle as oracle
import test_gpu_parity as T
from stress_fast4 import same_quals

QA = np.frombuffer(bytes(range(33, 127)), dtype=np.uint8)
BASES = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)


def long_records(rng, n, lo, hi, base_quals=0.0, logu=False):
    parts = []
    for i in range(n):
        L = int(np.exp(rng.uniform(np.log(lo), np.log(hi)))) if logu else int(rng.integers(lo, hi + 1))
        h = b"r%d" % i + b" " + b"x" * int(rng.integers(0, 60))
        seq = rng.choice(BASES[:5] if rng.random() < 0.9 else BASES, size=L).tobytes()
        qual = rng.choice(BASES if rng.random() < base_quals else QA, size=L).tobytes()
        parts.append(b"@" + h + b"\n" + seq + b"\n+" + (h if rng.random() < 0.2 else b"") + b"\n" + qual + b"\n")
    return b"".join(parts)


def main():
    ctx = hip.default_context(0)
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad = 0
    paths = {}
    for seed in range(nseeds):
        rng = np.random.default_rng(77000 + int(os.environ.get("FFQ_STRESS_SEED0", "0")) + seed)
        kind = seed % 5
        if kind == 0: data = long_records(rng, int(rng.integers(100, 2000)), 600, 3000)
        elif kind == 1: data = long_records(rng, int(rng.integers(20, 200)), 5000, 40000)
        elif kind == 2: data = long_records(rng, int(rng.integers(50, 600)), 50, 60000, logu=True)
        elif kind == 3: data = long_records(rng, int(rng.integers(20, 120)), 15000, 30000, base_quals=0.2)
        else: data = long_records(rng, int(rng.integers(500, 4000)), 300, 700)
        nedits = (0, 0, 1, 3)[(seed // 5) % 4]
        data = T.mutate(rng, data, nedits)
        if seed % 3 == 0:
            data = data[:len(data) - int(rng.integers(1, 3000))]
        for kw in (dict(), dict(eof=False), dict(sentinel=False, offset=5)):
            ctx.forget()
            want, end, status, off = oracle.scan(data, **kw)
            wq, wqoff = oracle.decode_quals(data, want)
            for attempt in range(2):       # (the second: what the context remembers)
                table, res, qual, qoff = ctx.scan_host(data, flags=hip.F_DECODE_QUAL | hip.F_SINGLE_PASS, qual_room=hip.INPLACE_STRIDE,
                                                       table_cap=len(want) + 8, **kw)
                ok = (table.shape == want.shape and (table == want).all() and int(res.end_state) == end and
                      int(res.last_status) == status and int(res.end_offset) == off and same_quals(res, want, qual, qoff, wq, wqoff))
                key = (kind, nedits, int(res.path))
                paths[key] = paths.get(key, 0) + 1
                if not ok:
                    bad += 1
                    print("MISMATCH seed", seed, kw, "kind", kind, "edits", nedits, "path", res.path, "n", len(want), int(res.n_records), flush=True)
    print("seeds", nseeds, "mismatches", bad, "paths (kind, edits, path) -> count", dict(sorted(paths.items())))


if __name__ == "__main__":
    main()
