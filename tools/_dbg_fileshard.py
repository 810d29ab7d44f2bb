import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fastqandfurious_amd
from fastqandfurious_amd import hip, sharded
from oracle import ffq_oracle
from test_sharded import make_stream, expected
from test_fileshard import run_ranks

stream = make_stream("long-wrapped")
want, err = expected(ffq_oracle, stream)
path = "/dev/shm/dbg_fs.fq"
open(path, "wb").write(stream.tobytes())
world = 3
for big in (False,):
    for decode in (True,):
        def work(rank, ctx, sw):
            sh = sharded.FileShard(ctx, path, rank, world, comm=sw)
            sh.load()
            if big:
                sh._alloc(sh.n_view // 160 + 1024, True, 64 << 20)
            res = sh.scan(decode=decode)
            rows = sh.rows()
            return dict(rows=rows, rounds=int(res.rounds), path=int(res.scan.path), head=int(res.head), lo=sh.lo, hi=sh.hi, base=int(res.record_base), nrows=int(res.n_rows), row_lo=int(res.row_lo), row_hi=int(res.row_hi))
        res = run_ranks(world, work)
        got = np.concatenate([r["rows"] for r in res])
        ok = got.shape == want.shape and (got == want).all()
        print("big", big, "decode", decode, "ok", ok, [(r["rounds"], r["path"], r["head"], r["rows"].shape[0], r["base"], r["row_lo"], r["row_hi"], r["nrows"]) for r in res])
        if not ok:
            base = 0
            for r in res:
                n = r["rows"].shape[0]
                w = want[base:base + n]
                if w.shape == r["rows"].shape:
                    bad = np.nonzero((w != r["rows"]).any(axis=1))[0]
                    print("  rank lo", r["lo"], "bad rows", bad[:5], len(bad))
                    for b in bad[:3]:
                        print("   got", r["rows"][b], "want", w[b])
                else:
                    print("  shape", w.shape, r["rows"].shape)
                base += n
