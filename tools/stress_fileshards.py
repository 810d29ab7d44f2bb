"""Differential run of the file-backed byte-range shards (round 6): seeded streams (S-single, S-wrapped, long wrapped records,
hostile messes, truncations), a random world of 1 ... 8 logical ranks with RANDOM cut points (any byte), random halos of
1 ... 4096 bytes, through (a) the resident step (pipelined or serial), (b) slabs of 64 KiB ... 1 MiB, (c) the same bytes as a BGZF
file with members of 200 ... 65000 bytes read by ranges (sharded.BgzfFileShard: the cut points are the members'); the rows
concatenated over the ranks, the ordinals and the stream's error against the oracle's scan of the whole file.
   tools/stress_fileshards.py [seeds]      FFQ_STRESS_SEED0=n"""
import os, sys, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import fastqandfurious_amd  # noqa: F401
from fastqandfurious_amd import bgzf, hip, sharded, synth
from oracle import ffq_oracle as oracle
import test_gpu_parity as T
from test_sharded import expected

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SEED0 = int(os.environ.get("FFQ_STRESS_SEED0", "0"))
path = "/dev/shm/ffq_stress_fileshards.%d.fq" % os.getpid()
bad, modes = 0, {}
for seed in range(SEED0, SEED0 + nseeds):
    rng = np.random.default_rng(424200 + seed)
    kind = seed % 6
    if kind == 0:
        data = synth.single(int(rng.integers(0, 1000)), int(rng.integers(2000, 20000)), seed=42).tobytes()
    elif kind == 1:
        data = synth.wrapped(int(rng.integers(0, 1000)), int(rng.integers(2000, 15000)), seed=43)[0].tobytes()
    elif kind == 2:
        data = T.random_records(rng, 60, 5000, 60000, wrap=int(rng.integers(50, 100)))
    elif kind == 3:
        data = T._mess(rng, 6000, fatal=False)
    elif kind == 4:
        data = T._mess(rng, 6000, fatal=True)
    else:
        data = T.random_records(rng, 8000, 1, 40, hdr_hi=5)
    if rng.random() < 0.3:
        data = data[:-int(rng.integers(1, 400))]
    a = np.frombuffer(data, dtype=np.uint8)
    want, err = expected(oracle, a)
    with open(path, "wb") as fh:
        fh.write(data)
    world = int(rng.integers(1, 9))
    cuts = sorted(int(x) for x in rng.integers(0, a.size + 1, world - 1))
    bounds = [0] + cuts + [int(a.size)]
    tail, head = int(rng.integers(1, 4097)), int(rng.integers(1, 4097))
    zpath = path + ".gz"
    with open(zpath, "wb") as fh:
        fh.write(bgzf.compress(data, block_bytes=int(rng.integers(200, 65001)), level=1, eof_marker=bool(seed & 1)))
    for mode in ("pipelined", "serial", "slabs", "bgzf"):
        slab = int(rng.integers(1 << 16, 1 << 20)) if mode == "slabs" else None
        sw = hip.ShardWorld(world)
        lw = sharded.LocalWorld(world)
        res, errs = [None] * world, [None] * world

        def work(rank):
            ctx = None
            try:
                ctx = hip.Context(0)
                if mode == "bgzf":
                    sh = sharded.BgzfFileShard(ctx, zpath, rank, world, comm=sw, exchange=sharded.LocalTransport(lw, rank).allgather,
                                               tail_bytes=tail, head_bytes=head, threads=2)
                else:
                    sh = sharded.FileShard(ctx, path, rank, world, comm=sw, bounds=bounds, tail_bytes=tail, head_bytes=head, slab_bytes=slab,
                                           serial=True if mode == "serial" else None)
                try:
                    r = sh.scan()
                    res[rank] = (sh.rows(), int(r.record_base), int(r.total_records))
                finally:
                    sh.close()
            except BaseException as e:      # noqa: BLE001
                errs[rank] = e
                sw.abort()
                lw.abort()
            finally:
                if ctx is not None:
                    ctx.close()
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in th: t.start()
        for t in th: t.join()
        sw.close()
        real = [e for e in errs if e is not None and "another logical rank failed" not in str(e) and not isinstance(e, threading.BrokenBarrierError)]
        ok = True
        if err is not None:
            ok = len(real) == world and all(isinstance(e, ValueError) and str(e) == err for e in real)
        elif real:
            ok = False
        else:
            got = np.concatenate([r[0] for r in res])
            ok = got.shape == want.shape and bool((got == want).all()) and all(r[2] == len(want) for r in res) and \
                [r[1] for r in res] == [sum(len(q[0]) for q in res[:k]) for k in range(world)]
        modes[(kind, mode, "err" if err else "ok")] = modes.get((kind, mode, "err" if err else "ok"), 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "kind", kind, mode, "world", world, "bounds", bounds, "halos", tail, head, "slab", slab, "err", err, [str(e)[:120] for e in real][:2], flush=True)
os.unlink(path)
os.unlink(zpath)
print("seeds", nseeds, "runs", 4 * nseeds, "mismatches", bad, dict(sorted(modes.items())))
