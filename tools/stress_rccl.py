"""Differential run of the DEVICE step over RCCL WITH REAL PEERS (round 6): N processes on one GPU that claim different hosts
(NCCL_HOSTID per rank -> RCCL's duplicate-GPU check lets them into one communicator, socket transport), under
torch.distributed.run.  Every seed: a stream (S-single, S-wrapped, long wrapped records, hostile messes, truncations -- the same
bytes on every rank, seeded), random 16-byte aligned cut points, random halos of 16 ... 8192 bytes; the step with hand-offs
between the ranks (ncclSend / ncclRecv; look-aheads that grow, entries that are re-entered: repair rounds), pipelined or serial,
with or without the decode, and the same stream as a FILE (any cut points; resident or slabs).  Rank 0 gathers every rank's rows
(gloo side group) and compares with the oracle's scan of the whole stream; stream errors must be raised alike on every rank.
   FFQ_TEST_RANKS_ON_ONE_GPU=1 python -m torch.distributed.run --nproc-per-node N tools/stress_rccl.py [seeds]"""
import os, sys
ONE_GPU = os.environ.get("FFQ_TEST_RANKS_ON_ONE_GPU") == "1"
if ONE_GPU:
    os.environ["NCCL_HOSTID"] = "ffq-rank-as-host-%s" % os.environ.get("RANK", "0")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import datetime, pickle
import numpy as np, torch, torch.distributed as dist
import fastqandfurious_amd  # noqa: F401
from fastqandfurious_amd import hip, sharded, synth
import test_gpu_parity as T
from test_sharded import expected

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = 0 if ONE_GPU else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group(backend="nccl", device_id=dev)
ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SEED0 = int(os.environ.get("FFQ_STRESS_SEED0", "0"))
ctx = hip.Context(local)
if rank == 0:
    from oracle import ffq_oracle as oracle
path = "/dev/shm/ffq_stress_rccl.%s.fq" % os.environ.get("MASTER_PORT", "0")
bad, tally = 0, {}


def gather_obj(obj):
    out = [None] * world if rank == 0 else None
    dist.gather_object(obj, out, dst=0, group=ctl)
    return out


for seed in range(SEED0, SEED0 + nseeds):
    rng = np.random.default_rng(777000 + seed)
    kind = seed % 6
    if kind == 0:
        data = synth.single(int(rng.integers(0, 1000)), int(rng.integers(3000, 20000)), seed=42).tobytes()
    elif kind == 1:
        data = synth.wrapped(int(rng.integers(0, 1000)), int(rng.integers(3000, 15000)), seed=43)[0].tobytes()
    elif kind == 2:
        data = T.random_records(rng, 50, 5000, 60000, wrap=int(rng.integers(50, 100)))
    elif kind == 3:
        data = T._mess(rng, 6000, fatal=False)
    elif kind == 4:
        data = T._mess(rng, 6000, fatal=True)
    else:
        data = T.random_records(rng, 8000, 1, 40, hdr_hi=5)
    if rng.random() < 0.3:
        data = data[:-int(rng.integers(1, 400))]
    a = np.frombuffer(data, dtype=np.uint8)
    n = int(a.size)
    cuts = sorted(int(x) // 16 * 16 for x in rng.integers(0, n + 1, world - 1))
    bounds = [0] + cuts + [n]
    tail, head = int(rng.integers(16, 8193)), int(rng.integers(16, 8193))
    mode = ("pipelined", "serial")[int(rng.integers(0, 2))]
    decode = bool(rng.integers(0, 2))
    if rank == 0:
        with open(path, "wb") as fh:
            fh.write(data)
        want, err = expected(oracle, a)
    dist.barrier(group=ctl)
    lo, hi = bounds[rank], bounds[rank + 1]
    # ---- (1) the step with hand-offs between the ranks --------------------------------------------------------------------
    got = None
    try:
        sc = sharded.NativeShardScanner(ctx, bounds, rank, world, tail_bytes=tail, head_bytes=head, unique_id=sharded.native_unique_id(dist, dev, ctl),
                                        serial=(mode == "serial"))
        t_, h_ = sc.halo()
        ext = torch.zeros(t_ + (hi - lo) + h_ + 64, dtype=torch.uint8, device=dev)
        ext[t_:t_ + hi - lo] = torch.from_numpy(a[lo:hi].copy()).to(dev)
        table = torch.empty((n // 20 + 64, 6), dtype=torch.int64, device=dev)
        qual = torch.empty(ext.numel() + (1 << 20), dtype=torch.int8, device=dev) if decode else None
        qoff = torch.empty(table.shape[0] + 1, dtype=torch.int64, device=dev) if decode else None
        torch.cuda.synchronize()
        ctx.reserve(ext.numel())
        out = sc.scan(ext, t_, h_, table, hip.F_DECODE_QUAL if decode else 0, qual, qoff)
        rows = table[out.row_lo:out.row_hi].cpu().numpy()
        q = None
        if decode and out.row_hi > out.row_lo:
            qo = qoff[out.row_lo:out.row_hi + 1].cpu().numpy()
            q = qual[int(qo[0]):int(qo[-1])].cpu().numpy()
        got = ("ok", rows, out.record_base, out.total_records, out.rounds, q)
        sc.close()
    except ValueError as e:
        got = ("err", str(e))
        sc.close()
    res1 = gather_obj(got)
    # ---- (2) the same stream as a FILE: any cut points, resident or through slabs ----------------------------------------------
    cuts2 = sorted(int(x) for x in rng.integers(0, n + 1, world - 1))
    bounds2 = [0] + cuts2 + [n]
    slab = int(rng.integers(1 << 16, 1 << 20)) if rng.random() < 0.5 else None
    try:
        sh = sharded.FileShard(ctx, path, rank, world, comm=sharded.native_unique_id(dist, dev, ctl), bounds=bounds2, tail_bytes=tail, head_bytes=head, slab_bytes=slab)
        try:
            r = sh.scan()
            got = ("ok", sh.rows(), int(r.record_base), int(r.total_records), int(r.rounds), None)
        finally:
            sh.close()
    except ValueError as e:
        got = ("err", str(e))
    res2 = gather_obj(got)
    if rank == 0:
        for name, res in (("step-%s%s" % (mode, "+decode" if decode else ""), res1), ("file%s" % ("-slabs" if slab else ""), res2)):
            ok = True
            if err is not None:
                ok = all(r[0] == "err" and r[1] == err for r in res)
            elif any(r[0] != "ok" for r in res):
                ok = False
            else:
                rows = np.concatenate([r[1] for r in res])
                ok = rows.shape == want.shape and bool((rows == want).all()) and all(r[3] == len(want) for r in res) and \
                    [r[2] for r in res] == [sum(len(x[1]) for x in res[:k]) for k in range(world)]
                if ok and name.startswith("step") and decode:
                    wq, _ = oracle.decode_quals(a, want)
                    qs = [r[5] for r in res if r[5] is not None]
                    ok = bool((np.concatenate(qs) == wq).all()) if qs else wq.size == 0
            key = (kind, name.split("+")[0], "err" if err else "ok")
            tally[key] = tally.get(key, 0) + 1
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, "kind", kind, name, "world", world, "halos", tail, head, "slab", slab, "err", err, [r[:2] if r[0] == "err" else (r[0], len(r[1]), r[2], r[3]) for r in res], flush=True)
dist.barrier(group=ctl)
if rank == 0:
    try:
        os.unlink(path)
    except OSError:
        pass
    print("world", world, "seeds", nseeds, "runs", 2 * nseeds, "mismatches", bad, dict(sorted(tally.items())), flush=True)
dist.barrier()
dist.destroy_process_group()
