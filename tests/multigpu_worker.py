"""Worker of tests/test_multigpu.py: one process per GPU under torch.distributed.run, backend nccl (= RCCL).

Every rank builds the same stream (the generators are counter-based), takes its byte range, and runs the library's own
step (ffq_shard_*: halo hand-off by ncclSend / ncclRecv between DIFFERENT ranks, scan, one ncclAllGather of the eight
words) -- plain, pipelined over two lanes with the hand-off on its own stream, with and without the decode, the same in
SERIAL mode (one communicator, one stream), and once through the WATCHDOG: one rank's gather is stalled, every rank's step
comes back with hip.FFQTimeout naming the stage, the communicators are aborted, a new one is built for the serial step and
the step taken again -- and the
file-backed form (every rank preads its range of one file; only the gather is RCCL).  Each rank leaves its rows in the
scratch directory; rank 0 puts them together and compares with the oracle's scan of the whole stream, and with what k
logical ranks in ONE process (the in-process transport the single-GPU tests use) give for the same ranges: same rows,
same repair rounds.  The invariant is the reference's own: results do not depend on how the stream is cut
(/root/reference/tests.py:219-226)."""
import json
import os
import sys

# FFQ_TEST_RANKS_ON_ONE_GPU=1: every rank on GPU 0.  RCCL refuses two ranks of one communicator on the same device of the same
# HOST ("Duplicate GPU detected"); ranks that say they sit on different hosts (NCCL_HOSTID) pass, and talk over the socket
# transport (loopback) instead of xGMI.  Nothing of that is fast and nothing of it is xGMI -- but it IS librccl with real
# peers: communicators of N ranks, ncclSend / ncclRecv between different ranks, the all-gather, two communicators driven
# from two streams at once, a collective whose peer never arrives and ncclCommAbort on it -- on a box with one GPU.
ONE_GPU = os.environ.get("FFQ_TEST_RANKS_ON_ONE_GPU") == "1"
if ONE_GPU:
    os.environ["NCCL_HOSTID"] = "ffq-rank-as-host-%s" % os.environ.get("RANK", "0")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist

import fastqandfurious_amd  # noqa: F401
from fastqandfurious_amd import fastqandfurious as F, hip, sharded
from test_sharded import bounds_for, expected, make_stream

KINDS = ("single", "wrapped", "tricky", "long-wrapped", "long", "small")


def peers_ok(info, world):
    """sharded.check_peers -- or, every rank on ONE GPU on purpose: the rank counts, and the bus ids all the same."""
    if not ONE_GPU:
        return sharded.check_peers(info, world)
    assert info["nranks_handoff"] == world and len(set(info["bus_ids"])) == 1 and info["bus_ids"][0], info
    if world > 1:
        try:
            sharded.check_peers(info, world)
            raise AssertionError("check_peers let %d ranks on one GPU pass" % world)
        except RuntimeError as e:
            assert "share" in str(e)


def main(scratch):
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if ONE_GPU:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend="nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == world
    # (what bench.py does at N > 1: a gloo side group for the communicator ids -- it still works when the GPUs' fabric has just
    # swallowed a collective, which is when a NEW id is needed)
    import datetime
    ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
    assert dist.get_backend(ctl) == "gloo"
    ctx = hip.Context(local)
    lane_ctx = hip.Context(share=ctx)
    report = {}
    for kind in KINDS:
        stream = make_stream(kind)
        t = torch.from_numpy(stream.copy()).to(dev)
        for origin, shift in ((0, 0), (5 * (1 << 32) + 123457, 48)):
            bounds = bounds_for(stream.size, world, origin, shift)
            lo, hi = bounds[rank], bounds[rank + 1]
            sc = sharded.NativeShardScanner(ctx, bounds, rank, world, unique_id=sharded.native_unique_id(dist, dev))
            assert sc.sh.transport() == "rccl"
            info = sc.info()
            peers_ok(info, world)                     # the communicators count `world` ranks, on distinct GPUs
            assert info["nranks_handoff"] == world and info["nranks_gather"] == world and info["mode"] == "pipelined"
            assert len(info["bus_ids"]) == world and all(b is not None for b in info["bus_ids"]), info
            lanes = [sc, sc.lane(lane_ctx)]
            tail, head = sc.halo()
            n_rows = stream.size // 40 + 64
            exts, tabs = [], []
            for _ in lanes:
                e = torch.zeros(tail + (hi - lo) + head + 64, dtype=torch.uint8, device=dev)
                e[tail:tail + hi - lo] = t[lo - origin:hi - origin]          # own bytes only: the halos must come from the peers
                exts.append(e)
                tabs.append(torch.empty((n_rows, 6), dtype=torch.int64, device=dev))
            qual = torch.empty(exts[0].numel() + (8 << 20), dtype=torch.int8, device=dev)
            qoff = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            ctx.reserve(exts[0].numel())
            lane_ctx.reserve(exts[0].numel())
            outs = []
            # plain step, then the decode, then three pipelined steps over the two lanes (hand-off of step i + 1 beside
            # the scan of step i)
            outs.append(("plain", 0, sc.scan(exts[0], tail, head, tabs[0])))
            for e in exts:                                                    # (wipe the halos again: each step must fetch them)
                e[:tail].zero_()
                e[tail + hi - lo:].zero_()
            torch.cuda.synchronize()
            outs.append(("decode", 0, sc.scan(exts[0], tail, head, tabs[0], hip.F_DECODE_QUAL, qual, qoff)))
            dq = None
            o = outs[-1][2]
            if o.row_hi > o.row_lo:
                qo = qoff[o.row_lo:o.row_hi + 1].cpu().numpy()
                dq = (qual[int(qo[0]):int(qo[-1])].cpu().numpy(), qo - qo[0])
            for e in exts:
                e[:tail].zero_()
                e[tail + hi - lo:].zero_()
            torch.cuda.synchronize()
            lanes[0].submit(exts[0], tail, head, tabs[0], overlap=True)
            for i in range(1, 3):
                lanes[i & 1].submit(exts[i & 1], tail, head, tabs[i & 1], overlap=True)
                outs.append(("lane", (i - 1) & 1, lanes[(i - 1) & 1].finish()))
            outs.append(("lane", 0, lanes[0].finish()))
            assert all(o.comm["mode"] == "pipelined" and o.comm["nranks"] == world for _n, _l, o in outs)
            # the same in SERIAL mode: ONE communicator, ONE stream (hand-off, scan, words, gather in order) -- same rows, same rounds
            sc.sh.set_serial(True)
            for e in exts:
                e[:tail].zero_()
                e[tail + hi - lo:].zero_()
            torch.cuda.synchronize()
            outs.append(("serial", 0, sc.scan(exts[0], tail, head, tabs[0])))
            for e in exts:
                e[:tail].zero_()
                e[tail + hi - lo:].zero_()
            torch.cuda.synchronize()
            lanes[0].submit(exts[0], tail, head, tabs[0], overlap=True)
            lanes[1].submit(exts[1], tail, head, tabs[1], overlap=True)
            outs.append(("serial-lane", 0, lanes[0].finish()))
            outs.append(("serial-lane", 1, lanes[1].finish()))
            assert all(o.comm["mode"] == "serial" and o.comm["nranks"] == world for _n, _l, o in outs[-3:])
            sc.sh.set_serial(False)
            key = "%s@%d" % (kind, origin)
            rep = report[key] = {"steps": []}
            for name, li, o in outs:
                rows = (tabs[li] if name.endswith("lane") else tabs[0])[o.row_lo:o.row_hi].cpu().numpy()
                rep["steps"].append({"name": name, "base": o.record_base, "total": o.total_records, "rounds": o.rounds,
                                     "handoff_bytes": o.comm["handoff_bytes"], "n": int(rows.shape[0])})
                np.save(os.path.join(scratch, "rows_%s_%s%d_%d.npy" % (key, name, len(rep["steps"]), rank)), rows)
                # the view the rows refer to is the stream's bytes: the hand-off delivered them
                got = o.ext[:o.tail + hi - lo + o.head].cpu().numpy()
                assert (got == stream[lo - o.tail - origin:hi + o.head - origin]).all(), "%s rank %d: halo bytes differ" % (key, rank)
            if dq is not None:
                np.save(os.path.join(scratch, "qual_%s_%d.npy" % (key, rank)), dq[0])
            for ln in reversed(lanes):
                ln.close()
            dist.barrier()
    # ---- the watchdog and the recovery, with real peers: the LAST rank's gather stalls (a kernel that waits for a host flag
    # in front of its all-gather), so no rank's all-gather completes; every rank's step must come back with FFQTimeout at
    # stage 'gather' within the deadline, abort its communicators, join a NEW one -- ONE, serial mode -- and get the rows
    stream = make_stream("wrapped")
    t = torch.from_numpy(stream.copy()).to(dev)
    bounds = bounds_for(stream.size, world, 0, 48)
    lo, hi = bounds[rank], bounds[rank + 1]
    sc = sharded.NativeShardScanner(ctx, bounds, rank, world, unique_id=sharded.native_unique_id(dist, dev))
    sc.sh.set_timeout(4.0)
    tail, head = sc.halo()
    ext = torch.zeros(tail + (hi - lo) + head + 64, dtype=torch.uint8, device=dev)
    ext[tail:tail + hi - lo] = t[lo:hi]
    tab = torch.empty((stream.size // 40 + 64, 6), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    if rank == world - 1:
        sc.sh.inject_stall(hip.STAGE_GATHER, 60.0)
    import time
    t0 = time.perf_counter()
    try:
        sc.scan(ext, tail, head, tab)
        raise AssertionError("rank %d: the stalled step came back" % rank)
    except hip.FFQTimeout as e:
        waited = time.perf_counter() - t0
        assert "stage 'gather'" in str(e) and "transport rccl" in str(e), str(e)
        assert 3.5 < waited < 30, waited
        report["watchdog"] = {"message": str(e), "waited_s": waited, "stage": sc.info()["last_stage"]}
    # (the ranks meet over gloo, THEN abort: a rank that aborts after its peers have torn their ends down waits in RCCL)
    assert sharded.abort_together(dist, ctl, [sc]), "rank %d: the abort did not complete: %s" % (rank, hip.last_error())
    sc.close()
    ext[:tail].zero_()
    ext[tail + hi - lo:].zero_()
    torch.cuda.synchronize()
    sc = sharded.NativeShardScanner(ctx, bounds, rank, world, unique_id=sharded.native_unique_id(dist, dev, ctl), serial=True)      # (the new id over gloo)
    info = sc.info()
    peers_ok(info, world)
    assert info["mode"] == "serial" and info["nranks_gather"] == 0 and info["nranks_handoff"] == world
    o = sc.scan(ext, tail, head, tab)
    assert o.comm["mode"] == "serial" and o.comm["nranks"] == world
    np.save(os.path.join(scratch, "rows_recovered_%d.npy" % rank), tab[o.row_lo:o.row_hi].cpu().numpy())
    sc.close()
    dist.barrier()
    # ---- one FILE read by all ranks: every rank preads its range, only the eight words travel ----------------------
    fpath = os.path.join(scratch, "shared.fq")
    fstream = make_stream("wrapped")
    if rank == 0:
        with open(fpath, "wb") as fh:
            fh.write(fstream.tobytes())
    dist.barrier()
    # (the third form: the range through SLABS of 256 KiB -- ffq_shard_scan_fd_slabs, a range that would not fit the GPU --,
    # its eight words over the same RCCL gather)
    for kw in ({}, dict(tail_bytes=200, head_bytes=64), dict(slab_bytes=1 << 18, tail_bytes=300, head_bytes=100)):
        it = F.readfastq_iter_range(fpath, rank, world, F.entryfunc_abspos, ctx=ctx, comm=sharded.native_unique_id(dist, dev), **kw)
        assert it.comm["transport"] == "rccl" and it.comm["halo_source"] == "file"
        rows = np.array([list(p) for p in it], dtype=np.int64).reshape(-1, 6)
        np.save(os.path.join(scratch, "file_rows_%d_%d.npy" % (len(kw), rank)), rows)
        report["file%d" % len(kw)] = {"base": it.record_base, "total": it.total_records, "n": it.n_records}
    # ---- FileShard's own recovery: the ranks found each other through torch.distributed, the last rank's gather stalls, every
    # rank's scan() trips, aborts, draws a new id over the side group and takes the serial step by itself
    # (a world of one has no communicator to replace: its FileShard is the in-process one)
    fs = sharded.FileShard(ctx, fpath, rank, world, group=ctl)
    if world > 1:
        fs.sh.set_timeout(4.0)
        if rank == world - 1:
            fs.sh.inject_stall(hip.STAGE_GATHER, 60.0)
    res = fs.scan()
    if world > 1:
        assert fs.recovered and "stage 'gather'" in fs.recovered and fs.sh.info()["mode"] == "serial", fs.recovered
    np.save(os.path.join(scratch, "file_rows_recovered_%d.npy" % rank), fs.rows())
    fs.close()
    dist.barrier()
    # ---- the same stream as a BGZF file read by ranges (sharded.BgzfFileShard): every rank inflates the members that begin in its
    # share of the COMPRESSED file, the sizes go round over the side group, the step hands the halos over RCCL
    zpath = os.path.join(scratch, "shared.fq.gz")
    if rank == 0:
        from fastqandfurious_amd import bgzf
        with open(zpath + ".tmp", "wb") as fh:
            fh.write(bgzf.compress(fstream.tobytes(), block_bytes=30000, level=1))
        os.rename(zpath + ".tmp", zpath)
    dist.barrier()
    for kw in ({}, dict(tail_bytes=200, head_bytes=64)):
        bz = sharded.BgzfFileShard(ctx, zpath, rank, world, group=ctl, **kw)
        try:
            bz.load()
            if kw and world > 1:
                # (and the shard's own recovery over inflated ranges: the last rank's gather stalls, every rank trips, the ranks meet,
                # abort, join one new communicator, inflate and load again, take the serial step)
                bz.sh.set_timeout(4.0)
                if rank == world - 1:
                    bz.sh.inject_stall(hip.STAGE_GATHER, 60.0)
            res = bz.scan(decode=True)
            if kw and world > 1:
                assert bz.recovered and "stage 'gather'" in bz.recovered and bz.sh.info()["mode"] == "serial", bz.recovered
            assert bz.sh.transport() == ("rccl" if world > 1 else "in-process") and int(res.halo_source) == 0
            rows = bz.rows()
            np.save(os.path.join(scratch, "bgzf_rows_%d_%d.npy" % (len(kw), rank)), rows)
            if rows.shape[0]:
                q, qo = bz.quals(0, rows.shape[0], rows)
                ln = rows[:, 5] - rows[:, 4]
                ix = np.repeat(qo[:len(rows)], ln) + (np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln))
                np.save(os.path.join(scratch, "bgzf_qual_%d_%d.npy" % (len(kw), rank)), q[ix])
            report["bgzf%d" % len(kw)] = {"base": int(res.record_base), "total": int(res.total_records), "n": int(rows.shape[0]),
                                          "bounds": [int(b) for b in bz.bounds]}
        finally:
            bz.close()
        dist.barrier()
    with open(os.path.join(scratch, "report_%d.json" % rank), "w") as fh:
        json.dump(report, fh)
    dist.barrier()
    if rank == 0:
        check(scratch, world)
        print("multi-gpu shards ok: world %d" % world, flush=True)
    dist.barrier()
    dist.destroy_process_group()


def check(scratch, world):
    from oracle import ffq_oracle as oracle
    from test_sharded import _hip_backends, run_local
    reports = [json.load(open(os.path.join(scratch, "report_%d.json" % r))) for r in range(world)]
    dev = torch.device("cuda", 0 if ONE_GPU else int(os.environ.get("LOCAL_RANK", "0")))
    for kind in KINDS:
        stream = make_stream(kind)
        want0, err = expected(oracle, stream)
        assert err is None
        wq, wqoff = oracle.decode_quals(stream, want0)
        for origin, shift in ((0, 0), (5 * (1 << 32) + 123457, 48)):
            key = "%s@%d" % (kind, origin)
            want = want0 + origin
            bounds = bounds_for(stream.size, world, origin, shift)
            # the same ranges as k logical ranks of ONE process (in-process transport): the rounds must agree
            make, made = _hip_backends(None)
            local = run_local(torch.from_numpy(stream.copy()).to(dev), bounds, make, native=True)
            for c in made.values():
                c.close()
            nsteps = len(reports[0][key]["steps"])
            for si in range(nsteps):
                name = reports[0][key]["steps"][si]["name"]
                parts = [np.load(os.path.join(scratch, "rows_%s_%s%d_%d.npy" % (key, name, si + 1, r))) for r in range(world)]
                got = np.concatenate(parts)
                assert got.shape == want.shape and (got == want).all(), "%s step %d (%s): rows over the ranks differ from the oracle's" % (key, si, name)
                base = 0
                for r in range(world):
                    st = reports[r][key]["steps"][si]
                    assert st["base"] == base and st["total"] == len(want) and st["n"] == parts[r].shape[0]
                    assert st["rounds"] == local[r][0].rounds, "%s step %d rank %d: %d repair rounds over RCCL, %d in process" % (key, si, r, st["rounds"], local[r][0].rounds)
                    if world > 1 and stream.size > 64 * world:
                        assert st["handoff_bytes"] > 0, "%s rank %d: no bytes were handed off" % (key, r)
                    base += parts[r].shape[0]
            qs = [np.load(os.path.join(scratch, "qual_%s_%d.npy" % (key, r))) for r in range(world)
                  if os.path.exists(os.path.join(scratch, "qual_%s_%d.npy" % (key, r)))]
            assert (np.concatenate(qs) == wq).all(), "%s: decoded qualities over the ranks differ from the oracle's" % key
    fwant, _ = expected(oracle, make_stream("wrapped"))
    got = np.concatenate([np.load(os.path.join(scratch, "rows_recovered_%d.npy" % r)) for r in range(world)])
    assert got.shape == fwant.shape and (got == fwant).all(), "after the watchdog trip: the serial step's rows differ from the oracle's"
    assert all(rep["watchdog"]["stage"] == "gather" for rep in reports)
    got = np.concatenate([np.load(os.path.join(scratch, "file_rows_recovered_%d.npy" % r)) for r in range(world)])
    assert got.shape == fwant.shape and (got == fwant).all(), "FileShard after its own recovery: rows over the ranks differ from the oracle's"
    fq, _ = oracle.decode_quals(make_stream("wrapped"), fwant)
    for k in (0, 2):
        got = np.concatenate([np.load(os.path.join(scratch, "bgzf_rows_%d_%d.npy" % (k, r))) for r in range(world)])
        assert got.shape == fwant.shape and (got == fwant).all(), "BGZF ranges: rows over the ranks differ from the oracle's scan of the inflated stream"
        qs = [np.load(os.path.join(scratch, "bgzf_qual_%d_%d.npy" % (k, r))) for r in range(world)
              if os.path.exists(os.path.join(scratch, "bgzf_qual_%d_%d.npy" % (k, r)))]
        assert (np.concatenate(qs) == fq).all(), "BGZF ranges: decoded qualities differ from the oracle's"
        assert reports[0]["bgzf%d" % k]["bounds"][0] == 0 and reports[0]["bgzf%d" % k]["bounds"][-1] == make_stream("wrapped").size
        assert [reports[r]["bgzf%d" % k]["base"] for r in range(world)] == [sum(reports[q]["bgzf%d" % k]["n"] for q in range(r)) for r in range(world)]
    for k in (0, 2, 3):
        got = np.concatenate([np.load(os.path.join(scratch, "file_rows_%d_%d.npy" % (k, r))) for r in range(world)])
        assert got.shape == fwant.shape and (got == fwant).all(), "file-backed ranges: rows over the ranks differ from the oracle's"
        assert [reports[r]["file%d" % k]["base"] for r in range(world)] == \
            [sum(reports[q]["file%d" % k]["n"] for q in range(r)) for r in range(world)]


if __name__ == "__main__":
    main(sys.argv[1])
