"""Byte-range sharding (fastq-and-furious_amd/sharded.py; the protocol itself: csrc/ffq_shard_proto.h).

CPU (`-m "not gpu"`): the HOST step of the library (ffq_shard_host_step: the same protocol functions the device step
runs) -- halo hand-off by owner, row ownership cut, hand-off verification, look-ahead growth for records longer than the
halo, re-entry from the left neighbour's exit, stream errors raised by every rank -- with the CPU oracle as the scan
engine (test infrastructure, handed in as the step's scan callback), over gloo (world 2 and 3, separate processes) and
over the in-process transport (k logical ranks as threads).

GPU (`-m gpu`): the DEVICE step (ffq_shard_step_*) with k in {2, 3, 8} logical ranges of one resident buffer, S-single and
S-wrapped, with and without decode, stream offsets past 2^32 (so that `add` and the cut see the offsets of BASELINE
config 5), records longer than the halo across an edge -- and the host step with the GPU scanning (ffq_scan_host).  The
unit that shards is the record chain of /root/reference/src/fastqandfurious.py:251-279; the reference grows its buffer
until a record fits (:274-279).

Concatenated shard rows must equal the oracle's single-range table, bit for bit."""
import os
import sys
import threading

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_scan(data, sentinel, offset, eof, add, rows, small=False):
    """The host step's scan callback on the CPU oracle: (rc, n_records, end_state, end_offset, last_status, last_pos0)."""
    from oracle import ffq_oracle as o
    t, end, status, off = o.scan(data, sentinel=sentinel, offset=offset, eof=eof, add=add)
    n = len(t)
    if small or n > rows.shape[0]:
        return -5, n, end, off, status, -1              # E_TABLE_FULL, as the HIP engine reports it
    rows[:n] = t
    pos0 = -1
    if end != 0:
        st, pos = o.entrypos(np.concatenate([np.array([10], dtype=np.uint8), data]) if sentinel else data, off, 0)
        pos0 = int(pos[0]) + add if pos[0] >= 0 else -1
    return 0, n, end, off, status, pos0


class HostEngine:
    """What run_local hands a rank: the scan of its host step (None: the GPU, through `ctx`)."""

    def __init__(self, scan=oracle_scan, ctx=None):
        self.scan, self.ctx = scan, ctx


# ---- streams ------------------------------------------------------------------------------------
def _long_record(seq_len, wrap=0):
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[np.arange(seq_len) % 4]
    q = (33 + (np.arange(seq_len) * 7) % 41).astype(np.uint8)
    if wrap:
        def w(a):
            n = (a.size + wrap - 1) // wrap
            out = np.full(a.size + n - 1, 10, dtype=np.uint8)
            idx = np.arange(a.size) + np.arange(a.size) // wrap
            out[idx] = a
            return out
        seq, q = w(seq), w(q)
    return b"@long/1\n" + seq.tobytes() + b"\n+\n" + q.tobytes() + b"\n"


def _fake_fastq_quality(L):
    """A record whose quality block is itself FASTQ-looking text: guesses that start inside it
    follow a chain of records that do not exist."""
    unit = b"@fake\nACGTACGT\n+\nIIIIIIII\n"
    q = (unit * (L // len(unit) + 1))[:L]
    if q.endswith(b"\n"):
        q = q[:-1] + b"I"
    k, m = divmod(L, 81)
    if m == 0:
        k, m = k - 1, 81
    seq = (b"A" * 80 + b"\n") * k + b"C" * m       # same byte length as the quality block
    assert len(seq) == L
    return b"@tricky/1\n" + seq + b"\n+\n" + q + b"\n"


def make_stream(kind):
    import fastqandfurious_amd  # noqa: F401
    from fastqandfurious_amd import synth
    if kind == "single":
        return synth.single(0, 14000, seed=42)
    if kind == "wrapped":
        return synth.wrapped(0, 12000, seed=43)[0]
    if kind == "long":           # a 3 MiB record in the middle: longer than the 1 MiB halo on either side
        a = synth.single(0, 5000, seed=42).tobytes()
        b = synth.single(5000, 5000, seed=42).tobytes()
        return np.frombuffer(a + _long_record(1536 << 10) + b, dtype=np.uint8)
    if kind == "long-wrapped":
        a = synth.wrapped(0, 3000, seed=43)[0].tobytes()
        return np.frombuffer(a + _long_record(1200 << 10, wrap=80) + a, dtype=np.uint8)
    if kind == "tricky":         # guesses inside the 1.5 MiB quality block are wrong
        a = synth.single(0, 4000, seed=42).tobytes()
        b = synth.single(4000, 3000, seed=42).tobytes()
        return np.frombuffer(a + _fake_fastq_quality(1536 << 10) + b, dtype=np.uint8)
    if kind == "small":
        return synth.single(0, 40, seed=42)
    if kind == "truncated":      # 'Incomplete final quality string at byte'
        return synth.single(0, 9000, seed=42)[:-100]
    if kind == "cut-header":     # 'Incomplete entry at byte %i'
        return synth.single(0, 9000, seed=42)[:-315]
    if kind == "invalid":        # 'Entry is invalid at byte %i': a '+' line of the wrong length
        s = synth.single(0, 9000, seed=42).tobytes()
        k = 6000 * 322
        return np.frombuffer(s[:k + 169] + b"+SYN\n" + s[k + 171:], dtype=np.uint8)
    raise KeyError(kind)


def expected(oracle, stream, origin=0):
    from conftest import rows_of  # noqa: F401
    want, end, st, off = oracle.scan(stream)
    err = None
    if end == 2:
        err = "Incomplete final quality string at byte"
    elif end == 3:
        err = "Incomplete entry at byte %i" % (off - 1 + origin)
    elif end == 4:
        err = "Entry is invalid at byte %i" % (off - 1 + origin)
    return want + origin, err


# ---- k logical ranks as threads of one process ---------------------------------------------------
def run_local(stream_t, bounds, make_backend, tail_bytes=None, head_bytes=None, flags=0, lanes=False,
              table_rows=None, decode=False, native=False, serial=None):
    """Every rank: ext = [zeros | own bytes | zeros] -> one step.  native: the device step (NativeShardScanner over
    hip.ShardWorld, `stream_t` on the GPU); else the host step (HostShardScanner over the thread transport, buffers in
    host memory, the engine's scan).  Returns the list of (ScanOutput, table, qual, qoff) per rank, or raises what the
    ranks raised (all the same)."""
    from fastqandfurious_amd import sharded
    world = len(bounds) - 1
    lw = sharded.LocalWorld(world)
    kw = {}
    if tail_bytes is not None:
        kw = dict(tail_bytes=tail_bytes, head_bytes=head_bytes)
    results, errors = [None] * world, [None] * world
    origin = bounds[0]
    n_rows = table_rows or (stream_t.numel() // 40 + 64)
    host_bytes = None if native else stream_t.cpu().numpy()

    def work(rank):
        try:
            eng = make_backend(rank)
            lo, hi = bounds[rank], bounds[rank + 1]
            qual = qoff = None
            if native:
                sc = sharded.NativeShardScanner(eng.ctx, bounds, rank, world, local_world=lw.native_world(), serial=serial, **kw)
                tail, head = sc.halo()
                ext = torch.zeros(tail + (hi - lo) + head + 64, dtype=torch.uint8, device=stream_t.device)
                ext[tail:tail + hi - lo] = stream_t[lo - origin:hi - origin]
                torch.cuda.synchronize()
                table = torch.empty((n_rows, 6), dtype=torch.int64, device=stream_t.device)
                if decode:
                    qual = torch.empty(ext.numel() + (8 << 20), dtype=torch.int8, device=stream_t.device)
                    qoff = torch.empty(n_rows + 1, dtype=torch.int64, device=stream_t.device)
                args = (ext, tail, head, table, flags, qual, qoff)
            else:
                assert not decode
                sc = sharded.HostShardScanner(lw.transport(rank), bounds, ctx=eng.ctx, scan=eng.scan, **kw)
                tail, head = sc.halo()
                ext = np.zeros(tail + (hi - lo) + head + 64, dtype=np.uint8)
                ext[tail:tail + hi - lo] = host_bytes[lo - origin:hi - origin]
                table = np.empty((n_rows, 6), dtype=np.int64)
                args = (ext, tail, head, table)
            if lanes:
                sc.submit(*args)
                out = sc.finish()
            else:
                out = sc.scan(*args)
            # the view the rows refer to is the stream's bytes: the hand-off delivered them
            got = out.ext[:out.tail + hi - lo + out.head]
            got = got.cpu().numpy() if native else np.asarray(got)
            ref = (stream_t[lo - out.tail - origin:hi + out.head - origin].cpu().numpy() if native
                   else host_bytes[lo - out.tail - origin:hi + out.head - origin])
            assert (got == ref).all(), "halo bytes differ"
            results[rank] = (out, table, qual, qoff)
        except BaseException as e:   # noqa: BLE001
            errors[rank] = e
            lw.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)
            and "another logical rank failed" not in str(e)]
    if real:
        if all(isinstance(e, (ValueError, RuntimeError)) for e in real) and len(real) == world:
            assert len({str(e) for e in real}) == 1, "ranks disagree on the error: %r" % real
        raise real[0]
    assert not any(errors), errors
    return results


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def check_rows(results, bounds, want):
    world = len(bounds) - 1
    parts, base = [], 0
    for r in range(world):
        out, table = results[r][0], results[r][1]
        rows = _np(table[out.row_lo:out.row_hi])
        mine = want[(want[:, 0] >= bounds[r]) & (want[:, 0] < bounds[r + 1])]
        assert rows.shape == mine.shape and (rows == mine).all(), "rank %d: shard rows differ from the single-range table" % r
        assert out.record_base == base and out.total_records == len(want)
        base += len(mine)
        parts.append(rows)
    got = np.concatenate(parts) if parts else np.zeros((0, 6), np.int64)
    assert got.shape == want.shape and (got == want).all()


def bounds_for(total, world, origin=0, shift=0):
    from fastqandfurious_amd import sharded
    b = sharded.shard_bounds(total, world)
    b = [b[0]] + [min(max(x + shift, 0), total) // 16 * 16 for x in b[1:-1]] + [b[-1]]
    return [x + origin for x in b]


LOCAL_CASES = [
    ("single", 2, {}), ("single", 3, {}), ("single", 8, {}),
    ("wrapped", 2, {}), ("wrapped", 3, {}), ("wrapped", 8, {}),
    # halos shorter than a record: every edge grows its look-ahead; run-ins too short to synchronise
    ("single", 3, dict(tail_bytes=64, head_bytes=48)),
    ("wrapped", 8, dict(tail_bytes=256, head_bytes=64)),
    ("wrapped", 3, dict(tail_bytes=1, head_bytes=16)),
    # a 3 MiB record across an edge with the product's 1 MiB halos
    ("long", 2, {}), ("long", 3, {}), ("long", 8, {}), ("long-wrapped", 2, {}), ("long-wrapped", 8, {}),
    # ranges that start inside a quality block made of FASTQ-looking text: wrong guesses, re-entry
    ("tricky", 2, {}), ("tricky", 3, {}), ("tricky", 8, {}),
    # shards smaller than the halo: the halo comes from several ranks
    ("small", 8, {}), ("small", 5, dict(tail_bytes=700, head_bytes=900)),
]


@pytest.mark.parametrize("kind,world,kw", LOCAL_CASES)
def test_local_ranks_oracle_engine(oracle, pkg, kind, world, kw):
    stream = make_stream(kind)
    want, err = expected(oracle, stream)
    assert err is None
    t = torch.from_numpy(stream.copy())
    for origin, shift in ((0, 0), (5 * (1 << 32) + 123457, 48)):
        bounds = bounds_for(stream.size, world, origin, shift)
        res = run_local(t, bounds, lambda r: HostEngine(), lanes=(origin != 0), **kw)
        check_rows(res, bounds, want + origin)
    if kind in ("long", "long-wrapped") and not kw:
        assert any(r[0].rounds > 0 and r[0].head > (1 << 20) for r in res), "no rank grew its look-ahead"
    if kind == "tricky":
        assert any(r[0].rounds > 0 for r in res)


@pytest.mark.parametrize("block", range(6))
def test_local_ranks_oracle_engine_random_streams(oracle, pkg, block):
    """The host step's protocol (csrc/ffq_shard_host.h through HostShardScanner) over seeded random streams, on the CPU: records
    of random lengths, wrapped or not, a few bytes turned into newlines / '@' / '+' or cut out (tests/test_gpu_parity.mutate: the
    hostile inputs the kernels are stressed with), cut into 2-7 ranges at random 16-byte aligned points with random halos of
    16 ... 4096 bytes -- rows of every rank against the single-range scan, stream errors with the oracle's text on every rank.
    (The device step over the same kind of input: tools/stress_fileshards.py, tools/stress_rccl.py on the GPU box.)"""
    import test_gpu_parity as T
    for seed in range(block * 6, block * 6 + 6):
        rng = np.random.default_rng(77000 + seed)
        lo, hi = ((100, 160), (20, 60), (250, 400), (1, 30), (800, 2500))[seed % 5]
        nrec = int(rng.integers(100, 1500)) if hi < 800 else int(rng.integers(30, 200))
        data = T.random_records(rng, nrec, lo, hi, wrap=int(rng.integers(60, 101)) if seed % 2 else 0, repeat_hdr=bool(seed & 4))
        data = T.mutate(rng, data, (0, 1, 3, 8)[(seed // 5) % 4])
        if seed % 3 == 0:
            data = data[:len(data) - int(rng.integers(1, 200))]
        stream = np.frombuffer(bytes(data), dtype=np.uint8)
        want, err = expected(oracle, stream)
        world = int(rng.integers(2, 8))
        cuts = sorted(int(x) // 16 * 16 for x in rng.integers(0, stream.size + 1, world - 1))
        bounds = [0] + cuts + [int(stream.size)]
        kw = dict(tail_bytes=int(rng.integers(1, 257)) * 16, head_bytes=int(rng.integers(1, 257)) * 16)
        t = torch.from_numpy(stream.copy())
        if err is None:
            res = run_local(t, bounds, lambda r: HostEngine(), lanes=bool(seed & 1), **kw)
            check_rows(res, bounds, want)
        else:
            with pytest.raises(ValueError) as ei:
                run_local(t, bounds, lambda r: HostEngine(), **kw)
            assert str(ei.value).startswith(err), (seed, str(ei.value), err)


@pytest.mark.parametrize("kind", ("truncated", "cut-header", "invalid"))
@pytest.mark.parametrize("world", (2, 3, 8))
def test_local_ranks_stream_errors(oracle, pkg, kind, world):
    """The iterator's ValueErrors (fastqandfurious.py:262, :269, :272), raised by every rank with
    the text the single-range scan gives."""
    stream = make_stream(kind)
    _want, err = expected(oracle, stream)
    assert err is not None
    t = torch.from_numpy(stream.copy())
    with pytest.raises(ValueError) as ei:
        run_local(t, bounds_for(stream.size, world), lambda r: HostEngine())
    assert str(ei.value) == err


def test_local_ranks_stream_error_byte_with_tiny_halos(oracle, pkg):
    """The byte a stream error names is the iterator's `offset` (pos5 - 1 of the last COMPLETE record,
    fastqandfurious.py:254, :269).  When that record straddles in from the left and the failing rank
    owns no row of its own, the byte must come from the left neighbour's chain, not from the failing
    rank's scan start (9 of 1100 fuzzed streams named the wrong byte)."""
    import fastqandfurious_amd  # noqa: F401
    from fastqandfurious_amd import synth
    rng = np.random.default_rng(20260929)
    checked = 0
    for it in range(160):
        nrec = int(rng.integers(1, 12))
        data = synth.wrapped(int(rng.integers(0, 1000)), nrec, seed=43)[0] if it & 1 else synth.single(int(rng.integers(0, 1000)), nrec, seed=42)
        cut = int(rng.integers(1, min(data.size, 700)))
        stream = np.ascontiguousarray(data[:data.size - cut])
        _want, err = expected(oracle, stream)
        if err is None:
            continue
        t = torch.from_numpy(stream.copy())
        for world in (2, 3, 5):
            for tail, head in ((1, 16), (64, 48), (256, 64)):
                with pytest.raises(ValueError) as ei:
                    run_local(t, bounds_for(stream.size, world, 0, int(rng.integers(-40, 40))), lambda r: HostEngine(),
                              tail_bytes=tail, head_bytes=head)
                assert str(ei.value) == err, (it, world, tail, head)
                checked += 1
    assert checked > 300


def test_local_ranks_table_too_small(oracle, pkg):
    """one rank's table cannot hold its rows: every rank raises, nobody is left in a collective"""
    stream = make_stream("single")
    t = torch.from_numpy(stream.copy())
    with pytest.raises(RuntimeError) as ei:
        run_local(t, bounds_for(stream.size, 3), lambda r: HostEngine(lambda *a: oracle_scan(*a, small=True)) if r == 1 else HostEngine())
    assert "offset table too small" in str(ei.value)


def test_local_ranks_empty_and_tiny(oracle, pkg):
    for n in (0, 1, 17, 100, 322, 323, 700):
        stream = make_stream("small")[:n]
        want, err = expected(oracle, stream)
        t = torch.from_numpy(stream.copy())
        for world in (2, 4):
            bounds = bounds_for(stream.size, world)
            if err is None:
                check_rows(run_local(t, bounds, lambda r: HostEngine()), bounds, want)
            else:
                with pytest.raises(ValueError) as ei:
                    run_local(t, bounds, lambda r: HostEngine())
                assert str(ei.value) == err


# ---- separate processes over gloo -----------------------------------------------------------------
def _worker(rank, world, port, kind, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fastqandfurious_amd  # noqa: F401
        from fastqandfurious_amd import sharded
        from oracle import ffq_oracle
        stream = make_stream(kind)
        total = stream.size
        S = sharded.shard_bounds(total, world)
        lo, hi = S[rank], S[rank + 1]
        # the library's host step (ffq_shard_host_step): its protocol, gloo as the transport, the oracle as the scan
        tr = sharded.DistTransport(dist)
        sc = sharded.HostShardScanner(tr, S, scan=oracle_scan)
        tail, head = sc.halo()

        def fresh():
            e = np.zeros(tail + (hi - lo) + head + 64, dtype=np.uint8)
            e[tail:tail + hi - lo] = stream[lo:hi]
            return e
        ext = fresh()
        table = np.empty((100000, 6), dtype=np.int64)
        out = sc.scan(ext, tail, head, table)
        assert bytes(out.ext[:out.tail + hi - lo + out.head]) == bytes(stream[lo - out.tail:hi + out.head]), "halo bytes differ"
        if world > 1:
            assert out.comm["handoff_bytes"] > 0
        # the same step again through submit / finish, twice (the queue bench.py keeps): same rows, same words
        tabs = [table, np.empty_like(table)]
        for i in range(1, 3):
            sc.submit(fresh(), tail, head, tabs[i & 1])
            o2 = sc.finish()
            assert (o2.row_lo, o2.row_hi, o2.exit_pos, o2.first_pos, o2.record_base, o2.rounds) == \
                   (out.row_lo, out.row_hi, out.exit_pos, out.first_pos, out.record_base, out.rounds)
            assert (tabs[i & 1][:o2.n_rows] == table[:out.n_rows]).all()
        want, end, st, off = ffq_oracle.scan(stream)
        mine = want[(want[:, 0] >= lo) & (want[:, 0] < hi)]
        got = table[out.row_lo:out.row_hi]
        assert got.shape == mine.shape and (got == mine).all(), "shard rows differ from the single-range table"
        first = int(np.searchsorted(want[:, 0], lo))
        assert out.record_base == first
        assert out.total_records == len(want)
        if kind == "long":
            assert out.rounds > 0
        np.save(os.path.join(tmpdir, "rows_%d.npy" % rank), got)
    finally:
        dist.destroy_process_group()


from _ports import free_port as _free_port      # noqa: E402  (below the ephemeral range: tests/_ports.py)


@pytest.mark.parametrize("world,kind", ((2, "single"), (2, "wrapped"), (3, "wrapped"), (2, "long"), (3, "tricky")))
def test_sharded_scan_gloo(tmp_path, oracle, world, kind):
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path)), nprocs=world, join=True)
    stream = make_stream(kind)
    want, *_ = oracle.scan(stream)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "rows_%d.npy" % r)) for r in range(world)])
    assert (got == want).all()


def _error_worker(rank, world, port, kind, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fastqandfurious_amd  # noqa: F401
        from fastqandfurious_amd import sharded
        stream = make_stream(kind)
        S = sharded.shard_bounds(stream.size, world)
        lo, hi = S[rank], S[rank + 1]
        sc = sharded.HostShardScanner(sharded.DistTransport(dist), S, scan=oracle_scan)
        tail, head = sc.halo()
        ext = np.zeros(tail + (hi - lo) + head + 64, dtype=np.uint8)
        ext[tail:tail + hi - lo] = stream[lo:hi]
        msg = "none"
        try:
            sc.scan(ext, tail, head, np.empty((100000, 6), dtype=np.int64))
        except ValueError as e:
            msg = str(e)
        with open(os.path.join(tmpdir, "err_%d.txt" % rank), "w") as fh:
            fh.write(msg)
    finally:
        dist.destroy_process_group()


def test_sharded_stream_error_gloo(tmp_path, oracle):
    mp.spawn(_error_worker, args=(2, _free_port(), "invalid", str(tmp_path)), nprocs=2, join=True)
    _want, err = expected(oracle, make_stream("invalid"))
    for r in range(2):
        assert open(os.path.join(str(tmp_path), "err_%d.txt" % r)).read() == err


def _failing_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        import fastqandfurious_amd  # noqa: F401
        from fastqandfurious_amd import hip, sharded
        stream = make_stream("wrapped")
        S = sharded.shard_bounds(stream.size, world)
        lo, hi = S[rank], S[rank + 1]

        def scan(*a):
            if rank == 1:
                raise MemoryError("rank 1's engine gives up")
            return oracle_scan(*a)
        sc = sharded.HostShardScanner(sharded.DistTransport(dist), S, scan=scan)
        tail, head = sc.halo()
        ext = np.zeros(tail + (hi - lo) + head + 64, dtype=np.uint8)
        ext[tail:tail + hi - lo] = stream[lo:hi]
        try:
            sc.scan(ext, tail, head, np.empty((100000, 6), dtype=np.int64))
            msg = "came back"
        except MemoryError as e:
            msg = "MemoryError: %s" % e
        except hip.FFQError as e:
            msg = "FFQError %d: %s" % (e.code, e)
        with open(os.path.join(tmpdir, "fail_%d.txt" % rank), "w") as fh:
            fh.write(msg)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_failing_rank_does_not_strand_its_peers_gloo(tmp_path):
    """A rank whose OWN scan fails (not the stream's error: its engine's) used to return at once and leave the others in the
    next all-gather for ever (round-5 advisor).  Now its words say "failed", every rank sees them in the gather that was
    due anyway and comes back with an error: the failing rank with its own, the others naming it."""
    mp.spawn(_failing_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    msgs = [open(os.path.join(str(tmp_path), "fail_%d.txt" % r)).read() for r in range(3)]
    assert msgs[1].startswith("MemoryError"), msgs
    for r in (0, 2):
        assert msgs[r].startswith("FFQError -6") and "rank 1" in msgs[r], msgs


def _meeting_worker(rank, world, port, tmpdir, who_leaves):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime, time
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))
    import fastqandfurious_amd  # noqa: F401
    from fastqandfurious_amd import sharded

    class Tripped:          # (a shard whose step ran into the watchdog: abort() is what is left)
        aborted_at = None

        def abort(self):
            self.aborted_at = time.time()
            return True
    sh = Tripped()
    if rank == who_leaves:
        os._exit(0)                         # gone before the meeting: no goodbye
    if rank == world - 1:
        time.sleep(1.5)                     # the rank whose deadline ran out later
    t_in = time.time()
    try:
        ok = sharded.abort_together(dist, None, [sh])
        msg = "ok %s %.3f %.3f" % (ok, t_in, sh.aborted_at)
    except RuntimeError as e:
        msg = "RuntimeError %s: %s" % (sh.aborted_at, e)
    with open(os.path.join(tmpdir, "meet_%d.txt" % rank), "w") as fh:
        fh.write(msg)
    os._exit(0)                             # (a group whose peer is gone does not shut down cleanly)


@pytest.mark.timeout(300)
def test_ranks_meet_before_they_abort_gloo(tmp_path):
    """sharded.abort_together: nobody aborts before the LAST rank's step has tripped (a rank that aborts after its peers tore
    their ends down waits in RCCL's teardown: world 8), and a peer that never reports fails the meeting
    with nothing aborted."""
    d = str(tmp_path)
    mp.spawn(_meeting_worker, args=(3, _free_port(), d, -1), nprocs=3, join=True)
    rows = [open(os.path.join(d, "meet_%d.txt" % r)).read().split() for r in range(3)]
    assert all(r[0] == "ok" and r[1] == "True" for r in rows), rows
    late_in = float(rows[2][2])
    assert all(float(r[3]) >= late_in for r in rows), rows          # every abort after the late rank arrived
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    mp.spawn(_meeting_worker, args=(3, _free_port(), d, 1), nprocs=3, join=True)
    for r in (0, 2):
        msg = open(os.path.join(d, "meet_%d.txt" % r)).read()
        assert msg.startswith("RuntimeError None") and "did not report" in msg, msg


def test_shard_bounds(pkg):
    from fastqandfurious_amd import sharded
    b = sharded.shard_bounds(1000003, 4)
    assert b[0] == 0 and b[-1] == 1000003 and all(x % 16 == 0 for x in b[:-1]) and b == sorted(b)
    # every byte of every halo has exactly one provider: the pieces the host step asks its transport for
    B = [0, 1600, 3200, 4800, 5000]
    seen = []

    class Recorder(sharded.SoloTransport):
        world = 4

        def exchange(self, pieces):
            seen.append([(s_, d, a, c) for s_, d, a, c, _p in pieces])
            raise StopIteration                     # (the plan is all this test wants)
    sc = sharded.HostShardScanner(Recorder(), B, tail_bytes=1000, head_bytes=2000, scan=oracle_scan)
    with pytest.raises(StopIteration):
        sc.scan(np.zeros(1600 + 2000 + 64, dtype=np.uint8), 0, 2000, np.empty((10, 6), dtype=np.int64))
    plan = seen[0]
    for q in range(4):
        got = sorted((a, c) for s_, d, a, c in plan if d == q)
        lo, hi = B[q:q + 2]
        need = [(max(lo - 1000, 0), lo), (hi, min(hi + 2000, 5000))]
        assert sum(c - a for a, c in got) == sum(c - a for a, c in need)
        assert all(s_ != q for s_, d, a, c in plan if d == q)
    assert all(B[s_] <= a and c <= B[s_ + 1] for s_, d, a, c in plan)          # ... and the provider owns what it provides


# ---- the HIP engine ---------------------------------------------------------------------------------
def _hip_backends(gpu_ctx):
    """One context (= stream + scratch) per logical rank: contexts are not shared between threads."""
    from fastqandfurious_amd import hip, sharded
    made = {}

    def make(rank):
        made[rank] = hip.Context(0)
        return HostEngine(scan=None, ctx=made[rank])          # (the host step's scan: ffq_scan_host on this context)
    return make, made


GPU_CASES = [
    ("single", 2, {}), ("single", 3, {}), ("single", 8, {}),
    ("wrapped", 2, {}), ("wrapped", 3, {}), ("wrapped", 8, {}),
    ("single", 3, dict(tail_bytes=64, head_bytes=48)),
    ("wrapped", 8, dict(tail_bytes=256, head_bytes=64)),
    ("long", 2, {}), ("long", 8, {}), ("long-wrapped", 3, {}),
    ("tricky", 2, {}), ("tricky", 8, {}),
    ("small", 8, {}),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world,kw", GPU_CASES)
@pytest.mark.parametrize("native,decode", ((False, False), (True, False), (True, True)))      # (the host step has no decode)
def test_local_ranks_hip_engine(gpu_ctx, oracle, kind, world, kw, decode, native):
    """native: the DEVICE step (ffq_shard_step_submit / _wait with the in-process transport); else the HOST step
    (ffq_shard_host_step over the thread transport, the GPU scanning through ffq_scan_host) -- same protocol functions,
    same ranges, same rows, same rounds."""
    from fastqandfurious_amd import hip
    stream = make_stream(kind)
    want, err = expected(oracle, stream)
    assert err is None
    t = torch.from_numpy(stream.copy()).cuda()
    wq, wqoff = oracle.decode_quals(stream, want) if decode else (None, None)
    for origin, shift, lanes in ((0, 0, False), (5 * (1 << 32) + 123457, 48, True)):
        bounds = bounds_for(stream.size, world, origin, shift)
        make, made = _hip_backends(gpu_ctx)
        res = run_local(t, bounds, make, lanes=lanes, decode=decode, flags=hip.F_DECODE_QUAL if decode else 0, native=native, **kw)
        check_rows(res, bounds, want + origin)
        if decode:
            # every rank decoded the records of its whole view; its own records' qualities, in
            # order, are the stream's
            base = 0
            for r in range(world):
                out, table, qual, qoff = res[r]
                n_own = out.row_hi - out.row_lo
                qo = qoff[out.row_lo:out.row_hi + 1].cpu().numpy()
                q = qual[int(qo[0]):int(qo[-1])].cpu().numpy() if n_own else np.zeros(0, np.int8)
                assert (qo - qo[0] == wqoff[base:base + n_own + 1] - wqoff[base]).all()
                assert (q == wq[int(wqoff[base]):int(wqoff[base + n_own])]).all(), "rank %d: decoded qualities differ" % r
                base += n_own
        for c in made.values():
            c.close()
    if kind in ("long", "long-wrapped") and not kw:
        assert any(r[0].rounds > 0 and r[0].head > (1 << 20) for r in res), "no rank grew its look-ahead"
    if kind == "tricky":
        assert any(r[0].rounds > 0 for r in res)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world,kw", [c for c in GPU_CASES if c[1] != 8 or c[0] in ("wrapped", "tricky")])
@pytest.mark.parametrize("decode", (False, True))
def test_serial_step_equals_the_pipelined_one(gpu_ctx, oracle, kind, world, kw, decode):
    """The fallback mode of the device step (include/ffq.h: ONE communicator, ONE stream -- hand-off, scan, words, gather in
    order on the scan stream): the same rows, the same repair rounds, the same grown views as the pipelined step on the same
    ranges, and the oracle's rows; ffq_shard_result says which step ran."""
    from fastqandfurious_amd import hip
    stream = make_stream(kind)
    want, err = expected(oracle, stream)
    assert err is None
    t = torch.from_numpy(stream.copy()).cuda()
    bounds = bounds_for(stream.size, world, 5 * (1 << 32) + 123457, 48)
    got = {}
    for serial in (False, True):
        make, made = _hip_backends(gpu_ctx)
        res = run_local(t, bounds, make, lanes=True, decode=decode, flags=hip.F_DECODE_QUAL if decode else 0, native=True, serial=serial, **kw)
        check_rows(res, bounds, want + bounds[0])
        assert all(r[0].comm["mode"] == ("serial" if serial else "pipelined") and r[0].comm["nranks"] == world for r in res)
        got[serial] = [(r[0].rounds, r[0].head, r[0].row_lo, r[0].row_hi) for r in res]
        if decode:
            wq, wqoff = oracle.decode_quals(stream, want)
            base = 0
            for r in range(world):
                out, table, qual, qoff = res[r]
                n_own = out.row_hi - out.row_lo
                qo = qoff[out.row_lo:out.row_hi + 1].cpu().numpy()
                q = qual[int(qo[0]):int(qo[-1])].cpu().numpy() if n_own else np.zeros(0, np.int8)
                assert (q == wq[int(wqoff[base]):int(wqoff[base + n_own])]).all(), "rank %d: decoded qualities differ" % r
                base += n_own
        for c in made.values():
            c.close()
    assert got[True] == got[False], "the serial step took other rounds / views than the pipelined one"


@pytest.mark.gpu
@pytest.mark.parametrize("native", (False, True))
@pytest.mark.parametrize("kind", ("truncated", "cut-header", "invalid"))
def test_local_ranks_hip_engine_stream_errors(gpu_ctx, oracle, kind, native):
    stream = make_stream(kind)
    _want, err = expected(oracle, stream)
    t = torch.from_numpy(stream.copy()).cuda()
    for world in (2, 8):
        make, made = _hip_backends(gpu_ctx)
        with pytest.raises(ValueError) as ei:
            run_local(t, bounds_for(stream.size, world), make, native=native)
        assert str(ei.value) == err
        for c in made.values():
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("native", (False, True))
@pytest.mark.parametrize("kind", ("single", "wrapped"))
def test_synthetic_shards_local_ranks(gpu_ctx, kind, native):
    """bench.py's SyntheticShard, k = 4 logical ranks on one GPU (the N > 1 code path of bench.py
    without RCCL): generation per rank, halo hand-off, pipelined submit / finish lanes, and the
    closed-form checks bench.py applies to its measured output."""
    from fastqandfurious_amd import hip, sharded
    world = 4
    lw = sharded.LocalWorld(world)
    dev = torch.device("cuda", 0)
    errors = [None] * world
    totals = [None] * world

    def work(rank):
        try:
            ctx = hip.Context(0)
            sh = sharded.SyntheticShard(ctx, kind, 24 << 20, rank, world, dev, transport=lw.transport(rank), native=native)
            ctx.reserve(sh.ext.numel())
            table = torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device=dev)
            qual = torch.empty(sh.ext.numel(), dtype=torch.int8, device=dev)
            qoff = torch.empty(table.shape[0] + 1, dtype=torch.int64, device=dev)
            out = sh.scan(table, flags=hip.F_DECODE_QUAL if native else 0, qual=qual, qoff=qoff)
            sh.verify(table, out)
            if native:
                sh.verify_decode(table, out, qual, qoff)
            sh.make_lanes(2)
            # with peers every lane has a buffer of its own and its hand-off runs on the hand-off stream,
            # beside the other lane's scan: wipe the halos, so that rows can only come out right if
            # each lane's hand-off has landed before its scan reads them
            assert not native or (sh._overlap and sh._exts[0].data_ptr() != sh._exts[1].data_ptr())
            for e in sh._exts:
                e[:sh.tail].zero_()
                e[sh.tail + sh.n_own_bytes:].zero_()
            torch.cuda.synchronize()
            tabs = (table, torch.empty_like(table))
            sh.submit(0, tabs[0])
            for i in range(1, 4):
                sh.submit(i & 1, tabs[i & 1])
                o2 = sh.finish((i - 1) & 1)
                sh.verify(tabs[(i - 1) & 1], o2)
                assert (o2.row_lo, o2.row_hi, o2.record_base) == (out.row_lo, out.row_hi, out.record_base)
            o2 = sh.finish(3 & 1)
            totals[rank] = (out.total_records, out.record_base, out.n_own_records)
        except BaseException as e:   # noqa: BLE001
            errors[rank] = e
            lw.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)
            and "another logical rank failed" not in str(e)]
    if real:
        raise real[0]
    assert sum(t[2] for t in totals) == totals[0][0]
    assert [t[1] for t in totals] == [sum(x[2] for x in totals[:r]) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world,flag", (("long-wrapped", 3, 0), ("long-wrapped", 2, "ranked"), ("small", 3, "serial"), ("wrapped", 3, "ranked")))
def test_device_step_when_the_scan_ends_in_a_host_driven_tier(gpu_ctx, oracle, kind, world, flag):
    """A scan whose result only exists after the host's wait -- the list-ranking tier and the one-wave walker are driven
    from ffq_scan_wait; a context that has met long records goes straight there on its NEXT scans -- must not hand the
    step the previous scan's result block: its words are "not ready" and gathered again (k_publish marks the block).
    Three consecutive steps on the same shard objects; the second and third start in the ranking tier."""
    from fastqandfurious_amd import hip, sharded
    stream = make_stream(kind)
    want, err = expected(oracle, stream)
    assert err is None
    t = torch.from_numpy(stream.copy()).cuda()
    bounds = bounds_for(stream.size, world)
    lw = sharded.LocalWorld(world)
    flags = {0: 0, "ranked": hip.F_FORCE_RANKED, "serial": hip.F_FORCE_SERIAL}[flag]
    results, errors = [None] * world, [None] * world

    def work(rank):
        try:
            ctx = hip.Context(0)
            sc = sharded.NativeShardScanner(ctx, bounds, rank, world, local_world=lw.native_world())
            tail, head = sc.halo()
            lo, hi = bounds[rank], bounds[rank + 1]
            ext = torch.zeros(tail + (hi - lo) + head + 64, dtype=torch.uint8, device="cuda")
            ext[tail:tail + hi - lo] = t[lo:hi]
            table = torch.empty((stream.size // 40 + 64, 6), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            outs = []
            for step in range(3):
                table.fill_(-7)
                torch.cuda.synchronize()
                out = sc.scan(ext, tail, head, table, flags)
                outs.append((out.res.path, out.rounds, out.record_base, table[out.row_lo:out.row_hi].cpu().numpy()))
            results[rank] = outs
            sc.close()
            ctx.close()
        except BaseException as e:   # noqa: BLE001
            errors[rank] = e
            lw.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    real = [e for e in errors if e is not None and "another logical rank failed" not in str(e)]
    if real:
        raise real[0]
    for step in range(3):
        got = np.concatenate([results[r][step][3] for r in range(world)])
        assert got.shape == want.shape and (got == want).all(), "step %d: rows over the ranks differ from the oracle's" % step
        assert [results[r][step][2] for r in range(world)] == [sum(results[q][step][3].shape[0] for q in range(r)) for r in range(world)]
    if flag == "serial":
        assert all(results[r][0][0] == 1 for r in range(world))
    elif kind == "long-wrapped" or flag == "ranked":
        assert any(results[r][2][0] == 5 for r in range(world)), "no rank's third scan took the ranking tier"
