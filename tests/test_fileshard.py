"""File-backed byte-range shards ("file-range sharding across the GPUs").

`world` ranks read ONE file together: rank r loads bytes [S_r - tail, S_r+1 + head) of it straight into its GPU's
memory (ffq_shard_load_fd: pread -> pinned slots -> two copy streams), one native step (ffq_shard_step_*) scans the
view, cuts the rank's rows out and proves them against the neighbours' with one gather of eight words; nothing else
passes between ranks.  What is replaced is the reference's single reader: /root/reference/src/fastqandfurious.py
:30-36 (read), :241-245 (first fill + sentinel), :274-279 (the carry, here per range edge).

The invariant is the reference's own "results do not depend on how the stream is cut" (/root/reference/tests.py
:219-226, bufsize 100/200/600/700): rows concatenated over the ranks == the oracle's scan of the whole file, the
entries of the ranks' iterators concatenated == readfastq_iter's golden tuples.  k logical ranks run as threads of
this process on one GPU (hip.ShardWorld); the same calls over RCCL: tests/test_multigpu.py."""
import os
import threading
from array import array

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_file
from test_sharded import expected, make_stream

pytestmark = pytest.mark.gpu


def run_ranks(world, fn):
    """fn(rank, ctx, shard_world) on `world` threads, a context each; what the ranks raised is raised here (all alike)."""
    from fastqandfurious_amd import hip
    sw = hip.ShardWorld(world)
    results, errors = [None] * world, [None] * world

    def work(rank):
        ctx = None
        try:
            ctx = hip.Context(0)
            results[rank] = fn(rank, ctx, sw)
        except BaseException as e:      # noqa: BLE001
            errors[rank] = e
            sw.abort()
        finally:
            if ctx is not None:
                ctx.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sw.close()
    real = [e for e in errors if e is not None and "another logical rank failed" not in str(e)]
    if real:
        if len(real) == world:
            assert len({str(e) for e in real}) == 1, "ranks disagree on the error: %r" % real
        raise real[0]
    assert not any(errors), errors
    return results


def shard_rows(path, world, bounds=None, decode=False, **kw):
    """Every rank: FileShard -> load -> scan; (rows, record_base, total, rounds, halo_source, view bytes == file bytes)."""
    from fastqandfurious_amd import sharded

    def work(rank, ctx, sw):
        sh = sharded.FileShard(ctx, path, rank, world, comm=sw, bounds=bounds, **kw)
        try:
            assert sh.load() == sh.n_view
            res = sh.scan(decode=decode)
            view = np.empty(sh.tail + (sh.hi - sh.lo) + int(res.head), dtype=np.uint8)
            if view.size:
                ctx.d2h(view, int(res.d_ext))
            with open(path, "rb") as fh:
                fh.seek(sh.lo - sh.tail)
                assert fh.read(view.size) == view.tobytes(), "rank %d: the view is not the file's bytes" % rank
            rows = sh.rows()
            q = sh.quals(0, rows.shape[0], rows) if decode and rows.shape[0] else None
            return dict(rows=rows, base=int(res.record_base), total=int(res.total_records), rounds=int(res.rounds),
                        source=int(res.halo_source), head=int(res.head), bounds=list(sh.bounds), quals=q,
                        transport=sh.sh.transport())
        finally:
            sh.close()
    return run_ranks(world, work)


def check(results, want):
    got = np.concatenate([r["rows"] for r in results])
    assert got.shape == want.shape and (got == want).all(), "rows over the ranks differ from the scan of the whole file"
    base = 0
    for r, res in enumerate(results):
        b = res["bounds"]
        # (ownership is by '@': the first range that starts the stream also owns what lies in front of it -- nothing --,
        # the last one everything up to the end)
        lo = -1 if b[r] == b[0] else b[r]
        hi = (1 << 62) if b[r + 1] == b[-1] else b[r + 1]
        mine = want[(want[:, 0] >= lo) & (want[:, 0] < hi)] if b[r + 1] > b[r] else want[:0]
        assert res["rows"].shape == mine.shape and (res["rows"] == mine).all(), "rank %d owns other records than those starting in its range" % r
        assert res["base"] == base and res["total"] == len(want)
        assert res["source"] == 1 and res["transport"] == "in-process"
        base += len(mine)


@pytest.fixture()
def shm_file():
    made = []

    def make(data, name="ffq_fileshard_test.fq"):
        d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        p = os.path.join(d, "%s.%d" % (name, os.getpid()))
        with open(p, "wb") as fh:
            fh.write(bytes(data))
        made.append(p)
        return p
    yield make
    for p in made:
        try:
            os.unlink(p)
        except OSError:
            pass


@pytest.mark.parametrize("world", (2, 3, 8))
@pytest.mark.parametrize("name", ("test.fq", "test_longqualityheader.fq", "test_multiline.fq"))
def test_golden_files_over_ranks(gpu_ctx, oracle, golden, name, world):
    """The reference's own data files (tests/golden/data), cut into 2 / 3 / 8 ranges (with the product's 1 MiB halos
    every view is the whole file; with 40-byte halos every edge has to grow its look-ahead out of the file)."""
    path = os.path.join(GOLDEN_DIR, "data", name)
    data = np.frombuffer(golden_file(name), dtype=np.uint8)
    want, err = expected(oracle, data)
    assert err is None
    assert [list(map(int, r)) for r in want] == golden["files"][name]["bufsizes"]["65536"]["c"]["rows"]
    for kw in ({}, dict(tail_bytes=40, head_bytes=24)):
        res = shard_rows(path, world, **kw)
        check(res, want)
    assert any(r["rounds"] > 0 for r in res), "40-byte halos: no edge grew its look-ahead"


@pytest.mark.parametrize("kind,world,kw", [
    ("single", 2, {}), ("single", 3, {}), ("single", 8, {}),
    ("wrapped", 2, {}), ("wrapped", 3, {}), ("wrapped", 8, {}),
    ("single", 3, dict(tail_bytes=64, head_bytes=48)),
    ("wrapped", 8, dict(tail_bytes=256, head_bytes=64)),
    ("wrapped", 3, dict(tail_bytes=1, head_bytes=16)),
    ("long", 2, {}), ("long", 8, {}), ("long-wrapped", 3, {}),
    ("long", 3, dict(qual_room=16384)), ("single", 2, dict(qual_room=16384)),       # (room for the in-place single pass)
    ("tricky", 2, {}), ("tricky", 8, {}),
    ("small", 8, {}),
])
@pytest.mark.parametrize("decode", (False, True))
def test_synthetic_files_over_ranks(gpu_ctx, oracle, shm_file, kind, world, kw, decode):
    """S-single / S-wrapped files in /dev/shm, a 3 MiB record across an edge (the look-ahead grows out of the file),
    a quality block of FASTQ-looking text (wrong entry guesses, re-entered from the left neighbour's exit)."""
    stream = make_stream(kind)
    want, err = expected(oracle, stream)
    assert err is None
    path = shm_file(stream)
    res = shard_rows(path, world, decode=decode, **kw)
    check(res, want)
    if kind in ("long", "long-wrapped") and "tail_bytes" not in kw:
        assert any(r["rounds"] > 0 and r["head"] > (1 << 20) for r in res), "no rank grew its look-ahead"
    if kind == "tricky":
        assert any(r["rounds"] > 0 for r in res)
    if decode:
        wq, wqoff = oracle.decode_quals(stream, want)
        base = 0
        for r in res:
            n = r["rows"].shape[0]
            if not n:
                continue
            qual, qoff = r["quals"]
            lens = r["rows"][:, 5] - r["rows"][:, 4]
            idx = np.repeat(qoff[:n], lens) + (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
            assert (qual[idx] == wq[int(wqoff[base]):int(wqoff[base + n])]).all(), "decoded qualities differ"
            base += n


def test_edges_forced_into_record_parts(gpu_ctx, oracle, shm_file):
    """Cut points placed ON the bytes where a guess can go wrong: inside a header, on the '+' of a '+' line, on the
    newline behind it, on the first byte of a quality line that begins with '@', on a record's '@' and on the newline
    in front of it -- with halos too short to run in (a guess from 100 bytes back starts inside the record)."""
    from fastqandfurious_amd import synth
    n = 6000
    data = synth.single(0, n, seed=42)
    raw = data.tobytes()
    want, err = expected(oracle, data)
    assert err is None
    at_q = [k for k in range(100, n - 100) if raw[322 * k + 171] == ord("@")]
    plus_q = [k for k in range(100, n - 100) if raw[322 * k + 171] == ord("+")]
    assert len(at_q) > 20 and len(plus_q) > 20
    path = shm_file(data)
    k0, k1, k2 = at_q[3], at_q[len(at_q) // 2], at_q[-3]
    cases = [
        [0, 322 * k0 + 5, 322 * k1 + 169, 322 * k2 + 171, data.size],          # header / '+' / '@'-leading quality
        [0, 322 * k0 + 170, 322 * k1 + 171, 322 * k2 + 172, data.size],        # "\n" behind '+' / '@' quality / one byte in
        [0, 322 * k0, 322 * k0 + 321, 322 * k1 + 17, data.size],               # a record's '@' / the newline before the next / header end
        [0, 322 * plus_q[5] + 171, 322 * plus_q[9] + 168, 322 * k2 + 18, data.size],   # '+'-leading quality / "\n" before '+' / first base
        [0, 322 * k0 + 171, 322 * k0 + 171, 322 * k0 + 171, data.size],        # empty ranges between equal cut points
    ]
    for bounds in cases:
        for kw in ({}, dict(tail_bytes=100, head_bytes=64), dict(tail_bytes=1, head_bytes=1), dict(tail_bytes=160, head_bytes=400)):
            res = shard_rows(path, 4, bounds=bounds, **kw)
            check(res, want)
    # the same cuts through S-wrapped: '+' lines that repeat the header, sequence and quality lines of 80 columns
    wdata, _ = synth.wrapped(0, 5000, seed=43)
    wwant, err = expected(oracle, wdata)
    assert err is None
    wpath = shm_file(wdata, "ffq_fileshard_w.fq")
    rows = wwant
    mid = [len(rows) // 5, len(rows) // 2, 4 * len(rows) // 5]
    for pick in (lambda r: r[0] + 3, lambda r: r[3] + 1, lambda r: r[4], lambda r: r[4] - 1, lambda r: r[2] + 81, lambda r: r[5]):
        bounds = [0] + [int(pick(rows[i])) for i in mid] + [wdata.size]
        for kw in ({}, dict(tail_bytes=120, head_bytes=80)):
            check(shard_rows(wpath, 4, bounds=bounds, **kw), wwant)


@pytest.mark.parametrize("kind", ("truncated", "cut-header", "invalid"))
def test_stream_errors_on_every_rank(gpu_ctx, oracle, shm_file, kind):
    """The iterator's three ValueErrors (fastqandfurious.py:262, :269, :272): every rank raises the text the scan of
    the whole file gives, before any entry is handed out."""
    from fastqandfurious_amd import fastqandfurious as F
    stream = make_stream(kind)
    _want, err = expected(oracle, stream)
    assert err is not None
    path = shm_file(stream)
    for world in (2, 8):
        with pytest.raises(ValueError) as ei:
            shard_rows(path, world)
        assert str(ei.value) == err
        with pytest.raises(ValueError) as ei:
            run_ranks(world, lambda rank, ctx, sw: list(F.readfastq_iter_range(path, rank, world, comm=sw, ctx=ctx)))
        assert str(ei.value) == err


def test_range_iterator_matches_reference_tuples(gpu_ctx, golden):
    """readfastq_iter_range over the reference's files: the ranks' entries, concatenated, are the tuples the
    reference's readfastq_iter yields (captured by tests/golden/make_golden.py), each rank's record_base the ordinal
    of its first; entryfunc_abspos gives the golden absolute rows."""
    from fastqandfurious_amd import fastqandfurious as F
    for name, g in golden["files"].items():
        path = os.path.join(GOLDEN_DIR, "data", name)
        for world in (1, 2, 3, 8):
            for kw in ({}, dict(tail_bytes=33, head_bytes=17)):
                def work(rank, ctx, sw, entryfunc=F.entryfunc):
                    it = F.readfastq_iter_range(path, rank, world, entryfunc, comm=sw, ctx=ctx, **kw)
                    assert it.comm["halo_source"] == "file" and it.total_records == len(g["tuples"])
                    out = [e if not isinstance(e, array) else list(e) for e in it]
                    assert len(out) == it.n_records
                    return it.record_base, out
                res = run_ranks(world, work)
                assert [b for b, _ in res] == [sum(len(o) for _, o in res[:r]) for r in range(world)]
                got = [[h.hex(), s.hex(), q.hex()] for _, o in res for h, s, q in o]
                assert got == g["tuples"], (name, world, kw)
                res = run_ranks(world, lambda rank, ctx, sw: work(rank, ctx, sw, F.entryfunc_abspos))
                assert [r for _, o in res for r in o] == g["bufsizes"]["65536"]["c"]["rows"], (name, world, kw)
                res = run_ranks(world, lambda rank, ctx, sw: work(rank, ctx, sw, F.entryfunc_namedtuple))
                assert all(isinstance(e, F.Entry) for _, o in res for e in o)
                assert [[e.header.hex(), e.sequence.hex(), e.quality.hex()] for _, o in res for e in o] == g["tuples"]


def test_range_iterator_phred_and_world_of_one(gpu_ctx, oracle, shm_file):
    """entryfunc_phred: the qualities come from the step's own decode (device), as array('b'); a world of one needs
    no comm at all and equals readfastq_iter over the same file."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, synth
    data = synth.wrapped(0, 3000, seed=43)[0]
    path = shm_file(data)
    with open(path, "rb") as fh:
        ref = list(F.readfastq_iter(fh, 1 << 20, F.entryfunc_phred, C.entrypos))
    it = F.readfastq_iter_range(path, 0, 1, F.entryfunc_phred)
    assert (it.record_base, it.n_records, it.total_records) == (0, len(ref), len(ref))
    got = list(it)
    assert got == ref and all(isinstance(q, array) and q.typecode == "b" for _, _, q in got)
    for world in (3, 8):
        res = run_ranks(world, lambda rank, ctx, sw: list(F.readfastq_iter_range(path, rank, world, F.entryfunc_phred, comm=sw, ctx=ctx,
                                                                                 batch_rows=257)))
        assert [e for o in res for e in o] == ref
    # part of a file as the stream (start / end): offsets stay file offsets
    want, _ = expected(oracle, data)
    a, b = int(want[700][0]), int(want[2100][5]) + 1
    res = run_ranks(3, lambda rank, ctx, sw: [list(p) for p in F.readfastq_iter_range(path, rank, 3, F.entryfunc_abspos, comm=sw, ctx=ctx,
                                                                                      start=a, end=b)])
    assert [r for o in res for r in o] == [list(map(int, r)) for r in want[700:2101]]


def test_load_fd_and_short_file(gpu_ctx, shm_file):
    """ffq_load_fd: a byte range of a file in HBM, bit for bit (several 32 MiB slots, odd offsets); a file shorter
    than the bounds say is an error of the load, not garbage in the view."""
    from fastqandfurious_amd import hip, sharded, synth
    data = synth.single(0, 330000, seed=42)                    # 106 MB: four slots
    path = shm_file(data)
    fd = os.open(path, os.O_RDONLY)
    try:
        for pos, n in ((0, data.size), (12345, 70 << 20), (data.size - 1000, 5000), (7, 0)):
            d = gpu_ctx.dev_alloc(max(n, 1) + 64)
            got = gpu_ctx.load_fd(fd, pos, n, d)
            assert got == min(n, data.size - pos)
            back = np.empty(got, dtype=np.uint8)
            if got:
                gpu_ctx.d2h(back, d)
            assert (back == data[pos:pos + got]).all()
            gpu_ctx.dev_free(d)
    finally:
        os.close(fd)
    sh = sharded.FileShard(gpu_ctx, path, 0, 1)
    try:
        os.truncate(path, data.size - 4096)
        with pytest.raises(hip.FFQError, match="the file ends at byte"):
            sh.load()
    finally:
        sh.close()


_AFFINITY_PROBE = """
import os, sys
sys.path.insert(0, %r)
import numpy as np
import fastqandfurious_amd
from fastqandfurious_amd import hip
before = sorted(os.sched_getaffinity(0))
ctx = hip.Context(0)
fd = os.open(sys.argv[1], os.O_RDONLY)
n = os.fstat(fd).st_size
d = ctx.dev_alloc(n + 64)
assert ctx.load_fd(fd, 0, n, d) == n
back = np.empty(n, dtype=np.uint8)
ctx.d2h(back, d)
assert (back == np.fromfile(sys.argv[1], dtype=np.uint8)).all()
assert sorted(os.sched_getaffinity(0)) == before, "the calling thread's affinity was changed"
print("probe ok")
"""


@pytest.mark.parametrize("affinity", ("1", "0"))
def test_loader_helpers_next_to_the_gpu(shm_file, affinity):
    """The loader's helper threads bind themselves to the CPUs next to the GPU (sysfs local_cpulist; two-socket hosts lose a
    fifth of the rate on the other socket: profiles/r06_probes/loader_numa.txt) and the pinned slots are allocated from there;
    FFQ_POOL_AFFINITY=0 switches it off.  Either way: the same bytes, the CALLER's affinity untouched, and FFQ_POOL_DEBUG
    says what was done (a host without that sysfs entry: "not bound", nothing else changes)."""
    import subprocess
    import sys
    from fastqandfurious_amd import synth
    data = synth.single(0, 120000, seed=5)                     # 38 MB: two slots
    path = shm_file(data)
    env = dict(os.environ, FFQ_POOL_DEBUG="1", FFQ_POOL_AFFINITY=affinity)
    r = subprocess.run([sys.executable, "-c", _AFFINITY_PROBE % os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "probe ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("[ffq pool]")]
    assert len(lines) == 1, r.stderr[-2000:]
    if affinity == "0":
        assert "not bound" in lines[0], lines
    else:
        assert "next to the GPU" in lines[0] or "not bound" in lines[0], lines


@pytest.mark.parametrize("kind,world", (("wrapped", 3), ("long", 2), ("tricky", 8), ("small", 5)))
def test_device_step_over_a_hosted_transport(gpu_ctx, oracle, shm_file, kind, world):
    """ffq_shard_create_hosted: the device step (buffers in HBM, scan and words on the device) over the caller's own
    transport -- here the thread transport of the host step, as DistTransport over gloo would be for several processes
    sharing one GPU: hand-offs staged through host memory, the words gathered by the callback.  With halos handed off
    (a resident buffer) and with a file behind every rank (nothing handed off)."""
    import torch
    from fastqandfurious_amd import hip, sharded
    from test_sharded import bounds_for
    stream = make_stream(kind)
    want, err = expected(oracle, stream)
    assert err is None
    bounds = bounds_for(stream.size, world)
    lw = sharded.LocalWorld(world)
    path = shm_file(stream)
    t = torch.from_numpy(stream.copy()).cuda()
    results, errors = [None] * world, [None] * world

    def work(rank):
        try:
            ctx = hip.Context(0)
            sc = sharded.NativeShardScanner(ctx, bounds, rank, world, hosted=lw.transport(rank))
            assert sc.sh.transport() == "hosted"
            tail, head = sc.halo()
            lo, hi = bounds[rank], bounds[rank + 1]
            ext = torch.zeros(tail + (hi - lo) + head + 64, dtype=torch.uint8, device="cuda")
            ext[tail:tail + hi - lo] = t[lo:hi]
            table = torch.empty((stream.size // 40 + 64, 6), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            out = sc.scan(ext, tail, head, table)
            rows = table[out.row_lo:out.row_hi].cpu().numpy()
            assert world == 1 or stream.size < 64 * world or out.comm["handoff_bytes"] > 0
            sc.close()
            fs = sharded.FileShard(ctx, path, rank, world, comm=lw.transport(rank), bounds=bounds)
            fs.load()
            res = fs.scan()
            assert fs.sh.transport() == "hosted" and res.halo_source == 1 and res.handoff_bytes == 0
            frows = fs.rows()
            fs.close()
            ctx.close()
            results[rank] = (rows, frows, out.record_base, int(res.record_base), out.rounds, int(res.rounds))
        except BaseException as e:   # noqa: BLE001
            errors[rank] = e
            lw.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real:
        raise real[0]
    for k in (0, 1):
        got = np.concatenate([r[k] for r in results])
        assert got.shape == want.shape and (got == want).all()
        assert [r[2 + k] for r in results] == [sum(q[k].shape[0] for q in results[:i]) for i in range(world)]
    if kind in ("long", "tricky"):
        assert any(r[4] > 0 for r in results) and any(r[5] > 0 for r in results)


def test_file_shards_at_size_closed_form(gpu_ctx, shm_file):
    """A 3 GiB S-single file in /dev/shm read by 8 logical ranks (0.38 GiB each, resident): every rank's rows against the
    generator's closed form (row k = 322 k + the six column offsets), ordinals and counts adding up -- the file-backed
    counterpart of the config-5 property test, at a size the pool's boxes hold in host memory."""
    import torch
    from fastqandfurious_amd import sharded
    n = (3 << 30) // 322
    total = n * 322
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    if os.statvfs(d).f_bavail * os.statvfs(d).f_frsize < total + (1 << 30):
        pytest.skip("no room for a 3 GiB file in %s" % d)
    path = os.path.join(d, "ffq_fileshard_big.%d.fq" % os.getpid())
    dev = torch.device("cuda", 0)
    try:
        # the file, written piece by piece from the device generator (counter-based: any piece on its own)
        with open(path, "wb") as fh:
            piece = 1 << 20                                # records per piece (322 MB)
            buf = torch.empty(piece * 322, dtype=torch.uint8, device=dev)
            for first in range(0, n, piece):
                cnt = min(piece, n - first)
                gpu_ctx.synth_single(buf.data_ptr(), first, cnt, seed=42)
                gpu_ctx.sync()
                fh.write(buf[:cnt * 322].cpu().numpy().tobytes())
            del buf
        world = 8
        col = torch.tensor([0, 17, 18, 168, 171, 321], dtype=torch.int64, device=dev)

        def work(rank, ctx, sw):
            sh = sharded.FileShard(ctx, path, rank, world, comm=sw)
            try:
                assert sh.load() == sh.n_view
                res = sh.scan()
                assert res.scan.path == 3 and res.rounds == 0 and res.halo_source == 1
                n_own = int(res.row_hi - res.row_lo)
                k0 = -(-sh.lo // 322)
                assert n_own == -(-sh.hi // 322) - k0 and int(res.record_base) == k0 and int(res.total_records) == n
                rows = torch.as_tensor(sharded._DevView(sh.d_table + int(res.row_lo) * 48, n_own * 48), device=dev).view(torch.int64).view(-1, 6)
                k = torch.arange(k0, k0 + n_own, dtype=torch.int64, device=dev) * 322
                assert bool((rows == k[:, None] + col[None, :]).all()), "rank %d: rows differ from the closed form" % rank
                return n_own
            finally:
                sh.close()
        counts = run_ranks(world, work)
        assert sum(counts) == n
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
        torch.cuda.empty_cache()


def test_tiny_and_empty_files(gpu_ctx, oracle, shm_file):
    """An empty file, one record, a file smaller than the number of ranks: ranges of zero bytes, ranks that own nothing."""
    from fastqandfurious_amd import fastqandfurious as F, synth
    for nrec, worlds in ((0, (1, 3)), (1, (1, 2, 8)), (3, (8,))):
        data = synth.single(0, nrec, seed=42) if nrec else np.zeros(0, dtype=np.uint8)
        path = shm_file(data, "ffq_tiny_%d.fq" % nrec)
        want, err = expected(oracle, data)
        assert err is None and len(want) == nrec
        for world in worlds:
            res = run_ranks(world, lambda rank, ctx, sw: (lambda it: (it.record_base, it.total_records, [list(p) for p in it]))(
                F.readfastq_iter_range(path, rank, world, F.entryfunc_abspos, comm=sw, ctx=ctx)))
            assert [r for _, _, o in res for r in o] == [list(map(int, r)) for r in want]
            assert all(t == nrec for _, t, _ in res)
    # a file of a few bytes that is no FASTQ at all: the stream's error, on every rank
    path = shm_file(b"@r1\nAC", "ffq_tiny_bad.fq")
    with pytest.raises(ValueError, match="Incomplete entry at byte"):
        run_ranks(2, lambda rank, ctx, sw: list(F.readfastq_iter_range(path, rank, 2, comm=sw, ctx=ctx)))
