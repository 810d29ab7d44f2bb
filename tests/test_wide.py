"""The Phred decode of records of ANY layout in ONE pass over the input (round 6): FFQ_F_DECODE_QUAL | FFQ_F_SINGLE_PASS on
the general path with FFQ_INPLACE_STRIDE bytes of quality buffer per tile -- the index kernel also writes EVERY byte of the
buffer decoded at its own offset (k_scan_lines<.., WIDE>, csrc/ffq_kernels.h), no tier runs a decode kernel, d_qoff[i] = the
offset pos4 has in the buffer, res.path carries FFQ_PATH_IN_PLACE.

What is computed is the reference's: array('b').frombytes(buf[pos4:pos5]); arrayadd_b(q, -33) per record
(/root/reference/doc/user-guide.rst:126-141, /root/reference/src/_fastqandfurious.c:129, :161-185) -- for wrapped records
the slice keeps its embedded newlines, which decode to '\\n' - 33 = -23 -- against the oracle's restatement, record by
record, on every tier (group kernels, dense configuration, list ranking, one-wave walker), with truncated streams, search
offsets, a sentinel or none, stream offsets past 2^32, quality lines that begin with '@' and '+', and at BASELINE
configs[3]'s size through the generator's closed form."""
import numpy as np
import pytest

from test_gpu_parity import random_records

pytestmark = pytest.mark.gpu


@pytest.fixture
def hipmod(pkg):
    from fastqandfurious_amd import hip
    return hip


def wide_same(ctx, hipmod, oracle, data, flags=0, expect=None, qual_add=-33, **kw):
    """Rows and every record's decoded bytes against the oracle; the layout by res.path."""
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    want, *_ = oracle.scan(a, **kw)
    sentinel = kw.get("sentinel", True)
    add = kw.get("add", -1 if sentinel else 0)
    in_buf = want - add - (1 if sentinel else 0)                       # the rows as offsets into `a`
    wq, wqoff = oracle.decode_quals(a, in_buf, qual_add)
    table, res, qual, qoff = ctx.scan_host(a, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS | flags, qual_room=hipmod.INPLACE_STRIDE,
                                           qual_add=qual_add, **kw)
    assert table.shape == want.shape and (table == want).all(), "rows differ"
    n = len(want)
    lens = want[:, 5] - want[:, 4] if n else np.zeros(0, np.int64)
    assert qoff.shape[0] == n + 1
    if n:
        assert int(qoff[n]) == int(qoff[n - 1] + lens[n - 1]) == int(res.n_qual_bytes)
        idx = np.repeat(qoff[:n] - wqoff[:n], lens) + np.arange(wq.size)
        assert (qual[idx] == wq).all(), "decoded bytes differ"
    if res.path & hipmod.PATH_IN_PLACE:
        assert (qoff[:n] == in_buf[:, 4]).all(), "in place: qoff[i] must be pos4's offset in the buffer"
    elif res.path not in (6,):
        assert (qoff == wqoff).all()                                     # the two passes: packed
    if expect is not None:
        assert res.path == expect, "path %d, expected %d" % (res.path, expect)
    return res


def tier_flags(hipmod, tier):
    return {"general": hipmod.F_FORCE_GENERAL, "ranked": hipmod.F_FORCE_RANKED, "serial": hipmod.F_FORCE_SERIAL}[tier]


TIER_PATH = {"general": 8, "ranked": 13, "serial": 9}


@pytest.mark.parametrize("tier", ("general", "ranked", "serial"))
def test_wrapped_every_tier(gpu_ctx, hipmod, oracle, tier):
    from fastqandfurious_amd import synth
    for data in (synth.wrapped(0, 9000 if tier != "serial" else 1500, seed=43)[0], synth.wrapped(777, 1200, seed=43)[0], synth.single(0, 800, seed=42)):
        gpu_ctx.forget()
        res = wide_same(gpu_ctx, hipmod, oracle, data, flags=tier_flags(hipmod, tier))
        assert res.path == TIER_PATH[tier], res.path
        # embedded newlines of a wrapped quality decode like any other byte: '\n' - 33
    rng = np.random.default_rng(5)
    data = np.frombuffer(random_records(rng, 3000 if tier != "serial" else 600, 1, 700, wrap=61), dtype=np.uint8)
    gpu_ctx.forget()
    wide_same(gpu_ctx, hipmod, oracle, data, flags=tier_flags(hipmod, tier), expect=TIER_PATH[tier])
    gpu_ctx.forget()
    wide_same(gpu_ctx, hipmod, oracle, data[:-1], flags=tier_flags(hipmod, tier))               # 'Incomplete final quality string'
    gpu_ctx.forget()
    wide_same(gpu_ctx, hipmod, oracle, data[:data.size * 2 // 3], flags=tier_flags(hipmod, tier), eof=False)
    gpu_ctx.forget()
    wide_same(gpu_ctx, hipmod, oracle, data[:data.size * 2 // 3], flags=tier_flags(hipmod, tier), eof=True)


def test_a_context_that_has_met_wrapped_records_takes_the_one_pass(gpu_ctx, hipmod, oracle):
    """No forcing flag: the first scan of a fresh context tries the four-line fast path, finds out, and decodes in two passes
    (packed); the context remembers, and the next scans start on the general kernels with the wide index pass."""
    from fastqandfurious_amd import synth
    gpu_ctx.forget()
    data = synth.wrapped(0, 20000, seed=43)[0]
    first = wide_same(gpu_ctx, hipmod, oracle, data)
    assert first.path == 0                                               # two passes, packed
    for k in range(3):
        nxt = wide_same(gpu_ctx, hipmod, oracle, synth.wrapped(20000 * (k + 1), 20000, seed=43)[0])
        assert nxt.path == 8, nxt.path
    # ... and without room for the in-place layout: two passes, whatever the context remembers
    want, *_ = oracle.scan(data)
    table, res, qual, qoff = gpu_ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS)      # (SEG_STRIDE per tile)
    wq, wqoff = oracle.decode_quals(data, want)
    assert res.path == 0 and (table == want).all() and (qoff == wqoff).all() and (qual == wq).all()
    # four-line input again: the general kernels (one pass still) until the context's probe finds out, then the fast path --
    # two passes while the back-off of its own single pass (refused on the wrapped input) runs down, then that one again
    single = synth.single(0, 60000, seed=42)
    paths = [wide_same(gpu_ctx, hipmod, oracle, single).path for _ in range(48)]
    assert paths[0] == 8 and paths[-1] == 6 and 3 in paths, paths


def test_dense_configuration_and_short_lines(gpu_ctx, hipmod, oracle):
    """Wrapped at few columns: tiles of many short lines (the dense LDS budget, path 2 | 8), down to lines under 16 bytes
    (dense tiles: entries in the overflow pool -- the WIDE index kernel is k_scan_lines itself, so they are handled)."""
    rng = np.random.default_rng(11)
    for wrap, n in ((40, 4000), (12, 6000), (5, 8000)):
        data = np.frombuffer(random_records(rng, n, 1, 300, wrap=wrap, hdr_hi=6), dtype=np.uint8)
        gpu_ctx.forget()
        res = wide_same(gpu_ctx, hipmod, oracle, data, flags=hipmod.F_FORCE_GENERAL)
        assert res.path & hipmod.PATH_IN_PLACE, (wrap, res.path)
    # the reference's own test template repeated (tests.py:8-35): 27-byte records, every tile dense
    data = np.frombuffer(b"@foo#2\nAATTGCCG\n+\n3425@!#!\n" * 30000, dtype=np.uint8)
    gpu_ctx.forget()
    res = wide_same(gpu_ctx, hipmod, oracle, data, flags=hipmod.F_FORCE_GENERAL)
    assert res.path & hipmod.PATH_IN_PLACE


def test_offsets_sentinel_add_and_quality_add(gpu_ctx, hipmod, oracle):
    from fastqandfurious_amd import synth
    data = synth.wrapped(5, 6000, seed=43)[0]
    want, *_ = oracle.scan(data)
    F = hipmod.F_FORCE_GENERAL
    for off in (1, int(want[7, 0]) + 1, int(want[3000, 5]), int(want[3000, 4]) + 2):
        gpu_ctx.forget()
        wide_same(gpu_ctx, hipmod, oracle, data, flags=F, offset=off, expect=8)
    gpu_ctx.forget()
    wide_same(gpu_ctx, hipmod, oracle, np.concatenate([np.array([10], np.uint8), data]), flags=F, sentinel=False, expect=8)
    gpu_ctx.forget()
    wide_same(gpu_ctx, hipmod, oracle, data, flags=F, add=5 * (1 << 32) + 12345, expect=8)
    for v in (0, 1, -128, 127, 200, -129):
        gpu_ctx.forget()
        wide_same(gpu_ctx, hipmod, oracle, data[:200000], flags=F, qual_add=v, expect=8)
    # empty, a buffer shorter than a tile, exactly one tile, one byte more
    gpu_ctx.forget()
    wide_same(gpu_ctx, hipmod, oracle, np.zeros(0, np.uint8), flags=F)
    for n in (100, 16384, 16385, 32768 + 7):
        gpu_ctx.forget()
        wide_same(gpu_ctx, hipmod, oracle, data[:n], flags=F)
        gpu_ctx.forget()
        wide_same(gpu_ctx, hipmod, oracle, data[:n], flags=F, eof=False)


def test_quality_lines_that_look_like_headers(gpu_ctx, hipmod, oracle):
    """The multiline golden (its second record's quality starts with '@') and the "tricky" stream of the shard tests (a
    1.5 MiB quality block of FASTQ-looking text): which lines are quality lines is the chain's business; the wide pass
    decodes every byte and cannot be fooled."""
    from conftest import golden_file
    from test_sharded import make_stream
    for tier in ("general", "ranked", "serial"):
        gpu_ctx.forget()
        wide_same(gpu_ctx, hipmod, oracle, np.frombuffer(golden_file("test_multiline.fq"), dtype=np.uint8), flags=tier_flags(hipmod, tier),
                  expect=TIER_PATH[tier])
    for tier in ("general", "ranked"):
        gpu_ctx.forget()
        res = wide_same(gpu_ctx, hipmod, oracle, make_stream("tricky"), flags=tier_flags(hipmod, tier))
        assert res.path & hipmod.PATH_IN_PLACE
    gpu_ctx.forget()
    res = wide_same(gpu_ctx, hipmod, oracle, make_stream("long-wrapped"), flags=hipmod.F_FORCE_GENERAL)
    assert res.path & hipmod.PATH_IN_PLACE


def test_random_edits_differential(gpu_ctx, hipmod, oracle):
    """Wrapped records under random byte edits (newlines dropped and added, '@' / '+' planted): rows, end state and the
    bytes of every complete record against the oracle, on whatever tier the scan ends up."""
    from fastqandfurious_amd import synth
    base = synth.wrapped(0, 1500, seed=43)[0]
    rng = np.random.default_rng(2024)
    for seed in range(40):
        d = base.copy()
        for _ in range(int(rng.integers(1, 30))):
            at = int(rng.integers(0, d.size))
            d[at] = rng.choice(np.frombuffer(b"\n@+AI", dtype=np.uint8))
        gpu_ctx.forget()
        a = np.frombuffer(d.tobytes(), dtype=np.uint8)
        want, end, *_ = oracle.scan(a)
        table, res, qual, qoff = gpu_ctx.scan_host(a, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS | hipmod.F_FORCE_GENERAL,
                                                   qual_room=hipmod.INPLACE_STRIDE)
        assert (table == want).all() and res.end_state == end, seed
        wq, wqoff = oracle.decode_quals(a, want)
        n = len(want)
        if n:
            lens = want[:, 5] - want[:, 4]
            idx = np.repeat(qoff[:n] - wqoff[:n], lens) + np.arange(wq.size)
            assert (qual[idx] == wq).all(), seed


def test_file_shards_of_wrapped_records_decode_in_one_pass(gpu_ctx, hipmod, oracle, tmp_path):
    """FileShard(qual_room=INPLACE_STRIDE).scan(decode=True) over a wrapped file, three logical ranks: the ranks' scans take
    the wide pass once their contexts have met the input (second scan), rows and qualities are the oracle's."""
    import threading
    from fastqandfurious_amd import sharded, synth
    data = synth.wrapped(0, 30000, seed=43)[0]
    p = tmp_path / "w.fq"
    p.write_bytes(data.tobytes())
    want, *_ = oracle.scan(data)
    wq, wqoff = oracle.decode_quals(data, want)
    world = 3
    sw = hipmod.ShardWorld(world)
    got, errs = [None] * world, []

    def work(rank):
        try:
            ctx = hipmod.Context(0)
            sh = sharded.FileShard(ctx, str(p), rank, world, comm=sw, qual_room=hipmod.INPLACE_STRIDE)
            paths = []
            for _ in range(2):
                res = sh.scan(decode=True)
                paths.append(int(res.scan.path))
            n = int(res.row_hi - res.row_lo)
            rows = sh.rows()
            q, qo = sh.quals(0, n, rows)
            got[rank] = (rows, q, qo, paths, int(res.record_base))
            sh.close()
            ctx.close()
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
            sw.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sw.close()
    assert not errs, errs
    assert (np.concatenate([g[0] for g in got]) == want).all()
    for rows, q, qo, paths, base in got:
        assert paths[1] & hipmod.PATH_IN_PLACE, paths
        n = len(rows)
        lens = rows[:, 5] - rows[:, 4]
        w0 = int(wqoff[base])
        idx = np.repeat(qo[:n], lens) + (np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens))
        assert (q[idx] == wq[w0:w0 + int(lens.sum())]).all()


def test_wrapped_at_config_size_decoded_in_one_pass(gpu_ctx, hipmod):
    """BASELINE configs[3]'s input (10 GiB S-wrapped) WITH the decode, one pass: every row against the generator's closed
    form (sharded.SyntheticShard.verify) and a spread of records' decoded bytes against the buffer's own bytes - 33."""
    import torch
    from fastqandfurious_amd import synthshard
    free, _total = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip("needs 60 GiB of device memory")
    dev = torch.device("cuda", 0)
    ctx = hipmod.Context(0)
    shard = synthshard.SyntheticShard(ctx, "wrapped", 10 << 30, 0, 1, dev)
    n = shard.ext.numel()
    ctx.reserve(n)
    table = torch.empty((shard.max_records + 64, 6), dtype=torch.int64, device=dev)
    qual = torch.empty(((n + 16383) >> 14) * hipmod.INPLACE_STRIDE, dtype=torch.int8, device=dev)
    qoff = torch.empty(table.shape[0] + 1, dtype=torch.int64, device=dev)
    flags = hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS
    out = shard.scan(table, flags=flags, qual=qual, qoff=qoff)          # (the first scan of the context: two passes)
    out = shard.scan(table, flags=flags, qual=qual, qoff=qoff)
    assert out.res.path == 8, out.res.path
    shard.verify(table, out)
    nrow = int(out.n_rows)
    assert bool((qoff[:nrow] == table[:nrow, 4] - (shard.own_lo - shard.tail)).all()), "in place: qoff[i] is pos4's offset in the buffer"
    assert int(qoff[nrow].item()) == int(out.res.n_qual_bytes) == int((table[nrow - 1, 5] - (shard.own_lo - shard.tail)).item())
    idx = torch.unique(torch.cat([torch.arange(0, 64, device=dev), torch.linspace(0, nrow - 1, 20000, device=dev).long(), torch.arange(nrow - 64, nrow, device=dev)]))
    for i in idx[::16].tolist():
        a, b = int(table[i, 4].item()), int(table[i, 5].item())
        q0 = int(qoff[i].item())
        src = (shard.ext[a:b].to(torch.int16) - 33).to(torch.int8)
        assert bool((src == qual[q0:q0 + b - a]).all()), i
    del table, qual, qoff, shard
    ctx.close()
    torch.cuda.empty_cache()
