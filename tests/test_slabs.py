"""A rank's range of a file that does NOT fit its GPU goes through ONE device buffer, slab after slab
(ffq_shard_scan_fd_slabs, sharded.FileShard(slab_bytes=...); round 6, VERDICT r5 missing #3) -- and the length filter of the
reference's user guide pushed down into the range iterator (missing #4).

What the reference does for any size of stream is the loop of /root/reference/src/fastqandfurious.py:251-279 -- scan the
buffer, keep buf[offset:] (the unfinished entry), read more (:274-279) --; over slabs that loop runs INSIDE each rank, the
rank's two edges are proven against its neighbours' as ever (eight words, one gather).  The invariant is the reference's
own: the entries do not depend on how the stream is cut into buffers (/root/reference/tests.py:219-226) -- rows over the
ranks, with slabs of 64 KiB ... 64 MiB, == the oracle's scan of the whole file; the ranks' iterators concatenated == the
reference's golden tuples.  The filter: /root/reference/doc/user-guide.rst:153-180, against the goldens the reference
itself produced running the guide's function (tests/golden/lengthfilter.json) and against the same object on the CPU."""
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_file
from test_fileshard import check, run_ranks, shm_file  # noqa: F401
from test_sharded import expected, make_stream

pytestmark = pytest.mark.gpu


def slab_rows(path, world, slab_bytes, bounds=None, **kw):
    from fastqandfurious_amd import sharded

    def work(rank, ctx, sw):
        sh = sharded.FileShard(ctx, path, rank, world, comm=sw, bounds=bounds, slab_bytes=slab_bytes, **kw)
        try:
            assert sh.d_ext is None and sh.load() == 0              # nothing resident
            res = sh.scan()
            assert res.d_ext is None and int(res.halo_source) == 1
            rows = sh.rows()
            return dict(rows=rows, base=int(res.record_base), total=int(res.total_records), rounds=int(res.rounds), source=1,
                        head=int(res.head), bounds=list(sh.bounds), transport=sh.sh.transport(), n_slabs=int(res.n_slabs),
                        bytes_read=int(res.bytes_read), n_view=sh.n_view)
        finally:
            sh.close()
    return run_ranks(world, work)


@pytest.mark.parametrize("name", ("test.fq", "test_longqualityheader.fq", "test_multiline.fq"))
@pytest.mark.parametrize("world", (1, 2, 3))
def test_golden_files_through_slabs(gpu_ctx, oracle, name, world):
    """The reference's three fixtures (458 ... 1617 bytes): the slab is larger than the file here -- one slab per rank, the
    edges forced into the records by tiny halos as well."""
    path = os.path.join(GOLDEN_DIR, "data", name)
    want, err = expected(oracle, np.frombuffer(golden_file(name), dtype=np.uint8))
    assert err is None
    for kw in ({}, dict(tail_bytes=40, head_bytes=24), dict(tail_bytes=1, head_bytes=1)):
        res = slab_rows(path, world, 1 << 16, **kw)
        check(res, want)


@pytest.mark.parametrize("kind,world", (("single", 1), ("single", 3), ("wrapped", 2), ("wrapped", 3), ("long", 2), ("long-wrapped", 3), ("tricky", 2), ("tricky", 3)))
def test_streams_through_small_slabs(gpu_ctx, oracle, shm_file, kind, world):
    """Slabs of 64 KiB ... 1 MiB over 4-8 MiB streams: dozens of slabs per rank, a record across every slab edge; "long":
    one record (3 MiB) longer than the slab -- the slab doubles until it holds it --; "tricky": a rank that enters inside a
    quality block of FASTQ-looking text streams its range again from its left neighbour's exit."""
    stream = make_stream(kind)
    want, err = expected(oracle, stream)
    assert err is None
    path = shm_file(stream, "ffq_slab_%s.fq" % kind)
    for slab in (1 << 16, 200000, 1 << 20):
        for kw in ({}, dict(tail_bytes=300, head_bytes=200)):
            res = slab_rows(path, world, slab, **kw)
            check(res, want)
            if slab == 1 << 16 and kind in ("single", "wrapped"):
                assert all(r["n_slabs"] >= r["n_view"] // (1 << 16) for r in res), [r["n_slabs"] for r in res]
    # the resident step and the slabs agree on the repair rounds (same words, same decision) -- except that a look-ahead
    # that must grow costs the resident step a round and the slabs none (the pass reads on)
    from test_fileshard import shard_rows
    a, b = shard_rows(path, world), slab_rows(path, world, 1 << 20)
    assert all(rb["rounds"] <= ra["rounds"] for ra, rb in zip(a, b))
    if kind in ("single", "wrapped"):
        assert [r["rounds"] for r in a] == [r["rounds"] for r in b] == [0] * world
    if kind in ("long", "long-wrapped"):
        assert any(ra["rounds"] > 0 for ra in a)
    if kind == "tricky":
        assert any(r["rounds"] > 0 for r in b)                        # a contradicted entry: the range streamed again


@pytest.mark.parametrize("kind", ("truncated", "cut-header", "invalid"))
def test_stream_errors_through_slabs(gpu_ctx, oracle, shm_file, kind):
    """The iterator's three ValueErrors, raised on every rank alike, whichever rank's slab meets the bad entry."""
    stream = make_stream(kind)
    _want, err = expected(oracle, stream)
    path = shm_file(stream, "ffq_slab_err.fq")
    for world in (1, 3):
        with pytest.raises(ValueError) as ei:
            slab_rows(path, world, 1 << 18)
        assert str(ei.value) == err


def test_table_too_small_is_grown_and_the_pass_repeated(gpu_ctx, oracle, shm_file):
    from fastqandfurious_amd import sharded, synth
    rng = np.random.default_rng(3)
    from test_gpu_parity import random_records
    data = np.frombuffer(random_records(rng, 40000, 1, 30, hdr_hi=4), dtype=np.uint8)      # ~45-byte records: 3.5 x the default estimate
    want, err = expected(oracle, data)
    path = shm_file(data, "ffq_slab_small.fq")

    def work(rank, ctx, sw):
        sh = sharded.FileShard(ctx, path, rank, 2, comm=sw, slab_bytes=1 << 17)
        try:
            res = sh.scan()
            return dict(rows=sh.rows(), base=int(res.record_base), total=int(res.total_records), rounds=0, source=1, head=0, bounds=list(sh.bounds),
                        transport="in-process", cap=sh.table_cap)
        finally:
            sh.close()
    res = run_ranks(2, work)
    check(res, want)
    assert all(r["cap"] > r["rows"].shape[0] for r in res)


def test_one_gib_file_three_ranks_64m_slabs(gpu_ctx, shm_file):
    """A 1 GiB S-single file, k = 1, 2, 3 ranks, 64 MiB slabs (the VERDICT's acceptance case): every row against the
    generator's closed form, ordinals, slab counts; the bytes each rank read are its range's (+ halos), once."""
    import torch
    from fastqandfurious_amd import synth
    n = (1 << 30) // 322
    p = shm_file(b"", "ffq_slab_1g.fq")
    blk = 200000
    with open(p, "wb") as fh:
        for k0 in range(0, n, blk):
            fh.write(synth.single(k0, min(blk, n - k0), seed=42).tobytes())
    size = os.path.getsize(p)
    assert size == n * 322
    cols = np.array([0, 17, 18, 168, 171, 321], dtype=np.int64)
    for world in (1, 2, 3):
        res = slab_rows(p, world, 64 << 20)
        base = 0
        for r, out in enumerate(res):
            rows = out["rows"]
            b = out["bounds"]
            k0, k1 = -(-b[r] // 322), (-(-b[r + 1] // 322) if r + 1 < world else n)
            assert rows.shape[0] == k1 - k0 and out["base"] == base == k0 and out["total"] == n
            assert (rows == (np.arange(k0, k1, dtype=np.int64) * 322)[:, None] + cols[None, :]).all()
            assert out["n_slabs"] == -(-out["n_view"] // (64 << 20)) or out["n_slabs"] == -(-out["n_view"] // (64 << 20)) + 1
            assert out["n_view"] <= out["bytes_read"] <= out["n_view"] + out["n_slabs"] * 400       # (the carried record of every slab, again)
            base += rows.shape[0]
    torch.cuda.empty_cache()


def test_range_iterator_through_slabs_matches_the_reference_tuples(gpu_ctx, golden, oracle, shm_file):
    """readfastq_iter_range(..., slab_bytes=) == the reference's golden tuples; entryfunc_phred over slabs (nothing resident)
    has its qualities decoded on the device batch by batch (FileShard.quals_from_file) and equals the oracle's decode and the
    resident range's entries."""
    from array import array
    from fastqandfurious_amd import fastqandfurious as F, hip, synth

    def entries(path, world, ef, **kw):
        def work(rank, ctx, sw):
            return list(F.readfastq_iter_range(path, rank, world, ef, comm=sw, ctx=ctx, **kw))
        return [e for part in run_ranks(world, work) for e in part]
    for name, g in golden["files"].items():
        path = os.path.join(GOLDEN_DIR, "data", name)
        for world in (1, 2, 3):
            got = entries(path, world, F.entryfunc, slab_bytes=1 << 16, tail_bytes=50, head_bytes=30)
            assert [[h.hex(), s_.hex(), q.hex()] for h, s_, q in got] == g["tuples"], (name, world)
            rows = entries(path, world, lambda buf, pos, off: [int(p) + off for p in pos], slab_bytes=1 << 16)
            assert rows == g["bufsizes"]["65536"]["c"]["rows"], (name, world)
    data = synth.wrapped(0, 6000, seed=43)[0]
    p = shm_file(data, "ffq_slab_phred.fq")
    table, *_ = oracle.scan(data)
    wq, wqoff = oracle.decode_quals(data, table)
    got = entries(p, 3, F.entryfunc_phred, slab_bytes=1 << 17)
    assert len(got) == len(table)
    for i in (0, 1, 2999, 5999):
        assert got[i][2] == array("b", wq[int(wqoff[i]):int(wqoff[i + 1])].tobytes())
    resident = entries(p, 3, F.entryfunc_phred)
    assert resident == got


# ---- the length filter in the range iterator ------------------------------------------------------------------------------
def _lf_golden():
    import json
    with open(os.path.join(GOLDEN_DIR, "lengthfilter.json")) as fh:
        return json.load(fh)


def _enc(x):
    if x is None:
        return None
    if isinstance(x, tuple):
        return [bytes(y).hex() for y in x]
    return bytes(x).hex()


@pytest.mark.parametrize("slab", (None, 1 << 16))
def test_lengthfilter_pushed_down_into_the_range_iterator(gpu_ctx, shm_file, slab):
    """readfastq_iter_range(path, rank, world, entryfunc_lengthfilter(...)): rows filtered on the device, only the kept rows
    (and, from a resident range, their gathered component) cross the link; the ranks' items concatenated == what the
    REFERENCE's iterator yielded running the user guide's function (golden), for every threshold, column, and with the
    dropped records left out; a resident range and slabs alike."""
    from fastqandfurious_amd import fastqandfurious as F, synth
    lf = _lf_golden()

    def items(path, world, ef):
        def work(rank, ctx, sw):
            kw = dict(slab_bytes=slab) if slab else {}
            it = F.readfastq_iter_range(path, rank, world, ef, comm=sw, ctx=ctx, tail_bytes=64, head_bytes=48, **kw)
            return list(it)
        return [e for part in run_ranks(world, work) for e in part]
    for fn, d in lf["files"].items():
        path = os.path.join(GOLDEN_DIR, "data", fn)
        for th, v in d.items():
            for world in (1, 3):
                assert [_enc(x) for x in items(path, world, F.entryfunc_lengthfilter(int(th)))] == v["items"], (fn, th, world)
            for c in ("header", "quality", "entry"):
                assert [_enc(x) for x in items(path, 2, F.entryfunc_lengthfilter(int(th), column=c))] == v["columns"][c], (fn, th, c)
                assert [_enc(x) for x in items(path, 2, F.entryfunc_lengthfilter(int(th), column=c, yield_dropped=False))] == \
                    [x for x in v["columns"][c] if x is not None], (fn, th, c)
    # a larger file: against the same object called per record by the pure-Python scanner
    data = synth.wrapped(0, 30000, seed=43)[0]
    p = shm_file(data, "ffq_range_lf.fq")
    for kw in (dict(threshold=120), dict(min_len=100, max_len=250, column="quality"), dict(threshold=90, column="entry", yield_dropped=False),
               dict(min_len=10 ** 6), dict(min_len=0)):
        ef = F.entryfunc_lengthfilter(**kw)
        want = list(F.readfastq_iter(io.BytesIO(data.tobytes()), 1 << 20, ef, F.entrypos))
        for world in (1, 3):
            assert items(p, world, ef) == want, (kw, world)
