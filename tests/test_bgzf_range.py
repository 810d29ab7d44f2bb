"""BGZF files read by RANGES (missing #4: "BGZF files -- block-seekable, the natural compressed shard -- cannot
be range-read").

bgzip writes a gzip file whose members hold at most 64 KiB each and say in their header how long they are; k ranks take the
members that BEGIN in their share of the compressed file (ffq_bgzf_range finds the first one by its signature and the chain
of headers behind it, inflates them side by side on the host), one exchange of the sizes gives the cut points of the
uncompressed stream, and the ordinary sharded step runs over the inflated ranges with the halos handed over between the ranks
(sharded.BgzfFileShard, fastqandfurious.readfastq_iter_range).  What it stands for: the reference's loop over
gzip.open(...) (/root/reference/src/fastqandfurious.py:241-279; :290-334 automagic_open maps .gz to gzip.open) -- rows and
entries are those of the UNCOMPRESSED stream; invariance under the cut: /root/reference/tests.py:219-226.

CPU: ffq_bgzf_range against zlib / gzip on random ranges, and what it says about files that are not BGZF, cut short or
corrupt.  GPU: the ranks' rows against the oracle's scan of the inflated bytes, the ranks' entries against the reference's
golden tuples."""
import gzip
import os
import struct
import threading
import zlib
from array import array

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_file


@pytest.fixture()
def tmp_bgzf(tmp_path):
    def make(data, name="x.fq.gz", **kw):
        from fastqandfurious_amd import bgzf
        p = str(tmp_path / name)
        with open(p, "wb") as fh:
            fh.write(bgzf.compress(bytes(data), **kw))
        return p
    return make


def ranges(fd, world, threads=3):
    from fastqandfurious_amd import hip
    size = os.fstat(fd).st_size
    parts, meta = [], []
    for r in range(world):
        lo, hi = size * r // world, size * (r + 1) // world
        sized = hip.bgzf_range(fd, lo, hi)
        out = np.empty(sized[2], dtype=np.uint8)
        assert hip.bgzf_range(fd, lo, hi, out=out, threads=threads) == sized
        parts.append(out.tobytes())
        meta.append(sized)
    return parts, meta


@pytest.fixture(params=("mapped", "read"))
def how(request, monkeypatch):
    """ffq_bgzf_range walks and inflates the members where the page cache holds them (mmap); a descriptor that cannot be mapped is
    read in windows (FFQ_BGZF_NO_MMAP=1 forces that path)."""
    if request.param == "read":
        monkeypatch.setenv("FFQ_BGZF_NO_MMAP", "1")
    return request.param


@pytest.mark.parametrize("block_bytes", (100, 7000, 65280))
def test_ranges_of_members_add_up_to_the_file(pkg, tmp_bgzf, block_bytes, how):
    """Whatever the number of ranks and the size of the members: the ranks' members follow each other without a gap (rank r's
    end is rank r + 1's first), their bytes concatenated are what gzip makes of the whole file, and the sizing pass (nothing
    inflated) promises exactly what the inflating pass delivers."""
    rng = np.random.default_rng(block_bytes)
    data = bytes(rng.integers(33, 75, 400000 if block_bytes > 100 else 30000, dtype=np.uint8))
    path = tmp_bgzf(data, block_bytes=block_bytes)
    assert gzip.open(path).read() == data
    fd = os.open(path, os.O_RDONLY)
    try:
        size = os.fstat(fd).st_size
        for world in (1, 2, 3, 7, 64):
            parts, meta = ranges(fd, world)
            assert b"".join(parts) == data
            assert meta[0][0] == 0 and meta[-1][1] == size
            assert all(meta[r][1] == meta[r + 1][0] for r in range(world - 1))
            assert sum(m[3] for m in meta) == -(-len(data) // block_bytes) + 1          # (+ the empty end-of-file member)
    finally:
        os.close(fd)


def test_members_with_other_extra_subfields_and_no_eof_marker(pkg, tmp_path, how):
    """The "BC" subfield need not be the only one in a member's extra field, nor the first; the empty member at the end is a
    convention, not a must; a range that begins in the last member's tail holds nothing."""
    from fastqandfurious_amd import bgzf, hip
    chunks = [bytes([65 + i % 20]) * (500 + 37 * i) for i in range(40)]
    blob = b"".join(bgzf.block(c, extra_before=b"XY\x03\x00abc" if i % 3 == 0 else b"", extra_after=b"ZZ\x01\x00q" if i % 4 == 0 else b"")
                    for i, c in enumerate(chunks))
    path = str(tmp_path / "extra.gz")
    open(path, "wb").write(blob)
    fd = os.open(path, os.O_RDONLY)
    try:
        for world in (1, 3, 11):
            parts, meta = ranges(fd, world)
            assert b"".join(parts) == b"".join(chunks)
        assert hip.bgzf_range(fd, len(blob) - 5, len(blob)) == (len(blob), len(blob), 0, 0)
        assert hip.bgzf_range(fd, len(blob), len(blob) + 100)[2:] == (0, 0)
    finally:
        os.close(fd)


def test_what_is_not_bgzf_is_said_so(pkg, tmp_path, tmp_bgzf, how):
    """A plain gzip file, a BGZF file cut short, one whose member lies about its length or does not match its CRC-32: errors
    of the kinds Python's gzip raises (OSError / EOFError), never bytes."""
    from fastqandfurious_amd import hip, sharded
    data = bytes(np.random.default_rng(3).integers(33, 75, 200000, dtype=np.uint8))
    plain = str(tmp_path / "plain.gz")
    with gzip.open(plain, "wb") as fh:
        fh.write(data)
    good = tmp_bgzf(data, block_bytes=9000)
    assert sharded.is_bgzf(good) and not sharded.is_bgzf(plain)
    blob = open(good, "rb").read()

    def try_file(content, lo_frac=0.0, hi_frac=1.0):
        p = str(tmp_path / "bad.gz")
        open(p, "wb").write(content)
        fd = os.open(p, os.O_RDONLY)
        try:
            n = len(content)
            sized = hip.bgzf_range(fd, int(n * lo_frac), int(n * hi_frac))
            out = np.empty(sized[2], dtype=np.uint8)
            return hip.bgzf_range(fd, int(n * lo_frac), int(n * hi_frac), out=out)
        finally:
            os.close(fd)
    fd = os.open(plain, os.O_RDONLY)
    try:
        with pytest.raises(hip.FFQGzipError, match="no BGZF member"):
            hip.bgzf_range(fd, 0, 1 << 30)
    finally:
        os.close(fd)
    with pytest.raises(EOFError, match="ended before the end-of-stream marker"):
        try_file(blob[:len(blob) - 4000])
    with pytest.raises(EOFError):
        try_file(blob[:len(blob) - 4000], 0.5, 1.0)                    # (the rank that holds the torn member says so)
    assert try_file(blob[:len(blob) - 4000], 0.0, 0.3)[2] > 0           # (the others have their bytes)
    first = struct.unpack("<H", blob[16:18])[0] + 1
    bad_crc = bytearray(blob)
    bad_crc[first - 8] ^= 0x55                                           # CRC-32 of the first member
    with pytest.raises(hip.FFQGzipError, match="does not inflate to what its trailer says"):
        try_file(bytes(bad_crc))
    bad_len = bytearray(blob)
    bad_len[first - 4:first] = struct.pack("<I", 9001)                   # ISIZE of the first member: one byte too many
    with pytest.raises(hip.FFQGzipError, match="does not inflate to what its trailer says"):
        try_file(bytes(bad_len))
    at = 0
    for _ in range(8):                                                   # (behind the eighth member: the chain that proves byte 0 holds)
        at += struct.unpack("<H", blob[at + 16:at + 18])[0] + 1
    garbage = blob[:at] + b"\x00" * 50 + blob[at:]
    with pytest.raises(hip.FFQGzipError, match="not a BGZF member"):
        try_file(garbage)
    small = np.empty(10, dtype=np.uint8)
    fd = os.open(good, os.O_RDONLY)
    try:
        with pytest.raises(hip.FFQError, match="inflates to more than"):
            hip.bgzf_range(fd, 0, len(blob), out=small)
    finally:
        os.close(fd)


# ---- the sharded step over the inflated ranges (GPU) -------------------------------------------------------------------------
def run_ranks(world, fn):
    """fn(rank, ctx, shard_world, exchange) on `world` threads, a context each (tests/test_fileshard.py's runner plus the
    in-process exchange of the ranks' sizes)."""
    from fastqandfurious_amd import hip, sharded
    sw = hip.ShardWorld(world)
    lw = sharded.LocalWorld(world)
    results, errors = [None] * world, [None] * world

    def work(rank):
        ctx = None
        try:
            ctx = hip.Context(0)
            results[rank] = fn(rank, ctx, sw, sharded.LocalTransport(lw, rank).allgather)
        except BaseException as e:      # noqa: BLE001
            errors[rank] = e
            sw.abort()
            lw.abort()
        finally:
            if ctx is not None:
                ctx.close()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sw.close()
    real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)
            and "another logical rank failed" not in str(e)]
    if real:
        raise real[0]
    assert not any(errors), errors
    return results


def shard_rows(path, world, decode=False, **kw):
    from fastqandfurious_amd import sharded

    def work(rank, ctx, sw, exchange):
        sh = sharded.BgzfFileShard(ctx, path, rank, world, comm=sw, exchange=exchange, threads=2, **kw)
        try:
            sh.load()
            res = sh.scan(decode=decode)
            rows = sh.rows()
            q = sh.quals(0, rows.shape[0], rows) if decode and rows.shape[0] else None
            base, host = sh.host_bytes()
            return dict(rows=rows, base=int(res.record_base), total=int(res.total_records), rounds=int(res.rounds), source=int(res.halo_source),
                        bounds=list(sh.bounds), quals=q, host=(base, host.tobytes()), members=list(sh.members))
        finally:
            sh.close()
    return run_ranks(world, work)


def check(results, want, data):
    got = np.concatenate([r["rows"] for r in results])
    assert got.shape == want.shape and (got == want).all(), "rows over the ranks differ from the scan of the inflated file"
    base = 0
    for r, res in enumerate(results):
        b = res["bounds"]
        assert b[0] == 0 and b[-1] == len(data) and res["source"] == 0              # (halos from the neighbours, not from a file)
        lo = -1 if b[r] == b[0] else b[r]
        hi = (1 << 62) if b[r + 1] == b[-1] else b[r + 1]
        mine = want[(want[:, 0] >= lo) & (want[:, 0] < hi)] if b[r + 1] > b[r] else want[:0]
        assert res["rows"].shape == mine.shape and (res["rows"] == mine).all(), "rank %d owns other records than those starting in its range" % r
        assert res["base"] == base and res["total"] == len(want)
        base += len(mine)
        hb, hbytes = res["host"]
        assert hb == b[r] and hbytes == bytes(data[hb:hb + len(hbytes)])
        if len(mine):
            assert hb + len(hbytes) >= int(mine[-1][5]) + 1, "rank %d: its last record's bytes are not all on the host" % r


@pytest.mark.gpu
@pytest.mark.parametrize("world", (1, 2, 3, 8))
@pytest.mark.parametrize("name", ("test.fq", "test_longqualityheader.fq", "test_multiline.fq"))
def test_golden_files_compressed_over_ranks(gpu_ctx, oracle, golden, tmp_bgzf, name, world):
    """The reference's data files as BGZF with members of 300 bytes (dozens of members: every rank has some), cut into 1 / 2 / 3
    / 8 shares of the COMPRESSED file: the ranks' rows are the oracle's scan of the inflated bytes = the reference's golden
    rows; with 40-byte halos the look-aheads grow over the hand-off."""
    from test_sharded import expected
    data = golden_file(name)
    want, err = expected(oracle, np.frombuffer(data, dtype=np.uint8))
    assert err is None and [list(map(int, r)) for r in want] == golden["files"][name]["bufsizes"]["65536"]["c"]["rows"]
    path = tmp_bgzf(data, block_bytes=300)
    for kw in ({}, dict(tail_bytes=40, head_bytes=24)):
        res = shard_rows(path, world, **kw)
        check(res, want, data)
        assert sum(res[0]["members"]) == -(-len(data) // 300) + 1
    if world > 1:
        assert any(r["rounds"] > 0 for r in res), "40-byte halos: no edge grew its look-ahead"


@pytest.mark.gpu
@pytest.mark.parametrize("kind,world", (("single", 3), ("wrapped", 5), ("long", 2), ("tricky", 8)))
def test_synthetic_files_compressed_over_ranks(gpu_ctx, oracle, tmp_bgzf, kind, world):
    """Larger inputs at bgzip's own member size, with the decode: rows and decoded qualities against the oracle."""
    from test_sharded import expected, make_stream
    stream = make_stream(kind)
    want, err = expected(oracle, stream)
    assert err is None
    path = tmp_bgzf(stream.tobytes(), level=1)
    wq, wqoff = oracle.decode_quals(stream, want)
    for decode in (False, True):
        res = shard_rows(path, world, decode=decode)
        check(res, want, stream.tobytes())
        if decode:
            got = []
            for r in res:
                if r["quals"] is None:
                    continue
                q, qo = r["quals"]
                rows = r["rows"]
                ln = rows[:, 5] - rows[:, 4]
                ix = np.repeat(qo[:len(rows)], ln) + (np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln))
                got.append(q[ix])
            assert np.array_equal(np.concatenate(got), wq)


@pytest.mark.gpu
def test_stream_errors_inside_a_compressed_file(gpu_ctx, oracle, tmp_bgzf):
    """The iterator's ValueErrors name bytes of the UNCOMPRESSED stream, on every rank alike."""
    from test_sharded import expected, make_stream
    for kind in ("truncated", "invalid"):
        stream = make_stream(kind)
        want, err = expected(oracle, stream)
        assert err is not None
        path = tmp_bgzf(stream.tobytes(), block_bytes=20000)
        with pytest.raises(ValueError) as ei:
            shard_rows(path, 3)
        assert str(ei.value).startswith(err), (str(ei.value), err)


@pytest.mark.gpu
def test_range_iterator_over_bgzf_matches_reference_tuples(gpu_ctx, golden, tmp_bgzf):
    """readfastq_iter_range over the reference's files compressed with BGZF (recognised by their first member): the ranks'
    entries, concatenated, are the tuples the reference's readfastq_iter yields over the plain file -- what it yields over
    gzip.open(...) of this one --, entryfunc_abspos the golden rows, entryfunc_phred the decoded arrays, the user guide's length
    filter pushed down."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C
    for name, g in golden["files"].items():
        data = golden_file(name)
        path = tmp_bgzf(data, name=name + ".gz", block_bytes=400)
        with gzip.open(path, "rb") as fh:
            ref_phred = list(F.readfastq_iter(fh, 1 << 20, F.entryfunc_phred, C.entrypos))
        for world in (1, 2, 3, 8):
            for kw in ({}, dict(tail_bytes=33, head_bytes=17)):
                def work(rank, ctx, sw, exchange, entryfunc=F.entryfunc):
                    it = F.readfastq_iter_range(path, rank, world, entryfunc, comm=sw, ctx=ctx, exchange=exchange, batch_rows=5, **kw)
                    assert it.comm["halo_source"] == "ranks" and it.total_records == len(g["tuples"])
                    out = [e if not isinstance(e, array) else list(e) for e in it]
                    assert len(out) == it.n_records
                    return it.record_base, out
                res = run_ranks(world, work)
                assert [b for b, _ in res] == [sum(len(o) for _, o in res[:r]) for r in range(world)]
                assert [[h.hex(), s.hex(), q.hex()] for _, o in res for h, s, q in o] == g["tuples"], (name, world, kw)
                res = run_ranks(world, lambda rank, ctx, sw, ex: work(rank, ctx, sw, ex, F.entryfunc_abspos))
                assert [r for _, o in res for r in o] == g["bufsizes"]["65536"]["c"]["rows"], (name, world, kw)
            res = run_ranks(world, lambda rank, ctx, sw, ex: work(rank, ctx, sw, ex, F.entryfunc_phred))
            assert [e for _, o in res for e in o] == ref_phred
            flt = F.entryfunc_lengthfilter(30, column="sequence")
            with gzip.open(path, "rb") as fh:
                ref_f = list(F.readfastq_iter(fh, 1 << 20, F.entryfunc_lengthfilter(30, column="sequence"), C.entrypos))
            res = run_ranks(world, lambda rank, ctx, sw, ex: work(rank, ctx, sw, ex, flt))
            assert [e for _, o in res for e in o] == ref_f
    with pytest.raises(ValueError, match="do not apply to a BGZF file"):
        F.readfastq_iter_range(path, 0, 1, start=5)
    plain_gz = str(tmp_bgzf(b"", name="dir.marker")) + ".plain.gz"
    with gzip.open(plain_gz, "wb") as fh:
        fh.write(golden_file("test.fq"))
    with pytest.raises(ValueError, match="is a gzip file but not BGZF"):
        F.readfastq_iter_range(plain_gz, 0, 1)
