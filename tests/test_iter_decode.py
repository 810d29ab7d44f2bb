"""Iterator-level Phred decode, coalesced reads and the three kinds of stream source.

The reference's documented decode is an entryfunc of the user's own (doc/user-guide.rst:126-141,
:206-214): array('b').frombytes(buf[pos4:pos5]); arrayadd_b(quality, -33).  tests/golden/phred.json
holds what the REAL reference yields with that entryfunc (make_golden_phred.py: reference iterator,
reference C scanner, reference arrayadd_b).  Here `entryfunc_phred` must yield the same entries
through the GPU stream's bulk decode -- from a plain file, a gzip file (inflated by the library's
reader thread), and from objects only Python can read (BytesIO, bz2, lzma: chunks pushed into
pinned memory) -- at the reference's own buffer sizes (20-50 kB, coalesced) and at tiny ones (every
fill carries an unfinished entry over).
"""
import bz2
import gzip
import hashlib
import io
import json
import lzma
import os
from array import array

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_file


@pytest.fixture(scope="module")
def phred():
    with open(os.path.join(GOLDEN_DIR, "phred.json")) as fh:
        return json.load(fh)


def _hexed(entries):
    return [[h.hex(), s.hex(), q.tobytes().hex()] for h, s, q in entries]


def _digest(entries):
    h = hashlib.sha256()
    for a, b, c in entries:
        for x in (a, b, c.tobytes()):
            h.update(len(x).to_bytes(8, "little"))
            h.update(x)
    return h.hexdigest()


def _sources(tmp_path, data, tag):
    """(name, opener) pairs: every kind of stream source over the same bytes."""
    plain = tmp_path / ("%s.fq" % tag)
    plain.write_bytes(data)
    gz = tmp_path / ("%s.fq.gz" % tag)
    with gzip.open(gz, "wb") as fh:
        fh.write(data)
    # two members + zero padding, as `cat a.gz b.gz` / bgzip produce
    gz2 = tmp_path / ("%s.2.fq.gz" % tag)
    cut = len(data) // 2
    gz2.write_bytes(gzip.compress(data[:cut]) + gzip.compress(data[cut:]) + b"\0" * 37)
    # BGZF (bgzip): members that say how long they are -- the library inflates them side by side
    from fastqandfurious_amd import bgzf
    bg = tmp_path / ("%s.fq.bgz" % tag)
    bg.write_bytes(bgzf.compress(data, block_bytes=min(65280, max(64, len(data) // 7))))
    bz = tmp_path / ("%s.fq.bz2" % tag)
    bz.write_bytes(bz2.compress(data))
    xz = tmp_path / ("%s.fq.xz" % tag)
    xz.write_bytes(lzma.compress(data))
    return [("file", lambda: open(plain, "rb")), ("gzip", lambda: gzip.open(gz, "rb")),
            ("gzip-members", lambda: gzip.open(gz2, "rb")), ("bgzf", lambda: gzip.open(bg, "rb")),
            ("bytesio", lambda: io.BytesIO(data)),
            ("bz2", lambda: bz2.open(bz, "rb")), ("xz", lambda: lzma.open(xz, "rb")),
            ("gzip-over-bytesio", lambda: gzip.GzipFile(fileobj=io.BytesIO(gz.read_bytes())))]


def test_entryfunc_phred_per_record_shape(pkg):
    """The per-record form is the user guide's entryfunc (it needs the device for arrayadd_b: here only
    that it is wired to this package's arrayadd_b and cuts the same slices)."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C
    import inspect
    src = inspect.getsource(F.entryfunc_phred)
    assert "arrayadd_b(quality, -33)" in src and "frombytes(buf[pos[4]:pos[5]])" in src
    assert C.entrypos.chunk_bytes(50000) % 50000 == 0 and C.entrypos.chunk_bytes(50000) >= C.entrypos.coalesce_bytes
    assert C.entrypos.chunk_bytes(1 << 26) == 1 << 26


@pytest.mark.gpu
@pytest.mark.parametrize("coalesce", (0, 8 << 20))
def test_phred_goldens_every_source(gpu_ctx, phred, tmp_path, coalesce):
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C
    C.entrypos.coalesce_bytes = coalesce
    cases = [(fn, golden_file(fn), g) for fn, g in phred["files"].items()]
    cases += [(name, bytes.fromhex(g["data"]), g) for name, g in phred["edge"].items()]
    for name, data, g in cases:
        for src, opener in _sources(tmp_path, data, name.replace(".", "_")):
            for bs in ((100, 600, 20000) if not coalesce else (20000,)):
                got, err = [], None
                try:
                    with opener() as fh:
                        for e in F.readfastq_iter(fh, bs, F.entryfunc_phred, C.entrypos):
                            assert isinstance(e[2], array) and e[2].typecode == "b"
                            got.append(e)
                except ValueError as e:
                    err = str(e)
                want_err = g["error"]
                if want_err == "hang":              # INVALID at eof: the reference never leaves its loop; this build raises
                    assert err is not None and err.startswith("Entry is invalid at byte"), (name, src, bs, err)
                else:
                    assert err == want_err, (name, src, bs, err)
                assert _hexed(got) == g["entries"], (name, src, bs)


@pytest.mark.gpu
@pytest.mark.parametrize("name,maker", (("single_3000", lambda s: s.single(0, 3000, seed=42)),
                                        ("wrapped_3000", lambda s: s.wrapped(0, 3000, seed=43)[0]),
                                        ("single_3000_at_7", lambda s: s.single(7, 3000, seed=42))))
@pytest.mark.parametrize("pgz", (False, True))
def test_phred_synthetic_streams(gpu_ctx, phred, tmp_path, name, maker, pgz, monkeypatch):
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, synth, hip
    data = maker(synth).tobytes()
    g = phred["synth"][name]
    if pgz:
        # plain gzip members through the several-thread inflate (csrc/ffq_pgz.h) although they are small: chunks of 20 KB
        monkeypatch.setenv("FFQ_PGZ_MIN", "1")
        monkeypatch.setenv("FFQ_PGZ_CHUNK", "20000")
        monkeypatch.setenv("FFQ_GZ_THREADS", "4")
    stats0 = hip.gunzip_stats()
    for coalesce in (0, 8 << 20):
        C.entrypos.coalesce_bytes = coalesce
        for src, opener in _sources(tmp_path, data, name):
            for bs in (50000, 20000, 3000):
                with opener() as fh:
                    got = list(F.readfastq_iter(fh, bs, F.entryfunc_phred, C.entrypos))
                assert len(got) == g["n"] and _digest(got) == g["sha256"], (name, src, bs, coalesce)
                assert [x.hex() for x in (got[0][0], got[0][1], got[0][2].tobytes())] == g["first"]
                assert [x.hex() for x in (got[-1][0], got[-1][1], got[-1][2].tobytes())] == g["last"]
    stats1 = hip.gunzip_stats()
    if pgz:
        assert stats1["chunks"] > stats0["chunks"] + 100 and stats1["giveups"] == stats0["giveups"]
    else:
        assert stats1 == stats0


@pytest.mark.gpu
def test_live_source_is_not_held_back_by_coalescing(gpu_ctx):
    """A source that trickles (a socket, stdin: every read comes back short) with the default coalescing (8 MiB per
    device call): the first entries must come out once fbufsize bytes are in, not after 8 MiB have arrived -- the
    reference's loop yields after every read of fbufsize bytes (fastqandfurious.py:222-232)."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, synth
    data = synth.single(0, 4000, seed=42).tobytes()          # 1.29 MB

    class Trickle:
        def __init__(self, blob, piece):
            self.blob, self.piece, self.at = blob, piece, 0

        def readinto(self, mv):
            n = min(len(mv), self.piece, len(self.blob) - self.at)
            mv[:n] = self.blob[self.at:self.at + n]
            self.at += n
            return n

    C.entrypos.coalesce_bytes = 8 << 20
    src = Trickle(data, 997)
    it = F.readfastq_iter(src, 20000, F.entryfunc, C.entrypos)
    first = next(it)
    assert src.at <= 3 * 20000, src.at                 # (a fill or two of the caller's size, not the whole megabyte)
    got = [first] + list(it)
    want = list(F.readfastq_iter(__import__("io").BytesIO(data), 20000, F.entryfunc, F.entrypos))
    assert got == want


@pytest.mark.gpu
def test_phred_iterator_takes_the_single_pass(gpu_ctx, phred, tmp_path):
    """What the decoding iterator runs on: the stream readfastq_iter opens for entryfunc_phred asks for the single pass
    (index + decoded qualities from one read of the fill, res.path 6) and gets it on four-line input; wrapped records
    come through the two passes, packed -- the entries are the reference's either way (phred.json)."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, synth
    C.entrypos.coalesce_bytes = 8 << 20
    for name, want_path in (("single_3000", (6,)), ("wrapped_3000", (0, 2))):
        data = (synth.single(0, 3000, seed=42) if name.startswith("single") else synth.wrapped(0, 3000, seed=43)[0]).tobytes()
        p = tmp_path / (name + ".fq")
        p.write_bytes(data)
        g = phred["synth"][name]
        with open(p, "rb") as fh:
            st = C.entrypos.open_stream(fh, 50000, True)
            paths, got = set(), []
            for rows, fill, fill_offset, end_state, err in st:
                paths.add(st.path())
                for chunk in F._phred_entries(st, fill, rows, fill_offset):      # (one list of entries per native call)
                    got.extend(chunk)
            st.close()
        assert paths <= set(want_path) and paths, (name, paths)
        assert len(got) == g["n"] and _digest(got) == g["sha256"], name


@pytest.mark.gpu
def test_default_entryfunc_every_source_and_coalescing(gpu_ctx, golden, tmp_path):
    """The default entryfunc through every kind of source, coalesced and not: the golden tuples of the
    reference's fixtures (captured from the reference, golden.json)."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C
    for fn, g in golden["files"].items():
        data = golden_file(fn)
        for coalesce in (0, 8 << 20):
            C.entrypos.coalesce_bytes = coalesce
            for src, opener in _sources(tmp_path, data, fn.replace(".", "_")):
                for bs in (100, 700, 50000):
                    with opener() as fh:
                        got = [[h.hex(), s.hex(), q.hex()] for h, s, q in F.readfastq_iter(fh, bs, F.entryfunc, C.entrypos)]
                    assert got == g["tuples"], (fn, src, bs, coalesce)
                    with opener() as fh:
                        rows = [list(p) for p in F.readfastq_iter(fh, bs, F.entryfunc_abspos, C.entrypos)]
                    assert rows == g["bufsizes"]["65536"]["c"]["rows"], (fn, src, bs, coalesce)


@pytest.mark.gpu
def test_gzip_errors_and_pipe_stop(gpu_ctx, tmp_path):
    """A truncated / corrupt gzip file raises (it does not hang or return a short stream); a pipe whose
    writer stays open and idle does not block close()."""
    import time
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, hip, synth
    data = synth.single(0, 2000, seed=42).tobytes()
    blob = gzip.compress(data)
    bad = tmp_path / "cut.fq.gz"
    bad.write_bytes(blob[:len(blob) // 2])
    with pytest.raises(hip.FFQError, match="gzip"):
        with gzip.open(bad, "rb") as fh:
            list(F.readfastq_iter(fh, 50000, F.entryfunc, C.entrypos))
    # ... as the exceptions a caller of gzip.open() catches: EOFError for a file cut short, an OSError (BadGzipFile is one)
    # for corrupt data
    with pytest.raises(EOFError):
        with gzip.open(bad, "rb") as fh:
            list(F.readfastq_iter(fh, 50000, F.entryfunc, C.entrypos))
    notgz = tmp_path / "not.fq.gz"
    notgz.write_bytes(data)
    with pytest.raises(hip.FFQError, match="gzip"):
        st = hip.FileStream(gpu_ctx, os.open(notgz, os.O_RDONLY), 1 << 20, gzip=True)
        list(st)
    with pytest.raises(OSError):
        st = hip.FileStream(gpu_ctx, os.open(notgz, os.O_RDONLY), 1 << 20, gzip=True)
        list(st)
    # a gzip file object the library inflated itself is left exhausted, as the reference's loop leaves it
    good = tmp_path / "good.fq.gz"
    good.write_bytes(blob)
    with gzip.open(good, "rb") as fh:
        assert len(list(F.readfastq_iter(fh, 50000, F.entryfunc, C.entrypos))) == 2000
        assert fh.read() == b""
        assert list(F.readfastq_iter(fh, 50000, F.entryfunc, C.entrypos)) == []
    # ... and one the caller ABANDONS after a few records behind the fills that were handed out, as the reference's
    # generator leaves it behind its last read(): what follows is still there for fh.read()
    big = synth.single(0, 40000, seed=42).tobytes()                 # 12.9 MB: two fills of the coalesced 8 MiB
    good.write_bytes(gzip.compress(big, 1))
    with gzip.open(good, "rb") as fh:
        it = F.readfastq_iter(fh, 1 << 23, F.entryfunc, C.entrypos)
        first = [next(it) for _ in range(3)]
        it.close()
        assert [h for h, s, q in first] == [b"SYN.%010d/1" % i for i in range(3)]
        assert fh.tell() == 1 << 23 and fh.read() == big[1 << 23:]
    # an idle pipe: some records arrive, the writer keeps the pipe open; the first fill is handed over
    # short (not the end of the stream) and close() returns at once
    r, w = os.pipe()
    os.write(w, data[:322 * 10])
    st = hip.FileStream(gpu_ctx, r, 1 << 20)
    t0 = time.time()
    rows, fill, off, end, err = st._next()
    assert rows.shape[0] == 9 and end == hip.END_REFILL            # the tenth record may still grow: carried over
    st.close()
    assert time.time() - t0 < 5.0
    os.close(w)
    os.close(r)


@pytest.mark.gpu
def test_stream_tell_is_behind_the_last_fill_handed_out(gpu_ctx, tmp_path):
    """ffq_stream_tell = end of the last chunk handed out, not of the read-ahead: an iterator closed
    early leaves the file where the reference's loop would (fastqandfurious.py:274-277)."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, synth
    data = synth.single(0, 20000, seed=42).tobytes()
    p = tmp_path / "t.fq"
    p.write_bytes(data)
    C.entrypos.coalesce_bytes = 0
    bs = 1 << 20
    with open(p, "rb") as fh:
        it = F.readfastq_iter(fh, bs, F.entryfunc, C.entrypos)
        for _ in range(5000):           # inside the second fill
            next(it)
        it.close()
        assert fh.tell() == 2 * bs


@pytest.mark.gpu
def test_edge_and_fuzz_corpora_through_pushed_and_gzip_sources(gpu_ctx, golden, tmp_path):
    """The reference-captured runs of the edge corpus (29 inputs x 4 buffer sizes) and of the fuzz corpus (400
    seeded inputs, half of them mutated) -- rows and ValueError texts of the reference iterator with its C
    scanner -- through the sources that are not plain files: chunks pushed from a BytesIO, a gzip file inflated by
    the library's reader thread, gzip over BytesIO (pushed).  Every fill boundary falls somewhere else than in
    the reference's own runs; the entries must not care."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C

    def run(opener, bs):
        rows, err = [], None
        try:
            with opener() as fh:
                for p in F.readfastq_iter(fh, bs, F.entryfunc_abspos, C.entrypos):
                    rows.append([int(x) for x in p])
        except ValueError as e:
            err = str(e)
        return rows, err

    def check(data, want, tag, sizes):
        if want.get("hang") or want.get("skipped"):
            return 0
        gz = tmp_path / "c.gz"
        gz.write_bytes(gzip.compress(data, 1))
        n = 0
        for src, opener in (("bytesio", lambda: io.BytesIO(data)), ("gzip", lambda: gzip.open(gz, "rb")),
                            ("gzip-bytesio", lambda: gzip.GzipFile(fileobj=io.BytesIO(gz.read_bytes())))):
            for bs in sizes:
                rows, err = run(opener, bs)
                assert rows == want["rows"] and err == want["error"], (tag, src, bs, err, want["error"])
                n += 1
        return n

    n = 0
    C.entrypos.coalesce_bytes = 0
    for name, ent in golden["edge"].items():
        data = bytes.fromhex(ent["data"])
        for bs, runs in ent["runs"].items():
            n += check(data, runs["c"], name, (int(bs),))
    for i, ent in enumerate(golden["fuzz"]):
        if "c" in ent:
            n += check(bytes.fromhex(ent["data"]), ent["c"], "fuzz%d" % i, (64, 1000) if i % 8 == 0 else (257,))
    C.entrypos.coalesce_bytes = 8 << 20
    for i, ent in enumerate(golden["fuzz"][:60]):
        if "c" in ent:
            n += check(bytes.fromhex(ent["data"]), ent["c"], "fuzz%d-coalesced" % i, (50000,))
    assert n > 1500
