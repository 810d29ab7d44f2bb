"""The native stream front end (ffq_stream_*: reference read() + the refill loop of
readfastq_iter, src/fastqandfurious.py:30-36, :241-279) against the golden runs of the
reference iterator: same rows, same slices, same error texts, at every buffer size."""
import hashlib
import io
import os
import threading

import numpy as np
import pytest

from conftest import golden_file

pytestmark = pytest.mark.gpu
FILES = ("test.fq", "test_longqualityheader.fq", "test_multiline.fq")


@pytest.fixture(scope="module")
def mods(pkg, gpu_ctx):
    from fastqandfurious_amd import fastqandfurious as F, hip, index
    return F, hip, index


def run_stream(mods, gpu_ctx, fd, bufsize):
    """(rows, tuples, error text) of a whole stream"""
    F, hip, _ = mods
    rows, tuples, err = [], [], None
    st = hip.FileStream(gpu_ctx, fd, bufsize)
    try:
        for t, fill, off, end_state, err_off in st:
            for r in t.tolist():
                rows.append(r)
                p = [x - off for x in r]
                b = fill.tobytes()
                tuples.append([b[p[0] + 1:p[1]].hex(), b[p[2]:p[3]].hex(), b[p[4]:p[5]].hex()])
            if end_state not in (hip.END_OK, hip.END_REFILL):
                try:
                    F._raise_for_end(end_state, err_off)
                except ValueError as e:
                    err = str(e)
    finally:
        st.close()
    return rows, tuples, err


def with_file(tmp_path, data, fn):
    path = str(tmp_path / "in.fq")
    with open(path, "wb") as fh:
        fh.write(data)
    fd = os.open(path, os.O_RDONLY)
    try:
        return fn(fd)
    finally:
        os.close(fd)


@pytest.mark.parametrize("fn", FILES)
@pytest.mark.parametrize("bufsize", (100, 200, 600, 700, 65536))
def test_stream_golden_files(mods, gpu_ctx, golden, tmp_path, fn, bufsize):
    rows, tuples, err = with_file(tmp_path, golden_file(fn), lambda fd: run_stream(mods, gpu_ctx, fd, bufsize))
    assert err is None
    assert rows == golden["files"][fn]["bufsizes"]["65536"]["c"]["rows"]
    assert tuples == golden["files"][fn]["tuples"]


def test_stream_edge_corpus(mods, gpu_ctx, golden, tmp_path):
    """rows and ValueError texts of the reference iterator (C scanner) on the edge corpus"""
    n = 0
    for name, ent in golden["edge"].items():
        data = bytes.fromhex(ent["data"])
        for bs, run in ent["runs"].items():
            want = run["c"]
            if want.get("hang") or want.get("skipped"):
                continue
            rows, _, err = with_file(tmp_path, data, lambda fd: run_stream(mods, gpu_ctx, fd, int(bs)))
            assert rows == want["rows"], (name, bs)
            assert err == want["error"], (name, bs)
            n += 1
    assert n > 40


def test_stream_synthetic_chunks_and_index(mods, gpu_ctx, oracle, golden, tmp_path, pkg):
    F, hip, index = mods
    from fastqandfurious_amd import synth
    blob = synth.single(0, 60000, seed=42).tobytes()          # 19 MB
    want, *_ = oracle.scan(blob)
    for bs in (1 << 20, (1 << 22) + 37, 1 << 25):
        rows, _, err = with_file(tmp_path, blob, lambda fd: run_stream(mods, gpu_ctx, fd, bs))
        assert err is None and np.array_equal(np.array(rows, dtype=np.int64), want)
    # build_index over a real file takes the native stream: same bytes as the reference's index
    small = synth.single(0, 2000, seed=42).tobytes()
    path = str(tmp_path / "s.fq")
    open(path, "wb").write(small)
    fi = io.BytesIO()
    with open(path, "rb") as fh:
        assert index.build_index(fh, fi, 1 << 16) == 2000
    assert hashlib.sha256(fi.getvalue()).hexdigest() == golden["index"]["synth_single_2000"]["sha256"]
    for fn in FILES:
        p2 = str(tmp_path / fn)
        open(p2, "wb").write(golden_file(fn))
        fi = io.BytesIO()
        with open(p2, "rb") as fh:
            index.build_index(fh, fi, 600)
        assert fi.getvalue().hex() == golden["index"][fn]["index_hex"]


def test_stream_long_record_grows_the_carry(mods, gpu_ctx, oracle, tmp_path):
    """a record longer than the chunk and than the initial 1 MiB carry space"""
    rng = np.random.default_rng(3)
    L = 3 << 20
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
    qual = rng.choice(np.frombuffer(bytes(range(35, 74)), dtype=np.uint8), size=L).tobytes()
    data = b"@a\nACGT\n+\nIIII\n@long\n" + seq + b"\n+\n" + qual + b"\n@b\nAC\n+\nII\n"
    want, *_ = oracle.scan(data)
    rows, _, err = with_file(tmp_path, data, lambda fd: run_stream(mods, gpu_ctx, fd, 1 << 18))
    assert err is None and np.array_equal(np.array(rows, dtype=np.int64), want)


def test_stream_pipe_and_empty(mods, gpu_ctx, golden, tmp_path):
    """a descriptor that cannot seek (read() instead of pread()); an empty file"""
    data = golden_file("test.fq") * 50
    r, w = os.pipe()

    def feed():
        with os.fdopen(w, "wb") as fh:
            for i in range(0, len(data), 777):
                fh.write(data[i:i + 777])
    t = threading.Thread(target=feed)
    t.start()
    try:
        rows, tuples, err = run_stream(mods, gpu_ctx, r, 4096)
    finally:
        os.close(r)
        t.join()
    assert err is None and len(rows) == 200
    assert tuples[:4] == golden["files"]["test.fq"]["tuples"]
    rows, _, err = with_file(tmp_path, b"", lambda fd: run_stream(mods, gpu_ctx, fd, 4096))
    assert rows == [] and err is None


@pytest.mark.parametrize("fn", FILES)
@pytest.mark.parametrize("bufsize", (100, 600, 65536))
def test_iterator_over_real_files_takes_the_stream(mods, gpu_ctx, golden, tmp_path, fn, bufsize, monkeypatch):
    """readfastq_iter(open(path, 'rb'), ..., GPU scanner): same entries, and it IS the native
    stream that produced them (the per-fill Python path is made to fail)"""
    F, hip, _ = mods
    from fastqandfurious_amd import _fastqandfurious as C
    path = str(tmp_path / fn)
    open(path, "wb").write(golden_file(fn))

    def boom(*a, **k):
        raise AssertionError("the Python buffer loop must not run for a real file")
    monkeypatch.setattr(F, "_iter_batched", boom)
    with open(path, "rb") as fh:
        got = [[h.hex(), s.hex(), q.hex()] for h, s, q in F.readfastq_iter(fh, bufsize, F.entryfunc, C.entrypos)]
    assert got == golden["files"][fn]["tuples"]
    with open(path, "rb") as fh:
        rows = [list(p) for p in F.readfastq_iter(fh, bufsize, F.entryfunc_abspos, C.entrypos)]
    assert rows == golden["files"][fn]["bufsizes"]["65536"]["c"]["rows"]
    with open(path, "rb") as fh:
        fh.seek(0)
        ents = list(F.readfastq_iter(fh, bufsize, F.entryfunc_namedtuple, C.entrypos))
    assert [e.sequence.hex() for e in ents] == [t[1] for t in golden["files"][fn]["tuples"]]


def test_iterator_errors_over_real_files(mods, gpu_ctx, golden, tmp_path):
    F, hip, _ = mods
    from fastqandfurious_amd import _fastqandfurious as C
    n = 0
    for name, ent in golden["edge"].items():
        want = ent["runs"]["100"]["c"]
        if want.get("hang") or want.get("skipped") or want["error"] is None:
            continue
        path = str(tmp_path / "e.fq")
        open(path, "wb").write(bytes.fromhex(ent["data"]))
        rows = []
        with open(path, "rb") as fh:
            with pytest.raises(ValueError) as ei:
                for p in F.readfastq_iter(fh, 100, F.entryfunc_abspos, C.entrypos):
                    rows.append(list(p))
        assert str(ei.value) == want["error"] and rows == want["rows"], name
        n += 1
    assert n >= 5


@pytest.mark.parametrize("single_pass", (False, True))
@pytest.mark.parametrize("bufsize", (4096, 1 << 20))
def test_stream_with_decode(mods, gpu_ctx, oracle, tmp_path, pkg, bufsize, single_pass):
    """fills with FFQ_F_DECODE_QUAL: every record's bytes qual[qoff[i] : qoff[i] + pos5 - pos4] against arrayadd_b(-33)
    over its quality slice; packed streams (and fills the single pass declines: wrapped records) also as the
    concatenated int8 streams with CSR offsets.  single_pass: the stream's default, segmented where the input allows"""
    F, hip, _ = mods
    from fastqandfurious_amd import synth
    for blob in (synth.single(0, 3000, seed=42).tobytes(), synth.wrapped(0, 3000, seed=43)[0].tobytes(),
                 golden_file("test_multiline.fq")):
        want, *_ = oracle.scan(blob)
        wq, wqoff = oracle.decode_quals(blob, want)
        path = str(tmp_path / "d.fq")
        open(path, "wb").write(blob)
        fd = os.open(path, os.O_RDONLY)
        try:
            st = hip.FileStream(gpu_ctx, fd, bufsize, decode=True, single_pass=single_pass)
            quals, lens, nrows, packed = [], [], 0, True
            for rows, fill, off, end_state, err in st:
                q, qo = st.quals()
                n = rows.shape[0]
                ln = rows[:, 5] - rows[:, 4]
                assert qo.shape[0] == n + 1 and qo[-1] == q.shape[0]
                assert n == 0 or (qo[1:n] >= qo[:n - 1] + ln[:n - 1]).all() and qo[n] == qo[n - 1] + ln[n - 1]
                if n and not (qo[0] == 0 and (np.diff(qo) == ln).all()):
                    packed = False
                    assert single_pass                 # (gaps only where the caller accepted them)
                idx = np.repeat(qo[:n], ln) + (np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln))
                quals.append(q[idx].copy())
                lens.append(ln.copy())
                nrows += n
                assert end_state in (hip.END_OK, hip.END_REFILL)
            st.close()
        finally:
            os.close(fd)
        assert nrows == len(want)
        assert np.array_equal(np.concatenate(quals), wq)
        assert np.array_equal(np.concatenate(lens), np.diff(wqoff))
        if not single_pass:
            assert packed


def test_stream_long_record_after_short_ones(mods, gpu_ctx, oracle, tmp_path):
    """A fill that holds many complete records and ends in the start of a record longer than the
    carry space: the carry grows (every slot is reallocated) while the previous fill's rows have
    just been handed out.  Tuples are cut from every fill's bytes, not only rows compared."""
    from fastqandfurious_amd import synth
    rng = np.random.default_rng(11)
    L = 3 << 20
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
    qual = rng.choice(np.frombuffer(bytes(range(35, 74)), dtype=np.uint8), size=L).tobytes()
    short = synth.single(0, 9000, seed=42).tobytes()                  # ~2.9 MB of short records
    data = short + b"@long\n" + seq + b"\n+\n" + qual + b"\n" + short[:322 * 50]
    want, *_ = oracle.scan(data)
    for bufsize in (4 << 20, 1 << 20, 5 << 20):
        rows, tuples, err = with_file(tmp_path, data, lambda fd: run_stream(mods, gpu_ctx, fd, bufsize))
        assert err is None and np.array_equal(np.array(rows, dtype=np.int64), want)
        for r, t in zip(want[8990:9060].tolist(), tuples[8990:9060]):
            assert t == [data[r[0] + 1:r[1]].hex(), data[r[2]:r[3]].hex(), data[r[4]:r[5]].hex()]
        h = hashlib.sha256()
        for t in tuples:
            h.update(bytes.fromhex(t[2]))
        h2 = hashlib.sha256()
        for r in want.tolist():
            h2.update(data[r[4]:r[5]])
        assert h.hexdigest() == h2.hexdigest()


def test_stream_leaves_the_file_object_where_reading_stopped(mods, gpu_ctx, oracle, tmp_path):
    """The stream reads with pread from the object's position and does not move the shared
    descriptor under a BufferedReader; afterwards the object stands at the end of what was read
    (the reference's loop leaves `fh` at EOF)."""
    F, hip, index = mods
    from fastqandfurious_amd import _fastqandfurious as C, synth
    blob = synth.single(0, 5000, seed=42).tobytes()
    junk = b"#" * 1000
    path = str(tmp_path / "pos.fq")
    open(path, "wb").write(junk + blob)
    want, *_ = oracle.scan(blob)
    with open(path, "rb") as fh:
        assert fh.read(1000) == junk               # the reader's buffer now holds read-ahead data
        rows = [list(p) for p in F.readfastq_iter(fh, 1 << 16, F.entryfunc_abspos, C.entrypos)]
        assert np.array_equal(np.array(rows, dtype=np.int64), want)      # offsets count from the start position
        assert fh.tell() == 1000 + len(blob) and fh.read() == b""
    with open(path, "rb") as fh:
        fh.seek(1000)
        fi = io.BytesIO()
        assert index.build_index(fh, fi, 1 << 16) == len(want)
        assert fh.tell() == 1000 + len(blob)
        assert np.array_equal(np.frombuffer(fi.getvalue(), dtype=np.int64).reshape(-1, 6), want)


def test_stream_many_fills_and_reuse(mods, gpu_ctx, oracle, tmp_path):
    """hundreds of fills through the three-slot pipeline, twice (the second stream takes the
    first one's parked buffers), at chunk sizes around the slice size of the reader pool"""
    from fastqandfurious_amd import synth
    blob = synth.wrapped(0, 60000, seed=43)[0].tobytes()
    want, *_ = oracle.scan(blob)
    for bufsize in (1 << 16, (1 << 20) + 4096, 3 << 20, 1 << 16):
        rows, _, err = with_file(tmp_path, blob, lambda fd: run_stream(mods, gpu_ctx, fd, bufsize))
        assert err is None and np.array_equal(np.array(rows, dtype=np.int64), want)


def test_plain_c_host_counts_like_the_oracle(gpu_ctx, oracle, tmp_path):
    """examples/ffq_count.c: a host written in C on the same ABI (ffq_stream_open / _next): records and bases of a file."""
    import subprocess
    from fastqandfurious_amd import synth
    from test_abi import build_c_example
    exe = build_c_example(tmp_path)
    if exe is None:
        pytest.skip("no gcc")
    data = bytes(synth.wrapped(0, 40000, seed=5)[0])
    p = tmp_path / "c_host.fq"
    p.write_bytes(data)
    want, *_ = oracle.scan(data)
    r = subprocess.run([exe, str(p)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == [str(len(want)), "records,", str(int((want[:, 3] - want[:, 2]).sum())), "bases"]
    bad = tmp_path / "c_host_bad.fq"
    bad.write_bytes(data[:len(data) // 2])                       # cut inside a record: the reference raises, the C host reports
    r = subprocess.run([exe, str(bad)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "stream error" in r.stderr
