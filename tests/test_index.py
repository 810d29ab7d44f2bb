"""Offset-index files: the bytes the reference stores with
`pos.tofile` (src/demo/benchmark.py:277-283) and the tuples its replay yields
(benchmark.py:62-71), against golden vectors captured from the reference."""
import hashlib
import io

import numpy as np
import pytest

from conftest import golden_file
from test_host_iter import _BatchedOracleScanner

FILES = ("test.fq", "test_longqualityheader.fq", "test_multiline.fq")


@pytest.fixture(scope="module")
def X(pkg):
    from fastqandfurious_amd import index
    return index


@pytest.fixture(scope="module")
def F(pkg):
    from fastqandfurious_amd import fastqandfurious
    return fastqandfurious


def _check_file(X, golden, fn, entrypos, bufsize):
    data = golden_file(fn)
    want = golden["index"][fn]
    fi = io.BytesIO()
    n = X.build_index(io.BytesIO(data), fi, bufsize, entrypos=entrypos)
    assert fi.getvalue().hex() == want["index_hex"]
    assert n == len(want["replay"])
    fi.seek(0)
    got = [[h.hex(), s.hex(), q.hex()] for (h, s, q) in X.iter_indexed(io.BytesIO(data), fi, chunk_records=3)]
    assert got == want["replay"]


@pytest.mark.parametrize("fn", FILES)
@pytest.mark.parametrize("bufsize", (100, 600, 65536))
def test_index_python_scanner(X, F, golden, fn, bufsize):
    """record-by-record path, exactly the reference's loop"""
    _check_file(X, golden, fn, F.entrypos, bufsize)


@pytest.mark.parametrize("fn", FILES)
@pytest.mark.parametrize("bufsize", (100, 600, 65536))
def test_index_batched_scanner(X, golden, oracle, fn, bufsize):
    """table-at-a-time path (the one the GPU scanner takes), oracle as the engine"""
    _check_file(X, golden, fn, _BatchedOracleScanner(oracle), bufsize)


def test_index_synthetic_sha(X, golden, oracle, pkg):
    from fastqandfurious_amd import synth
    for name, blob in (("synth_single_2000", synth.single(0, 2000, seed=42).tobytes()),
                       ("synth_wrapped_2000", synth.wrapped(0, 2000, seed=43)[0].tobytes())):
        fi = io.BytesIO()
        n = X.build_index(io.BytesIO(blob), fi, 1 << 16, entrypos=_BatchedOracleScanner(oracle))
        assert n == 2000 and len(fi.getvalue()) == golden["index"][name]["bytes"]
        assert hashlib.sha256(fi.getvalue()).hexdigest() == golden["index"][name]["sha256"]


def test_select_rows_and_replay_subset(X, golden, oracle):
    """Filtering reads = deleting rows (doc/user-guide.rst:199-204); the replay of the edited
    index yields exactly the kept records."""
    data = golden_file("test.fq")
    fi = io.BytesIO()
    X.build_index(io.BytesIO(data), fi, 600, entrypos=_BatchedOracleScanner(oracle))
    fi.seek(0)
    t = X.read_index(fi)
    lens = (t[:, 3] - t[:, 2]).tolist()
    assert lens == [85, 412, 133, 54]
    kept = X.select_rows(t, min_seq_len=60, max_seq_len=200)
    assert kept.tolist() == [t[0].tolist(), t[2].tolist()]
    replay = list(X.iter_indexed(io.BytesIO(data), io.BytesIO(kept.tobytes())))
    full = golden["index"]["test.fq"]["replay"]
    assert [[h.hex(), s.hex(), q.hex()] for h, s, q in replay] == [full[0], full[2]]
    with pytest.raises(ValueError):
        X.read_index(io.BytesIO(b"\0" * 50))


@pytest.mark.gpu
@pytest.mark.parametrize("fn", FILES)
def test_index_gpu_scanner(X, golden, gpu_ctx, fn):
    from fastqandfurious_amd import _fastqandfurious as C
    _check_file(X, golden, fn, C.entrypos, 65536)
    _check_file(X, golden, fn, C.entrypos, 600)


@pytest.mark.gpu
def test_index_gpu_synthetic(X, golden, gpu_ctx, pkg):
    from fastqandfurious_amd import synth
    for name, blob in (("synth_single_2000", synth.single(0, 2000, seed=42).tobytes()),
                       ("synth_wrapped_2000", synth.wrapped(0, 2000, seed=43)[0].tobytes())):
        fi = io.BytesIO()
        assert X.build_index(io.BytesIO(blob), fi, 1 << 18) == 2000          # default scanner: the GPU one
        assert hashlib.sha256(fi.getvalue()).hexdigest() == golden["index"][name]["sha256"]


@pytest.mark.gpu
def test_select_rows_device(X, gpu_ctx, oracle, pkg):
    """ffq_table_select_seqlen == the numpy filter, on a table straight from a device scan;
    the small table queries in between must leave the context fit for the next scan."""
    import torch
    from fastqandfurious_amd import synth
    data = synth.wrapped(0, 30000, seed=43)[0]            # sequence lengths 50..300
    want, *_ = oracle.scan(data)
    dbuf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    cap = len(want) + 8
    table = torch.empty((cap, 6), dtype=torch.int64, device="cuda")
    rc, res = gpu_ctx.scan_device(dbuf.data_ptr(), len(data), table.data_ptr(), cap)
    assert rc == 0 and int(res.n_records) == len(want)
    t = table[:len(want)]
    for lo, hi in ((None, None), (100, None), (None, 120), (75, 250), (301, None), (50, 50)):
        got = X.select_rows_device(gpu_ctx, t, lo, hi).cpu().numpy()
        exp = oracle.select_seqlen(want, -(1 << 62) if lo is None else lo, (1 << 62) if hi is None else hi)
        assert got.shape == exp.shape and (got == exp).all(), (lo, hi)
        assert (X.select_rows(want, lo, hi) == exp).all()
    assert gpu_ctx.table_lower_bound(t.data_ptr(), len(want), 0, int(want[1234][0])) == 1234
    assert X.select_rows_device(gpu_ctx, t[:0]).shape[0] == 0
    # a dense-tile buffer right after: its index kernel bump-allocates from the control block
    short = b"".join(b"@r%d\nA\n+\n#\n" % i for i in range(20000))
    w2, *_ = oracle.scan(short)
    t2, r2 = gpu_ctx.scan_host(short)
    assert (t2 == w2).all()


def test_oracle_column_slices_are_the_entryfuncs(F, golden, oracle):
    """the checker's packed columns == what the reference's entryfunc cuts, record by record
    (golden tuples captured from the reference)"""
    for fn in FILES:
        data = golden_file(fn)
        want, *_ = oracle.scan(data)
        tup = golden["files"][fn]["tuples"]
        for j, which in enumerate(("header", "sequence", "quality")):
            out, off = oracle.gather_column(data, want, which)
            for i, t in enumerate(tup):
                assert out[off[i]:off[i + 1]].tobytes().hex() == t[j]
    # the user guide's length-filter entryfunc (doc/user-guide.rst:153-180), on the table
    data = golden_file("test.fq")
    want, *_ = oracle.scan(data)
    ref = [data[p[2]:p[3]] for p in want.tolist() if p[3] - p[2] < 100]
    kept = oracle.select_seqlen(want, -(1 << 62), 99)
    out, off = oracle.gather_column(data, kept, "sequence")
    assert [out[off[i]:off[i + 1]].tobytes() for i in range(len(kept))] == ref


def _shapes():
    """tables with every kind of row for the column gather"""
    import fastqandfurious_amd  # noqa: F401
    from fastqandfurious_amd import synth
    rng = np.random.default_rng(17)
    yield "single", synth.single(0, 5000, seed=42).tobytes()
    yield "wrapped", synth.wrapped(0, 5000, seed=43)[0].tobytes()
    yield "multiline", golden_file("test_multiline.fq") * 30
    yield "tiny", b"".join(b"@%d\nA\n+\n#\n" % i for i in range(4000))
    yield "empty-header", b"".join(b"@\nACGTAC\n+\nIIIIII\n" for i in range(300))
    yield "one", b"@r\nACGT\n+\nIIII\n"
    parts = []
    for i in range(600):
        L = int(rng.integers(1, 40)) if i % 3 else int(rng.integers(1000, 70000))
        parts.append(b"@x%d some text\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L)) + b"\n+\n" +
                     bytes(rng.integers(35, 74, L).astype(np.uint8)) + b"\n")
    yield "mixed-lengths", b"".join(parts)


@pytest.mark.gpu
def test_gather_column_device(X, gpu_ctx, oracle, pkg):
    """ffq_table_gather_column == the oracle's packed slices: header / sequence / quality (with
    and without a value added), on whole tables, filtered tables and empty ones."""
    import torch
    from fastqandfurious_amd import hip
    for name, data in _shapes():
        want, *_ = oracle.scan(data)
        dbuf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        table = torch.empty((len(want) + 8, 6), dtype=torch.int64, device="cuda")
        rc, res = gpu_ctx.scan_device(dbuf.data_ptr(), len(data), table.data_ptr(), table.shape[0])
        assert rc == 0 and int(res.n_records) == len(want)
        t = table[:len(want)]
        for which, value in (("header", 0), ("sequence", 0), ("quality", 0), ("quality", -33), ("sequence", 7)):
            out, off = X.select_column_device(gpu_ctx, dbuf, t, which, value_add=value)
            wo, woff = oracle.gather_column(data, want, which, value)
            assert (off.cpu().numpy() == woff).all(), (name, which)
            assert (out.cpu().numpy() == wo).all(), (name, which, value)
        # the length-filter entryfunc in two device calls: delete rows, then gather the sequences
        lens = want[:, 3] - want[:, 2]
        thr = int(np.median(lens)) if len(lens) else 0
        kept = X.select_rows_device(gpu_ctx, t, None, thr)
        wk = oracle.select_seqlen(want, -(1 << 62), thr)
        assert (kept.cpu().numpy() == wk).all()
        out, off = X.select_column_device(gpu_ctx, dbuf, kept, "sequence")
        wo, woff = oracle.gather_column(data, wk, "sequence")
        assert (off.cpu().numpy() == woff).all() and (out.cpu().numpy() == wo).all(), name
        out, off = X.select_column_device(gpu_ctx, dbuf, t[:0], "header")
        assert out.numel() == 0 and off.tolist() == [0]
        # an output that is too small: the size needed comes back, nothing is written past the room given
        wo, woff = oracle.gather_column(data, want, "quality")
        if wo.size > 40:
            small = torch.full((wo.size,), 99, dtype=torch.int8, device="cuda")
            offs = torch.empty(len(want) + 1, dtype=torch.int64, device="cuda")
            rc, nb = gpu_ctx.table_gather_column(dbuf.data_ptr(), len(data), t.data_ptr(), len(want), "quality",
                                                 small.data_ptr(), wo.size - 33, offs.data_ptr())
            assert rc == hip.E_TABLE_FULL and nb == wo.size
            assert (small[wo.size - 33:] == 99).all() and (small[:wo.size - 33].cpu().numpy() == wo[:-33]).all()
    # rows of a shard: absolute offsets far from the buffer's (add), no sentinel
    data = golden_file("test.fq") * 200
    base = 7 * (1 << 32) + 12345
    want, *_ = oracle.scan(b"\n" + data, sentinel=False, add=base)
    dbuf = torch.from_numpy(np.frombuffer(b"\n" + data + b"\0" * 15, dtype=np.uint8).copy()).cuda()
    table = torch.empty((len(want) + 8, 6), dtype=torch.int64, device="cuda")
    rc, res = gpu_ctx.scan_device(dbuf.data_ptr(), len(data) + 1, table.data_ptr(), table.shape[0], sentinel=False, add=base)
    assert rc == 0 and int(res.n_records) == len(want) == 800
    out, off = X.select_column_device(gpu_ctx, dbuf, table[:800], "sequence", sentinel=False, add=base)
    wo, woff = oracle.gather_column(b"\n" + data, want - base, "sequence")
    assert (off.cpu().numpy() == woff).all() and (out.cpu().numpy() == wo).all()
