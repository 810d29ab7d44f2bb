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
        exp = X.select_rows(want, lo, hi)
        assert got.shape == exp.shape and (got == exp).all(), (lo, hi)
    assert gpu_ctx.table_lower_bound(t.data_ptr(), len(want), 0, int(want[1234][0])) == 1234
    assert X.select_rows_device(gpu_ctx, t[:0]).shape[0] == 0
    # a dense-tile buffer right after: its index kernel bump-allocates from the control block
    short = b"".join(b"@r%d\nA\n+\n#\n" % i for i in range(20000))
    w2, *_ = oracle.scan(short)
    t2, r2 = gpu_ctx.scan_host(short)
    assert (t2 == w2).all()
