"""Host-side mirror of the reference API (fastq-and-furious_amd/fastqandfurious.py)
against the golden vectors captured from the reference."""
import io
from array import array

import numpy as np
import pytest

from conftest import golden_file, rows_of

FILES = ("test.fq", "test_longqualityheader.fq", "test_multiline.fq")


@pytest.fixture(scope="module")
def F(pkg):
    from fastqandfurious_amd import fastqandfurious
    return fastqandfurious


def run_iter(F, data, bufsize, **kw):
    rows, err = [], None
    try:
        for p in F.readfastq_iter(io.BytesIO(data), bufsize, entryfunc=F.entryfunc_abspos, **kw):
            rows.append([int(x) for x in p])
    except ValueError as e:
        err = str(e)
    return rows, err


def test_constants(F):
    assert (F.INVALID, F.MISSING_SEQHEADER_BEGIN, F.MISSING_SEQHEADER_END, F.MISSING_SEQ_BEG,
            F.MISSING_SEQ_END, F.MISSING_QUAL_BEGIN, F.MISSING_QUAL_END, F.COMPLETE,
            F.MISSING_QUALHEADER_END) == (-1, 0, 1, 2, 3, 4, 5, 6, 7)
    assert F.Entry._fields == ("header", "sequence", "quality")


@pytest.mark.parametrize("fn", FILES)
@pytest.mark.parametrize("bufsize", (100, 200, 600, 700, 65536))
def test_readfastq_abspos_files(F, golden, fn, bufsize):
    """BASELINE.json config 1: data/*.fq via readfastq_iter() with the
    pure-Python entrypos on CPU (reference test: tests.py:219-226)."""
    data = golden_file(fn)
    rows, err = run_iter(F, data, bufsize)
    want = golden["files"][fn]["bufsizes"][str(bufsize)]["py"]
    assert err is None and rows == want["rows"]
    # header / sequence slices as the reference test reads them
    for p in rows:
        assert data[p[0]:p[0] + 1] == b"@" and data[p[1]:p[1] + 1] == b"\n"


@pytest.mark.parametrize("fn", FILES)
def test_entry_tuples(F, golden, fn):
    data = golden_file(fn)
    got = [[h.hex(), s.hex(), q.hex()] for h, s, q in F.readfastq_iter(io.BytesIO(data), 300)]
    assert got == golden["files"][fn]["tuples"]
    nt = list(F.readfastq_iter(io.BytesIO(data), 300, entryfunc=F.entryfunc_namedtuple))
    assert [[e.header.hex(), e.sequence.hex(), e.quality.hex()] for e in nt] == golden["files"][fn]["tuples"]


def test_python_entrypos_prefix_curves(F, golden):
    for tpl in golden["templates"]:
        buf = bytes.fromhex(tpl["buf"])
        for rec in tpl["curve"]:
            pos = array("q", [-1] * 6)
            st = F.entrypos(buf[:rec["cut"]], 0, pos)
            assert [st, list(pos)] == rec["py"], (tpl["name"], rec["cut"])


def test_python_entrypos_does_not_reset(F):
    pos = array("q", [7] * 6)
    assert F.entrypos(b"no header here", 0, pos) == F.MISSING_SEQHEADER_BEGIN
    assert list(pos) == [7] * 6


def test_edge_and_fuzz_outcomes(F, golden):
    for name, ent in golden["edge"].items():
        data = bytes.fromhex(ent["data"])
        for bs, runs in ent["runs"].items():
            run = runs["py"]
            rows, err = run_iter(F, data, int(bs))
            assert rows == run["rows"], (name, bs)
            if run["hang"]:
                assert err is not None and err.startswith("Entry is invalid at byte")
            else:
                assert err == run["error"], (name, bs)
    for i, ent in enumerate(golden["fuzz"]):
        data = bytes.fromhex(ent["data"])
        run = ent["py"]
        rows, err = run_iter(F, data, 65536)
        assert rows == run["rows"], i
        if run["hang"]:
            assert err is not None and err.startswith("Entry is invalid at byte")
        else:
            assert err == run["error"], i


def test_abspos_aliases_posbuffer(F):
    """entryfunc_abspos returns the SAME array every time (reference :193-195)."""
    it = F.readfastq_iter(io.BytesIO(golden_file("test.fq")), 700, entryfunc=F.entryfunc_abspos)
    a = next(it)
    b = next(it)
    assert a is b


def _oracle_scan_buffer(oracle):
    def scan_buffer(buf, offset, eof):
        table, end, status, off = oracle.scan(buf, sentinel=False, offset=offset, eof=eof, add=0)
        rows = array("q")
        rows.frombytes(np.ascontiguousarray(table).tobytes())
        return rows, end, off
    return scan_buffer


class _BatchedOracleScanner:
    """A scanner exposing scan_buffer: exercises the batched iterator (the
    code path the GPU scanner uses) with the CPU oracle as the engine."""

    def __init__(self, oracle):
        self.scan_buffer = _oracle_scan_buffer(oracle)

    def __call__(self, buf, offset, posbuffer):
        raise AssertionError("the batched iterator must not call per record")


@pytest.mark.parametrize("fn", FILES)
@pytest.mark.parametrize("bufsize", (100, 600, 65536))
def test_batched_iterator_files(F, golden, oracle, fn, bufsize):
    data = golden_file(fn)
    rows, err = run_iter(F, data, bufsize, entrypos=_BatchedOracleScanner(oracle))
    assert err is None and rows == golden["files"][fn]["bufsizes"][str(bufsize)]["c"]["rows"]


def test_batched_iterator_edge_and_fuzz(F, golden, oracle):
    sc = _BatchedOracleScanner(oracle)
    for name, ent in golden["edge"].items():
        data = bytes.fromhex(ent["data"])
        for bs, runs in ent["runs"].items():
            run = runs["c"]
            rows, err = run_iter(F, data, int(bs), entrypos=sc)
            assert rows == run["rows"], (name, bs)
            if run["hang"]:
                assert err is not None and err.startswith("Entry is invalid at byte")
            else:
                assert err == run["error"], (name, bs, err, run["error"])
    for i, ent in enumerate(golden["fuzz"]):
        if "c" not in ent:
            continue
        run = ent["c"]
        rows, err = run_iter(F, bytes.fromhex(ent["data"]), 97, entrypos=sc)
        assert rows == run["rows"], i
        if run["hang"]:
            assert err is not None and err.startswith("Entry is invalid at byte")
        else:
            assert err == run["error"], (i, err, run["error"])


def test_batched_iterator_bufsize_independent(F, oracle, pkg):
    from fastqandfurious_amd import synth
    data, _ = synth.wrapped(0, 400, seed=43)
    data = data.tobytes()
    want, *_ = oracle.scan(data)
    for bs in (512, 5000, 1 << 20):
        rows, err = run_iter(F, data, bs, entrypos=_BatchedOracleScanner(oracle))
        assert err is None and rows == rows_of(want)
