"""The batched default entryfunc (csrc/ffq_entries.c) against the reference's expression
    (buf[pos[0]+1:pos[1]], buf[pos[2]:pos[3]], buf[pos[4]:pos[5]])      fastqandfurious.py:161-171
on the golden tuples captured from the reference, on arbitrary rows (Python's slice rules), and
through both batched iterators.  CPU only: host glue."""
import io
from array import array

import numpy as np
import pytest

from conftest import golden_file

FILES = ("test.fq", "test_longqualityheader.fq", "test_multiline.fq")


@pytest.fixture(scope="module")
def E(pkg):
    from fastqandfurious_amd import build, entries
    build.build_entries()
    assert entries.native() is not None, "csrc/_ffq_entries.so did not build / load"
    return entries


def slices(buf, rows):
    return [(buf[r[0] + 1:r[1]], buf[r[2]:r[3]], buf[r[4]:r[5]]) for r in rows]


@pytest.mark.parametrize("fn", FILES)
def test_native_entries_equal_reference_tuples(E, golden, oracle, fn):
    data = golden_file(fn)
    buf = b"\n" + data
    want, *_ = oracle.scan(data, add=0)            # buffer coordinates (the sentinel is byte 0)
    got = E.entries(buf, np.ascontiguousarray(want))
    ref = [tuple(bytes.fromhex(x) for x in t) for t in golden["files"][fn]["tuples"]]      # captured from the reference
    assert got == ref == slices(buf, want.tolist())
    assert E.entries_python(buf, array("q", want.reshape(-1).tolist())) == ref


def test_any_row_follows_python_slicing(E):
    rng = np.random.default_rng(5)
    buf = bytes(rng.integers(0, 256, size=300, dtype=np.uint8))
    rows = rng.integers(-400, 400, size=(5000, 6), dtype=np.int64)
    rows[:50] = rng.integers(-2**62, 2**62, size=(50, 6), dtype=np.int64)
    want = slices(buf, rows.tolist())
    assert E.entries(buf, rows) == want
    # every kind of buffer, and stream coordinates with a shift
    assert E.entries(memoryview(buf), rows) == want
    assert E.entries(np.frombuffer(buf, dtype=np.uint8), rows) == want
    assert E.entries(bytearray(buf), array("q", rows.reshape(-1).tolist())) == want
    small = rows[50:]
    assert E.entries(buf, small + 12345, 12345) == slices(buf, small.tolist())
    # hskip = 0: the header slice keeps the '@' (index replay, benchmark.py:62-71)
    assert E.entries(buf, small, 0, 0) == [(buf[r[0]:r[1]], buf[r[2]:r[3]], buf[r[4]:r[5]]) for r in small.tolist()]
    assert E.entries_python(buf, array("q", small.reshape(-1).tolist()), 0, 0) == E.entries(buf, small, 0, 0)
    assert E.entries(buf, np.zeros((0, 6), dtype=np.int64)) == []
    with pytest.raises(ValueError):
        E.entries(buf, np.zeros(7, dtype=np.int64))
    with pytest.raises(TypeError):
        E.entries("text", rows)


def test_iterators_use_the_native_entries_and_agree_with_python(E, oracle, pkg, monkeypatch):
    """More rows than one native call takes (_ENTRY_CHUNK), through the batched iterator, with and
    without the compiled module."""
    from fastqandfurious_amd import fastqandfurious as F, synth
    data = synth.single(0, 5000, seed=3).tobytes()

    class Scanner:
        def __call__(self, *a):
            raise AssertionError("per-record protocol not expected")

        def scan_buffer(self, buf, offset, eof):
            table, end, st, off = oracle.scan(buf, sentinel=False, offset=offset, eof=eof, add=0)
            rows = array("q")
            rows.frombytes(np.ascontiguousarray(table).tobytes())
            return rows, int(end), int(off)

    calls = []
    real = E.native().entries

    class Spy:
        @staticmethod
        def entries(buf, rows, shift=0, hskip=1, cls=None):
            calls.append(len(rows) // 48)
            return real(buf, rows, shift, hskip, cls)

    monkeypatch.setattr(E, "_native", Spy)
    got = list(F.readfastq_iter(io.BytesIO(data), 200000, entrypos=Scanner()))
    assert calls and max(calls) <= F._ENTRY_CHUNK and sum(calls) == 5000
    monkeypatch.setattr(E, "_native", None)
    plain = list(F.readfastq_iter(io.BytesIO(data), 200000, entrypos=Scanner()))
    assert got == plain and len(got) == 5000
    ref = list(F.readfastq_iter(io.BytesIO(data), 200000))          # the pure-Python scanner, per record
    assert got == ref
    # entryfunc_namedtuple (:146-158): Entry instances, natively and in Python
    monkeypatch.setattr(E, "_native", Spy)
    del calls[:]
    nt = list(F.readfastq_iter(io.BytesIO(data), 200000, entryfunc=F.entryfunc_namedtuple, entrypos=Scanner()))
    assert sum(calls) == 5000 and all(type(e) is F.Entry for e in nt)
    assert nt == got and nt[7].header == got[7][0] and nt[7].quality == got[7][2]
    ref_nt = list(F.readfastq_iter(io.BytesIO(data), 200000, entryfunc=F.entryfunc_namedtuple))
    assert nt == ref_nt and type(ref_nt[0]) is F.Entry
    with pytest.raises(TypeError):
        E.entries(b"abc", np.zeros((1, 6), dtype=np.int64), 0, 1, dict)


def test_entries_phred_native_equals_the_user_guide_entryfunc(oracle, pkg):
    """csrc/ffq_entries.c::entries_phred wraps a bulk decode (here: the oracle's) into what the reference's
    documented decoding entryfunc builds per record (doc/user-guide.rst:126-141): (header, sequence, array('b'))."""
    from array import array
    from fastqandfurious_amd import entries, synth
    nat = entries.native()
    assert nat is not None and hasattr(nat, "entries_phred")
    for data in (synth.single(0, 700, seed=42), synth.wrapped(3, 500, seed=43)[0]):
        buf = data.tobytes()
        table, *_ = oracle.scan(data)
        qual, qoff = oracle.decode_quals(data, table)
        want = []
        for p in table:
            q = array("b")
            q.frombytes(buf[p[4]:p[5]])
            oracle.arrayadd_b(q, -33)
            want.append((buf[p[0] + 1:p[1]], buf[p[2]:p[3]], q))
        rows = np.ascontiguousarray(table)
        got = []
        for at in range(0, len(table), 256):          # (in pieces, as the iterator calls it: offsets not starting at 0)
            got += nat.entries_phred(buf, memoryview(rows[at:at + 256]).cast("B"), 0, qual,
                                     memoryview(np.ascontiguousarray(qoff[at:at + 257])).cast("B"), array)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g[0] == w[0] and g[1] == w[1] and isinstance(g[2], array) and g[2].typecode == "b" and g[2] == w[2]
    assert nat.entries_phred(b"", b"", 0, np.zeros(0, np.int8), memoryview(np.zeros(1, np.int64)).cast("B"), array) == []
    with pytest.raises(ValueError):
        nat.entries_phred(b"x", memoryview(np.zeros((1, 6), np.int64)).cast("B"), 0, np.zeros(4, np.int8),
                          memoryview(np.array([0, 9], np.int64)).cast("B"), array)


def test_entries_phred_arrays_are_arrays_and_calls_share_no_state(oracle, pkg):
    """Round 6: the records' array('b') objects are built directly (no process-global staging array any more -- the
    round-5 advisor's finding: a second thread's call could overwrite it mid-loop).  They must BE arrays -- mutable,
    growable, accepted by arrayadd_b, picklable, freed normally --, a subclass of array must work too (it takes the slicing
    path or its own probe), and calls from several threads at once must each get their own fill's qualities."""
    import gc
    import pickle
    import threading
    from array import array
    from fastqandfurious_amd import entries, synth
    nat = entries.native()
    data = synth.single(0, 400, seed=42)
    buf = data.tobytes()
    table, *_ = oracle.scan(data)
    qual, qoff = oracle.decode_quals(data, table)
    rows = memoryview(np.ascontiguousarray(table)).cast("B")
    qo = memoryview(np.ascontiguousarray(qoff)).cast("B")
    got = nat.entries_phred(buf, rows, 0, qual, qo, array)
    q = got[5][2]
    ref = array("b", qual[int(qoff[5]):int(qoff[6])].tobytes())
    assert type(q) is array and q.typecode == "b" and q.itemsize == 1 and q == ref and len(q) == 150
    assert pickle.loads(pickle.dumps(q)) == ref and q.tobytes() == ref.tobytes() and bytes(memoryview(q)) == ref.tobytes()
    q.append(-5); q.extend([1, 2, 3]); q[0] = 7
    assert len(q) == 154 and q[-4] == -5 and q[0] == 7 and got[6][2][0] == int(qual[int(qoff[6])])   # (its neighbours are untouched)
    oracle.arrayadd_b(q, 1)
    assert q[0] == 8
    del got, q
    gc.collect()

    class MyArray(array):
        pass
    sub = nat.entries_phred(buf, rows, 0, qual, qo, MyArray)
    assert all(isinstance(e[2], array) and e[2] == array("b", qual[int(qoff[i]):int(qoff[i]) + 150].tobytes()) for i, e in enumerate(sub))
    again = nat.entries_phred(buf, rows, 0, qual, qo, array)          # ... and back to the plain type
    assert type(again[0][2]) is array and again[3][2] == sub[3][2]

    # several threads, each with its OWN fill (qualities shifted by a per-thread constant): nobody sees another's bytes
    bad = []

    def work(k):
        mine = (qual.astype(np.int16) + k).astype(np.int8)
        for _ in range(60):
            out = nat.entries_phred(buf, rows, 0, mine, qo, array)
            for i in (0, 77, 399):
                if out[i][2] != array("b", mine[int(qoff[i]):int(qoff[i]) + 150].tobytes()):
                    bad.append((k, i))
    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad[:5]
