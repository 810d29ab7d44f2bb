"""The CPU oracle against the reference: committed golden vectors (captured by
tests/golden/make_golden.py from the reference's Python module and its C
extension) and, where /root/reference is mounted, the live reference."""
import io
from array import array

import numpy as np
import pytest

from conftest import end_matches, golden_file, rows_of

FILES = ("test.fq", "test_longqualityheader.fq", "test_multiline.fq")


@pytest.mark.parametrize("fn", FILES)
def test_files_abspos(golden, oracle, fn):
    data = golden_file(fn)
    for variant, key in ((oracle.VARIANT_C, "c"), (oracle.VARIANT_PY, "py")):
        table, end, status, off = oracle.scan(data, variant=variant)
        for bs, runs in golden["files"][fn]["bufsizes"].items():
            assert rows_of(table) == runs[key]["rows"], (fn, bs, key)
            assert runs[key]["error"] is None and end == 0


def test_survey_known_tables(oracle):
    # SURVEY.md 8c / BASELINE.md section 4
    t, *_ = oracle.scan(golden_file("test.fq"))
    assert rows_of(t) == [[0, 29, 30, 115, 118, 203], [204, 233, 234, 646, 649, 1061],
                          [1062, 1091, 1092, 1225, 1228, 1361], [1362, 1391, 1392, 1446, 1449, 1503]]
    t, *_ = oracle.scan(golden_file("test_multiline.fq"))
    assert rows_of(t)[-1] == [349, 379, 380, 417, 420, 457]


def test_template_prefix_curves(golden, oracle):
    n = 0
    for tpl in golden["templates"]:
        buf = bytes.fromhex(tpl["buf"])
        for rec in tpl["curve"]:
            b = buf[:rec["cut"]]
            st, pos = oracle.entrypos(b, 0, oracle.VARIANT_PY)
            assert [st, [int(x) for x in pos]] == rec["py"], (tpl["name"], rec["cut"])
            if "c" in rec:
                st, pos = oracle.entrypos(b, 0, oracle.VARIANT_C)
                assert [st, [int(x) for x in pos]] == rec["c"], (tpl["name"], rec["cut"])
                n += 1
    assert n > 300


def test_reference_live_test_expectations(oracle):
    """The expectations of the reference's own live tests (tests.py:110-166)."""
    H, S, Q = "foo#2", "AATTGCCG", "3425@!#!"
    MS, MQ = "AATTGCCG\nGCCGTA", "3425@!#!\n255212"
    FINAL = "\n@{header}\n{sequence}\n+\n{quality}\n"
    QUALHEAD = "\n@{header}\n{sequence}\n+\n{quality}\n@bar{header}\n"
    NOQUAL = "\n@{header}\n{sequence}\n+\n"
    for variant in (oracle.VARIANT_C, oracle.VARIANT_PY):
        for tpl in (FINAL, QUALHEAD):
            for s, q in ((S, Q), (MS, MQ)):
                e = tpl.format(header=H, sequence=s, quality=q).encode()
                for cut, want in ((len(H) - 2, 1), (len(H) + len(s) - 1, 3), (len(H) + len(s) + len(q), 5)):
                    st, _ = oracle.entrypos(e[:cut], 0, variant)
                    assert st == want
        # tests.py:56-80 (shadowed there, still the documented behaviour)
        for tpl, want in ((FINAL, 5), (QUALHEAD, 6)):
            for s, q in ((S, Q), (MS, MQ)):
                e = tpl.format(header=H, sequence=s, quality=q).encode()
                st, pos = oracle.entrypos(e, 0, variant)
                assert st == want
                assert e[pos[0] + 1:pos[1]] == H.encode() and e[pos[2]:pos[3]] == s.encode()
                if want == 6:
                    assert e[pos[4]:pos[5]] == q.encode()
    # tests.py:110-138: Python gives 4, C gives 7 (the reference's xfail)
    for s in (S, MS):
        e = NOQUAL.format(header=H, sequence=s).encode()
        assert oracle.entrypos(e, 0, oracle.VARIANT_PY)[0] == 4
        assert oracle.entrypos(e, 0, oracle.VARIANT_C)[0] == 7


def test_edge_corpus(golden, oracle):
    for name, ent in golden["edge"].items():
        data = bytes.fromhex(ent["data"])
        for variant, key in ((oracle.VARIANT_C, "c"), (oracle.VARIANT_PY, "py")):
            table, end, status, off = oracle.scan(data, variant=variant)
            for bs, runs in ent["runs"].items():
                run = runs[key]
                assert rows_of(table) == run["rows"], (name, bs, key)
                assert end_matches(run, end, off), (name, bs, key, run, end, off)


def test_fuzz_corpus(golden, oracle):
    nc = 0
    for i, ent in enumerate(golden["fuzz"]):
        data = bytes.fromhex(ent["data"])
        for variant, key in ((oracle.VARIANT_C, "c"), (oracle.VARIANT_PY, "py")):
            if key not in ent:
                continue
            table, end, status, off = oracle.scan(data, variant=variant)
            assert rows_of(table) == ent[key]["rows"], (i, key)
            assert end_matches(ent[key], end, off), (i, key, ent[key]["error"], end, off)
            nc += key == "c"
    assert nc > 350


def test_arrayadd_known_answers(golden, oracle):
    for k in golden["arrayadd"]["b"]:
        a = np.frombuffer(bytes.fromhex(k["in"]), dtype=np.int8).copy()
        oracle.arrayadd_b(a, k["value"])
        assert a.tobytes().hex() == k["out"]
    for k in golden["arrayadd"]["q"]:
        a = np.array(k["in"], dtype=np.int64)
        oracle.arrayadd_q(a, k["value"])
        assert [int(x) for x in a] == k["out"]
    # SURVEY.md 8c
    a = np.frombuffer(b"!I~5@+\n", dtype=np.int8).copy()
    oracle.arrayadd_b(a, -33)
    assert a.tolist() == [0, 40, 93, 20, 31, 10, -23]


def test_synthetic_tables(oracle, pkg):
    from fastqandfurious_amd import synth
    import os
    from conftest import GOLDEN_DIR
    want = np.load(os.path.join(GOLDEN_DIR, "synth_single_table.npy"))
    data = synth.single(0, 2000, seed=42)
    t, end, st, off = oracle.scan(data)
    assert end == 0 and (t == want).all()
    # closed form on clean 4-line input
    k = np.arange(2000, dtype=np.int64) * 322
    assert (t[:, 0] == k).all() and (t[:, 1] == k + 17).all() and (t[:, 3] == k + 168).all()
    assert (t[:, 4] == k + 171).all() and (t[:, 5] == k + 321).all()
    want = np.load(os.path.join(GOLDEN_DIR, "synth_wrapped_table.npy"))
    data, start = synth.wrapped(0, 2000, seed=43)
    t, end, st, off = oracle.scan(data)
    assert end == 0 and (t == want).all()
    assert (t[:, 0] == start[:-1]).all()


def test_decode_quals(oracle, pkg):
    from fastqandfurious_amd import synth
    data, _ = synth.wrapped(5, 50, seed=43)
    t, *_ = oracle.scan(data)
    q, qoff = oracle.decode_quals(data, t)
    for i in range(len(t)):
        a = array("b")
        a.frombytes(data[t[i, 4]:t[i, 5]].tobytes())
        want = [(x - 33 + 128) % 256 - 128 for x in a]
        assert q[qoff[i]:qoff[i + 1]].tolist() == want
    assert qoff[-1] == q.size


def test_nonsentinel_and_offset(oracle):
    buf = b"\n" + golden_file("test.fq")
    t0, end, st, off = oracle.scan(buf, sentinel=False, eof=True, add=-1)
    t1, *_ = oracle.scan(golden_file("test.fq"))
    assert (t0 == t1).all()
    # start in the middle: search from the second record on
    t2, *_ = oracle.scan(buf, sentinel=False, offset=int(t1[0, 5]), add=-1)
    assert (t2 == t1[1:]).all()
    # not eof: the last record cannot be COMPLETE (needs 2 bytes after pos5)
    t3, end, st, off = oracle.scan(buf, sentinel=False, eof=False, add=-1)
    assert (t3 == t1[:-1]).all() and end == 1 and st == 5 and off == t1[-2, 5] + 1 - 1


# ---- the live reference, when mounted (this container only) -----------------
def _live():
    from oracle import refload
    if not (refload.have_reference_py() and refload.have_reference_ext()):
        pytest.skip("reference checkout / oracle/_ref not present")
    return refload.load_py(), refload.load_ext()


def test_live_reference_random_prefixes(oracle):
    py, ext = _live()
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b"\n\n@+ACGT!I#", dtype=np.uint8)
    n = 0
    for _ in range(3000):
        L = int(rng.integers(0, 80))
        b = b"\n" + rng.choice(alpha, size=L).tobytes()
        off = int(rng.integers(0, max(L, 1)))
        pp = array("q", [-1] * 6)
        st = py.entrypos(b, off, pp)
        so, po = oracle.entrypos(b, off, oracle.VARIANT_PY)
        assert (st, list(pp)) == (so, po.tolist()), (b, off)
        i = b.find(b"\n@", off)
        if i >= 0 and i + 2 >= len(b):
            continue   # the C scanner reads out of bounds here (_fastqandfurious.c:70-71)
        pc = array("q", [-1] * 6)
        st = ext.entrypos(b, off, pc)
        so, po = oracle.entrypos(b, off, oracle.VARIANT_C)
        assert (st, list(pc)) == (so, po.tolist()), (b, off)
        n += 1
    assert n > 2500


def test_live_reference_iterator_bufsize_independence(oracle, pkg):
    py, ext = _live()
    from fastqandfurious_amd import synth
    data, _ = synth.wrapped(100, 300, seed=43)
    data = data.tobytes()
    want, end, st, off = oracle.scan(data)
    for bs in (700, 4096, 65536):
        got = [list(p) for p in py.readfastq_iter(io.BytesIO(data), bs, entryfunc=py.entryfunc_abspos,
                                                  entrypos=ext.entrypos)]
        assert got == rows_of(want)


def test_live_reference_arrayadd(oracle):
    py, ext = _live()
    rng = np.random.default_rng(3)
    for _ in range(50):
        n = int(rng.integers(0, 200))
        src = rng.integers(-128, 128, size=n, dtype=np.int8)
        v = int(rng.integers(-300, 300))
        a = array("b", src.tolist())
        ext.arrayadd_b(a, v)
        b = src.copy()
        oracle.arrayadd_b(b, v)
        assert a.tolist() == b.tolist()
        src = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
        v = int(rng.integers(-2**40, 2**40))
        a = array("q", src.tolist())
        ext.arrayadd_q(a, v)
        b = src.copy()
        oracle.arrayadd_q(b, v)
        assert a.tolist() == b.tolist()
