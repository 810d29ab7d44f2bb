"""Push-down of the user guide's length filter (/root/reference/doc/user-guide.rst:153-180).

The guide's `lengthfilter_entryfunc` -- the sequence of a read shorter than a threshold, None for the others -- is
fastqandfurious.entryfunc_lengthfilter(threshold).  Goldens (tests/golden/lengthfilter.json) were captured by running the
guide's function inside the REFERENCE's readfastq_iter with the reference's C scanner
(tests/golden/make_golden_lengthfilter.py): the three fixtures, hand-written cases (errors included) and synthetic samples,
several thresholds; plus the same filter building the header / quality / whole entry of the kept records.

CPU: the object called per record by this package's iterator with the pure-Python scanner.
GPU: the same call with the GPU scanner -- the stream front end filters every fill's table ON THE DEVICE
(ffq_stream_set_filter: k_sel_count / k_scan / k_sel_scatter, then the column kernels over the kept rows) before
anything is copied back -- through every kind of source (descriptor, gzip, BGZF, pushed chunks) and buffer size;
and the C ABI itself (ffq_stream_selected: kept rows, their ordinals, the gathered column) against the oracle."""
import hashlib
import io
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_file


@pytest.fixture(scope="module")
def lf():
    with open(os.path.join(GOLDEN_DIR, "lengthfilter.json")) as fh:
        return json.load(fh)


def enc(item):
    if item is None:
        return None
    if isinstance(item, tuple):
        return [x.hex() for x in item]
    return item.hex()


def digest(items):
    h = hashlib.sha256()
    for it in items:
        if it is None:
            h.update(b"\xff")
            continue
        for x in (it if isinstance(it, tuple) else (it,)):
            h.update(len(x).to_bytes(8, "little"))
            h.update(x)
    return h.hexdigest()


def run(F, opener, bufsize, flt, scanner):
    out, err = [], None
    try:
        with opener() as fh:
            for item in F.readfastq_iter(fh, bufsize, flt, scanner):
                out.append(item)
    except ValueError as e:
        err = str(e)
    return out, err


def synth_blob(name):
    import fastqandfurious_amd  # noqa: F401
    from fastqandfurious_amd import synth
    return {"single_3000": lambda: synth.single(0, 3000, seed=42).tobytes(),
            "wrapped_3000": lambda: synth.wrapped(0, 3000, seed=43)[0].tobytes(),
            "wrapped_20000_at_11": lambda: synth.wrapped(11, 20000, seed=43)[0].tobytes()}[name]()


def check_all(F, lf, scanner, openers_for, bufsizes):
    for fn, d in lf["files"].items():
        data = golden_file(fn)
        for th, v in d.items():
            for name, opener in openers_for(data, fn.replace(".", "_")):
                for bs in bufsizes:
                    got, err = run(F, opener, bs, F.entryfunc_lengthfilter(int(th)), scanner)
                    assert ([enc(x) for x in got], err) == (v["items"], v["error"]), (fn, th, name, bs)
                for c in ("header", "quality", "entry"):
                    got, err = run(F, opener, bufsizes[0], F.entryfunc_lengthfilter(int(th), column=c), scanner)
                    assert [enc(x) for x in got] == v["columns"][c], (fn, th, name, c)
                    # yield_dropped=False: the kept records' items alone, in order
                    got, err = run(F, opener, bufsizes[0], F.entryfunc_lengthfilter(int(th), column=c, yield_dropped=False), scanner)
                    assert [enc(x) for x in got] == [x for x in v["columns"][c] if x is not None], (fn, th, name, c)
    for case, d in lf["edge"].items():
        data = bytes.fromhex(d["data"])
        for th, v in d["thresholds"].items():
            for name, opener in openers_for(data, "edge_" + case):
                got, err = run(F, opener, bufsizes[0], F.entryfunc_lengthfilter(int(th)), scanner)
                if v["error"] == "hang":                       # (the reference never leaves its loop; this build raises)
                    assert err is not None and err.startswith("Entry is invalid at byte"), (case, th, name)
                else:
                    assert err == v["error"], (case, th, name)
                assert [enc(x) for x in got] == v["items"], (case, th, name)
    for sname, d in lf["synth"].items():
        data = synth_blob(sname)
        for th, v in d.items():
            for name, opener in openers_for(data, sname):
                got, err = run(F, opener, bufsizes[-1], F.entryfunc_lengthfilter(int(th)), scanner)
                kept = [x for x in got if x is not None]
                assert err is None and (len(got), len(kept), digest(got)) == (v["n"], v["kept"], v["sha256"]), (sname, th, name)
                assert run(F, opener, bufsizes[-1], F.entryfunc_lengthfilter(int(th), yield_dropped=False), scanner) == (kept, None)
                assert (enc(kept[0]) if kept else None, enc(kept[-1]) if kept else None) == (v["first_kept"], v["last_kept"])
            opener = openers_for(data, sname)[0][1]
            for c in ("header", "quality", "entry"):
                got, err = run(F, opener, bufsizes[-1], F.entryfunc_lengthfilter(int(th), column=c), scanner)
                assert digest(got) == v["columns"][c], (sname, th, c)


def test_lengthfilter_per_record_python_scanner(pkg, lf):
    """The object as a plain entryfunc (the reference's plug-in protocol), pure-Python scanner: no device."""
    from fastqandfurious_amd import fastqandfurious as F
    check_all(F, lf, F.entrypos, lambda data, tag: [("bytesio", lambda: io.BytesIO(data))], (100, 600, 20000))
    f = F.entryfunc_lengthfilter(25)
    assert (f.min_len, f.max_len, f.column) == (None, 24, "sequence")
    from array import array
    pos = array("q", [0, 3, 4, 8, 11, 15])
    assert f(b"@r1\nACGT\n+\nIIII\n", pos) == b"ACGT" == f(b"@r1\nACGT\n+\nIIII\n", pos, -1)      # the guide's two-argument form too
    assert F.entryfunc_lengthfilter(4)(b"@r1\nACGT\n+\nIIII\n", pos) is None
    with pytest.raises(ValueError):
        F.entryfunc_lengthfilter(4, max_len=7)
    with pytest.raises(ValueError):
        F.entryfunc_lengthfilter(4, column="nope")


@pytest.mark.gpu
def test_lengthfilter_pushed_down_every_source(gpu_ctx, lf, tmp_path):
    """readfastq_iter(fh, bufsize, entryfunc_lengthfilter(...), GPU scanner): rows filtered and the column gathered on the
    device, through a descriptor, gzip (one member / several / BGZF), pushed chunks; small buffers (every fill carries an
    unfinished record over) and coalesced ones."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C
    from test_iter_decode import _sources

    def openers(data, tag):
        src = _sources(tmp_path, data, tag)
        return [s for s in src if s[0] in ("file", "gzip", "gzip-members", "bgzf", "bytesio")]
    check_all(F, lf, C.entrypos, openers, (700, 50000))
    C.entrypos.coalesce_bytes = 8 << 20
    check_all(F, lf, C.entrypos, lambda data, tag: openers(data, tag)[:1], (50000,))


@pytest.mark.gpu
def test_stream_filter_c_abi_against_oracle(gpu_ctx, oracle, tmp_path):
    """ffq_stream_set_filter / ffq_stream_selected: per fill, the kept rows are the oracle's selection of the fill's
    rows, the ordinals say where they stood, the gathered column is the slice of the fill; min and max bounds, every
    column, value_add on the quality (the Phred decode of the kept records only)."""
    from fastqandfurious_amd import hip, synth
    data = synth.wrapped(3, 9000, seed=43)[0]
    want, *_ = oracle.scan(data)
    p = tmp_path / "w.fq"
    p.write_bytes(data.tobytes())
    ln = want[:, 3] - want[:, 2]
    for lo, hi, col, add in ((None, 75, "sequence", 0), (100, 180, "header", 0), (250, None, "quality", -33), (120, 120, None, 0),
                             (10**6, None, "sequence", 0)):
        keep = np.ones(len(want), dtype=bool)
        if lo is not None:
            keep &= ln >= lo
        if hi is not None:
            keep &= ln <= hi
        for fbuf in (1 << 16, 1 << 22):
            fd = os.open(p, os.O_RDONLY)
            st = hip.FileStream(gpu_ctx, fd, fbuf)
            st.set_filter(lo, hi, col, add)
            got_rows, base, scanned = [], 0, 0
            for rows, fill, off, end, err in st:
                idx, ns, c, coff = st.selected()
                assert end in (hip.END_OK, hip.END_REFILL)
                k = rows.shape[0]
                assert (idx[:k] == np.nonzero(keep[base:base + ns])[0]).all()
                assert (rows == want[base:base + ns][keep[base:base + ns]]).all()
                if col is not None and k:
                    ca, sh, cb = {"header": (0, 1, 1), "sequence": (2, 0, 3), "quality": (4, 0, 5)}[col]
                    lens = rows[:, cb] - rows[:, ca] - sh
                    assert (coff[1:k + 1] - coff[:k] == lens).all() and coff[0] == 0
                    for j in (0, k // 2, k - 1):
                        a = int(rows[j, ca]) + sh - off
                        src = fill[a:a + int(lens[j])].astype(np.int16) + add
                        assert (c[int(coff[j]):int(coff[j + 1])].astype(np.int16) == ((src + 128) % 256 - 128)).all()
                elif col is None:
                    assert c is None
                got_rows.append(rows.copy())
                base += ns
                scanned += ns
            st.close()
            os.close(fd)
            assert scanned == len(want) and sum(r.shape[0] for r in got_rows) == int(keep.sum())
    # a decoding stream refuses the filter (it decodes every record's qualities; the filter gathers the kept ones')
    fd = os.open(p, os.O_RDONLY)
    st = hip.FileStream(gpu_ctx, fd, 1 << 20, decode=True)
    with pytest.raises(hip.FFQError):
        st.set_filter(None, 10)
    st.close()
    os.close(fd)


class _OddOnly:
    """(mixin) keeps records whose sequence length is ODD as well as under the threshold: a filter the device knows nothing of."""

    def keeps(self, length):
        return length % 2 == 1 and super().keeps(length)


def _subclass_items(F, scanner, opener, **kw):
    class Odd(_OddOnly, F.entryfunc_lengthfilter):
        pass
    return list(F.readfastq_iter(opener(), 20000, Odd(120, **kw), scanner))


def test_a_subclassed_filter_is_called_not_pushed_down_python_scanner(pkg):
    from fastqandfurious_amd import fastqandfurious as F, synth
    data = synth.wrapped(0, 2000, seed=43)[0].tobytes()
    got = _subclass_items(F, F.entrypos, lambda: io.BytesIO(data))
    plain = list(F.readfastq_iter(io.BytesIO(data), 20000, F.entryfunc_lengthfilter(120), F.entrypos))
    assert len(got) == len(plain) == 2000
    # (keeps() sees pos3 - pos2: the sequence slice as the reference cuts it, embedded newlines included)
    assert got == [p if (p is not None and len(p) % 2 == 1) else None for p in plain]
    assert 0 < sum(g is not None for g in got) < sum(p is not None for p in plain)
    assert _subclass_items(F, F.entrypos, lambda: io.BytesIO(data), yield_dropped=False) == [g for g in got if g is not None]
    assert not F._pushes_down(type("X", (_OddOnly, F.entryfunc_lengthfilter), {})(120)) and F._pushes_down(F.entryfunc_lengthfilter(120))
    assert F._pushes_down(type("Y", (F.entryfunc_lengthfilter,), {"note": 1})(120))      # (a subclass that changes nothing that matters)


@pytest.mark.gpu
def test_a_subclassed_filter_yields_the_same_items_on_the_gpu_stream(gpu_ctx, tmp_path):
    """Round-5 advisor: isinstance() let a subclass with its own keeps() be evaluated on the device from min_len / max_len
    alone -- other items than on the CPU scanners.  Now only the library's own, unchanged filter is pushed down."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C, synth
    data = synth.wrapped(0, 2000, seed=43)[0].tobytes()
    p = tmp_path / "w.fq"
    p.write_bytes(data)
    want = _subclass_items(F, F.entrypos, lambda: io.BytesIO(data))
    assert _subclass_items(F, C.entrypos, lambda: open(p, "rb")) == want
    assert _subclass_items(F, C.entrypos, lambda: io.BytesIO(data)) == want
    kept = [w for w in want if w is not None]
    assert _subclass_items(F, C.entrypos, lambda: open(p, "rb"), yield_dropped=False) == kept
    assert _subclass_items(F, F.entrypos, lambda: io.BytesIO(data), yield_dropped=False) == kept
