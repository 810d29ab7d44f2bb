"""Golden vectors for the length-filter entryfunc of the reference's user guide, generated from the REAL reference.

Run in the build container only (needs /root/reference and oracle/_ref):

    make -C oracle && python tests/golden/make_golden_lengthfilter.py

The guide's example of "only build entry components as needed" (/root/reference/doc/user-guide.rst:153-180) is

    def lengthfilter_entryfunc(buf, posarray):
        if posarray[3] - posarray[2] < LENGTH_THRESHOLD:
            return buf[posarray[2]:posarray[3]]
        else:
            return None

handed to readfastq_iter as its entryfunc (which calls it with a third argument, globaloffset:
src/fastqandfurious.py:255): one item per record, the sequence or None.  This script runs exactly that -- the
reference's readfastq_iter, the reference's C scanner -- at several thresholds, and stores what comes out:
lengthfilter.json holds, per input and threshold, the list of yielded items (hex, or null), or for the larger
synthetic samples the counts and a sha256 over the yielded items.  Inputs: the three reference fixtures (data under
tests/golden/data), hand-written cases and the seeded generators of fastq-and-furious_amd/synth.py.  Also the same
filter building the header, the quality and the whole (header, sequence, quality) entry of the kept records (what the
build's column selector offers beyond the guide's example).  No reference source is stored.
"""
import hashlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refload  # noqa: E402
import fastqandfurious_amd  # noqa: E402,F401
from fastqandfurious_amd import synth  # noqa: E402

py = refload.load_py()
ext = refload.load_ext()


def guide_filter(threshold, column):
    """doc/user-guide.rst:166-170 with LENGTH_THRESHOLD = threshold (and, beyond the guide, another component)."""
    def lengthfilter_entryfunc(buf, posarray, globaloffset=None):
        if posarray[3] - posarray[2] < threshold:
            if column == "sequence":
                return buf[posarray[2]:posarray[3]]
            if column == "header":
                return buf[(posarray[0] + 1):posarray[1]]
            if column == "quality":
                return buf[posarray[4]:posarray[5]]
            return (buf[(posarray[0] + 1):posarray[1]], buf[posarray[2]:posarray[3]], buf[posarray[4]:posarray[5]])
        else:
            return None
    return lengthfilter_entryfunc


class Hang(Exception):
    pass


def guarded(scanner):
    """The reference iterator never leaves its loop on INVALID at eof (fastqandfurious.py:256-270):
    detect the repeated call and stop."""
    state = {"last": None, "n": 0}

    def f(buf, offset, posbuffer):
        key = (id(buf), offset)
        if key == state["last"]:
            state["n"] += 1
            if state["n"] > 50:
                raise Hang()
        else:
            state["last"], state["n"] = key, 0
        return scanner(buf, offset, posbuffer)
    return f


def run(data, bufsize, threshold, column="sequence"):
    out, err = [], None
    try:
        for item in py.readfastq_iter(io.BytesIO(data), bufsize, entryfunc=guide_filter(threshold, column), entrypos=guarded(ext.entrypos)):
            out.append(item)
    except ValueError as e:
        err = str(e)
    except Hang:
        err = "hang"             # (this build raises 'Entry is invalid at byte ...' there)
    return out, err


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


R1 = b"@r1\nACGT\n+\nIIII\n"
R2 = b"@r2 desc\nACGTACGT\n+\n@III+III\n"
R3 = b"@r3\nAC\nGT\n+r3\n!!\n!!\n"
EDGE = {
    "empty": (b"", (1, 5)),
    "three_mixed": (R1 + R2 + R3, (1, 5, 6, 9, 100)),             # lengths 4, 8, 5 (the wrapped one counts its newline)
    "no_trailing_newline": (R1 + R2[:-1], (5, 9)),
    "all_dropped": (R1 * 5, (4,)),
    "truncated_qual": (R1 + b"@r2\nACGT\n+\nII", (5,)),
    "plus_mismatch": (b"@r1\nACGT\n+zzzzzz\nIIII\n" + R2, (9,)),
}

golden = {"files": {}, "edge": {}, "synth": {}}
for fn, thresholds in (("test.fq", (55, 100, 134, 1000)), ("test_longqualityheader.fq", (100,)), ("test_multiline.fq", (37, 38))):
    data = open(os.path.join(HERE, "data", fn), "rb").read()
    golden["files"][fn] = {}
    for th in thresholds:
        ref, err = run(data, 65536, th)
        for bs in (100, 600, 20000):
            assert run(data, bs, th) == (ref, err)
        golden["files"][fn][str(th)] = {"items": [enc(x) for x in ref], "error": err,
                                        "columns": {c: [enc(x) for x in run(data, 65536, th, c)[0]] for c in ("header", "quality", "entry")}}
for name, (data, thresholds) in EDGE.items():
    golden["edge"][name] = {"data": data.hex(), "thresholds": {}}
    for th in thresholds:
        ref, err = run(data, 65536, th)
        golden["edge"][name]["thresholds"][str(th)] = {"items": [enc(x) for x in ref], "error": err}
for name, blob, thresholds in (("single_3000", synth.single(0, 3000, seed=42).tobytes(), (150, 151)),
                               ("wrapped_3000", synth.wrapped(0, 3000, seed=43)[0].tobytes(), (60, 76, 200, 400)),
                               ("wrapped_20000_at_11", synth.wrapped(11, 20000, seed=43)[0].tobytes(), (76,))):
    golden["synth"][name] = {}
    for th in thresholds:
        ref, err = run(blob, 50000, th)
        assert err is None and run(blob, 20000, th)[0] == ref
        kept = [x for x in ref if x is not None]
        golden["synth"][name][str(th)] = {
            "n": len(ref), "kept": len(kept), "sha256": digest(ref),
            "first_kept": enc(kept[0]) if kept else None, "last_kept": enc(kept[-1]) if kept else None,
            "columns": {c: digest(run(blob, 50000, th, c)[0]) for c in ("header", "quality", "entry")}}

with open(os.path.join(HERE, "lengthfilter.json"), "w") as fh:
    json.dump(golden, fh, indent=0, sort_keys=True)
print("lengthfilter.json: %d files, %d edge cases, %d synthetic samples" % (len(golden["files"]), len(golden["edge"]), len(golden["synth"])))
for name, d in golden["synth"].items():
    print(name, {th: (v["n"], v["kept"]) for th, v in d.items()})
