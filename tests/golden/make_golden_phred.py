"""Golden vectors for the iterator-level Phred decode, generated from the REAL reference.

Run in the build container only (needs /root/reference and oracle/_ref):

    make -C oracle && python tests/golden/make_golden_phred.py

The reference's only documented decode is an entryfunc of the user's own that does
    quality = array('b'); quality.frombytes(buf[posarray[4]:posarray[5]]); arrayadd_b(quality, -33)
(/root/reference/doc/user-guide.rst:126-141, :206-214; src/demo/benchmark.py:155-168).  This script
runs exactly that entryfunc -- with the reference's readfastq_iter, the reference's C scanner and the
reference's arrayadd_b -- and stores what it yields: phred.json holds, per input, the list of
(header, sequence, decoded quality) as hex (the int8 values as their bytes), or for the larger
synthetic samples the record count and a sha256 over the yielded fields.  Inputs are the three
reference fixtures (data under tests/golden/data), hand-written edge cases and the seeded
synthetic generators of fastq-and-furious_amd/synth.py.  No reference source is stored.
"""
import hashlib
import io
import json
import os
import sys
from array import array

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refload  # noqa: E402
import fastqandfurious_amd  # noqa: E402,F401
from fastqandfurious_amd import synth  # noqa: E402

py = refload.load_py()
ext = refload.load_ext()


def guide_entryfunc(buf, posarray, globaloffset):
    """doc/user-guide.rst:126-141 without the Biopython record around it."""
    quality = array('b')
    quality.frombytes(buf[posarray[4]:posarray[5]])
    ext.arrayadd_b(quality, -33)
    return (buf[(posarray[0] + 1):posarray[1]], buf[posarray[2]:posarray[3]], quality)


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


def run(data, bufsize):
    out, err = [], None
    try:
        for h, s, q in py.readfastq_iter(io.BytesIO(data), bufsize, entryfunc=guide_entryfunc, entrypos=guarded(ext.entrypos)):
            assert isinstance(q, array) and q.typecode == 'b'
            out.append((h, s, q.tobytes()))
    except ValueError as e:
        err = str(e)
    except Hang:
        err = "hang"             # (this build raises 'Entry is invalid at byte ...' there)
    return out, err


def digest(entries):
    h = hashlib.sha256()
    for a, b, c in entries:
        for x in (a, b, c):
            h.update(len(x).to_bytes(8, "little"))
            h.update(x)
    return h.hexdigest()


R1 = b"@r1\nACGT\n+\nIIII\n"
R2 = b"@r2 desc\nACGTACGT\n+\n@III+III\n"
R3 = b"@r3\nAC\nGT\n+r3\n!!\n!!\n"
EDGE = {
    "empty": b"",
    "one": R1,
    "three_mixed": R1 + R2 + R3,
    "no_trailing_newline": R1 + R2[:-1],
    "high_bytes": b"@hb\nACGT\n+\n\x7f\x80\xff~\n" + R1,
    "quality_at_start": b"@r1\nACGT\n+\n@@@@\n" + R1 + R2,
    "wrapped_with_at_quality": b"@w\nACGTAC\nGTAC\n+\n@IIIII\n@III\n" + R1,
    "truncated_qual": R1 + b"@r2\nACGT\n+\nII",
    "truncated_header": R1 + b"@r2 de",
    "plus_mismatch": b"@r1\nACGT\n+zzzzzz\nIIII\n" + R2,
}

golden = {"files": {}, "edge": {}, "synth": {}}
for fn in ("test.fq", "test_longqualityheader.fq", "test_multiline.fq"):
    data = open(os.path.join(HERE, "data", fn), "rb").read()
    ref, err = run(data, 65536)
    for bs in (100, 600, 20000):
        assert run(data, bs) == (ref, err)
    golden["files"][fn] = {"entries": [[a.hex(), b.hex(), c.hex()] for a, b, c in ref], "error": err}
for name, data in EDGE.items():
    ref, err = run(data, 65536)
    golden["edge"][name] = {"data": data.hex(), "entries": [[a.hex(), b.hex(), c.hex()] for a, b, c in ref], "error": err}
for name, blob in (("single_3000", synth.single(0, 3000, seed=42).tobytes()),
                   ("wrapped_3000", synth.wrapped(0, 3000, seed=43)[0].tobytes()),
                   ("single_3000_at_7", synth.single(7, 3000, seed=42).tobytes())):
    ref, err = run(blob, 50000)
    assert err is None and run(blob, 20000)[0] == ref
    golden["synth"][name] = {"n": len(ref), "sha256": digest(ref),
                             "first": [x.hex() for x in ref[0]], "last": [x.hex() for x in ref[-1]]}

with open(os.path.join(HERE, "phred.json"), "w") as fh:
    json.dump(golden, fh, indent=0, sort_keys=True)
print("phred.json: %d files, %d edge cases, %d synthetic samples" % (len(golden["files"]), len(golden["edge"]), len(golden["synth"])))
