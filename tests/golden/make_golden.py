"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference and oracle/_ref):

    make -C oracle && python tests/golden/make_golden.py

What is captured (inputs + the reference's outputs; no reference source):
  golden.json
    files      the three reference fixtures (copied as data under data/):
               abspos rows at several fbufsize, Python and C scanner
    templates  tests.py:8-35 strings: (status, pos) at every prefix length
    edge       hand-written edge cases: rows / exception text of readfastq_iter
    fuzz       seeded random FASTQ-like inputs and mutations: same
    arrayadd   known answers of arrayadd_b / arrayadd_q
    index      offset-index files (benchmark.py:277-283) of the fixtures and of the
               synthetic samples, and the tuples the reference's replay yields
  synth_single_table.npy / synth_wrapped_table.npy
               abspos tables of 2000 synthetic records (inputs are regenerated
               by fastq-and-furious_amd/synth.py from the seed)
"""
import io
import json
import os
import sys
from array import array

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refload  # noqa: E402
import fastqandfurious_amd  # noqa: E402,F401
from fastqandfurious_amd import synth  # noqa: E402

py = refload.load_py()
ext = refload.load_ext()


class Hang(Exception):
    pass


def guarded(scanner):
    """The reference iterator never leaves its loop on INVALID at eof
    (fastqandfurious.py:256-270): detect the repeated call and stop."""
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


def run_iter(data, bufsize, scanner):
    rows, err, hang = [], None, False
    try:
        for p in py.readfastq_iter(io.BytesIO(data), bufsize, entryfunc=py.entryfunc_abspos,
                                   entrypos=guarded(scanner)):
            rows.append([int(x) for x in p])
    except ValueError as e:
        err = str(e)
    except Hang:
        hang = True
    return {"rows": rows, "error": err, "hang": hang}


def run_tuples(data, bufsize, scanner):
    return [[h.hex(), s.hex(), q.hex()] for (h, s, q) in
            py.readfastq_iter(io.BytesIO(data), bufsize, entrypos=scanner)]


def safe_for_c(buf):
    """The C scanner reads out of bounds when the buffer ends right after a
    "\\n@" match (_fastqandfurious.c:70-71).  Such calls are not captured."""
    return True


golden = {}

# ---- (a) the reference's own fixtures -----------------------------------
files = {}
for fn in ("test.fq", "test_longqualityheader.fq", "test_multiline.fq"):
    data = open(os.path.join(refload.REF_ROOT, "data", fn), "rb").read()
    ent = {"bufsizes": {}, "tuples": run_tuples(data, 65536, ext.entrypos)}
    for bs in (100, 200, 600, 700, 65536):
        ent["bufsizes"][str(bs)] = {"py": run_iter(data, bs, py.entrypos),
                                    "c": run_iter(data, bs, ext.entrypos)}
    files[fn] = ent
golden["files"] = files

# ---- (b) template known answers (tests.py:8-35) ----------------------------
HEADER, SEQ, QUAL = "foo#2", "AATTGCCG", "3425@!#!"
MSEQ, MQUAL = "AATTGCCG\nGCCGTA", "3425@!#!\n255212"
TPL = {
    "FINAL": "\n@{h}\n{s}\n+\n{q}\n",
    "QUALHEAD": "\n@{h}\n{s}\n+\n{q}\n@bar{h}\n",
    "NOQUAL": "\n@{h}\n{s}\n+\n",
    "TWO": "\n@{h}\n{s}\n+\n{q}\n@bar{h}\n{s}\n+{h}x\n{q}\n",
    "LONGPLUS": "\n@{h}\n{s}\n+{h}\n{q}\n@bar{h}\n",
    "BADPLUS": "\n@{h}\n{s}\n+{h}xy\n{q}\n@bar{h}\n",
}
templates = []
for name, tpl in TPL.items():
    for s, q in ((SEQ, QUAL), (MSEQ, MQUAL)):
        full = tpl.format(h=HEADER, s=s, q=q).encode("ascii")
        curve = []
        for cut in range(len(full) + 1):
            b = full[:cut]
            pp = array("q", [-1] * 6)
            sp = py.entrypos(b, 0, pp)
            rec = {"cut": cut, "py": [int(sp), [int(x) for x in pp]]}
            # skip the C scanner where it would read out of bounds: buffer ends
            # right after the first "\n@" match
            i = b.find(b"\n@")
            if not (i >= 0 and i + 2 >= len(b)):
                pc = array("q", [-1] * 6)
                sc = ext.entrypos(b, 0, pc)
                rec["c"] = [int(sc), [int(x) for x in pc]]
            curve.append(rec)
        templates.append({"name": name, "multiline": s is MSEQ, "buf": full.hex(), "curve": curve})
golden["templates"] = templates

# ---- (c) edge corpus ----------------------------------------------------------
R1 = b"@r1\nACGT\n+\nIIII\n"
R2 = b"@r2 desc\nACGTACGT\n+\n@III+III\n"
R3 = b"@r3\nAC\nGT\n+r3\n!!\n!!\n"
EDGE = {
    "empty": b"",
    "newline_only": b"\n",
    "one": R1,
    "two": R1 + R2,
    "three_mixed": R1 + R2 + R3,
    "no_trailing_newline": R1 + R2[:-1],
    "crlf": R1.replace(b"\n", b"\r\n"),
    "leading_garbage": b"garbage line\n" + R1 + R2,
    "first_byte_not_at": b"xr1\nACGT\n+\nIIII\n" + R2,
    "short_quality": b"@r1\nACGT\n+\nII\n" + R2 + R1,
    "long_quality": b"@r1\nACGT\n+\nIIIIII\n" + R2 + R1,
    "plus_mismatch": b"@r1\nACGT\n+zzzzzz\nIIII\n" + R2,
    "plus_same_length": b"@r1\nACGT\n+zz\nIIII\n" + R2,
    "truncated_header": R1 + b"@r2 de",
    "truncated_seq": R1 + b"@r2\nACG",
    "truncated_plus": R1 + b"@r2\nACGT\n+",
    "truncated_plus_nl": R1 + b"@r2\nACGT\n+\n",
    "truncated_qual": R1 + b"@r2\nACGT\n+\nII",
    "empty_read": b"@e\n\n+\n\n" + R1,
    "empty_read_last": R1 + b"@e\n\n+\n\n",
    "quality_at_start": b"@r1\nACGT\n+\n@@@@\n" + R1 + R2,
    "quality_plus_start": b"@r1\nACGT\n+\n+III\n" + R1,
    "blank_lines_between": R1 + b"\n\n" + R2,
    "trailing_garbage": R1 + b"tail without at\n",
    "only_garbage": b"no records here\nat all\n",
    "at_only": b"@",
    "at_newline": b"@\n",
    "wrapped_with_at_quality": b"@w\nACGTAC\nGTAC\n+\n@IIIII\n@III\n" + R1,
    "seq_starts_with_plus_line": b"@p\n+CGT\n+\nIIII\n" + R1,
}
edge = {}
for name, data in EDGE.items():
    ent = {"data": data.hex(), "runs": {}}
    for bs in (7, 16, 100, 65536):
        ent["runs"][str(bs)] = {"py": run_iter(data, bs, py.entrypos),
                                "c": run_iter(data, bs, ext.entrypos)}
    edge[name] = ent
golden["edge"] = edge

# ---- (d) fuzz corpus ---------------------------------------------------------------
rng = np.random.default_rng(20240917)
ALPH_Q = np.frombuffer(bytes(range(33, 75)), dtype=np.uint8)


def rand_record(i):
    L = int(rng.integers(1, 60))
    wrap = int(rng.integers(0, 3))
    seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L).tobytes()
    qual = rng.choice(ALPH_Q, size=L).tobytes()
    head = b"@r%d" % i + (b" x" * int(rng.integers(0, 4)))
    if wrap:
        w = int(rng.integers(5, 25))
        seq = b"\n".join(seq[a:a + w] for a in range(0, L, w))
        qual = b"\n".join(qual[a:a + w] for a in range(0, L, w))
    plus = b"+" + (head[1:] if rng.integers(0, 3) == 0 else b"")
    return head + b"\n" + seq + b"\n" + plus + b"\n" + qual + b"\n"


def mutate(data):
    b = bytearray(data)
    kind = int(rng.integers(0, 6))
    if not b:
        return bytes(b)
    p = int(rng.integers(0, len(b)))
    if kind == 0:
        del b[p]
    elif kind == 1:
        b.insert(p, int(rng.choice(np.frombuffer(b"\n@+A!", dtype=np.uint8))))
    elif kind == 2:
        b[p] = int(rng.choice(np.frombuffer(b"\n@+A!", dtype=np.uint8)))
    elif kind == 3:
        del b[p:]
    elif kind == 4:
        q = int(rng.integers(0, len(b)))
        b[min(p, q):max(p, q)] = b""
    else:
        b[p:p] = b"\n@"
    return bytes(b)


fuzz = []
for case in range(400):
    nrec = int(rng.integers(1, 12))
    data = b"".join(rand_record(i) for i in range(nrec))
    if case % 2:
        for _ in range(int(rng.integers(1, 4))):
            data = mutate(data)
    # a stream ending right after "\n@" makes the C scanner read out of bounds
    # (the reference's own defect); keep the capture deterministic by skipping
    # chunk sizes... the whole-buffer size is always safe to ask about unless
    # the DATA itself ends that way.
    ent = {"data": data.hex(), "py": run_iter(data, 65536, py.entrypos)}
    if not (b"\n" + data).endswith(b"\n@"):
        ent["c"] = run_iter(data, 65536, ext.entrypos)
    fuzz.append(ent)
golden["fuzz"] = fuzz

# ---- (e) arrayadd ------------------------------------------------------------------------
kat_b = []
for src, val in ((b"!I~5@+\n", -33), (bytes([127, 128, 0]), 1), (b"abc", 128), (b"abc", 200),
                 (b"abc", -129), (b"abc", 300), (bytes(range(256)), -33), (b"", 5)):
    a = array("b")
    a.frombytes(src)
    ext.arrayadd_b(a, val)
    kat_b.append({"in": src.hex(), "value": val, "out": a.tobytes().hex()})
kat_q = []
for src, val in (([0, 10, -5, 2**62], -3), ([2**63 - 1], 1), ([-2**63], -1), ([1, 2, 3, 4, 5], 2**40),
                 ([], 7)):
    a = array("q", src)
    ext.arrayadd_q(a, val)
    kat_q.append({"in": src, "value": val, "out": [int(x) for x in a]})
golden["arrayadd"] = {"b": kat_b, "q": kat_q}

# ---- (e2) offset-index files: what the reference stores and what it replays ---------------------
# benchmark.py:277-283 writes `pos.tofile(fh_index)` for every entry of
# readfastq_iter(entryfunc_abspos, C scanner); the replay (benchmark.py:62-71) slices
# buf[pos0:pos1], buf[pos2:pos3], buf[pos4:pos5] with pos0 = the '@' -- i.e. the tuples of the
# reference iterator with the '@' put back in front of the header.
import hashlib  # noqa: E402
index = {}
for fn in ("test.fq", "test_longqualityheader.fq", "test_multiline.fq"):
    data = open(os.path.join(HERE, "data", fn), "rb").read()
    fi = io.BytesIO()
    for pos in py.readfastq_iter(io.BytesIO(data), 600, entryfunc=py.entryfunc_abspos, entrypos=ext.entrypos):
        pos.tofile(fi)
    replay = [[(b"@" + h).hex(), s.hex(), q.hex()] for (h, s, q) in
              py.readfastq_iter(io.BytesIO(data), 600, entrypos=ext.entrypos)]
    index[fn] = {"index_hex": fi.getvalue().hex(), "sha256": hashlib.sha256(fi.getvalue()).hexdigest(),
                 "replay": replay}
for name, blob in (("synth_single_2000", synth.single(0, 2000, seed=42).tobytes()),
                   ("synth_wrapped_2000", synth.wrapped(0, 2000, seed=43)[0].tobytes())):
    fi = io.BytesIO()
    for pos in py.readfastq_iter(io.BytesIO(blob), 65536, entryfunc=py.entryfunc_abspos, entrypos=ext.entrypos):
        pos.tofile(fi)
    index[name] = {"sha256": hashlib.sha256(fi.getvalue()).hexdigest(), "bytes": len(fi.getvalue())}
golden["index"] = index

# ---- (e3) FASTA scanner (reference :103-143; templates of tests.py:36-53): status and posbuffer
#      at every prefix length, Python scanner (the reference has no C one for FASTA)
FA_TPL = {
    "NOTFINAL": "\n>{h}\n{s}\n>{h}_2\n{s}\n",
    "FINAL": "\n>{h}\n{s}\n",
    "NOSEQ": "\n>{h}\n",
    "GT_IN_SEQ": "\n>{h}\n{s}>x\n{s}\n>{h}_2\n{s}",
}
fasta = []
for name, tpl in FA_TPL.items():
    for sq in (SEQ, MSEQ, ""):
        full = tpl.format(h=HEADER, s=sq).encode("ascii")
        curve = []
        for cut in range(len(full) + 1):
            for off in (0, 3):
                b = full[:cut]
                pp = array("q", [-1] * 6)
                if cut == 0:
                    # buf[-1] of an empty buffer raises in the reference only when "\n>" and a
                    # header end were found first: never for the empty prefix
                    pass
                st = py.entrypos_fasta(b, off, pp)
                curve.append({"cut": cut, "offset": off, "r": [int(st), [int(x) for x in pp]]})
        ent = None
        pp = array("q", [-1] * 6)
        if py.entrypos_fasta(full, 0, pp) in (py.COMPLETE, py.MISSING_SEQ_END):
            h, q = py.entryfunc_fasta(full, pp, 0)
            ent = [h.hex(), q.hex()]
        fasta.append({"name": name, "seq": sq, "buf": full.hex(), "curve": curve, "entry": ent})
golden["fasta"] = fasta

with open(os.path.join(HERE, "golden.json"), "w") as fh:
    json.dump(golden, fh, indent=0, sort_keys=True)

# ---- (f) synthetic samples: tables by the reference (C scanner, iterator) ----------------------
s1 = synth.single(0, 2000, seed=42).tobytes()
t1 = np.array([[int(x) for x in p] for p in
               py.readfastq_iter(io.BytesIO(s1), 65536, entryfunc=py.entryfunc_abspos,
                                 entrypos=ext.entrypos)], dtype=np.int64)
np.save(os.path.join(HERE, "synth_single_table.npy"), t1)
w1, _ = synth.wrapped(0, 2000, seed=43)
t2 = np.array([[int(x) for x in p] for p in
               py.readfastq_iter(io.BytesIO(w1.tobytes()), 65536, entryfunc=py.entryfunc_abspos,
                                 entrypos=ext.entrypos)], dtype=np.int64)
np.save(os.path.join(HERE, "synth_wrapped_table.npy"), t2)
# the same through the Python scanner must agree (bufsize independence too)
t1p = np.array([[int(x) for x in p] for p in
                py.readfastq_iter(io.BytesIO(s1), 1000, entryfunc=py.entryfunc_abspos)], dtype=np.int64)
assert (t1 == t1p).all() and t1.shape == (2000, 6), t1.shape
t2p = np.array([[int(x) for x in p] for p in
                py.readfastq_iter(io.BytesIO(w1.tobytes()), 1500, entryfunc=py.entryfunc_abspos)],
               dtype=np.int64)
assert (t2 == t2p).all() and t2.shape == (2000, 6), t2.shape
print("golden.json: %d files, %d templates, %d edge, %d fuzz" %
      (len(files), len(templates), len(edge), len(fuzz)))
