"""the FASTA plug-in scanner (reference
src/fastqandfurious.py:103-143, :174-183; templates of tests.py:36-107) against curves captured
from the reference, and the compressed-input opener (:282-334) feeding readfastq_iter."""
import bz2
import gzip
import io
import lzma
import os
from array import array

import pytest

from conftest import golden_file

FILES = ("test.fq", "test_longqualityheader.fq", "test_multiline.fq")


@pytest.fixture(scope="module")
def F(pkg):
    from fastqandfurious_amd import fastqandfurious
    return fastqandfurious


def test_fasta_prefix_curves(F, golden):
    n = 0
    for tpl in golden["fasta"]:
        full = bytes.fromhex(tpl["buf"])
        for rec in tpl["curve"]:
            pos = array("q", [-1] * 6)
            st = F.entrypos_fasta(full[:rec["cut"]], rec["offset"], pos)
            assert [st, list(pos)] == rec["r"], (tpl["name"], tpl["seq"], rec["cut"], rec["offset"])
            n += 1
        if tpl["entry"] is not None:
            pos = array("q", [-1] * 6)
            F.entrypos_fasta(full, 0, pos)
            h, s = F.entryfunc_fasta(full, pos, 0)
            assert [h.hex(), s.hex()] == tpl["entry"]
    assert n > 600


def test_fasta_reference_test_cases(F):
    """tests.py:83-107 as written there"""
    HEADER, SEQ, MSEQ = "foo#2", "AATTGCCG", "AATTGCCG\nGCCGTA"
    cases = (("\n>{header}\n{sequence}\n>{header}_2\n{sequence}\n", F.COMPLETE, SEQ),
             ("\n>{header}\n{sequence}\n>{header}_2\n{sequence}\n", F.COMPLETE, MSEQ),
             ("\n>{header}\n{sequence}\n", F.MISSING_SEQ_END, SEQ),
             ("\n>{header}\n{sequence}\n", F.MISSING_SEQ_END, MSEQ),
             ("\n>{header}\n", F.MISSING_SEQ_BEG, ""))
    for tpl, status, seq in cases:
        entries = tpl.format(header=HEADER, sequence=seq).encode("ascii")
        pos = array("q", [-1] * 6)
        assert F.entrypos_fasta(entries, 0, pos) == status
        header, sequence = F.entryfunc_fasta(entries, pos, 0)
        assert header == HEADER.encode("ascii")
        if status != F.MISSING_SEQ_BEG:
            assert sequence == seq.encode("ascii")


@pytest.mark.parametrize("ext,writer", (("gz", gzip.open), ("gzip", gzip.open), ("bz2", bz2.open),
                                        ("lzma", lzma.open), ("xz", lzma.open), ("fq", open), (None, open)))
def test_automagic_open_feeds_the_iterator(F, golden, tmp_path, ext, writer):
    for fn in FILES:
        data = golden_file(fn)
        path = str(tmp_path / (fn.replace(".", "_") + ("." + ext if ext else "")))
        with writer(path, "wb") as fh:
            fh.write(data)
        with F.automagic_open(path) as fh:
            got = [[h.hex(), s.hex(), q.hex()] for h, s, q in F.readfastq_iter(fh, 600)]
        assert got == golden["files"][fn]["tuples"]


def test_automagic_open_custom_openers(F, tmp_path):
    path = str(tmp_path / "x.rev")
    with open(path, "wb") as fh:
        fh.write(b"abc")

    class NS:
        @staticmethod
        def rev(filename, tag):
            return io.BytesIO(open(filename, "rb").read()[::-1] + tag)

    with F.automagic_open(path, {"rev": (NS, "rev", [b"!"])}) as fh:
        assert fh.read() == b"cba!"
    with F.automagic_open(path) as fh:                 # unknown extension: plain binary file
        assert fh.read() == b"abc"
    assert set(F.FORMAT_OPENERS) >= {"gz", "gzip", "bz2", "lzma"}      # reference :282-289


@pytest.mark.gpu
@pytest.mark.parametrize("ext,writer", (("gz", gzip.open), ("bz2", bz2.open), ("xz", lzma.open)))
def test_compressed_input_gpu_scanner(F, golden, gpu_ctx, tmp_path, ext, writer, pkg):
    """compressed file -> host decompression -> buffer fills -> GPU scan, same tuples"""
    from fastqandfurious_amd import _fastqandfurious as C, synth
    for fn in FILES:
        path = str(tmp_path / (fn + "." + ext))
        with writer(path, "wb") as fh:
            fh.write(golden_file(fn))
        with F.automagic_open(path) as fh:
            got = [[h.hex(), s.hex(), q.hex()] for h, s, q in F.readfastq_iter(fh, 65536, F.entryfunc, C.entrypos)]
        assert got == golden["files"][fn]["tuples"]
    blob = synth.single(0, 5000, seed=42).tobytes()
    path = str(tmp_path / ("syn.fq." + ext))
    with writer(path, "wb") as fh:
        fh.write(blob)
    with F.automagic_open(path) as fh:
        n = sum(1 for _ in F.readfastq_iter(fh, 1 << 18, F.entryfunc, C.entrypos))
    assert n == 5000


def test_fasta_oracle_matches_golden_curves(oracle, golden):
    """the C restatement of entrypos_fasta (oracle/) against the reference's own outputs"""
    n = 0
    for tpl in golden["fasta"]:
        full = bytes.fromhex(tpl["buf"])
        for rec in tpl["curve"]:
            if rec["cut"] == 0:
                continue
            st, pos = oracle.entrypos_fasta(full[:rec["cut"]], rec["offset"])
            assert [st, pos] == rec["r"], (tpl["name"], rec["cut"], rec["offset"])
            n += 1
    assert n > 600


def _fasta_blob(rng, n, gt_runs=True):
    import numpy as np
    parts = []
    for i in range(n):
        L = int(rng.integers(0, 400))
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L).tobytes()
        w = int(rng.integers(20, 90))
        body = b"\n".join(seq[k:k + w] for k in range(0, L, w))
        rec = b">s%d some description\n" % i + body + b"\n"
        u = rng.random()
        if gt_runs and u < 0.08:
            rec = b">empty%d\n" % i + rec                 # a header right after a header
        elif gt_runs and u < 0.12:
            rec = b">a\n>b\n>c\n" + rec                   # a run of three "\n>" lines
        elif u < 0.16:
            rec = b"\n" + rec                             # blank line
        parts.append(rec)
    return b"\n" + b"".join(parts)


@pytest.mark.gpu
def test_fasta_device_scan(gpu_ctx, oracle, golden, pkg):
    """ffq_scan_fasta_* == the repeated scanner call of the oracle: rows, last status, last
    posbuffer, offset of the last call; at prefixes, offsets and with the virtual sentinel"""
    import numpy as np
    rng = np.random.default_rng(11)
    blobs = [_fasta_blob(rng, 3000), _fasta_blob(rng, 200, gt_runs=False), b"", b"\n", b"\n>", b"\n>x", b"\n>x\n",
             b"\n>x\nAC", b"no entry here\n", b">first\nAC\n>second\nGT\n"]
    for tpl in golden["fasta"]:
        blobs.append(bytes.fromhex(tpl["buf"]))
    n = 0
    for blob in blobs:
        cuts = [len(blob)] if len(blob) > 4096 else range(len(blob) + 1)
        for cut in cuts:
            b = blob[:cut]
            for off in (0, 3, max(0, cut // 2)):
                want, st, last, loff = oracle.scan_fasta(b, offset=off, add=5) if len(b) else (np.zeros((0, 6), np.int64), 0, [-1] * 6, off)
                table, res = gpu_ctx.scan_fasta_host(b, offset=off, add=5)
                assert np.array_equal(table, want), (len(blob), cut, off)
                assert int(res.last_status) == st and list(res.last_pos) == last, (len(blob), cut, off)
                if st != 0:
                    assert int(res.end_offset) == loff
                n += 1
    assert n > 1000
    # the virtual sentinel: a file that starts with '>' scanned as b'\n' + file
    blob = _fasta_blob(rng, 500)[1:]
    want, st, last, loff = oracle.scan_fasta(b"\n" + blob, add=-1)
    table, res = gpu_ctx.scan_fasta_host(blob, sentinel=True, add=-1)
    assert np.array_equal(table, want) and int(res.last_status) == st and list(res.last_pos) == last
    # many entries, several tiles, long single-line sequences
    big = b"\n" + b"".join(b">chr%d\n" % i + b"ACGT" * int(rng.integers(1, 30000)) + b"\n" for i in range(300))
    want, st, last, loff = oracle.scan_fasta(big)
    table, res = gpu_ctx.scan_fasta_host(big)
    assert np.array_equal(table, want) and int(res.last_status) == st and list(res.last_pos) == last
    # a table that is too small: the rows it holds are the first rows, all but the last one complete (include/ffq.h)
    many = _fasta_blob(rng, 5000)
    want, st, last, loff = oracle.scan_fasta(many)
    for cap in (1, 2, 63, 64, 65, len(want) // 2, len(want) - 1, len(want)):
        table, res = gpu_ctx.scan_fasta_host(many, table_cap=cap)
        assert int(res.n_records) == len(want) and len(table) == cap
        assert np.array_equal(table[:cap - 1], want[:cap - 1]), cap
        assert np.array_equal(table[cap - 1, :3], want[cap - 1, :3]) and int(table[cap - 1, 3]) in (-1, int(want[cap - 1, 3])), cap
        if cap == len(want):
            assert np.array_equal(table, want)


@pytest.mark.gpu
def test_fasta_gpu_plugin_scanner(F, golden, gpu_ctx, pkg):
    """_fastqandfurious.entrypos_fasta keeps the reference's per-call protocol: the golden curves,
    the reference's own test cases (tests.py:83-107), and a chain of calls over a buffer"""
    from fastqandfurious_amd import _fastqandfurious as C
    n = 0
    for tpl in golden["fasta"]:
        full = bytes.fromhex(tpl["buf"])
        for rec in tpl["curve"]:
            pos = array("q", [-1] * 6)
            st = C.entrypos_fasta(full[:rec["cut"]], rec["offset"], pos)
            assert [st, list(pos)] == rec["r"], (tpl["name"], tpl["seq"], rec["cut"], rec["offset"])
            n += 1
    assert n > 600
    buf = b"\n>a desc\nACGT\nAC\n>b\n>c\nGG\n>d\nT"
    pos_g, pos_p = array("q", [-1] * 6), array("q", [-1] * 6)
    off_g = off_p = 0
    for _ in range(6):
        sg, sp = C.entrypos_fasta(buf, off_g, pos_g), F.entrypos_fasta(buf, off_p, pos_p)
        assert (sg, list(pos_g)) == (sp, list(pos_p))
        if sg != F.COMPLETE:
            break
        assert F.entryfunc_fasta(buf, pos_g, 0) == F.entryfunc_fasta(buf, pos_p, 0)
        off_g, off_p = pos_g[3], pos_p[3]


@pytest.mark.gpu
def test_fasta_dense_tiles_and_pool_growth(oracle, pkg):
    """Short FASTA lines (tiles over their slot -> pool) and more pooled entries than a fresh
    context's pool holds: the scan grows the pool and runs again."""
    import numpy as np
    from fastqandfurious_amd import hip
    ctx = hip.Context(0)
    body = b"".join(b">s%d\nACG\nTT\n" % i for i in range(150000))          # ~2 MB, lines of 3-8 bytes
    blob = b"\n" * (3 << 20) + body
    want, st, last, loff = oracle.scan_fasta(blob)
    table, res = ctx.scan_fasta_host(blob, table_cap=len(want) + 8)
    assert len(want) >= 149999
    assert np.array_equal(table, want) and int(res.last_status) == st and list(res.last_pos) == last
    assert res.retries >= 1
    table, res = ctx.scan_fasta_host(blob, offset=(3 << 20) + 12345)
    want, st, last, loff = oracle.scan_fasta(blob, offset=(3 << 20) + 12345)
    assert np.array_equal(table, want) and int(res.last_status) == st and list(res.last_pos) == last


@pytest.mark.gpu
def test_fasta_long_header_runs(gpu_ctx, oracle, pkg):
    """Runs of consecutive "\\n>" lines of every length, across chunk (64 entries) and tile
    borders, and a hostile buffer that is ONE run of a million of them (the parity of a run is
    looked up 64 entries at a time: a lane-by-lane walk back was quadratic in the run's length)."""
    import time
    import numpy as np
    rng = np.random.default_rng(23)
    parts = [b"\n"]
    for i in range(400):
        run = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 1000, 5000]))
        parts.append(b"".join(b">r%d\n" % k for k in range(run)))
        parts.append(b"ACGT" * int(rng.integers(0, 50)) + b"\n")
    blob = b"".join(parts)
    for off in (0, 1, len(blob) // 3, len(blob) - 100):
        want, st, last, loff = oracle.scan_fasta(blob, offset=off)
        table, res = gpu_ctx.scan_fasta_host(blob, offset=off)
        assert np.array_equal(table, want) and int(res.last_status) == st and list(res.last_pos) == last
    want, st, last, loff = oracle.scan_fasta(b"\n" + blob[1:], add=-1)
    table, res = gpu_ctx.scan_fasta_host(blob[1:], sentinel=True, add=-1)
    assert np.array_equal(table, want) and int(res.last_status) == st and list(res.last_pos) == last
    for hostile in (b"\n" + b">\n" * 1000001, b">\n" * 1000000, b"\n" + b">\n" * 999999 + b">"):
        want, st, last, loff = oracle.scan_fasta(hostile)
        t0 = time.perf_counter()
        table, res = gpu_ctx.scan_fasta_host(hostile, table_cap=len(want) + 8)
        assert time.perf_counter() - t0 < 5.0
        assert np.array_equal(table, want) and int(res.last_status) == st and list(res.last_pos) == last


@pytest.mark.gpu
def test_fasta_and_select_at_size(gpu_ctx, oracle, pkg):
    """Past 32 768 tiles / row blocks the offsets come from the two-level scan (launch_scan_u32): 640 MiB of FASTA (an
    8 MiB block of distinct entries repeated on the device) -- EVERY row against the oracle's rows of the block + the
    repeat's offset --, then the length filter over that table (9 M rows would do; here what the scan gave) against torch."""
    import numpy as np
    import torch
    from fastqandfurious_amd import index as X
    rng = np.random.default_rng(12)
    parts, tot = [], 0
    while tot < (8 << 20):
        L = int(np.exp(rng.uniform(np.log(30), np.log(20000))))
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L).tobytes()
        e = b">e%d some text\n" % len(parts) + b"\n".join(seq[k:k + 60] for k in range(0, L, 60)) + b"\n"
        parts.append(e); tot += len(e)
    block = b"".join(parts)
    nb = len(parts)
    want2, *_ = oracle.scan_fasta(b"\n" + block + block)
    want = want2[:nb]                                       # (the block's entries, the last one closed by the next block's first)
    assert len(want2) == 2 * nb - 1
    reps = 80
    d = torch.cat([torch.tensor([10], dtype=torch.uint8), torch.from_numpy(np.frombuffer(block, dtype=np.uint8).copy()).repeat(reps)]).cuda()
    assert d.numel() > (32768 << 14)
    n = nb * reps - 1
    table = torch.empty((n + 64, 6), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    rc, res = gpu_ctx.scan_fasta_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64)
    assert rc == 0 and int(res.n_records) == n
    wt = torch.from_numpy(want).cuda()
    shift = (torch.arange(reps, device="cuda", dtype=torch.int64) * len(block)).view(reps, 1, 1)
    exp = (wt.view(1, nb, 6) + shift).view(-1, 6)[:n].clone()
    exp[:, 4:] = -1                                         # (FASTA rows: no quality)
    assert bool((table[:n] == exp).all())
    # the filter, a table long enough for its own two-level scan: the same rows eight times over
    big = table[:n].repeat(8 * (1 + (9_000_000 // (8 * n))), 1)
    assert big.shape[0] > 32768 * 256
    lens = big[:, 3] - big[:, 2]
    for lo, hi in ((100, 5000), (None, 60), (20001, None)):
        got = X.select_rows_device(gpu_ctx, big, lo, hi)
        keep = (lens >= (lo if lo is not None else -(1 << 62))) & (lens <= (hi if hi is not None else (1 << 62)))
        assert got.shape[0] == int(keep.sum().item()) and bool((got == big[keep]).all()), (lo, hi)
