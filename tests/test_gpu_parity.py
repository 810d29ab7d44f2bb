"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
golden vectors.  Bit-exact: integer offsets and bytes."""
import io
import os
from array import array

import numpy as np
import pytest

from conftest import GOLDEN_DIR, end_matches, golden_file, rows_of

pytestmark = pytest.mark.gpu

FILES = ("test.fq", "test_longqualityheader.fq", "test_multiline.fq")


@pytest.fixture(scope="module")
def hipmod(pkg):
    from fastqandfurious_amd import hip
    return hip


def check_same(ctx, oracle, data, flags=0, **kw):
    """GPU scan == oracle scan on `data` (rows, end state, last status/pos)."""
    want, end, status, off = oracle.scan(data, **kw)
    table, res = ctx.scan_host(data, flags=flags, **kw)
    assert rows_of(table) == rows_of(want)
    assert int(res.n_records) == len(want)
    assert int(res.end_state) == end
    assert int(res.last_status) == status
    assert int(res.end_offset) == off
    return table, res


def tier_flags(hipmod, tier):
    """False: the usual tiers; True: the one-wave walker; "ranked": the list-ranking tier"""
    return hipmod.F_FORCE_RANKED if tier == "ranked" else (hipmod.F_FORCE_SERIAL if tier else 0)


def test_selftest(gpu_ctx):
    gpu_ctx.selftest()


@pytest.mark.parametrize("fn", FILES)
@pytest.mark.parametrize("serial", (False, True, "ranked"))
def test_golden_files(gpu_ctx, hipmod, golden, oracle, fn, serial):
    data = golden_file(fn)
    flags = tier_flags(hipmod, serial)
    table, res = check_same(gpu_ctx, oracle, data, flags=flags)
    assert rows_of(table) == golden["files"][fn]["bufsizes"]["65536"]["c"]["rows"]
    # four-line files take the fast path, the wrapped one the general kernels
    assert res.path == (5 if serial == "ranked" else 1 if serial else (0 if fn == "test_multiline.fq" else 3))


def test_template_prefix_curves_entrypos(gpu_ctx, golden):
    """ffq_entrypos == the C extension's entrypos at every prefix length."""
    n = 0
    for tpl in golden["templates"]:
        buf = bytes.fromhex(tpl["buf"])
        for rec in tpl["curve"]:
            if "c" not in rec:
                continue
            pos = array("q", [0] * 6)
            st = gpu_ctx.entrypos(buf[:rec["cut"]], 0, pos)
            assert [st, list(pos)] == rec["c"], (tpl["name"], rec["cut"])
            n += 1
    assert n > 300


@pytest.mark.parametrize("serial", (False, True, "ranked"))
def test_edge_corpus(gpu_ctx, hipmod, golden, oracle, serial, chain_path):
    flags = tier_flags(hipmod, serial)
    for name, ent in golden["edge"].items():
        data = bytes.fromhex(ent["data"])
        table, res = check_same(gpu_ctx, oracle, data, flags=flags)
        run = ent["runs"]["65536"]["c"]
        assert rows_of(table) == run["rows"], name
        assert end_matches(run, int(res.end_state), int(res.end_offset)), name


@pytest.mark.parametrize("serial", (False, True, "ranked"))
def test_fuzz_corpus(gpu_ctx, hipmod, golden, oracle, serial, chain_path):
    flags = tier_flags(hipmod, serial)
    for i, ent in enumerate(golden["fuzz"]):
        data = bytes.fromhex(ent["data"])
        table, res = check_same(gpu_ctx, oracle, data, flags=flags)
        if "c" in ent:
            assert rows_of(table) == ent["c"]["rows"], i
            assert end_matches(ent["c"], int(res.end_state), int(res.end_offset)), i


def test_not_eof_and_offsets(gpu_ctx, oracle, chain_path):
    buf = b"\n" + golden_file("test_multiline.fq")
    for eof in (False, True):
        for off in (0, 1, 2, 137, 200, len(buf) - 3, len(buf)):
            check_same(gpu_ctx, oracle, buf, sentinel=False, offset=off, eof=eof, add=0)
    check_same(gpu_ctx, oracle, golden_file("test.fq"), sentinel=True, eof=False)
    check_same(gpu_ctx, oracle, b"", sentinel=True)
    check_same(gpu_ctx, oracle, b"@", sentinel=True)
    check_same(gpu_ctx, oracle, b"\n@", sentinel=False, add=0)


@pytest.fixture(params=("fast4", "general"))
def chain_path(request, monkeypatch):
    """Run with the four-line fast path (default) and with the general chain kernels."""
    if request.param == "general":
        monkeypatch.setenv("FFQ_NO_FAST4", "1")
    return request.param


@pytest.mark.parametrize("nrec,first", ((1, 0), (50, 7), (51, 0), (2000, 0), (12345, 1000), (60000, 5)))
def test_synth_single(gpu_ctx, oracle, pkg, chain_path, nrec, first):
    from fastqandfurious_amd import synth
    data = synth.single(first, nrec, seed=42)
    table, res = check_same(gpu_ctx, oracle, data)
    assert res.path == (3 if chain_path == "fast4" else 0)
    if nrec == 2000 and first == 0:
        assert (table == np.load(os.path.join(GOLDEN_DIR, "synth_single_table.npy"))).all()


@pytest.mark.parametrize("nrec,first", ((3, 0), (2000, 0), (30000, 17)))
def test_synth_wrapped(gpu_ctx, oracle, pkg, nrec, first):
    from fastqandfurious_amd import synth
    data, _ = synth.wrapped(first, nrec, seed=43)
    table, res = check_same(gpu_ctx, oracle, data)
    assert res.path == 0
    if nrec == 2000 and first == 0:
        assert (table == np.load(os.path.join(GOLDEN_DIR, "synth_wrapped_table.npy"))).all()


def test_truncations_of_synthetic(gpu_ctx, oracle, pkg, chain_path):
    """Every way a stream can stop inside the last record."""
    from fastqandfurious_amd import synth
    data = synth.single(0, 120, seed=42).tobytes()       # > 2 tiles
    for cut in list(range(len(data) - 330, len(data) + 1, 7)) + [len(data) - 1]:
        for eof in (True, False):
            check_same(gpu_ctx, oracle, data[:cut], eof=eof)


def test_short_records_dense_tiles(gpu_ctx, oracle):
    """Lines shorter than 16 bytes on average: tiles overflow their slot and go
    through the pool; the chain falls back to the serial walker."""
    rec = b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(6000))
    table, res = check_same(gpu_ctx, oracle, rec)
    assert len(table) == 6000
    rec2 = b"\n" * 40000 + rec
    check_same(gpu_ctx, oracle, rec2)


def test_long_records(gpu_ctx, oracle):
    """Records longer than the chain kernel's window (long reads)."""
    rng = np.random.default_rng(5)
    parts = []
    for i in range(12):
        L = int(rng.integers(20000, 200000))
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes()
        qual = rng.choice(np.frombuffer(bytes(range(33, 74)), dtype=np.uint8), size=L).tobytes()
        parts.append(b"@long%d\n" % i + seq + b"\n+\n" + qual + b"\n")
    check_same(gpu_ctx, oracle, b"".join(parts))


def decode_same(ctx, hipmod, oracle, data, flags=0, **kw):
    want, *_ = oracle.scan(data, **kw)
    wq, wqoff = oracle.decode_quals(data, want)
    table, res, qual, qoff = ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL | flags, **kw)
    assert (table == want).all()
    assert (qoff == wqoff).all()
    assert int(res.n_qual_bytes) == wq.size
    assert (qual == wq).all()
    return res


def random_records(rng, n, seq_lo, seq_hi, wrap=0, hdr_hi=40, repeat_hdr=False):
    """n records with sequence lengths in [seq_lo, seq_hi]; wrap > 0 folds sequence and
    quality lines at `wrap` columns."""
    parts = []
    for i in range(n):
        L = int(rng.integers(seq_lo, seq_hi + 1))
        h = b"r%d" % i + b"x" * int(rng.integers(0, hdr_hi))
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L).tobytes()
        qual = rng.choice(np.frombuffer(bytes(range(35, 74)), dtype=np.uint8), size=L).tobytes()   # no '@', no '+'
        if wrap:
            seq = b"\n".join(seq[k:k + wrap] for k in range(0, L, wrap))
            qual = b"\n".join(qual[k:k + wrap] for k in range(0, L, wrap))
        parts.append(b"@" + h + b"\n" + seq + b"\n+" + (h if repeat_hdr else b"") + b"\n" + qual + b"\n")
    return b"".join(parts)


def test_decode_quals(gpu_ctx, hipmod, oracle, pkg, chain_path):
    from fastqandfurious_amd import synth
    for data in (synth.single(0, 3000, seed=42), synth.wrapped(0, 3000, seed=43)[0]):
        decode_same(gpu_ctx, hipmod, oracle, data)


@pytest.mark.parametrize("fn", FILES)
@pytest.mark.parametrize("serial", (False, True, "ranked"))
def test_decode_golden_files(gpu_ctx, hipmod, oracle, fn, serial):
    decode_same(gpu_ctx, hipmod, oracle, golden_file(fn), flags=tier_flags(hipmod, serial))


@pytest.mark.parametrize("shape", ("tiny", "short", "illumina", "mixed", "long", "wrapped", "huge"))
def test_decode_record_shapes(gpu_ctx, hipmod, oracle, chain_path, shape):
    """Quality streams of every granularity against the output blocks of k_decode_stream:
    thousands of records per 64 KiB block (several LDS windows, byte-wise tails), records
    that straddle chunks, records that span several blocks."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(shape.encode()))
    data = {
        "tiny": lambda: random_records(rng, 60000, 1, 3, hdr_hi=3),
        "short": lambda: random_records(rng, 40000, 1, 40, hdr_hi=8),
        "illumina": lambda: random_records(rng, 20000, 151, 151, repeat_hdr=True),
        "mixed": lambda: random_records(rng, 20000, 1, 400),
        "long": lambda: random_records(rng, 60, 30000, 300000),
        "wrapped": lambda: random_records(rng, 8000, 1, 700, wrap=61),
        "huge": lambda: random_records(rng, 3, 1 << 20, 3 << 20),
    }[shape]()
    res = decode_same(gpu_ctx, hipmod, oracle, data)
    if chain_path == "fast4" and shape in ("illumina", "mixed", "long", "huge"):
        assert res.path == 3
    # a stream that ends without the last newline: the final record is decoded too
    decode_same(gpu_ctx, hipmod, oracle, data[:-1])
    # not at eof: the cut record is not part of the stream
    decode_same(gpu_ctx, hipmod, oracle, data[:len(data) * 2 // 3], eof=False)


def test_decode_unaligned_destination_and_small_buffer(gpu_ctx, hipmod, oracle, pkg):
    """Device entry point with a destination at every alignment; a quality buffer that is
    too small reports the size needed and keeps what fits."""
    import torch
    from fastqandfurious_amd import synth
    data = synth.single(0, 2500, seed=7)
    want, *_ = oracle.scan(data)
    wq, wqoff = oracle.decode_quals(data, want)
    dbuf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    cap = len(want) + 8
    table = torch.empty((cap, 6), dtype=torch.int64, device="cuda")
    qoff = torch.empty(cap + 1, dtype=torch.int64, device="cuda")
    qbuf = torch.zeros(wq.size + 64, dtype=torch.int8, device="cuda")
    for shift in (0, 1, 7, 15):
        qbuf.fill_(99)
        torch.cuda.synchronize()      # (torch's stream is not the context's)
        rc, res = gpu_ctx.scan_device(dbuf.data_ptr(), len(data), table.data_ptr(), cap, flags=hipmod.F_DECODE_QUAL,
                                      d_qual=qbuf.data_ptr() + shift, qual_cap=wq.size, d_qoff=qoff.data_ptr())
        assert rc == 0 and int(res.n_qual_bytes) == wq.size
        got = qbuf.cpu().numpy()
        assert (got[shift:shift + wq.size] == wq).all()
        assert (got[:shift] == 99).all() and (got[shift + wq.size:] == 99).all()      # nothing outside
        assert (qoff[:len(want) + 1].cpu().numpy() == wqoff).all()
    small = wq.size // 2 + 5
    qbuf.fill_(99)
    torch.cuda.synchronize()
    rc, res = gpu_ctx.scan_device(dbuf.data_ptr(), len(data), table.data_ptr(), cap, flags=hipmod.F_DECODE_QUAL,
                                  d_qual=qbuf.data_ptr(), qual_cap=small, d_qoff=qoff.data_ptr())
    assert rc == hipmod.E_TABLE_FULL and int(res.n_qual_bytes) == wq.size
    got = qbuf.cpu().numpy()
    assert (got[:small] == wq[:small]).all() and (got[small:] == 99).all()


def test_decode_64mib(gpu_ctx, hipmod, oracle, pkg, chain_path):
    from fastqandfurious_amd import synth
    decode_same(gpu_ctx, hipmod, oracle, synth.single(5, (64 << 20) // synth.RECORD_BYTES, seed=11))


def test_arrayadd(gpu_ctx, golden, oracle, pkg):
    from fastqandfurious_amd import _fastqandfurious as C
    for k in golden["arrayadd"]["b"]:
        a = array("b")
        a.frombytes(bytes.fromhex(k["in"]))
        C.arrayadd_b(a, k["value"])
        assert a.tobytes().hex() == k["out"]
    for k in golden["arrayadd"]["q"]:
        a = array("q", k["in"])
        C.arrayadd_q(a, k["value"])
        assert [int(x) for x in a] == k["out"]
    rng = np.random.default_rng(11)
    for n in (1, 15, 16, 17, 1000, 100003):
        for shift in (0, 1, 3):
            src = rng.integers(-128, 128, size=n + shift, dtype=np.int8)
            a = src.copy()
            gpu_ctx.arrayadd_b(a[shift:], -33)
            b = src.copy()
            oracle.arrayadd_b(b[shift:], -33)
            assert (a == b).all()
        src = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
        a = src.copy()
        gpu_ctx.arrayadd_q(a, -12345678901)
        b = src.copy()
        oracle.arrayadd_q(b, -12345678901)
        assert (a == b).all()


def test_device_generators_match_numpy(gpu_ctx, pkg):
    from fastqandfurious_amd import synth
    n = 5000
    d = gpu_ctx.dev_alloc(n * 322)
    gpu_ctx.synth_single(d, 123, n, seed=42)
    out = np.empty(n * 322, dtype=np.uint8)
    gpu_ctx.d2h(out, d)
    gpu_ctx.dev_free(d)
    assert (out == synth.single(123, n, seed=42)).all()
    want, start = synth.wrapped(77, n, seed=43)
    d = gpu_ctx.dev_alloc(want.size)
    ds = gpu_ctx.dev_alloc(start.nbytes)
    gpu_ctx.h2d(ds, start)
    gpu_ctx.synth_wrapped(d, ds, 77, n, seed=43)
    out = np.empty(want.size, dtype=np.uint8)
    gpu_ctx.d2h(out, d)
    gpu_ctx.dev_free(d)
    gpu_ctx.dev_free(ds)
    assert (out == want).all()


# ---- the reference-shaped API on the GPU ------------------------------------------
def run_iter(F, data, bufsize, entrypos):
    rows, err = [], None
    try:
        for p in F.readfastq_iter(io.BytesIO(data), bufsize, entryfunc=F.entryfunc_abspos, entrypos=entrypos):
            rows.append([int(x) for x in p])
    except ValueError as e:
        err = str(e)
    return rows, err


@pytest.mark.parametrize("fn", FILES)
@pytest.mark.parametrize("bufsize", (100, 600, 65536))
def test_readfastq_iter_with_gpu_entrypos(gpu_ctx, golden, pkg, fn, bufsize):
    """readfastq_iter(..., entrypos=_fastqandfurious.entrypos): batched protocol."""
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C
    data = golden_file(fn)
    rows, err = run_iter(F, data, bufsize, C.entrypos)
    assert err is None and rows == golden["files"][fn]["bufsizes"][str(bufsize)]["c"]["rows"]
    got = [[h.hex(), s.hex(), q.hex()] for h, s, q in F.readfastq_iter(io.BytesIO(data), bufsize, entrypos=C.entrypos)]
    assert got == golden["files"][fn]["tuples"]


@pytest.mark.parametrize("fn", FILES)
def test_per_record_protocol(gpu_ctx, golden, pkg, fn):
    """The per-record plug-in protocol: a loop that calls entrypos(buf, offset,
    posbuffer) once per record, as the reference's iterator does."""
    from fastqandfurious_amd import _fastqandfurious as C
    buf = b"\n" + golden_file(fn)
    pos = array("q", [-1] * 6)
    rows, offset = [], 0
    while True:
        st = C.entrypos(buf, offset, pos)
        if st != C.COMPLETE:
            break
        rows.append([x - 1 for x in pos])
        offset = pos[5] - 1
    assert st == C.POS_QUAL_END
    assert rows == golden["files"][fn]["bufsizes"]["65536"]["c"]["rows"][:-1]
    # an arbitrary offset (not on the cached chain) rescans
    st = C.entrypos(buf, 5, pos)
    assert st == C.COMPLETE and pos[0] - 1 == rows[1][0]


def test_iterator_errors_on_gpu(gpu_ctx, golden, pkg):
    from fastqandfurious_amd import fastqandfurious as F, _fastqandfurious as C
    for name, ent in golden["edge"].items():
        data = bytes.fromhex(ent["data"])
        for bs in ("16", "65536"):
            run = ent["runs"][bs]["c"]
            rows, err = run_iter(F, data, int(bs), C.entrypos)
            assert rows == run["rows"], (name, bs)
            if run["hang"]:
                assert err is not None and err.startswith("Entry is invalid at byte")
            else:
                assert err == run["error"], (name, bs)


def test_submit_wait_two_contexts(gpu_ctx, hipmod, oracle, pkg):
    """ffq_scan_submit / ffq_scan_wait with two contexts that share their streams: scans are
    queued one ahead and complete in order, each with its own scratch and output."""
    import torch
    from fastqandfurious_amd import synth
    ctx2 = hipmod.Context(share=gpu_ctx)
    datas = [synth.single(0, 30000, seed=42), synth.wrapped(0, 20000, seed=43)[0],
             synth.single(500, 12345, seed=42), np.frombuffer(golden_file("test_multiline.fq"), dtype=np.uint8)]
    wants = [oracle.scan(d)[0] for d in datas]
    bufs = [torch.from_numpy(np.ascontiguousarray(d)).cuda() for d in datas]
    tabs = [torch.empty((len(w) + 8, 6), dtype=torch.int64, device="cuda") for w in wants]
    torch.cuda.synchronize()
    ctxs = (gpu_ctx, ctx2)
    def submit(i):
        ctxs[i & 1].scan_submit(bufs[i].data_ptr(), bufs[i].numel(), tabs[i].data_ptr(), tabs[i].shape[0])
    submit(0)
    for i in range(1, len(datas)):
        submit(i)
        rc, res = ctxs[(i - 1) & 1].scan_wait()
        assert rc == hipmod.OK and int(res.n_records) == len(wants[i - 1])
    rc, res = ctxs[(len(datas) - 1) & 1].scan_wait()
    assert rc == hipmod.OK and int(res.n_records) == len(wants[-1])
    for t, w in zip(tabs, wants):
        assert (t[:len(w)].cpu().numpy() == w).all()
    with pytest.raises(hipmod.FFQError):
        gpu_ctx.scan_wait()              # nothing pending
    ctx2.close()


@pytest.mark.parametrize("mib", (256, 2304))
def test_closed_form_at_size(gpu_ctx, pkg, mib):
    """A size-independent property at bench-like sizes: on S-single the table equals the
    generator's closed form (which the small tests prove equal to the reference).  2.25 GiB is
    past the point where the superblock sums become a kernel of their own."""
    import torch
    n = (mib << 20) // 322
    buf = torch.empty(n * 322 + 64, dtype=torch.uint8, device="cuda")
    gpu_ctx.synth_single(buf.data_ptr(), 0, n, seed=42)
    table = torch.empty((n + 8, 6), dtype=torch.int64, device="cuda")
    rc, res = gpu_ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 8)
    assert rc == 0 and int(res.n_records) == n and int(res.end_state) == 0 and res.path == 3
    k = torch.arange(n, dtype=torch.int64, device="cuda") * 322
    want = torch.stack([k, k + 17, k + 18, k + 168, k + 171, k + 321], dim=1)
    assert bool((table[:n] == want).all())
    # the same from an offset deep inside (past 2^31 in the large case): the search starts at the
    # '\n' in front of record k0 (buffer coordinate 322 k0 with the sentinel), rows k0.. follow;
    # also without eof: the last record is then held back (its quality could go on)
    k0 = n - n // 17
    rc, res = gpu_ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 8, offset=322 * k0)
    assert rc == 0 and int(res.n_records) == n - k0 and int(res.end_state) == 0
    assert bool((table[:n - k0] == want[k0:]).all())
    rc, res = gpu_ctx.scan_device(buf.data_ptr(), n * 322, table.data_ptr(), n + 8, offset=322 * k0, eof=False)
    assert rc == 0 and int(res.n_records) == n - k0 - 1 and int(res.end_state) == 1
    assert bool((table[:n - k0 - 1] == want[k0:n - 1]).all()) and int(res.end_offset) == 322 * (n - 1) - 1   # pos5 - 1 of the last complete record
    del buf, table, want, k
    torch.cuda.empty_cache()


def test_table_cut(gpu_ctx):
    """ffq_table_cut == searchsorted on column 0, for every kind of bound."""
    import torch
    rng = np.random.default_rng(5)
    for n in (0, 1, 63, 64, 65, 4096, 100003):
        p0 = np.sort(rng.choice(10 * n + 10, size=n, replace=False)).astype(np.int64) if n else np.zeros(0, np.int64)
        t = np.zeros((max(n, 1), 6), dtype=np.int64)
        t[:n, 0] = p0
        t[:n, 5] = p0 + 7
        d = torch.from_numpy(t).cuda()
        qs = [-(1 << 62), 0, 1 << 62] + ([int(p0[0]), int(p0[-1]), int(p0[-1]) + 1, int(p0[n // 2]), int(p0[n // 2]) + 1]
                                       if n else [])
        for lo in qs:
            for hi in qs:
                i0, i1, f0, f1, q0, q1 = gpu_ctx.table_cut(d.data_ptr(), n, lo, hi)
                e0, e1 = int(np.searchsorted(p0, lo, side="left")), int(np.searchsorted(p0, hi, side="left"))
                assert (i0, i1) == (e0, e1), (n, lo, hi)
                assert f0 == (int(p0[e0]) if e0 < n else -1) and f1 == (int(p0[e1]) if e1 < n else -1)
                assert q0 == (int(p0[e0 - 1]) + 7 if e0 > 0 else -1) and q1 == (int(p0[e1 - 1]) + 7 if e1 > 0 else -1)


def _mess(rng, n, fatal=True):
    """FASTQ-like text with everything that makes the chain hard: '@' and '+' heavy quality,
    wrapped and unwrapped records mixed, '+' lines that repeat (or garble) the header, empty
    lines, CRLF, stray text between records, records cut short."""
    qchars = np.frombuffer(b"@+@+IIII#5?@+", dtype=np.uint8)
    parts = []
    for i in range(n):
        L = int(rng.integers(1, 180))
        h = b"r%d" % i + (b" desc" if rng.random() < 0.3 else b"")
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L).tobytes()
        qual = rng.choice(qchars, size=L).tobytes()
        u = rng.random()
        if u < 0.35:
            w = int(rng.integers(7, 70))
            seq = b"\n".join(seq[k:k + w] for k in range(0, L, w))
            qual = b"\n".join(qual[k:k + w] for k in range(0, L, w))
        plus = b"+" + (h if rng.random() < 0.25 else b"")
        if fatal and rng.random() < 0.01:
            plus += b"xy"                                   # '+' line length rule -> INVALID
        # CRLF after the header makes a repeated '+' header one byte short of it -> INVALID
        nl = b"\r\n" if rng.random() < 0.02 and (fatal or len(plus) == 1) else b"\n"
        rec = b"@" + h + nl + seq + b"\n" + plus + b"\n" + qual + b"\n"
        v = rng.random()
        if fatal and v < 0.01:
            rec = rec[:int(rng.integers(1, len(rec)))] + b"\n"        # cut short
        elif v < 0.02:
            rec = b"\n" + rec                                          # empty line in front
        elif v < 0.03:
            rec = b"junk line\n" + rec
        parts.append(rec)
    return b"".join(parts)


@pytest.mark.parametrize("seed", range(6))
def test_differential_mess(gpu_ctx, hipmod, oracle, chain_path, seed):
    """Seeded differential test against the oracle on hostile input, every path, with the
    decode; then the same stream from a later offset, not at eof, and without the sentinel."""
    rng = np.random.default_rng(1000 + seed)
    data = _mess(rng, 20000, fatal=seed >= 4)      # seeds 0-3: the chain runs through all of it
    want, end, status, off = oracle.scan(data)
    if seed < 4:
        assert len(want) > 15000
    table, res, qual, qoff = gpu_ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL)
    assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off
    assert (table == want).all()
    wq, wqoff = oracle.decode_quals(data, want)
    assert (qoff == wqoff).all() and (qual == wq).all()
    cut = len(data) // 3
    for kw in (dict(offset=cut), dict(eof=False), dict(sentinel=False, offset=7, eof=False)):
        check_same(gpu_ctx, oracle, data[:len(data) - 11], **kw)


@pytest.mark.parametrize("L", (1000, 3000))
def test_wrapped_kilobase_records_are_repaired_not_serialised(gpu_ctx, oracle, L):
    """Records of a few KB wrapped at 80 columns: a group's 8 KiB run-in holds only a few of
    them, so some entry guesses are wrong.  The verification must send those groups through a
    repair pass (entered at the predecessor's exit) instead of giving the buffer to the serial
    walker -- same rows either way."""
    rng = np.random.default_rng(L)
    data = random_records(rng, (6 << 20) // (2 * L), L, L, wrap=80, hdr_hi=10)
    table, res = check_same(gpu_ctx, oracle, data)
    assert res.path in (0, 5)            # repaired, or ranked: never the one-wave walker
    check_same(gpu_ctx, oracle, data[:-1])
    check_same(gpu_ctx, oracle, data, offset=len(data) // 2)


@pytest.mark.parametrize("L,wrap", ((500, 60), (1000, 80), (3000, 80), (5000, 200)))
def test_wrapped_reads_of_kilobases_stay_on_the_group_kernels(gpu_ctx, oracle, L, wrap):
    """64 MiB of wrapped reads of 0.5-5 kbp whose quality lines begin with '@' and '+' now and then (false candidates: a
    call from one "reads" several records as one and its successor lies dozens of lines on -- found in the window since
    round 5, by a binary search of the node positions): the group kernels prove the chain without a repair pass, and the
    second scan of the context does too."""
    rng = np.random.default_rng(7 * L + wrap)
    data = random_records(rng, (64 << 20) // (2 * L + 2 * (L // wrap) + 60), L * 3 // 4, L, wrap=wrap, hdr_hi=20)
    gpu_ctx.forget()
    table, res = check_same(gpu_ctx, oracle, data)
    assert res.path == 0
    table, res = check_same(gpu_ctx, oracle, data)
    assert res.path == 0 and res.retries == 0
    check_same(gpu_ctx, oracle, data[:-1])
    check_same(gpu_ctx, oracle, data, offset=len(data) // 2, eof=False)
    gpu_ctx.forget()


@pytest.mark.parametrize("seed", range(6))
def test_differential_mess_ranked(gpu_ctx, hipmod, oracle, seed):
    """The hostile streams of test_differential_mess through the list-ranking tier (forced): the
    same chain, decode included; from a later offset, not at eof, without the sentinel."""
    rng = np.random.default_rng(1000 + seed)
    data = _mess(rng, 20000, fatal=seed >= 4)
    want, end, status, off = oracle.scan(data)
    fl = hipmod.F_FORCE_RANKED
    table, res, qual, qoff = gpu_ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL | fl)
    assert res.path == 5
    assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off
    assert (table == want).all()
    wq, wqoff = oracle.decode_quals(data, want)
    assert (qoff == wqoff).all() and (qual == wq).all()
    cut = len(data) // 3
    for kw in (dict(offset=cut), dict(eof=False), dict(sentinel=False, offset=7, eof=False)):
        check_same(gpu_ctx, oracle, data[:len(data) - 11], flags=fl, **kw)
    for n in (0, 1, 2, 5, 40, 41, 1000):
        check_same(gpu_ctx, oracle, data[:n], flags=fl)
        check_same(gpu_ctx, oracle, data[:n], flags=fl, eof=False)


@pytest.mark.parametrize("L", (5000, 20000, 60000))
def test_long_wrapped_records_take_the_ranked_tier(gpu_ctx, hipmod, oracle, L):
    """Wrapped records of 10-120 KB: a group of tiles lies inside one record, entry guesses are
    worthless.  The chain comes from list ranking over the "\\n@" matches (path 5), exact, and
    the context then starts its next scans there; short records afterwards send it back."""
    rng = np.random.default_rng(L)
    data = random_records(rng, (8 << 20) // (2 * L), L // 2, L, wrap=80, hdr_hi=10)
    gpu_ctx.forget()
    # (records of 2.5-5 KB: since round 5 the group kernels may prove them -- the guessed chain starts at a node whose call lands
    # exactly --, and the tier is then not needed: forced, so that it stays tested at this length too)
    want = (0, 5) if L == 5000 else (5,)
    table, res = check_same(gpu_ctx, oracle, data)
    assert res.path in want
    table, res = check_same(gpu_ctx, oracle, data[:-1])          # no trailing newline: the final-record rule
    assert res.path in want
    if L == 5000:
        table, res = check_same(gpu_ctx, oracle, data, flags=hipmod.F_FORCE_RANKED)
        assert res.path == 5
    check_same(gpu_ctx, oracle, data[:len(data) * 2 // 3])       # cut inside a record
    check_same(gpu_ctx, oracle, data, offset=len(data) // 2, eof=False)
    decode_same(gpu_ctx, hipmod, oracle, data)
    short = random_records(rng, 20000, 50, 150)
    table, res = check_same(gpu_ctx, oracle, short)
    assert res.path == 3 or (L == 5000 and res.path == 0)     # (a context whose last scans were the group kernels' asks the fast path again later)
    gpu_ctx.forget()


def mutate(rng, data, nedits):
    b = bytearray(data)
    for _ in range(nedits):
        kind = int(rng.integers(0, 8))
        p = int(rng.integers(0, max(1, len(b) - 400)))
        if kind == 0:
            b[p] = 10
        elif kind == 1:
            b[p] = 64
        elif kind == 2:
            b[p] = 43
        elif kind == 3:
            del b[p:p + int(rng.integers(1, 40))]
        elif kind == 4:
            b[p:p] = bytes(rng.integers(33, 75, size=int(rng.integers(1, 40))).astype(np.uint8))
        elif kind == 5:
            q = b.find(b"\n", p)
            if q > 0:
                b[q:q + 1] = b"\r\n"
        elif kind == 6:
            q = b.find(b"\n+\n", p)
            if q > 0:
                b[q:q + 3] = b"\n+extra\n"
        else:
            q = b.find(b"\n", p)
            if q > 0:
                b[q + 40:q + 40] = b"\n"           # wraps one line
    return bytes(b)


@pytest.mark.parametrize("seed", range(12))
def test_fast_path_edits(gpu_ctx, hipmod, oracle, seed):
    """Regular four-line records with a few random edits: the fast path must reproduce the
    oracle (rows, end state, decoded qualities) or decline; tools/stress_fast4.py is the long run."""
    rng = np.random.default_rng(9000 + seed)
    lo, hi = ((100, 160), (20, 60), (250, 400), (1, 30))[seed % 4]
    data = random_records(rng, int(rng.integers(500, 4000)), lo, hi, wrap=0, repeat_hdr=bool(seed & 1))
    data = mutate(rng, data, (0, 1, 3, 12)[(seed // 4) % 4])
    if seed % 3 == 0:
        data = data[:len(data) - int(rng.integers(1, 300))]
    for kw in (dict(), dict(eof=False), dict(offset=len(data) // 3), dict(sentinel=False, offset=5)):
        gpu_ctx.forget()
        want, end, status, off = oracle.scan(data, **kw)
        table, res, qual, qoff = gpu_ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL, **kw)
        assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off
        assert table.shape == want.shape and (table == want).all()
        wq, wqoff = oracle.decode_quals(data, want)
        assert (qoff == wqoff).all() and (qual == wq).all()


@pytest.mark.parametrize("kind", ("single", "wrapped"))
def test_bench_verifiers_accept_and_reject(gpu_ctx, hipmod, pkg, kind):
    """bench.py's full-size checks (closed form of the rows; CSR offsets and decoded bytes against
    the buffer) on a small shard: they pass on the scan's output and fail on a corrupted one."""
    import torch
    from fastqandfurious_amd import sharded
    dev = torch.device("cuda:0")
    sh = sharded.SyntheticShard(gpu_ctx, kind, 8 << 20, 0, 1, dev)
    table = torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device=dev)
    qual = torch.empty(sh.ext.numel(), dtype=torch.int8, device=dev)
    qoff = torch.empty(table.shape[0] + 1, dtype=torch.int64, device=dev)
    out = sh.scan(table, flags=hipmod.F_DECODE_QUAL, qual=qual, qoff=qoff)
    sh.verify(table, out)
    sh.verify_decode(table, out, qual, qoff)
    n = int(out.n_rows)
    k = int(qoff[8].item()) + 3            # (a record of the verifier's sample: the first 64 always are)
    qual[k] += 1
    with pytest.raises(AssertionError):
        sh.verify_decode(table, out, qual, qoff)
    qual[k] -= 1
    qoff[n // 3] += 1
    with pytest.raises(AssertionError):
        sh.verify_decode(table, out, qual, qoff)
    qoff[n // 3] -= 1
    table[n // 2, 3] += 1
    with pytest.raises(AssertionError):
        sh.verify(table, out)


@pytest.mark.parametrize("kind", ("single", "wrapped", "serial", "ranked"))
def test_table_too_small_with_decode(gpu_ctx, hipmod, oracle, pkg, kind):
    """A table with fewer rows than the buffer has records: E_TABLE_FULL with the count needed,
    the rows that fit are right, nothing is written past the table, the offsets or the quality
    buffer (the device-side scratch of the decode is sized from the table too); then a retry
    with enough room gives the whole answer."""
    import torch
    from fastqandfurious_amd import synth
    gpu_ctx.forget()
    data = synth.single(0, 3000, seed=5) if kind != "wrapped" else synth.wrapped(0, 3000, seed=6)[0]
    flags = hipmod.F_DECODE_QUAL | (hipmod.F_FORCE_SERIAL if kind == "serial" else hipmod.F_FORCE_RANKED if kind == "ranked" else 0)
    want, *_ = oracle.scan(data)
    wq, wqoff = oracle.decode_quals(data, want)
    n = len(want)
    dbuf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    for cap in (1, 7, n // 2, n - 1):
        table = torch.full((cap + 4, 6), -7, dtype=torch.int64, device="cuda")
        qoff = torch.full((cap + 1 + 4,), -7, dtype=torch.int64, device="cuda")
        qbuf = torch.full((wq.size + 64,), 99, dtype=torch.int8, device="cuda")
        rc, res = gpu_ctx.scan_device(dbuf.data_ptr(), len(data), table.data_ptr(), cap, flags=flags,
                                      d_qual=qbuf.data_ptr(), qual_cap=wq.size, d_qoff=qoff.data_ptr())
        assert rc == hipmod.E_TABLE_FULL and int(res.n_records) == n
        t = table.cpu().numpy()
        assert (t[:cap] == want[:cap]).all() and (t[cap:] == -7).all()
        assert (qoff[cap + 1:].cpu().numpy() == -7).all()
        assert (qbuf[wq.size:].cpu().numpy() == 99).all()
    table = torch.empty((n, 6), dtype=torch.int64, device="cuda")
    qoff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    qbuf = torch.empty(wq.size, dtype=torch.int8, device="cuda")
    rc, res = gpu_ctx.scan_device(dbuf.data_ptr(), len(data), table.data_ptr(), n, flags=flags,
                                  d_qual=qbuf.data_ptr(), qual_cap=wq.size, d_qoff=qoff.data_ptr())
    assert rc == 0 and int(res.n_records) == n
    assert (table.cpu().numpy() == want).all() and (qoff.cpu().numpy() == wqoff).all() and (qbuf.cpu().numpy() == wq).all()


def test_pool_grows_for_many_dense_tiles(hipmod, oracle):
    """More pooled index entries than a fresh context's pool holds (1 Mi): the scan sizes the
    pool for what was asked and runs again; same rows, with and without the decode."""
    ctx = hipmod.Context(0)                     # a fresh context: its pool is at the initial size
    rec = b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(120000))      # ~2.3 MB, 7 entries per 34 bytes
    data = b"\n" * (3 << 20) + rec
    want, end, status, off = oracle.scan(data)
    assert len(want) == 120000
    table, res = ctx.scan_host(data, table_cap=130000)       # (one call: no retry for a larger table)
    assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off
    assert (table == want).all() and res.retries >= 1
    table, res, qual, qoff = ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL)
    wq, wqoff = oracle.decode_quals(data, want)
    assert (table == want).all() and (qoff == wqoff).all() and (qual == wq).all()


def test_pool_overflow_behind_a_dense_hint(hipmod, oracle):
    """A context that remembers dense input starts the next scan with the dense-budget chain
    kernel queued right behind the index kernel; if that scan outgrows the pool, the queued kernel
    runs on an incomplete index (pooled entries that were not stored read as 0, never past the
    allocation) and the scan is redone with a larger pool."""
    ctx = hipmod.Context(0)
    small = b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(30000))        # dense tiles, fits the pool
    for _ in range(2):
        t, res = ctx.scan_host(small, table_cap=40000)
        assert len(t) == 30000
    big = b"\n" * (3 << 20) + b"".join(b"@q%d\nAC\n+\nII\n" % i for i in range(200000))
    want, end, status, off = oracle.scan(big)
    t, res = ctx.scan_host(big, table_cap=len(want) + 8)
    assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off
    assert (t == want).all() and res.retries >= 1


def test_wrapped_at_size_from_a_deep_offset(gpu_ctx, pkg):
    """S-wrapped, 2.25 GiB, general path: the chain from offset 0 and from the '\\n' in front of a
    record past 2^31 gives the generator's record starts and ends."""
    import torch
    from fastqandfurious_amd import sharded
    dev = torch.device("cuda:0")
    gpu_ctx.forget()
    sh = sharded.SyntheticShard(gpu_ctx, "wrapped", 2304 << 20, 0, 1, dev)
    n = sh.n_per
    st = torch.from_numpy(sh.starts).to(dev)
    table = torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device=dev)
    nbytes = sh.ext_scanned_bytes
    rc, res = gpu_ctx.scan_device(sh.ext.data_ptr(), nbytes, table.data_ptr(), table.shape[0])
    assert rc == 0 and int(res.n_records) == n and res.path == 0 and int(res.end_state) == 0
    assert bool((table[:n, 0] == st[:n]).all()) and bool((table[:n, 5] == st[1:n + 1] - 1).all())
    k0 = n - n // 13
    assert int(sh.starts[k0]) > (1 << 31)
    rc, res = gpu_ctx.scan_device(sh.ext.data_ptr(), nbytes, table.data_ptr(), table.shape[0], offset=int(sh.starts[k0]))
    assert rc == 0 and int(res.n_records) == n - k0 and int(res.end_state) == 0
    assert bool((table[:n - k0, 0] == st[k0:n]).all()) and bool((table[:n - k0, 5] == st[k0 + 1:n + 1] - 1).all())
    # not at eof (the last record is held back), without the sentinel (a shard that is not the
    # first one: the first record of the buffer has no newline in front and is not seen)
    rc, res = gpu_ctx.scan_device(sh.ext.data_ptr(), nbytes, table.data_ptr(), table.shape[0], offset=int(sh.starts[k0]),
                                  eof=False)
    assert rc == 0 and int(res.n_records) == n - k0 - 1 and int(res.end_state) == 1
    assert bool((table[:n - k0 - 1, 0] == st[k0:n - 1]).all())
    rc, res = gpu_ctx.scan_device(sh.ext.data_ptr(), nbytes, table.data_ptr(), table.shape[0], sentinel=False, add=0,
                                  eof=False)
    assert rc == 0 and int(res.n_records) == n - 2 and int(res.end_state) == 1 and res.path == 0
    assert bool((table[:n - 2, 0] == st[1:n - 1]).all()) and bool((table[:n - 2, 5] == st[2:n] - 1).all())
    del sh, table, st
    torch.cuda.empty_cache()


def _with_dense_regions(rng, where, kind):
    """~3 MiB of regular records (single-line or wrapped) with dense regions -- blank lines or
    tiny records, more than 1024 newlines per 16 KiB tile -- at the start / in the middle / at
    the end."""
    body = random_records(rng, 9000, 100, 160, wrap=(70 if kind == "wrapped" else 0))
    blank = b"\n" * int(rng.integers(20000, 150000))
    tiny = b"".join(b"@t%d\nAC\n+\nII\n" % i for i in range(int(rng.integers(3000, 9000))))
    cut = len(body) // 2
    cut = body.index(b"\n@", cut) + 1
    parts = []
    if "start" in where:
        parts.append(blank)
    parts.append(body[:cut])
    if "middle" in where:
        parts.append(tiny if "tiny" in where else blank)
    parts.append(body[cut:])
    if "end" in where:
        parts.append(blank if "tiny" not in where else tiny + blank)
    return b"".join(parts)


@pytest.mark.parametrize("kind", ("single", "wrapped"))
@pytest.mark.parametrize("where", ("start", "middle", "end", "start middle end", "middle tiny", "end tiny"))
def test_dense_regions_are_walked_group_by_group(hipmod, oracle, kind, where):
    """A dense region (tiles over their slot: k_chain_wave does not take them) is walked group by
    group, a window of index entries at a time (k_dense_walk), instead of sending the whole buffer
    to the serial walker: same rows, end state and decoded qualities as the oracle, parallel path."""
    ctx = hipmod.Context(0)
    rng = np.random.default_rng(len(where) * 7 + len(kind))
    data = _with_dense_regions(rng, where, kind)
    for kw in (dict(), dict(eof=False), dict(offset=len(data) // 3)):
        for trunc in (0, 37):
            d = data[:len(data) - trunc]
            want, end, status, off = oracle.scan(d, **kw)
            table, res, qual, qoff = ctx.scan_host(d, flags=hipmod.F_DECODE_QUAL, table_cap=len(want) + 8, **kw)
            assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off, (kw, trunc)
            assert table.shape == want.shape and (table == want).all(), (kw, trunc)
            wq, wqoff = oracle.decode_quals(d, want)
            assert (qoff == wqoff).all() and (qual == wq).all()
            if "tiny" not in where:
                assert res.path in (0, 2), (kw, trunc, res.path)      # not the serial walker


@pytest.mark.parametrize("general", (False, True))
@pytest.mark.parametrize("seed", range(4))
def test_very_short_reads_every_tile_dense(hipmod, oracle, seed, general):
    """Reads of a few bases with short headers: under 16 bytes per line, every tile over its slot
    (the shape of the reference's own test template, tests.py:8-35).  Plain four-line records: the
    fast path takes them (its row kernel's DENSE instantiation reads the overflow pool) -- and with
    the fast path switched off the groups are walked in parallel from guessed entries
    (k_dense_walk); qualities full of '@' and '+' make false candidates for the guesses.  Same
    result as the oracle either way, never through the whole-buffer serial walker."""
    ctx = hipmod.Context(0)
    fl = hipmod.F_FORCE_GENERAL if general else 0
    rng = np.random.default_rng(300 + seed)
    qch = np.frombuffer(b"@+I5@", dtype=np.uint8)
    parts = []
    for i in range(90000):
        L = int(rng.integers(1, 14))
        q = rng.choice(qch, size=L).tobytes()
        parts.append(b"@%d\n" % i + b"ACGTN"[:1] * L + b"\n+\n" + q + b"\n")
    data = b"".join(parts)
    if seed & 1:
        data = data[:len(data) - 5]
    for kw in (dict(), dict(eof=False), dict(offset=len(data) // 2)):
        want, end, status, off = oracle.scan(data, **kw)
        table, res, qual, qoff = ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL | fl, table_cap=len(want) + 8, **kw)
        assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off, kw
        assert table.shape == want.shape and (table == want).all(), kw
        wq, wqoff = oracle.decode_quals(data, want)
        assert (qoff == wqoff).all() and (qual == wq).all()
        # (a stream cut inside its last record ends in a call the closed form cannot vouch for -- no "\n+" behind the
        # sequence line: the general kernels take it, and the context remembers that for its next scans)
        assert res.path in ((0, 2) if general else (3,) if not (seed & 1) else (0, 2, 3)), (kw, res.path)
        table, res = ctx.scan_host(data, flags=fl, table_cap=len(want) + 8, **kw)          # (and without the decode)
        assert table.shape == want.shape and (table == want).all(), kw
        assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off, kw


def _template_records(n, multiline=False):
    """n copies of the reference's test template (/root/reference/tests.py:8-35: '@foo#2', 8 bases, '+', 8 qualities
    starting with a digit; 27 bytes, 6.75 per line), numbered so that no two are alike."""
    out = []
    for i in range(n):
        h = b"foo#%d" % (i % 977)
        out.append(b"@" + h + b"\nAATTGCCG\n+\n3425@!#!\n")
    return b"".join(out)


@pytest.mark.parametrize("general", (False, True))
def test_reference_template_shape_dense(hipmod, oracle, general):
    """The reference's own test template repeated (27-byte records): every tile dense, every '+' line's successor a
    '\n@'; a stretch of blank lines, a truncated tail and a search offset on top."""
    ctx = hipmod.Context(0)
    fl = hipmod.F_FORCE_GENERAL if general else 0
    base = _template_records(70000)
    for data, kw in ((base, dict()), (base[:len(base) - 3], dict()), (base, dict(eof=False)),
                     (base, dict(offset=len(base) // 3)), (base[:900000] + b"\n" * 70000 + base[900000 - 27 * 5:], dict()),
                     (base + b"\n" * 40000, dict())):
        want, end, status, off = oracle.scan(data, **kw)
        table, res, qual, qoff = ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL | fl, table_cap=len(want) + 8, **kw)
        assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off, kw
        assert table.shape == want.shape and (table == want).all(), kw
        wq, wqoff = oracle.decode_quals(data, want)
        assert (qoff == wqoff).all() and (qual == wq).all()
        assert res.path != 1, (kw, res.path)


def dense_mess(rng, target=400000, hostile=True):
    """~target bytes: blocks of tiny records (1-12 bases, headers of 0-5 characters), blank lines, tiny wrapped
    records, regular records; optionally random single-byte edits."""
    qh = np.frombuffer(b"@+@+II5#\n" if hostile else b"IIII5#?", dtype=np.uint8)
    parts, size, i = [], 0, 0
    while size < target:
        u = rng.random()
        if u < 0.55:                                        # a block of tiny four-line records
            blk = []
            for _ in range(int(rng.integers(50, 4000))):
                L = int(rng.integers(1, 13))
                h = (b"%d" % i)[:int(rng.integers(0, 6))]
                q = rng.choice(qh[qh != 10], size=L).tobytes()
                blk.append(b"@" + h + b"\n" + b"ACGTN"[i % 5:i % 5 + 1] * L + b"\n+\n" + q + b"\n")
                i += 1
            p = b"".join(blk)
        elif u < 0.70:                                      # blank lines
            p = b"\n" * int(rng.integers(1, 40000))
        elif u < 0.80:                                      # tiny records wrapped at 2-4 columns
            blk = []
            for _ in range(int(rng.integers(20, 800))):
                L = int(rng.integers(3, 20)); w = int(rng.integers(2, 5))
                s = b"A" * L; q = rng.choice(qh[qh != 10], size=L).tobytes()
                f = lambda a: b"\n".join(a[k:k + w] for k in range(0, L, w))
                blk.append(b"@w%d\n" % i + f(s) + b"\n+\n" + f(q) + b"\n")
                i += 1
            p = b"".join(blk)
        else:                                               # regular records
            blk = []
            for _ in range(int(rng.integers(20, 600))):
                L = int(rng.integers(100, 300))
                q = rng.choice(np.frombuffer(bytes(range(35, 74)), dtype=np.uint8), size=L).tobytes()
                blk.append(b"@reg%d\n" % i + b"C" * L + b"\n+\n" + q + b"\n")
                i += 1
            p = b"".join(blk)
        parts.append(p); size += len(p)
    a = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
    if hostile:
        for _ in range(int(rng.integers(0, 12))):
            a[int(rng.integers(0, a.size))] = int(rng.choice(np.frombuffer(b"\n@+A", dtype=np.uint8)))
    return a.tobytes()


@pytest.mark.parametrize("seed", range(6))
def test_differential_dense_mess(hipmod, oracle, seed):
    """Seeded differential test of the dense tiers against the oracle (tools/stress_dense.py is the long run of the
    same): fast path where it stands, general path forced, not at eof, from an offset."""
    rng = np.random.default_rng(77000 + seed)
    data = dense_mess(rng, int(rng.integers(100000, 900000)), hostile=(seed % 3 != 0))
    if seed % 2:
        data = data[:-int(rng.integers(1, 40))]
    ctx = hipmod.Context(0)
    for kw, fl in ((dict(), 0), (dict(), hipmod.F_FORCE_GENERAL), (dict(eof=False), 0),
                   (dict(offset=len(data) // 3), hipmod.F_FORCE_GENERAL)):
        want, end, status, off = oracle.scan(data, **kw)
        table, res, qual, qoff = ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL | fl, table_cap=len(want) + 8, **kw)
        assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off, (kw, fl)
        assert table.shape == want.shape and (table == want).all(), (kw, fl)
        wq, wqoff = oracle.decode_quals(data, want)
        assert (qoff == wqoff).all() and (qual == wq).all(), (kw, fl)


def test_dense_patch_in_regular_input(hipmod, oracle):
    """100 KB of 10-base reads + blank lines inside regular 150-base records, and blank lines at the very end
    (tools/cliffs.py at test size): the usual configuration of the general path with the dense groups walked."""
    from fastqandfurious_amd import synth
    ctx = hipmod.Context(0)
    reg = bytes(synth.single(0, 12000, seed=42))
    tiny = b"".join(b"@t%06d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(3200))
    mid = 6000 * 322
    for data in (reg[:mid] + tiny + b"\n" * 300 + reg[mid:], reg[:mid] + b"\n" * 100000 + reg[mid:],
                 reg[:len(reg) - 700000] + b"\n" * 700000, reg[:mid] + tiny[:32 * 40] + reg[mid:]):
        want, end, status, off = oracle.scan(data)
        for _ in range(2):
            table, res, qual, qoff = ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL, table_cap=len(want) + 8)
            assert int(res.end_state) == end and int(res.last_status) == status and int(res.end_offset) == off
            assert table.shape == want.shape and (table == want).all()
            wq, wqoff = oracle.decode_quals(data, want)
            assert (qoff == wqoff).all() and (qual == wq).all()
            assert res.path in (0, 2, 3), res.path


def test_short_wrapped_reads_take_the_dense_configuration(hipmod, oracle):
    """Reads of 100 bases wrapped at 80 columns: ~32 bytes per line, windows over the usual
    configuration's LDS budget but no dense tile.  They belong to the dense configuration of the
    chain kernel (path 2), not to the group walker (6x slower on such input)."""
    ctx = hipmod.Context(0)
    rng = np.random.default_rng(77)
    data = random_records(rng, 20000, 100, 100, wrap=80, hdr_hi=12)
    want, end, status, off = oracle.scan(data)
    table, res = ctx.scan_host(data, table_cap=len(want) + 8)
    assert (table == want).all() and int(res.end_state) == end and int(res.last_status) == status
    assert res.path == 2


# ---- BASELINE config sizes (configs[2], configs[3]) under -m gpu ------------------------------------
def test_decode_at_config_size_past_4g_of_qualities(gpu_ctx, hipmod, pkg):
    """configs[2]: 10 GiB of S-single with the quality -> int8 decode.  More than 2^32 decoded
    bytes (64-bit CSR offsets, stream directory, p4 copy); rows against the generator's closed
    form, offsets against the running sum of pos5 - pos4 over ALL rows, decoded bytes of ~4000
    records spread over the buffer against the buffer's own bytes - 33."""
    import torch
    from fastqandfurious_amd import sharded
    dev = torch.device("cuda:0")
    gpu_ctx.forget()
    sh = sharded.SyntheticShard(gpu_ctx, "single", 10 << 30, 0, 1, dev)
    table = torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device=dev)
    qual = torch.empty(sh.n_own_bytes // 2 + 4096, dtype=torch.int8, device=dev)
    qoff = torch.empty(table.shape[0] + 1, dtype=torch.int64, device=dev)
    out = sh.scan(table, flags=hipmod.F_DECODE_QUAL, qual=qual, qoff=qoff)
    assert out.res.path == 3 and int(out.n_rows) == sh.n_per == 33346019
    assert int(out.res.n_qual_bytes) == 150 * sh.n_per > (1 << 32)
    sh.verify(table, out)
    sh.verify_decode(table, out, qual, qoff)
    # the records on either side of the 2^32-th decoded byte
    k = (1 << 32) // 150
    for r in range(k - 2, k + 3):
        a = int(table[r, 4].item())
        src = (sh.ext[a:a + 150].to(torch.int16) - 33).to(torch.int8)
        q = int(qoff[r].item())
        assert q == 150 * r and bool((qual[q:q + 150] == src).all())
    # column selection at this size: the sequences, 5 GB of them, packed (offsets past 2^32)
    from fastqandfurious_amd import index
    n = int(out.n_rows)
    seqs, soff = index.select_column_device(gpu_ctx, sh.ext, table[:n], "sequence")
    assert seqs.numel() == 150 * n and bool((soff == 150 * torch.arange(n + 1, device=dev)).all())
    for r in list(range(k - 2, k + 3)) + [0, n - 1] + torch.randint(0, n, (200,)).tolist():
        a = int(table[r, 2].item())
        assert bool((seqs[150 * r:150 * r + 150].view(torch.uint8) == sh.ext[a:a + 150]).all())
    del seqs, soff
    # the general kernels on the same buffer (offsets through k_expand): same table, same stream
    gpu_ctx.forget()
    import os
    os.environ["FFQ_NO_FAST4"] = "1"
    try:
        t2 = torch.empty_like(table)
        q2 = torch.empty_like(qual)
        o2 = torch.empty_like(qoff)
        out2 = sh.scan(t2, flags=hipmod.F_DECODE_QUAL, qual=q2, qoff=o2)
    finally:
        del os.environ["FFQ_NO_FAST4"]
    n = int(out.n_rows)
    assert out2.res.path == 0 and int(out2.n_rows) == n
    assert bool((t2[:n] == table[:n]).all()) and bool((o2[:n + 1] == qoff[:n + 1]).all())
    nq = int(out.res.n_qual_bytes)
    assert bool((q2[:nq] == qual[:nq]).all())
    # ... and the single pass (FFQ_F_SINGLE_PASS, csrc/ffq_fused.h: the index kernel writes the decoded bytes
    # itself, segmented; the input is read once): same table, and for EVERY record the same 150 bytes
    gpu_ctx.forget()
    q3 = torch.zeros(((sh.ext.numel() + 16383) >> 14) * hipmod.SEG_STRIDE, dtype=torch.int8, device=dev)
    t2.zero_(); o2.zero_()
    torch.cuda.synchronize()      # (torch's fills run on torch's stream, the scan on the context's: they must be through)
    out3 = sh.scan(t2, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS, qual=q3, qoff=o2)
    assert out3.res.path == 6 and int(out3.n_rows) == n
    assert bool((t2[:n] == table[:n]).all())
    sh.verify_decode(t2, out3, q3, o2)
    ar = torch.arange(150, device=dev)
    for c0 in range(0, n, 1 << 20):
        c1 = min(n, c0 + (1 << 20))
        got = q3[(o2[c0:c1, None] + ar[None, :]).reshape(-1)]
        assert bool((got == qual[150 * c0:150 * c1]).all()), "segmented decode differs from the packed stream"
    del q3
    del sh, table, qual, qoff, t2, q2, o2
    torch.cuda.empty_cache()


def test_wrapped_at_config_size_all_columns(gpu_ctx, hipmod, pkg):
    """configs[3]: 10 GiB of S-wrapped (50-300 bp, 80-column wrap, general path): all six columns
    of every row against the generator's closed form, with and without the decode."""
    import torch
    from fastqandfurious_amd import sharded
    dev = torch.device("cuda:0")
    gpu_ctx.forget()
    sh = sharded.SyntheticShard(gpu_ctx, "wrapped", 10 << 30, 0, 1, dev)
    table = torch.empty((sh.max_records + 64, 6), dtype=torch.int64, device=dev)
    out = sh.scan(table)
    assert out.res.path == 0 and int(out.n_rows) == sh.n_per
    sh.verify(table, out)
    table.zero_()
    torch.cuda.synchronize()
    qual = torch.empty(sh.n_own_bytes // 2 + 4096, dtype=torch.int8, device=dev)
    qoff = torch.empty(table.shape[0] + 1, dtype=torch.int64, device=dev)
    out = sh.scan(table, flags=hipmod.F_DECODE_QUAL, qual=qual, qoff=qoff)
    assert int(out.res.n_qual_bytes) > (1 << 32)
    sh.verify(table, out)
    sh.verify_decode(table, out, qual, qoff)
    del sh, table, qual, qoff
    torch.cuda.empty_cache()


def test_poll_result_instead_of_an_end_event(gpu_ctx, hipmod, golden, oracle, pkg):
    """FFQ_F_POLL_RESULT: the last kernel of a scan says "done" through host-mapped memory and
    ffq_scan_wait polls; same results on every tier, alone and with two contexts one scan ahead."""
    import torch
    from fastqandfurious_amd import synth
    fl = hipmod.F_POLL_RESULT
    for fn in FILES:
        check_same(gpu_ctx, oracle, golden_file(fn), flags=fl)
        check_same(gpu_ctx, oracle, golden_file(fn), flags=fl | hipmod.F_FORCE_SERIAL)
        check_same(gpu_ctx, oracle, golden_file(fn), flags=fl | hipmod.F_FORCE_RANKED)
        decode_same(gpu_ctx, hipmod, oracle, golden_file(fn), flags=fl)        # (ignored with the decode)
    for name, ent in list(golden["edge"].items()):
        check_same(gpu_ctx, oracle, bytes.fromhex(ent["data"]), flags=fl)
    for ent in golden["fuzz"][:120]:
        check_same(gpu_ctx, oracle, bytes.fromhex(ent["data"]), flags=fl)
    datas = [synth.single(0, 40000, seed=42), synth.wrapped(0, 30000, seed=43)[0], synth.single(7, 25000, seed=5)[:-50]]
    wants = [oracle.scan(d) for d in datas]
    ctx2 = hipmod.Context(share=gpu_ctx)
    ctxs = (gpu_ctx, ctx2)
    bufs = [torch.from_numpy(np.array(d)).cuda() for d in datas]
    tabs = [torch.empty((60000, 6), dtype=torch.int64, device="cuda") for _ in range(2)]
    order = [i % 3 for i in range(12)]
    ctxs[0].scan_submit(bufs[order[0]].data_ptr(), len(datas[order[0]]), tabs[0].data_ptr(), 60000, flags=fl)
    for i in range(1, len(order) + 1):
        if i < len(order):
            ctxs[i & 1].scan_submit(bufs[order[i]].data_ptr(), len(datas[order[i]]), tabs[i & 1].data_ptr(), 60000, flags=fl)
        rc, res = ctxs[(i - 1) & 1].scan_wait()
        want, end, status, off = wants[order[i - 1]]
        assert rc == 0 and int(res.n_records) == len(want) and int(res.end_state) == end and int(res.end_offset) == off
        assert (tabs[(i - 1) & 1][:len(want)].cpu().numpy() == want).all()
        assert res.ms_index > 0
    ctx2.close()


@pytest.mark.parametrize("wrap,hdr_hi", ((8, 4), (12, 4), (20, 8), (30, 8), (40, 20), (45, 40), (60, 40), (80, 40)))
def test_wrap_widths_through_the_group_kernel(gpu_ctx, hipmod, oracle, wrap, hdr_hi):
    """Wrapped records at line lengths from 9 to 81 bytes: 200 to 1800 index entries per 16 KiB tile.  The group
    kernel takes a tile's entries 384 per pass (six per lane): one pass (up to 384 entries), two (up to 768), the
    redo loop (more), and the dense configuration beyond 1024 -- rows, end state and the decoded qualities
    against the oracle, with quality lines that start with '@' and '+' mixed in."""
    rng = np.random.default_rng(wrap * 131 + hdr_hi)
    blob = bytearray(random_records(rng, 6000, 50, 300, wrap=wrap, hdr_hi=hdr_hi, repeat_hdr=(wrap % 3 == 0)))
    # quality lines that begin with '@' / '+': overwrite the first byte of some lines in front of which no '+' line ends
    nl = np.flatnonzero(np.frombuffer(bytes(blob), dtype=np.uint8) == 10)
    for p in rng.choice(nl[:-2], size=400, replace=False):
        b = blob[p + 1]
        if b not in (ord("@"), ord("+"), 10) and 35 <= b < 74 and chr(b) not in "ACGTN":
            blob[p + 1] = ord("@") if rng.integers(2) else ord("+")
    data = np.frombuffer(bytes(blob), dtype=np.uint8)
    gpu_ctx.forget()
    res = decode_same(gpu_ctx, hipmod, oracle, data)
    assert res.path in (0, 2, 5)
    gpu_ctx.forget()
    decode_same(gpu_ctx, hipmod, oracle, data, eof=False)
