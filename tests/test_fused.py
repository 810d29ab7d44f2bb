"""The single-pass index + decode (csrc/ffq_fused.h) against the oracle, and its refusals.

With FFQ_F_DECODE_QUAL | FFQ_F_SINGLE_PASS the caller accepts the decoded qualities SEGMENTED (record i =
qual[qoff[i] : qoff[i] + pos5 - pos4], gaps between records allowed); on plain four-line input the index
kernel itself then writes them (res.path == 6).  That is speculation -- every fourth line is a record's
quality, whole -- verified by the row kernel; whatever it cannot vouch for must come out of the two-pass
kernels instead (path 3 / 0, packed), with the same table and the same bytes per record as the oracle's
restatement of array('b').frombytes(buf[pos4:pos5]); arrayadd_b(q, -33)
(/root/reference/doc/user-guide.rst:126-141).
"""
import numpy as np
import pytest

from test_gpu_parity import decode_same as _decode_same, random_records

pytestmark = pytest.mark.gpu


def decode_same(ctx, hipmod, oracle, data, flags=0, qual_room=None, **kw):
    """Table, offsets of every record's bytes and the bytes themselves against the oracle; the stream may have gaps."""
    want, *_ = oracle.scan(data, **kw)
    wq, wqoff = oracle.decode_quals(data, want)
    table, res, qual, qoff = ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS | flags, qual_room=qual_room, **kw)
    assert table.shape == want.shape and (table == want).all()
    n = len(want)
    lens = want[:, 5] - want[:, 4] if n else np.zeros(0, np.int64)
    assert qoff.shape[0] == n + 1
    if n:
        assert (qoff[1:n] >= qoff[:n - 1] + lens[:n - 1]).all(), "records overlap or are out of order"
        assert int(qoff[n]) == int(qoff[n - 1] + lens[n - 1]) == int(res.n_qual_bytes)
        idx = np.repeat(qoff[:n] - wqoff[:n], lens) + np.arange(wq.size)
        assert (qual[idx] == wq).all(), "decoded bytes differ"
    if res.path & hipmod.PATH_IN_PLACE:               # the general path's one pass (round 6, tests/test_wide.py): every byte in place
        assert n == 0 or (qoff[:n] == want[:, 4] - kw.get("add", -1 if kw.get("sentinel", True) else 0) - (1 if kw.get("sentinel", True) else 0)).all()
    elif res.path != 6:                               # the two passes: packed
        assert (qoff == wqoff).all()
    return res


@pytest.fixture
def hipmod(pkg):
    from fastqandfurious_amd import hip
    return hip


@pytest.mark.parametrize("nrec,first", ((1, 0), (3, 5), (50, 7), (51, 0), (2000, 0), (12345, 1000), (60000, 5), (400000, 0)))
def test_fused_synth_single(gpu_ctx, hipmod, oracle, nrec, first):
    from fastqandfurious_amd import synth
    data = synth.single(first, nrec, seed=42)
    res = decode_same(gpu_ctx, hipmod, oracle, data)
    # (a tile -- here: the short last one -- without a sequence line followed by a '+' line inside it gives the
    # single pass nothing to tell the line types by: such a buffer goes to the two passes)
    tail = len(data) % 16384
    if nrec > 1 and (tail == 0 or tail > 700):
        assert res.path == 6
    else:
        assert res.path in (3, 6)
    assert hipmod.SEG_STRIDE == 8704
    gpu_ctx.forget()
    assert _decode_same(gpu_ctx, hipmod, oracle, data).path == 3          # without the flag: the two passes


def test_fused_shapes(gpu_ctx, hipmod, oracle):
    rng = np.random.default_rng(77)
    for name, mk, want in (
            ("illumina-repeat-header", lambda: random_records(rng, 30000, 151, 151, repeat_hdr=True), 6),
            ("variable", lambda: random_records(rng, 30000, 20, 400), 6),
            ("short-lines", lambda: random_records(rng, 40000, 1, 40, hdr_hi=8), None),       # lines under 16 bytes: byte-wise tails
            ("kilobase", lambda: random_records(rng, 600, 2000, 6000), None),                # few lines per tile
            ("lines-longer-than-a-tile", lambda: random_records(rng, 40, 30000, 90000), 3),
            ("wrapped", lambda: random_records(rng, 5000, 1, 700, wrap=61), 0)):
        data = mk()
        gpu_ctx.forget()
        res = decode_same(gpu_ctx, hipmod, oracle, data)
        if want is not None:
            assert res.path == want, (name, res.path)
        gpu_ctx.forget()
        decode_same(gpu_ctx, hipmod, oracle, data[:-1])                      # no trailing newline: 'Incomplete final quality string'
        gpu_ctx.forget()
        decode_same(gpu_ctx, hipmod, oracle, data[:len(data) * 2 // 3], eof=False)
        gpu_ctx.forget()
        decode_same(gpu_ctx, hipmod, oracle, data[:len(data) * 2 // 3], eof=True)


def test_fused_refuses_what_it_cannot_vouch_for(gpu_ctx, hipmod, oracle):
    """Valid for the reference, not for the single pass: each must come out right through the other kernels."""
    rng = np.random.default_rng(78)
    good = random_records(rng, 20000, 100, 150)
    cases = {
        # a quality line LONGER than its sequence line: the reference cuts it at pos5 = pos4 + len(seq)
        # (_fastqandfurious.c:129) and finds the next "\n@" behind it -- the single pass decoded the whole line
        "long-quality-line": good[:3000000].rsplit(b"\n@", 1)[0] + b"\n@odd\nACGT\n+\nIIIIIIII\n" + good[:1000000],
        # text in front of the first record that could pass for lines of one
        "leading-text": b"# produced by a tool\n# and a second line\n# third\n# fourth\n# fifth\n" + good,
        # a blank line between two records
        "blank-line": good[:2000000].rsplit(b"\n@", 1)[0] + b"\n\n" + good[:500000],
        # one wrapped record among thousands of plain ones
        "one-wrapped": good[:1500000].rsplit(b"\n@", 1)[0] + b"\n@w\nACGTAC\nGTAC\n+\nIIIIII\nIIII\n" + good[:700000],
    }
    for name, data in cases.items():
        gpu_ctx.forget()
        res = decode_same(gpu_ctx, hipmod, oracle, data)
        assert res.path != 6, name
        # (the context remembers the refusal: the next scan goes to the two passes directly and is right too)
        decode_same(gpu_ctx, hipmod, oracle, data)


def test_fused_search_offsets_and_sentinel(gpu_ctx, hipmod, oracle):
    from fastqandfurious_amd import synth
    data = synth.single(3, 5000, seed=42).tobytes()
    for kw in (dict(sentinel=False, add=0), dict(sentinel=False, add=0, offset=1), dict(sentinel=False, add=0, offset=15),
               dict(sentinel=True, offset=1), dict(sentinel=False, add=0, offset=400)):
        gpu_ctx.forget()
        decode_same(gpu_ctx, hipmod, oracle, data, **kw)
    # what the stream front end scans: the fill starts with the last byte of the previous record's quality
    carry = b"I\n" + data
    gpu_ctx.forget()
    res = decode_same(gpu_ctx, hipmod, oracle, carry, sentinel=False, add=0)
    assert res.path == 6


def test_fused_remembers_and_recovers(gpu_ctx, hipmod, oracle):
    """After a refusal the context skips the attempt for a while, then tries again."""
    from fastqandfurious_amd import synth
    rng = np.random.default_rng(79)
    longs = random_records(rng, 30, 40000, 60000)
    plain = synth.single(0, 20000, seed=42)
    gpu_ctx.forget()
    assert decode_same(gpu_ctx, hipmod, oracle, longs).path == 3
    paths = [decode_same(gpu_ctx, hipmod, oracle, plain).path for _ in range(18)]
    assert paths[0] == 3 and paths[-1] == 6


def test_fused_beside_other_work_on_the_device(gpu_ctx, hipmod, oracle):
    """Kernels of another stream running beside it (the persistent-grid version of round 3 deadlocked there
    until its timeout): one workgroup per tile waits for nobody."""
    import time
    import torch
    from fastqandfurious_amd import synth
    data = synth.single(0, 2000000, seed=42)              # 644 MB
    want, *_ = oracle.scan(data)
    wq, wqoff = oracle.decode_quals(data, want)
    dbuf = torch.from_numpy(data.copy()).cuda()
    n = len(want)
    table = torch.empty((n + 8, 6), dtype=torch.int64, device="cuda")
    qual = torch.empty(((data.size + 16383) >> 14) * hipmod.SEG_STRIDE, dtype=torch.int8, device="cuda")
    qoff = torch.empty(n + 9, dtype=torch.int64, device="cuda")
    big = torch.empty(4 << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        gpu_ctx.forget()
        for _ in range(4):
            big.fill_(rep)                                  # torch's stream: beside the scan, not in front of it
        t0 = time.time()
        rc, res = gpu_ctx.scan_device(dbuf.data_ptr(), data.size, table.data_ptr(), n + 8, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS,
                                      d_qual=qual.data_ptr(), qual_cap=qual.numel(), d_qoff=qoff.data_ptr())
        assert time.time() - t0 < 5.0
        torch.cuda.synchronize()
        assert rc == 0 and res.path == 6 and int(res.n_records) == n
        assert (table[:n].cpu().numpy() == want).all()
        qo = qoff[:n].cpu().numpy()
        idx = np.repeat(qo - wqoff[:n], want[:, 5] - want[:, 4]) + np.arange(wq.size)
        assert (qual.cpu().numpy()[idx] == wq).all()


@pytest.mark.parametrize("world", (2, 8))
def test_fused_in_byte_range_shards(gpu_ctx, hipmod, oracle, world):
    """The single pass inside the sharded scan: every logical rank scans its [tail | own | head] view with
    FFQ_F_SINGLE_PASS; the rows it owns carry exact starts into ITS OWN segmented quality buffer."""
    import torch
    from test_sharded import make_stream, expected, run_local, check_rows, bounds_for, _hip_backends
    stream = make_stream("single")
    want, err = expected(oracle, stream)
    assert err is None
    wq, wqoff = oracle.decode_quals(stream, want)
    t = torch.from_numpy(stream.copy()).cuda()
    bounds = bounds_for(stream.size, world, 0, 48)
    make, made = _hip_backends(gpu_ctx)
    res = run_local(t, bounds, make, decode=True, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS, native=True)
    check_rows(res, bounds, want)
    base = 0
    for r in range(world):
        out, table, qual, qoff = res[r]
        assert out.res.path == 6
        n_own = out.row_hi - out.row_lo
        rows = table[out.row_lo:out.row_hi].cpu().numpy()
        lens = rows[:, 5] - rows[:, 4]
        qo = qoff[out.row_lo:out.row_hi].cpu().numpy()
        q = qual.cpu().numpy()
        idx = np.repeat(qo - (wqoff[base:base + n_own] - wqoff[base]), lens) + np.arange(int(lens.sum()))
        assert (q[idx] == wq[int(wqoff[base]):int(wqoff[base + n_own])]).all(), "rank %d: decoded qualities differ" % r
        base += n_own
    for c in made.values():
        c.close()


@pytest.mark.parametrize("shift", (1, 7, 8))
def test_unaligned_quality_buffer_gets_the_packed_stream(gpu_ctx, hipmod, oracle, shift):
    """The segmented output is written in whole 16-byte pieces: a d_qual that is not 16-byte aligned is served by
    the two passes (packed, any alignment) even if the single pass was asked for."""
    from fastqandfurious_amd import synth
    data = synth.single(0, 20000, seed=42)
    want, *_ = oracle.scan(data)
    wq, wqoff = oracle.decode_quals(data, want)
    n, nb = len(want), data.size
    cap = (nb // 16384 + 1) * hipmod.SEG_STRIDE + 64
    ctx = gpu_ctx
    d_buf, d_tab = ctx.dev_alloc(nb + 16), ctx.dev_alloc(48 * (n + 8))
    d_qual, d_qoff = ctx.dev_alloc(cap + 32), ctx.dev_alloc(8 * (n + 9))
    try:
        ctx.h2d(d_buf, data)
        ctx.forget()
        for flags in (hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS, hipmod.F_DECODE_QUAL):
            rc, res = ctx.scan_device(d_buf, nb, d_tab, n + 8, flags=flags, d_qual=d_qual + shift, qual_cap=cap, d_qoff=d_qoff)
            assert rc == 0 and res.path != 6 and res.n_records == n
            qual, qoff, table = np.empty(wq.size, np.int8), np.empty(n + 1, np.int64), np.empty((n, 6), np.int64)
            ctx.d2h(qual, d_qual + shift); ctx.d2h(qoff, d_qoff); ctx.d2h(table, d_tab)
            assert (table == want).all() and (qoff == wqoff).all() and (qual == wq).all()
        # aligned: the single pass
        rc, res = ctx.scan_device(d_buf, nb, d_tab, n + 8, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS, d_qual=d_qual,
                                  qual_cap=cap, d_qoff=d_qoff)
        assert rc == 0 and res.path == 6
    finally:
        for p in (d_buf, d_tab, d_qual, d_qoff):
            ctx.dev_free(p)


@pytest.mark.parametrize("value", (0, 1, -1, 31, -128, 127, 200, -129))
def test_quality_add_values_wrap_like_arrayadd_b(gpu_ctx, hipmod, oracle, value):
    """arrayadd_b adds (int8)value with two's-complement wrap (/root/reference/src/_fastqandfurious.c:161-185) to
    ANY byte: quality lines that hold every byte value but '\\n', through the single pass and the two passes."""
    rng = np.random.default_rng(1234 + value)
    recs = []
    for i in range(3000):
        n = int(rng.integers(1, 200))
        q = rng.integers(0, 256, size=n, dtype=np.uint8)
        q[q == 10] = 11
        if q[0] in (ord("@"), ord("+")) and i % 3:
            q[0] = 200
        recs.append(b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)) + b"\n+\n" + q.tobytes() + b"\n")
    data = np.frombuffer(b"".join(recs), dtype=np.uint8)
    want, *_ = oracle.scan(data)
    assert len(want) == 3000
    wq, wqoff = oracle.decode_quals(data, want, value)
    lens = want[:, 5] - want[:, 4]
    for flags in (hipmod.F_SINGLE_PASS, 0):
        gpu_ctx.forget()
        table, res, qual, qoff = gpu_ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL | flags, qual_add=value)
        assert (table == want).all() and res.path == (6 if flags else 3)
        idx = np.repeat(qoff[:3000] - wqoff[:3000], lens) + np.arange(wq.size)
        assert (qual[idx] == wq).all()


# ---- the in-place layout (k_scan_ident): room for FFQ_INPLACE_STRIDE bytes per tile admits lines of any length ----------

def long_records(rng, n, lo, hi, base_quals=0.0):
    """Four-line records with reads of lo..hi bases; qualities over the whole printable range ('@' and '+' at line starts
    included), or -- with probability base_quals -- made of base letters only."""
    qa = np.frombuffer(bytes(range(33, 127)), dtype=np.uint8)
    ba = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)
    parts = []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        h = b"r%d " % i + b"x" * int(rng.integers(0, 60))
        seq = rng.choice(ba[:5], size=L).tobytes()
        qual = rng.choice(ba if rng.random() < base_quals else qa, size=L).tobytes()
        parts.append(b"@" + h + b"\n" + seq + b"\n+" + (h if rng.random() < 0.2 else b"") + b"\n" + qual + b"\n")
    return b"".join(parts)


@pytest.mark.parametrize("n,lo,hi", ((3000, 300, 700), (600, 2000, 6000), (60, 30000, 90000), (400, 50, 40000), (3, 200000, 300000)))
def test_in_place_takes_long_lines(gpu_ctx, hipmod, oracle, n, lo, hi):
    rng = np.random.default_rng(n + lo)
    data = long_records(rng, n, lo, hi)
    room = hipmod.INPLACE_STRIDE
    gpu_ctx.forget()
    res = decode_same(gpu_ctx, hipmod, oracle, data, qual_room=room)
    assert res.path == 6 and res.retries == 1            # (the segmented pass refused the shape first)
    res = decode_same(gpu_ctx, hipmod, oracle, data, qual_room=room)
    assert res.path == 6 and res.retries == 0            # ... and the context remembers
    # in place means in place: record i's bytes lie at the offset pos4 has in the buffer
    table, res, qual, qoff = gpu_ctx.scan_host(data, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS, qual_room=room)
    assert res.path == 6 and (qoff[:-1] == table[:, 4]).all()
    for cut in (data[:-1], data[:len(data) * 2 // 3]):
        for eof in (True, False):
            gpu_ctx.forget()
            decode_same(gpu_ctx, hipmod, oracle, cut, qual_room=room, eof=eof)
    gpu_ctx.forget()
    assert decode_same(gpu_ctx, hipmod, oracle, data).path == 3          # room for segments only: the two passes, as before
    gpu_ctx.forget()


def test_in_place_guess_is_verified(gpu_ctx, hipmod, oracle):
    """A tile inside one long line takes 64 bytes of base letters at its beginning (end) for a stretch of a sequence line;
    quality lines made of base letters are false evidence: the pass must be refused, not wrong."""
    rng = np.random.default_rng(80)
    room = hipmod.INPLACE_STRIDE
    data = long_records(rng, 60, 20000, 50000, base_quals=0.3)
    gpu_ctx.forget()
    res = decode_same(gpu_ctx, hipmod, oracle, data, qual_room=room)
    assert res.path == 3
    decode_same(gpu_ctx, hipmod, oracle, data, qual_room=room)
    good = long_records(rng, 300, 3000, 9000)
    cases = {
        "long-quality-line": good[:1500000].rsplit(b"\n@", 1)[0] + b"\n@odd\nACGT\n+\nIIIIIIII\n" + good[:600000],
        "leading-text": b"# produced by a tool\n# and a second line\n# third\n# fourth\n# fifth\n" + good,
        "blank-line": good[:1000000].rsplit(b"\n@", 1)[0] + b"\n\n" + good[:500000],
        "one-wrapped": good[:900000].rsplit(b"\n@", 1)[0] + b"\n@w\nACGTAC\nGTAC\n+\nIIIIII\nIIII\n" + good[:700000],
        "short-lines-in-between": good[:900000].rsplit(b"\n@", 1)[0] + b"\n" + random_records(rng, 3000, 1, 9, hdr_hi=3) + good[:700000],
    }
    for name, d in cases.items():
        gpu_ctx.forget()
        res = decode_same(gpu_ctx, hipmod, oracle, d, qual_room=room)
        assert res.path != 6, name
        decode_same(gpu_ctx, hipmod, oracle, d, qual_room=room)
    gpu_ctx.forget()


def test_in_place_then_short_reads_again(gpu_ctx, hipmod, oracle):
    """The context starts with the layout that stood last; forget() goes back to segments (fewer bytes written on short reads)."""
    from fastqandfurious_amd import synth
    rng = np.random.default_rng(81)
    room = hipmod.INPLACE_STRIDE
    longs = long_records(rng, 40, 30000, 60000)
    plain = synth.single(0, 20000, seed=42)
    gpu_ctx.forget()
    assert decode_same(gpu_ctx, hipmod, oracle, longs, qual_room=room).path == 6
    table, res, qual, qoff = gpu_ctx.scan_host(plain, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS, qual_room=room)
    assert res.path == 6 and (qoff[:-1] == table[:, 4]).all()            # in place, remembered
    decode_same(gpu_ctx, hipmod, oracle, plain, qual_room=room)
    gpu_ctx.forget()
    table, res, qual, qoff = gpu_ctx.scan_host(plain, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS, qual_room=room)
    assert res.path == 6 and not (qoff[:-1] == table[:, 4]).all()        # segments
    gpu_ctx.forget()


def test_in_place_at_size_past_4g(gpu_ctx, hipmod, oracle):
    """6 GiB of long reads (500 ... 40000 bases, a 32 MiB block of distinct records repeated on the device), one pass, in
    place: EVERY row against the oracle's rows of the block (+ the repeat's offset), every record's offset = pos4, every
    quality byte of every repeat against the block's own bytes - 33."""
    import torch
    rng = np.random.default_rng(90)
    parts, tot, i = [], 0, 0
    qa = np.frombuffer(bytes(range(33, 127)), dtype=np.uint8)
    while tot < (32 << 20):
        L = int(np.exp(rng.uniform(np.log(500), np.log(40000))))
        r = (b"@read%d len=%d\n" % (i, L) + rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L).tobytes() + b"\n+\n" +
             rng.choice(qa, size=L).tobytes() + b"\n")
        parts.append(r); tot += len(r); i += 1
    block = np.frombuffer(b"".join(parts), dtype=np.uint8)
    rows, end, status, off = oracle.scan(block)
    assert len(rows) == i and end == 0
    reps = (6 << 30) // block.size + 1
    dev = torch.device("cuda:0")
    d = torch.from_numpy(block.copy()).to(dev).repeat(reps)
    n = i * reps
    table = torch.empty((n + 64, 6), dtype=torch.int64, device=dev)
    ntiles = (d.numel() + 16383) >> 14
    qual = torch.zeros(ntiles * hipmod.INPLACE_STRIDE, dtype=torch.int8, device=dev)
    qoff = torch.zeros(n + 65, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()          # (torch's stream; the scan runs on the context's)
    gpu_ctx.forget()
    rc, res = gpu_ctx.scan_device(d.data_ptr(), d.numel(), table.data_ptr(), n + 64, flags=hipmod.F_DECODE_QUAL | hipmod.F_SINGLE_PASS,
                                  d_qual=qual.data_ptr(), qual_cap=qual.numel(), d_qoff=qoff.data_ptr())
    assert rc == 0 and res.path == 6 and int(res.n_records) == n and res.end_state == 0
    assert d.numel() > (1 << 32)
    want = torch.from_numpy(rows).to(dev)
    shift = (torch.arange(reps, device=dev, dtype=torch.int64) * block.size).view(reps, 1, 1)
    assert bool((table[:n].view(reps, i, 6) == want.view(1, i, 6) + shift).all())
    assert bool((qoff[:n] == table[:n, 4]).all())
    assert int(qoff[n].item()) == int(res.n_qual_bytes) == int(table[n - 1, 5].item())
    mask = np.zeros(block.size, dtype=bool)
    for a, b in zip(rows[:, 4], rows[:, 5]):
        mask[a:b] = True
    at = torch.from_numpy(np.nonzero(mask)[0]).to(dev)
    src = (torch.from_numpy(block.copy()).to(dev)[at].to(torch.int16) - 33).to(torch.int8)
    for r in range(reps):
        assert bool((qual[r * block.size:(r + 1) * block.size][at] == src).all()), r
    gpu_ctx.forget()
