"""The gzip reader of the stream front end (csrc/ffq_stream.h), on the host alone (ffq_gunzip_fd).

What the reference hands readfastq_iter for a compressed file is `gzip.open(...)`
(/root/reference/src/fastqandfurious.py:282-334): Python's gzip module is the statement of what the
bytes of a gzip file are -- concatenated members, zero padding -- and of what is an error.  The
library's reader inflates one member at a time with zlib, and BGZF members (bgzip; they carry their
own length) side by side on several threads; both must give gzip.decompress's bytes, and must refuse
what it refuses.  No device is involved: these run in the CPU suite.
"""
import gzip
import os
import struct
import threading
import zlib

import numpy as np
import pytest

from fastqandfurious_amd import bgzf


@pytest.fixture(scope="module")
def hip(pkg):
    from fastqandfurious_amd import hip as H
    H.lib()
    return H


def _data(n, seed=5):
    rng = np.random.default_rng(seed)
    # compressible, FASTQ-like: four letters and a few newlines
    return rng.choice(np.frombuffer(b"ACGT\n@+I", dtype=np.uint8), size=n, p=[.22, .22, .22, .22, .03, .03, .03, .03]).tobytes()


def _gunzip(hip, tmp_path, blob, cap, chunk=1 << 20, threads=4):
    f = tmp_path / "x.gz"
    f.write_bytes(blob)
    fd = os.open(f, os.O_RDONLY)
    try:
        out, npar = hip.gunzip_fd(fd, cap, chunk, threads)
    finally:
        os.close(fd)
    return out.tobytes(), npar


@pytest.mark.parametrize("threads", (1, 2, 5))
@pytest.mark.parametrize("chunk", (1, 777, 65280, 1000003, 1 << 24))
def test_bgzf_equals_python_gzip(hip, tmp_path, threads, chunk):
    data = _data(700001 if chunk > 1 else 20011)
    blob = bgzf.compress(data, block_bytes=65280 if chunk > 1 else 997)
    assert gzip.decompress(blob) == data
    out, npar = _gunzip(hip, tmp_path, blob, len(data), chunk, threads)
    assert out == data
    if threads == 1:
        assert npar == 0                         # one thread: the pool is never started
    elif chunk >= 1000003:
        assert npar >= len(data) // 65280 - 3    # (all but the members a chunk boundary cut through)


def test_plain_and_mixed_members(hip, tmp_path):
    a, b, c = _data(150000, 1), _data(90000, 2), _data(200000, 3)
    for blob in (gzip.compress(a),                                   # one ordinary member: nothing to do side by side
                 gzip.compress(a) + gzip.compress(b),
                 bgzf.compress(a, eof_marker=False) + gzip.compress(b) + bgzf.compress(c),
                 gzip.compress(b) + bgzf.compress(c) + b"\0" * 100,
                 bgzf.compress(a) + b"\0" * 5 + bgzf.compress(b, block_bytes=1000),
                 bgzf.compress(b""), b""):
        want = gzip.decompress(blob) if blob else b""
        for chunk in (4099, 1 << 20):
            out, _ = _gunzip(hip, tmp_path, blob, len(want), chunk, 4)
            assert out == want


def test_bgzf_extra_subfields_and_odd_headers(hip, tmp_path):
    """The BC subfield among others; members whose flags are not BGZF's go one at a time."""
    data = _data(300000, 7)
    blocks = [data[i:i + 50000] for i in range(0, len(data), 50000)]
    blob = b"".join(bgzf.block(b, extra_before=struct.pack("<BBH", 65, 66, 3) + b"xyz") for b in blocks)
    out, npar = _gunzip(hip, tmp_path, blob, len(data))
    assert out == data and npar == len(blocks)
    blob = b"".join(bgzf.block(b, extra_after=struct.pack("<BBH", 90, 90, 0)) for b in blocks)
    out, npar = _gunzip(hip, tmp_path, blob, len(data))
    assert out == data and npar == len(blocks)
    # an extra field without BC, and a member with a file name: ordinary members
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    cd = raw.compress(blocks[0]) + raw.flush()
    noBC = struct.pack("<BBBBIBBH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 4) + struct.pack("<BBH", 1, 2, 0) + cd + \
        struct.pack("<II", zlib.crc32(blocks[0]), len(blocks[0]))
    assert gzip.decompress(noBC) == blocks[0]
    out, npar = _gunzip(hip, tmp_path, noBC + bgzf.compress(blocks[1]), 100000)
    assert out == blocks[0] + blocks[1]


def test_members_that_lie(hip, tmp_path):
    """A member whose header or trailer is wrong: the same answer as one member at a time gives."""
    data = _data(400000, 9)
    good = bgzf.compress(data, eof_marker=False)
    first = struct.unpack_from("<H", good, 16)[0] + 1
    # (1) BSIZE of the second member too small: the side-by-side inflate finds the deflate data cut short and
    #     hands over; the serial inflate does not read BSIZE at all
    bad = bytearray(good)
    struct.pack_into("<H", bad, first + 16, struct.unpack_from("<H", good, first + 16)[0] - 7)
    assert gzip.decompress(bytes(bad)) == data
    out, _ = _gunzip(hip, tmp_path, bytes(bad), len(data))
    assert out == data
    # (2) a flipped bit in the deflate data, (3) a wrong CRC, (4) a wrong length: errors, as for Python
    for at, what in ((first + 40, "data"), (first - 8, "crc"), (first - 4, "isize")):
        bad = bytearray(good)
        bad[at] ^= 0x10
        with pytest.raises((OSError, EOFError, zlib.error)):
            gzip.decompress(bytes(bad))
        with pytest.raises(hip.FFQError, match="gzip"):
            _gunzip(hip, tmp_path, bytes(bad), len(data))
    # (5) truncated inside a member, and right behind a header
    for cut in (len(good) - 11, first + 18, first + 5):
        with pytest.raises(hip.FFQError, match="gzip"):
            _gunzip(hip, tmp_path, good[:cut], len(data))
    # (6) garbage where a member should begin
    with pytest.raises(hip.FFQError, match="gzip"):
        _gunzip(hip, tmp_path, good + b"garbage, not a member", len(data) + 100)


def test_capacity(hip, tmp_path):
    data = _data(100000, 4)
    blob = bgzf.compress(data, block_bytes=10000)
    out, _ = _gunzip(hip, tmp_path, blob, len(data))
    assert out == data
    with pytest.raises(hip.FFQError, match="more than"):
        _gunzip(hip, tmp_path, blob, len(data) - 1)
    out, _ = _gunzip(hip, tmp_path, blob, len(data) + 12345)
    assert out == data


def test_from_a_pipe(hip):
    """A descriptor that cannot seek: the reader takes what arrives (in dribbles here)."""
    data = _data(500000, 6)
    blob = bgzf.compress(data, block_bytes=30000)
    r, w = os.pipe()

    def feed():
        for i in range(0, len(blob), 7001):
            os.write(w, blob[i:i + 7001])
        os.close(w)
    t = threading.Thread(target=feed)
    t.start()
    try:
        out, npar = hip.gunzip_fd(r, len(data), 1 << 20, 3)
    finally:
        t.join()
        os.close(r)
    assert out.tobytes() == data


def test_random_layouts(hip, tmp_path):
    rng = np.random.default_rng(77)
    for it in range(25):
        parts, want = [], b""
        for _ in range(int(rng.integers(1, 6))):
            d = _data(int(rng.integers(0, 120000)), int(rng.integers(1 << 30)))
            kind = int(rng.integers(3))
            if kind == 0:
                parts.append(gzip.compress(d, int(rng.integers(1, 9))))
            else:
                parts.append(bgzf.compress(d, block_bytes=int(rng.integers(100, 65281)), level=int(rng.integers(0, 9)),
                                           eof_marker=bool(kind == 1)))
            want += d
            if rng.integers(4) == 0:
                parts.append(b"\0" * int(rng.integers(1, 50)))
        blob = b"".join(parts)
        assert gzip.decompress(blob) == want
        out, _ = _gunzip(hip, tmp_path, blob, len(want), int(rng.integers(1, 300000)), int(rng.integers(1, 7)))
        assert out == want, it


# ---- one plain member inflated by several threads (csrc/ffq_pgz.h) -------------------------------------------------------
def _fastq(n, seed=3):
    rng = np.random.default_rng(seed)
    out = bytearray()
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, 100))
    q = (np.clip(rng.normal(36, 4, size=(n, 100)).astype(np.int64), 2, 41) + 33).astype(np.uint8)
    for i in range(n):
        out += b"@SRR0000001.%d %d/1\n" % (i + 1, i + 1) + seq[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n"
    return bytes(out)


def _gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, name=b""):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    hdr = struct.pack("<BBBBIBB", 0x1F, 0x8B, 8, 8 if name else 0, 0, 0, 3) + (name + b"\0" if name else b"")
    return hdr + c.compress(data) + c.flush() + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)


@pytest.fixture
def pgz_env(monkeypatch):
    def set_(chunk=65536, minimum=1, **kw):
        monkeypatch.setenv("FFQ_PGZ_CHUNK", str(chunk))
        monkeypatch.setenv("FFQ_PGZ_MIN", str(minimum))
        for k, v in kw.items():
            monkeypatch.setenv("FFQ_PGZ_" + k.upper(), str(v))
    return set_


@pytest.mark.parametrize("level", (1, 6, 9))
@pytest.mark.parametrize("pchunk", (16384, 100000, 1 << 20))
def test_plain_member_by_several_threads(hip, tmp_path, pgz_env, level, pchunk):
    data = _fastq(12000)
    blob = _gz(data, level, name=b"reads.fq")
    assert gzip.decompress(blob) == data
    pgz_env(chunk=pchunk)
    for threads, chunk in ((2, 1 << 24), (4, 65280), (5, 777777)):
        s0 = hip.gunzip_stats()
        out, _ = _gunzip(hip, tmp_path, blob, len(data), chunk, threads)
        s1 = hip.gunzip_stats()
        assert out == data
        assert s1["giveups"] == s0["giveups"] and s1["members"] == s0["members"] + 1
        if pchunk < (1 << 20):
            assert s1["chunks"] - s0["chunks"] > (s1["batches"] - s0["batches"])      # chunks entered mid-stream were taken
    # one thread: zlib alone
    s0 = hip.gunzip_stats()
    out, _ = _gunzip(hip, tmp_path, blob, len(data), 1 << 20, 1)
    assert out == data and hip.gunzip_stats() == s0


def test_plain_member_block_kinds(hip, tmp_path, pgz_env):
    """Stored and fixed-Huffman blocks are never entered mid-stream (only chunk 0 of a batch meets them); data that
    mixes all three kinds; a member that inflates to a thousand times its size."""
    rng = np.random.default_rng(11)
    fq = _fastq(4000, 5)
    noise = rng.integers(0, 256, size=300000, dtype=np.uint8).tobytes()
    mixed = fq[:200000] + noise + fq[200000:] + noise[:70000] + b"A" * 500000 + fq[:100000]
    pgz_env(chunk=32768)
    for data, level, strategy in ((fq, 0, zlib.Z_DEFAULT_STRATEGY), (fq, 6, zlib.Z_FIXED), (mixed, 6, zlib.Z_DEFAULT_STRATEGY),
                                  (mixed, 1, zlib.Z_DEFAULT_STRATEGY), (mixed, 9, zlib.Z_HUFFMAN_ONLY), (fq, 6, zlib.Z_RLE),
                                  (b"\0" * 40_000_000, 9, zlib.Z_DEFAULT_STRATEGY), (b"", 6, zlib.Z_DEFAULT_STRATEGY), (b"x", 6, zlib.Z_DEFAULT_STRATEGY)):
        blob = _gz(data, level, strategy)
        assert gzip.decompress(blob) == data
        for threads in (2, 4):
            out, _ = _gunzip(hip, tmp_path, blob, len(data), 1 << 22, threads)
            assert out == data


def test_plain_members_in_a_row(hip, tmp_path, pgz_env):
    a, b = _fastq(5000, 1), _fastq(3000, 2)
    pgz_env(chunk=50000)
    blob = _gz(a, 6) + _gz(b, 1) + b"\0" * 77 + bgzf.compress(a[:300000]) + _gz(b, 9) + b"\0" * 3
    want = a + b + a[:300000] + b
    assert gzip.decompress(blob) == want
    for chunk in (4099, 1 << 20, 1 << 26):
        out, _ = _gunzip(hip, tmp_path, blob, len(want), chunk, 4)
        assert out == want


def test_plain_member_handed_over_to_zlib(hip, tmp_path, pgz_env):
    """The engine gives up (here: told to, after k batches; or a chunk that may not grow) and zlib goes on from the
    block boundary it stopped at -- a bit position inside a byte, with the 32 KiB in front as its dictionary."""
    data = _fastq(15000, 8)
    blob = _gz(data, 6)
    for k in (1, 2, 5):
        pgz_env(chunk=40000, giveup_after=k)
        s0 = hip.gunzip_stats()
        out, _ = _gunzip(hip, tmp_path, blob, len(data), 1 << 20, 3)
        assert out == data
        assert hip.gunzip_stats()["giveups"] == s0["giveups"] + 1
    pgz_env(chunk=40000, max_out=65536 + 300, giveup_after=1000000)
    out, _ = _gunzip(hip, tmp_path, blob, len(data), 1 << 20, 3)
    assert out == data
    zeros = b"\0" * 30_000_000                         # blocks of megabytes: nothing fits a chunk that may not grow
    out, _ = _gunzip(hip, tmp_path, _gz(zeros, 9), len(zeros), 1 << 22, 3)
    assert out == zeros


def test_plain_member_errors(hip, tmp_path, pgz_env):
    data = _fastq(9000, 4)
    good = _gz(data, 6)
    pgz_env(chunk=30000)
    n = len(good)
    for at, what in ((n // 2, "data"), (n // 7, "data"), (n - 8, "crc"), (n - 4, "isize"), (n - 3000, "data")):
        bad = bytearray(good)
        bad[at] ^= 0x04
        with pytest.raises((OSError, EOFError, zlib.error)):
            gzip.decompress(bytes(bad))
        with pytest.raises(hip.FFQError, match="gzip"):
            _gunzip(hip, tmp_path, bytes(bad), len(data) + 100, 1 << 20, 4)
    for cut in (n - 1, n - 5, n - 9, n // 2, n // 3, 100, 12):
        with pytest.raises(hip.FFQError, match="ended before the end-of-stream marker"):
            _gunzip(hip, tmp_path, good[:cut], len(data), 1 << 20, 4)
    with pytest.raises(hip.FFQError, match="gzip"):
        _gunzip(hip, tmp_path, good + b"garbage, not a member", len(data) + 100, 1 << 20, 4)
    # what comes out in front of the error is what zlib gives: the good bytes of a file cut short
    out, _ = _gunzip(hip, tmp_path, good + b"\0" * 9, len(data), 1 << 20, 4)
    assert out == data


def test_engine_under_sanitizers_with_corrupt_files(tmp_path):
    """The engine reads files it was not promised anything about: its harness (tools/pgz_main.cpp = ffq_pgz.h alone)
    built with ASan + UBSan and run over flipped bits, cut and overwritten stretches -- any exit but 'done', 'failed'
    (trailer mismatch) or 'gave up' (zlib's turn) is a memory error or undefined behaviour."""
    import random
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = tmp_path / "pgz_asan"
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-pthread",
                        "-o", str(exe), os.path.join(root, "tools", "pgz_main.cpp"), "-lz"], capture_output=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime: " + r.stderr.decode()[-200:])
    blob = _gz(_fastq(9000, 13), 6)
    random.seed(7)
    env = dict(os.environ, FFQ_PGZ_MIN="1", ASAN_OPTIONS="detect_leaks=0")
    f = tmp_path / "fz.gz"
    seen = set()
    for it in range(40):
        b = bytearray(blob)
        kind = it % 4
        if kind == 0:
            for _ in range(random.randint(1, 4)):
                b[random.randrange(10, len(b))] ^= 1 << random.randrange(8)
        elif kind == 1:
            b = b[:random.randrange(10, len(b))]
        elif kind == 2:
            at, n = random.randrange(10, len(b)), random.randrange(1, 3000)
            b[at:at + n] = bytes(random.randrange(256) for _ in range(n))
        else:
            at = random.randrange(10, len(b))
            del b[at:at + random.randrange(1, 500)]
        f.write_bytes(bytes(b))
        env["FFQ_PGZ_CHUNK"] = random.choice(["4096", "16384", "100000"])
        r = subprocess.run([str(exe), str(f), str(random.randint(2, 5))], env=env, capture_output=True, timeout=300)
        assert r.returncode in (0, 1, 3), r.stderr.decode()[-2000:]
        seen.add(r.returncode)
    f.write_bytes(blob)
    r = subprocess.run([str(exe), str(f), "3", "check"], env=env, capture_output=True, timeout=300)
    assert r.returncode == 0 and b"EQUAL" in r.stdout, r.stderr.decode()[-2000:]
    assert seen - {0}                 # (the corruptions were noticed)
    # a batch the end of the file cuts short, its last chunk a short tail without a block header in it (zero padding
    # behind the member): the candidate scan of that chunk must stop where the input does
    env["FFQ_PGZ_CHUNK"] = "16384"
    for tail in (1, 100, 1500):
        f.write_bytes(blob + b"\0" * ((tail - len(blob)) % 16384))
        r = subprocess.run([str(exe), str(f), "32", "check"], env=env, capture_output=True, timeout=300)
        assert r.returncode == 0 and b"EQUAL" in r.stdout, r.stderr.decode()[-2000:]
    # the file ends inside the header of a dynamic-Huffman block (the first one starts right behind the gzip header;
    # further ones at the boundaries a chunked run reports): every cut over the first header, and cuts a few bytes
    # behind random later positions
    for cut in list(range(11, 140)) + [random.randrange(200, len(blob)) for _ in range(40)]:
        f.write_bytes(blob[:cut])
        r = subprocess.run([str(exe), str(f), "2"], env=env, capture_output=True, timeout=300)
        assert r.returncode in (1, 3), r.stderr.decode()[-2000:]


def test_plain_member_default_settings(hip, tmp_path, monkeypatch):
    """No switches: a member of more than 4 MiB goes to the engine with its 1 MiB chunks and the small first batch;
    one of less than that is zlib's."""
    for k in ("FFQ_PGZ_CHUNK", "FFQ_PGZ_MIN", "FFQ_PGZ_CPT", "FFQ_PGZ_MAX_OUT", "FFQ_PGZ_GIVEUP_AFTER", "FFQ_GZ_THREADS"):
        monkeypatch.delenv(k, raising=False)
    big = _fastq(12000, 21) * 8                      # ~22 MB, ~9 MB compressed
    blob = _gz(big, 6)
    assert len(blob) > (5 << 20)
    s0 = hip.gunzip_stats()
    out, _ = _gunzip(hip, tmp_path, blob, len(big), 16 << 20, 4)
    s1 = hip.gunzip_stats()
    assert out == big
    assert s1["members"] == s0["members"] + 1 and s1["giveups"] == s0["giveups"] and s1["chunks"] >= s0["chunks"] + 6
    small = _gz(big[:3_000_000], 6)
    out, _ = _gunzip(hip, tmp_path, small, 3_000_000, 16 << 20, 4)
    assert out == big[:3_000_000] and hip.gunzip_stats() == s1
