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
