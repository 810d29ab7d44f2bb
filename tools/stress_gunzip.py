"""Differential stress of the several-thread gzip inflate (csrc/ffq_pgz.h) against Python's gzip module, host only:
random FASTQ-like / noisy / repetitive data, every zlib level and strategy, random engine chunk sizes, thread counts and
reader chunk sizes, members in a row.  python tools/stress_gunzip.py [iterations] [seed]"""
import gzip, os, struct, sys, tempfile, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fastqandfurious_amd import hip


def piece(rng):
    kind = rng.integers(0, 5)
    n = int(rng.integers(1, 400000))
    if kind == 0:
        L = int(rng.integers(20, 300))
        m = max(n // (2 * L + 30), 1)
        seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=(m, L), p=[.24, .24, .24, .24, .04])
        q = rng.integers(33, 75, size=(m, L), dtype=np.uint8)
        return b"".join(b"@r%d len=%d\n" % (i, L) + seq[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n" for i in range(m))
    if kind == 1:
        return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    if kind == 2:
        return bytes([int(rng.integers(0, 256))]) * n
    if kind == 3:
        return rng.choice(np.frombuffer(b"ACGT\n", dtype=np.uint8), size=n).tobytes()
    unit = rng.integers(0, 256, size=int(rng.integers(2, 3000)), dtype=np.uint8).tobytes()
    return (unit * (n // len(unit) + 1))[:n]


def member(rng, data):
    level = int(rng.integers(0, 10))
    strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))] if rng.random() < .3 else zlib.Z_DEFAULT_STRATEGY
    c = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.integers(1, 10)), strategy)
    body = bytearray()
    at = 0
    while at < len(data):                      # (flushes put empty stored blocks and byte alignment into the stream)
        n = int(rng.integers(1, 300000))
        body += c.compress(data[at:at + n])
        if rng.random() < .2:
            body += c.flush([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_PARTIAL_FLUSH][int(rng.integers(0, 3))])
        at += n
    body += c.flush()
    return struct.pack("<BBBBIBB", 0x1F, 0x8B, 8, 0, 0, 0, 3) + bytes(body) + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "x.gz")
        for it in range(iters):
            blob, want = b"", b""
            for _ in range(int(rng.integers(1, 4))):
                data = b"".join(piece(rng) for _ in range(int(rng.integers(1, 8))))
                blob += member(rng, data) + b"\0" * int(rng.integers(0, 3))
                want += data
            assert gzip.decompress(blob) == want
            os.environ["FFQ_PGZ_CHUNK"] = str(int(rng.choice([4096, 20000, 65536, 300000, 1 << 20])))
            os.environ["FFQ_PGZ_MIN"] = "1"
            os.environ["FFQ_PGZ_CPT"] = str(int(rng.integers(1, 4)))
            if rng.random() < .2:
                os.environ["FFQ_PGZ_GIVEUP_AFTER"] = str(int(rng.integers(1, 6)))
            else:
                os.environ.pop("FFQ_PGZ_GIVEUP_AFTER", None)
            open(f, "wb").write(blob)
            fd = os.open(f, os.O_RDONLY)
            try:
                out, _ = hip.gunzip_fd(fd, len(want) + 5, int(rng.choice([1, 4099, 65280, 1 << 20, 1 << 26])) if len(want) < 200000 else int(rng.choice([65280, 1 << 20, 1 << 26])), int(rng.integers(2, 7)))
            finally:
                os.close(fd)
            if out.tobytes() != want:
                bad += 1
                print("MISMATCH at iteration", it, {k: v for k, v in os.environ.items() if k.startswith("FFQ_PGZ")})
    print("%d iterations, %d mismatches; engine: %s" % (iters, bad, hip.gunzip_stats()))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
