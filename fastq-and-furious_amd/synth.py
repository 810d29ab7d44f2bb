"""Synthetic FASTQ inputs, numpy side.

Counter-based: byte values depend only on (seed, record index, position), so
this generator and the device kernels (ffq_synth_single / ffq_synth_wrapped in
csrc/ffq_kernels.h) produce identical bytes.

S-single : "@SYN.%010d/1\\n" + 150 bases + "\\n+\\n" + 150 quals + "\\n" = 322 B
S-wrapped: length 50 + h % 251, 80-column wrap of sequence and quality,
           '+' line repeats the header text when (h >> 32) % 4 == 0.
"""
import numpy as np

RECORD_BYTES = 322
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _headers(idx):
    """(n, 18) uint8: '@SYN.%010d/1\\n' for every index."""
    n = idx.shape[0]
    out = np.empty((n, 18), dtype=np.uint8)
    out[:, 0:5] = np.frombuffer(b"@SYN.", dtype=np.uint8)
    x = idx.astype(np.int64).copy()
    for k in range(9, -1, -1):
        out[:, 5 + k] = 48 + (x % 10)
        x //= 10
    out[:, 15:18] = np.frombuffer(b"/1\n", dtype=np.uint8)
    return out


def single(first, count, seed=42):
    """`count` S-single records starting at record `first`, as uint8[count*322]."""
    out = np.empty((count, RECORD_BYTES), dtype=np.uint8)
    step = 1 << 16
    for a in range(0, count, step):
        b = min(a + step, count)
        idx = np.arange(first + a, first + b, dtype=np.uint64)
        out[a:b, 0:18] = _headers(idx)
        key = (idx[:, None] << np.uint64(9)) | np.arange(150, dtype=np.uint64)[None, :]
        h = splitmix64(np.uint64(seed) ^ key)
        out[a:b, 18:168] = _ACGT[(h & np.uint64(3)).astype(np.int64)]
        out[a:b, 171:321] = (33 + (h >> np.uint64(8)) % np.uint64(41)).astype(np.uint8)
    out[:, 168] = 10
    out[:, 169] = 43
    out[:, 170] = 10
    out[:, 321] = 10
    return out.reshape(-1)


def single_records_for(target_bytes):
    return int(target_bytes) // RECORD_BYTES


def wrapped_sizes(first, count, seed=43):
    idx = np.arange(first, first + count, dtype=np.uint64)
    h = splitmix64(np.uint64(seed) ^ idx)
    length = 50 + (h % np.uint64(251)).astype(np.int64)
    nl = (length + 79) // 80
    rep = np.where((h >> np.uint64(32)) % np.uint64(4) == 0, 16, 0).astype(np.int64)
    return 18 + length + nl + 1 + rep + 1 + length + nl


def wrapped_fields(first, count, seed=43):
    """(read length, bytes the '+' line repeats of the header: 0 or 16) of S-wrapped records."""
    idx = np.arange(first, first + count, dtype=np.uint64)
    h = splitmix64(np.uint64(seed) ^ idx)
    length = 50 + (h % np.uint64(251)).astype(np.int64)
    rep = np.where((h >> np.uint64(32)) % np.uint64(4) == 0, 16, 0).astype(np.int64)
    return length, rep


def wrapped(first, count, seed=43):
    """`count` S-wrapped records starting at record `first` (uint8 array)."""
    sizes = wrapped_sizes(first, count, seed)
    start = np.zeros(count + 1, dtype=np.int64)
    np.cumsum(sizes, out=start[1:])
    out = np.empty(int(start[-1]), dtype=np.uint8)
    idx = np.arange(first, first + count, dtype=np.uint64)
    hh = splitmix64(np.uint64(seed) ^ idx)
    heads = _headers(idx)
    for r in range(count):
        i = idx[r]
        length = 50 + int(hh[r] % np.uint64(251))
        rep = int((hh[r] >> np.uint64(32)) % np.uint64(4)) == 0
        key = (i << np.uint64(9)) | np.arange(length, dtype=np.uint64)
        h = splitmix64(np.uint64(seed) ^ key)
        bases = _ACGT[(h & np.uint64(3)).astype(np.int64)]
        quals = (33 + (h >> np.uint64(8)) % np.uint64(41)).astype(np.uint8)
        parts = [heads[r]]
        for a in range(0, length, 80):
            parts.append(bases[a:a + 80])
            parts.append(np.array([10], dtype=np.uint8))
        parts.append(np.array([43], dtype=np.uint8))
        if rep:
            parts.append(heads[r][1:17])
        parts.append(np.array([10], dtype=np.uint8))
        for a in range(0, length, 80):
            parts.append(quals[a:a + 80])
            parts.append(np.array([10], dtype=np.uint8))
        rec = np.concatenate(parts)
        assert rec.size == sizes[r]
        out[start[r]:start[r + 1]] = rec
    return out, start
