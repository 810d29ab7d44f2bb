"""A BGZF writer (what bgzip produces; SAM specification section 4.1) in a few lines of zlib: for tests, the bench

A BGZF file is a series of gzip members of at most 64 KiB of data each, every one with a "BC" extra
subfield that holds the member's total length - 1, and an empty member as the end-of-file marker.
"""
import struct
import zlib


def block(chunk, level=6, extra_before=b"", extra_after=b""):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    cd = c.compress(chunk) + c.flush()
    xlen = 6 + len(extra_before) + len(extra_after)
    total = 12 + xlen + len(cd) + 8
    assert total <= 65536
    head = struct.pack("<BBBBIBBH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, xlen)
    bc = struct.pack("<BBHH", 66, 67, 2, total - 1)
    return head + extra_before + bc + extra_after + cd + struct.pack("<II", zlib.crc32(chunk), len(chunk))


def compress(data, block_bytes=65280, level=6, eof_marker=True):
    out = bytearray()
    for i in range(0, len(data), block_bytes):
        out += block(data[i:i + block_bytes], level)
    if eof_marker:
        out += block(b"", level)
    return bytes(out)
