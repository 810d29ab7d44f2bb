"""ctypes front-end of oracle/libffq_oracle.so (TEST INFRASTRUCTURE ONLY).

Restates, on the CPU, the reference path
  /root/reference/src/_fastqandfurious.c:25-153   (entrypos, C)
  /root/reference/src/fastqandfurious.py:39-100   (entrypos, Python)
  /root/reference/src/fastqandfurious.py:198-279  (readfastq_iter record chain)
  /root/reference/src/_fastqandfurious.c:161-217  (arrayadd_b, arrayadd_q)
Parity: pinned against the reference extension (oracle/_ref) and the golden
vectors in tests/golden/ by tests/test_oracle.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

END_OK, END_REFILL, END_ERR_FINAL_QUAL, END_ERR_INCOMPLETE, END_ERR_INVALID, END_TABLE_FULL = range(6)

VARIANT_C = 0
VARIANT_PY = 1

_lib = None


def build(asan=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    target = ["asan"] if asan else []
    subprocess.run(["make", "-C", _HERE] + target, check=True,
                   stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libffq_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        u8p = ctypes.c_void_p
        i64 = ctypes.c_int64
        L.ffq_oracle_entrypos_c.argtypes = [u8p, i64, i64, ctypes.c_void_p]
        L.ffq_oracle_entrypos_c.restype = ctypes.c_int
        L.ffq_oracle_entrypos_py.argtypes = [u8p, i64, i64, ctypes.c_void_p]
        L.ffq_oracle_entrypos_py.restype = ctypes.c_int
        L.ffq_oracle_scan.argtypes = [u8p, i64, ctypes.c_int, i64, ctypes.c_int,
                                      ctypes.c_int, i64, ctypes.c_void_p, i64,
                                      ctypes.c_void_p]
        L.ffq_oracle_scan.restype = None
        L.ffq_oracle_entrypos_fasta.argtypes = [u8p, i64, i64, ctypes.c_void_p]
        L.ffq_oracle_entrypos_fasta.restype = ctypes.c_int
        L.ffq_oracle_scan_fasta.argtypes = [u8p, i64, i64, i64, ctypes.c_void_p, i64, ctypes.c_void_p, ctypes.c_void_p]
        L.ffq_oracle_scan_fasta.restype = None
        L.ffq_oracle_arrayadd_b.argtypes = [ctypes.c_void_p, i64, ctypes.c_int]
        L.ffq_oracle_arrayadd_b.restype = None
        L.ffq_oracle_arrayadd_q.argtypes = [ctypes.c_void_p, i64, i64]
        L.ffq_oracle_arrayadd_q.restype = None
        L.ffq_oracle_decode_quals.argtypes = [u8p, ctypes.c_void_p, i64, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p]
        L.ffq_oracle_decode_quals.restype = None
        L.ffq_oracle_gather_column.argtypes = [u8p, ctypes.c_void_p, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ffq_oracle_gather_column.restype = None
        L.ffq_oracle_select_seqlen.argtypes = [ctypes.c_void_p, i64, i64, i64, ctypes.c_void_p]
        L.ffq_oracle_select_seqlen.restype = i64
        _lib = L
    return _lib


def _as_u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return a


def entrypos(buf, offset, variant=VARIANT_C, pos=None):
    """(status, pos[6]) of one scanner call on `buf` (bytes-like)."""
    a = _as_u8(buf)
    if pos is None:
        pos = np.full(6, -1, dtype=np.int64)
    f = lib().ffq_oracle_entrypos_py if variant else lib().ffq_oracle_entrypos_c
    st = f(a.ctypes.data, a.size, int(offset), pos.ctypes.data)
    return st, pos


def scan(data, sentinel=True, offset=0, eof=True, variant=VARIANT_C, add=None, cap=None):
    """Record chain over one buffer.

    Returns (table int64[n,6], end_state, last_status, end_offset).  With
    sentinel=True and add=None the rows are absolute file offsets (add=-1), as
    readfastq_iter(..., entryfunc=entryfunc_abspos) yields them.
    """
    a = _as_u8(data)
    if add is None:
        add = -1 if sentinel else 0
    if cap is None:
        cap = a.size // 4 + 2     # a record needs >= 4 bytes after its '@'
    table = np.empty((cap, 6), dtype=np.int64)
    out = np.zeros(4, dtype=np.int64)
    lib().ffq_oracle_scan(a.ctypes.data, a.size, int(bool(sentinel)), int(offset),
                          int(bool(eof)), int(variant), int(add),
                          table.ctypes.data, cap, out.ctypes.data)
    n = int(out[0])
    return table[:n].copy(), int(out[1]), int(out[2]), int(out[3])


def arrayadd_b(arr, value):
    a = arr if isinstance(arr, np.ndarray) else np.frombuffer(arr, dtype=np.int8)
    assert a.itemsize == 1
    lib().ffq_oracle_arrayadd_b(a.ctypes.data, a.size, int(value))
    return arr


def arrayadd_q(arr, value):
    a = arr if isinstance(arr, np.ndarray) else np.frombuffer(arr, dtype=np.int64)
    assert a.itemsize == 8
    v = int(value)
    v = (v + 2**63) % 2**64 - 2**63
    lib().ffq_oracle_arrayadd_q(a.ctypes.data, a.size, v)
    return arr


def decode_quals(base, table, value=-33):
    """Packed int8 qualities + CSR offsets for the records of `table` (indices
    into `base`)."""
    b = _as_u8(base)
    t = np.ascontiguousarray(table, dtype=np.int64)
    n = t.shape[0]
    total = int((t[:, 5] - t[:, 4]).sum()) if n else 0
    out = np.empty(total, dtype=np.int8)
    qoff = np.empty(n + 1, dtype=np.int64)
    lib().ffq_oracle_decode_quals(b.ctypes.data, t.ctypes.data, n, int(value),
                                  out.ctypes.data, qoff.ctypes.data)
    return out, qoff


COLUMNS = {"header": (0, 1, 1), "sequence": (2, 0, 3), "quality": (4, 0, 5)}    # (begin column, shift, end column)


def gather_column(base, table, which, value=0):
    """Packed buf[pos[ca] + shift:pos[cb]] (+ value, int8) of every row + CSR offsets: the
    component an entryfunc that builds only `which` would return (positions index `base`)."""
    b = _as_u8(base)
    t = np.ascontiguousarray(table, dtype=np.int64).reshape(-1, 6)
    ca, sh, cb = COLUMNS[which]
    n = t.shape[0]
    total = int(np.maximum(t[:, cb] - t[:, ca] - sh, 0).sum()) if n else 0
    out = np.empty(total, dtype=np.int8)
    coff = np.empty(n + 1, dtype=np.int64)
    lib().ffq_oracle_gather_column(b.ctypes.data, t.ctypes.data, n, ca, sh, cb, int(value), out.ctypes.data,
                                   coff.ctypes.data)
    return out, coff


def select_seqlen(table, lo, hi):
    """Rows with lo <= pos3 - pos2 <= hi, in order (doc/user-guide.rst:153-180 on a table)."""
    t = np.ascontiguousarray(table, dtype=np.int64).reshape(-1, 6)
    out = np.empty_like(t)
    k = lib().ffq_oracle_select_seqlen(t.ctypes.data, t.shape[0], int(lo), int(hi), out.ctypes.data)
    return out[:k]


def entrypos_fasta(buf, offset):
    """(status, [6 positions]) of one FASTA scanner call (reference fastqandfurious.py:103-143)."""
    b = _as_u8(buf)
    pos = np.full(6, -1, dtype=np.int64)
    st = lib().ffq_oracle_entrypos_fasta(b.ctypes.data if b.size else None, b.size, int(offset), pos.ctypes.data)
    return int(st), [int(x) for x in pos]


def scan_fasta(data, offset=0, add=0, cap=None):
    """(table int64[n,6] of the COMPLETE entries, last status, last posbuffer, offset of the last call)."""
    b = _as_u8(data)
    cap = int(cap) if cap is not None else b.size // 4 + 8
    table = np.empty((cap, 6), dtype=np.int64)
    last = np.full(6, -1, dtype=np.int64)
    out = np.zeros(4, dtype=np.int64)
    lib().ffq_oracle_scan_fasta(b.ctypes.data if b.size else None, b.size, int(offset), int(add),
                                table.ctypes.data, cap, last.ctypes.data, out.ctypes.data)
    return table[:int(out[0])], int(out[1]), [int(x) for x in last], int(out[2])
