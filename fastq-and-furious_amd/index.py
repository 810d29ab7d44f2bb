"""Offset-index files: store the table of record positions once, replay it without parsing.

The reference keeps such an index as the concatenation of `pos.tofile(fh_index)` over
`readfastq_iter(fh, bufsize, entryfunc=entryfunc_abspos, ...)`
(/root/reference/src/demo/benchmark.py:268-287): raw native int64 x 6 per record, ABSOLUTE
stream offsets (pos0 = the '@').  It replays it record by record with `array.fromfile`,
`fh.seek`, `fh.read` and `arrayadd_q(posarray, -offset)`
(/root/reference/src/demo/benchmark.py:47-83, doc/user-guide.rst:182-204).

The GPU scan already produces exactly this table for a whole buffer fill, so the index is
written table by table (`build_index`) and replayed in chunks of rows (`iter_indexed`).
Filtering or trimming reads is editing rows (`select_rows`), as the user guide suggests.
"""
from array import array

import numpy as np

from . import entries as _entries
from . import fastqandfurious as _F

ROW_BYTES = 48          # 6 x int64 per record


def _fileno(fh):
    """(fd, start) of a plain binary file object -- its descriptor and the position the object is
    at -- else None.  The descriptor itself is not touched: the native stream reads it with pread
    from `start` (start None: it cannot seek and is read in order)."""
    import io
    # only plain files: a GzipFile also has a fileno() -- that of the COMPRESSED file
    raw = fh.raw if isinstance(fh, io.BufferedReader) else fh
    if not isinstance(raw, io.FileIO):
        return None
    try:
        fd = fh.fileno()
    except (AttributeError, OSError, ValueError):
        return None
    try:
        start = fh.tell() if fh.seekable() else None
    except (OSError, ValueError, AttributeError):
        start = None
    if start is None and fh is not raw:
        return None              # a buffered pipe: bytes may already sit in the object's buffer, past the descriptor
    return fd, start


def _leave_at(fh, st):
    """Leave a shared file object where the stream stopped reading (the reference's loop leaves
    `fh` behind the last chunk it read, /root/reference/src/fastqandfurious.py:274-277)."""
    try:
        if fh.seekable():
            fh.seek(st.tell())
    except (OSError, ValueError, AttributeError):
        pass


def iter_tables(fh, fbufsize, scan_buffer):
    """One (buf, rows, globaloffset) per buffer fill: `rows` is an array('q') of 6*n
    buffer-relative positions, `rows[i] + globaloffset` the absolute ones.  Same refill,
    sentinel, globaloffset and error behaviour as readfastq_iter
    (/root/reference/src/fastqandfurious.py:241-279)."""
    globaloffset = -1
    offset = 0
    buf, eof = _F.read(fh, fbufsize)
    buf = b'\n' + buf
    while True:
        rows, end_state, end_offset = scan_buffer(buf, offset, eof)
        if len(rows):
            yield buf, rows, globaloffset
        offset = end_offset
        if end_state == _F._END_OK:
            return
        if end_state != _F._END_REFILL:
            _F._raise_for_end(end_state, globaloffset + offset)
        globaloffset += offset
        tmp_buf, eof = _F.read(fh, fbufsize)
        buf = buf[offset:] + tmp_buf
        del tmp_buf
        offset = 0


def build_index(fh, fh_index, fbufsize=1 << 24, entrypos=None):
    """Write the offset index of the FASTQ stream `fh` to `fh_index`; returns the number of
    records.  `entrypos` is a scanner as readfastq_iter takes it; one that offers
    `scan_buffer` (the GPU scanner of this package, the default) writes a whole table per
    buffer fill, any other goes record by record exactly like the reference
    (benchmark.py:277-283).  The bytes written are the same either way."""
    if entrypos is None:
        from . import _fastqandfurious
        entrypos = _fastqandfurious.entrypos
        f = _fileno(fh)
        if f is not None:
            # a real file and the GPU scanner: the native stream front end (ffq_stream_*) reads
            # ahead into pinned memory and hands back whole tables; no per-fill Python copies
            from . import hip
            n = 0
            st = hip.FileStream(hip.default_context(), f[0], fbufsize, start=f[1])
            try:
                for rows, _fill, _off, end_state, err in st:
                    if rows.shape[0]:
                        fh_index.write(memoryview(rows).cast("B"))
                        n += rows.shape[0]
                    if end_state not in (_F._END_OK, _F._END_REFILL):
                        _F._raise_for_end(end_state, err)
            finally:
                _leave_at(fh, st)
                st.close()
            return n
    scan_buffer = getattr(entrypos, 'scan_buffer', None)
    n = 0
    if scan_buffer is None:
        for pos in _F.readfastq_iter(fh, fbufsize, _F.entryfunc_abspos, entrypos):
            pos.tofile(fh_index)
            n += 1
        return n
    for _buf, rows, globaloffset in iter_tables(fh, fbufsize, scan_buffer):
        t = np.frombuffer(rows, dtype=np.int64) + np.int64(globaloffset)     # entryfunc_abspos, all rows
        fh_index.write(t.tobytes())
        n += t.size // 6
    return n


def read_index(fh_index, count=-1):
    """The index as int64[n][6] (count = -1: all of it)."""
    raw = fh_index.read() if count < 0 else fh_index.read(count * ROW_BYTES)
    if len(raw) % ROW_BYTES:
        raise ValueError('The index must hold 6 int64 per entry.')
    return np.frombuffer(raw, dtype=np.int64).reshape(-1, 6)


def iter_indexed(fh, fh_index, chunk_records=1 << 16):
    """Replay: yields (header, sequence, quality) per index row, read from `fh` by position
    without parsing.  As in the reference's replay loop (benchmark.py:62-71) the slices are
    `buf[pos0:pos1]`, `buf[pos2:pos3]`, `buf[pos4:pos5]` -- pos0 is the '@', so the header
    slice starts with it.  Rows may have been filtered or edited; they must be in stream
    order within a chunk only if `fh` cannot seek backwards."""
    while True:
        t = read_index(fh_index, chunk_records)
        if t.shape[0] == 0:
            return
        lo = int(t[:, 0].min())
        hi = int(t[:, 5].max())
        fh.seek(lo)
        buf = fh.read(hi - lo + 1)
        if _entries.native() is not None:
            # arrayadd_q(posarray, -offset) and the three slices of every row, natively
            # (csrc/ffq_entries.c; a thousand rows per call so that consumed tuples are reused)
            t = np.ascontiguousarray(t)
            for at in range(0, t.shape[0], 1024):
                yield from _entries.entries(buf, t[at:at + 1024], lo, 0)
            continue
        rel = (t - lo).tolist()                 # arrayadd_q(posarray, -offset), whole chunk
        for p0, p1, p2, p3, p4, p5 in rel:
            yield (buf[p0:p1], buf[p2:p3], buf[p4:p5])


def select_rows(table, min_seq_len=None, max_seq_len=None):
    """Rows whose sequence length pos3 - pos2 lies in [min_seq_len, max_seq_len]: the length
    filter of doc/user-guide.rst:153-180 evaluated on the table, before any per-record
    object exists."""
    t = np.asarray(table, dtype=np.int64).reshape(-1, 6)
    ln = t[:, 3] - t[:, 2]
    keep = np.ones(t.shape[0], dtype=bool)
    if min_seq_len is not None:
        keep &= ln >= min_seq_len
    if max_seq_len is not None:
        keep &= ln <= max_seq_len
    return t[keep]


def select_rows_device(ctx, table, min_seq_len=None, max_seq_len=None):
    """select_rows on the GPU: `table` is a CUDA int64[n][6] torch tensor (e.g. the output of a
    device scan); returns a new tensor with the kept rows, in order.  One C-ABI call
    (ffq_table_select_seqlen: count, scan, scatter kernels)."""
    import torch
    n = int(table.shape[0])
    out = torch.empty_like(table)
    lo = -(1 << 62) if min_seq_len is None else int(min_seq_len)
    hi = (1 << 62) if max_seq_len is None else int(max_seq_len)
    k = ctx.table_select_seqlen(table.data_ptr(), n, lo, hi, out.data_ptr()) if n else 0
    return out[:k]


def select_column_device(ctx, buf, table, which, sentinel=True, add=None, value_add=0):
    """One component of every row, packed, on the GPU: `buf` is the CUDA uint8 tensor the rows of
    `table` (CUDA int64[n][6]) were scanned from; which = "header" | "sequence" | "quality".
    Returns (int8 tensor with the bytes, int64 tensor with n + 1 offsets): what an entryfunc that
    builds only that component returns for every entry (doc/user-guide.rst:153-180), before any
    per-record Python object exists.  With value_add = -33 on "quality": the Phred decode."""
    import torch
    n = int(table.shape[0])
    off = torch.empty(n + 1, dtype=torch.int64, device=table.device)
    ca, sh, cb = ctx.COLUMNS[which] if isinstance(which, str) else which
    total = int((table[:, cb] - table[:, ca] - sh).clamp_(min=0).sum().item()) if n else 0
    out = torch.empty(max(total, 16), dtype=torch.int8, device=table.device)
    rc, nb = ctx.table_gather_column(buf.data_ptr(), buf.numel(), table.data_ptr(), n, which, out.data_ptr(), total,
                                     off.data_ptr(), sentinel=sentinel, add=add, value_add=value_add)
    assert rc == 0 and nb == total
    return out[:total], off
