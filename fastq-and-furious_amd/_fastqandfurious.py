"""Mirror of the reference C extension `fastqandfurious._fastqandfurious`
(/root/reference/src/_fastqandfurious.c) on the MI355X.

    entrypos(blob, offset, posbuffer) -> int     (:25-153)
    arrayadd_b(a, value)                         (:161-185)
    arrayadd_q(a, value)                         (:193-217)
    INVALID, POS_HEAD_BEG .. POS_QUAL_END, COMPLETE, MISSING_QUALHEADER_END (:254-262)

Everything computes on the GPU through libffq_hip.so (include/ffq.h).  There is
no CPU fallback: without the library or a gfx950 device the calls raise.

`entrypos` is a callable object.  Called per record (the reference's plug-in
protocol) it scans the whole buffer once on the GPU and serves the following
calls of the same chain from the resulting table; handed to this package's
readfastq_iter it is used through `scan_buffer`, one GPU call per buffer fill.
"""
from array import array

import numpy as np

from . import hip as _hip

INVALID = _hip.INVALID
POS_HEAD_BEG = _hip.POS_HEAD_BEG
POS_HEAD_END = _hip.POS_HEAD_END
POS_SEQ_BEG = _hip.POS_SEQ_BEG
POS_SEQ_END = _hip.POS_SEQ_END
POS_QUAL_BEG = _hip.POS_QUAL_BEG
POS_QUAL_END = _hip.POS_QUAL_END
COMPLETE = _hip.COMPLETE
MISSING_QUALHEADER_END = _hip.MISSING_QUALHEADER_END


def _writable_view(obj, itemsize, fmt_name):
    """Writable view of a buffer-protocol object with the itemsize the
    reference checks (:38-43, :170-174, :202-206)."""
    try:
        m = memoryview(obj)
    except TypeError:
        raise TypeError("a bytes-like object is required, not '%s'" % type(obj).__name__)
    if m.itemsize != itemsize:
        raise ValueError("The buffer must be of format type %s." % fmt_name)
    if m.readonly:
        # the reference's "y*" writes through read-only buffers; we refuse
        raise TypeError("a writable bytes-like object is required")
    if not m.c_contiguous:
        raise BufferError("the buffer must be C-contiguous")
    return m


class _GpuEntrypos:
    """entrypos(blob, offset, posbuffer) -> status, computed on the MI355X."""

    def __init__(self, device=None):
        self._device = device
        self._ctx = None
        self._blob = None
        self._table = None
        self._row = 0
        self._next_offset = None
        self._term = None

    def _context(self):
        if self._ctx is None:
            self._ctx = _hip.default_context(self._device)
        return self._ctx

    # -- the reference's plug-in protocol ---------------------------------
    def __call__(self, blob, offset, posbuffer):
        pos = _writable_view(posbuffer, 8, 'q')
        if pos.nbytes < 48:
            raise ValueError("posbuffer must hold 6 positions")
        out = np.frombuffer(pos, dtype=np.int64)
        if isinstance(blob, str):
            blob = blob.encode('utf-8')
        offset = int(offset)
        cacheable = isinstance(blob, bytes)
        if not (cacheable and self._blob is blob and offset == self._next_offset):
            table, res = self._context().scan_host(blob, sentinel=False, offset=offset, eof=False, add=0)
            self._blob = blob if cacheable else None
            self._table = table
            self._row = 0
            self._term = (int(res.last_status), [int(x) for x in res.last_pos])
        if self._row < len(self._table):
            row = self._table[self._row]
            out[:6] = row
            self._row += 1
            self._next_offset = int(row[5]) - 1
            return COMPLETE
        status, last = self._term
        out[:6] = last
        self._next_offset = None        # the chain has ended: rescan on the next call
        return status

    # -- stream protocol: the source is read, carried and scanned by the library itself ----------
    # Buffer fills are scanned whole, and the entries do not depend on how the stream is cut into fills
    # (fastqandfurious.py:274-279 carries every unfinished entry over): reads of the reference's
    # advised 20-50 kB (fastqandfurious.py:229, benchmark.py:415) are therefore COALESCED -- the
    # library reads k * fbufsize bytes per fill, the smallest multiple that reaches this many bytes --
    # instead of paying a host-to-device copy, four kernel launches and a copy back per 150 records.
    coalesce_bytes = 8 << 20

    def chunk_bytes(self, fbufsize):
        fbufsize = max(int(fbufsize), 1)
        if fbufsize >= self.coalesce_bytes:
            return fbufsize
        return fbufsize * -(-self.coalesce_bytes // fbufsize)

    def open_stream(self, fh, fbufsize, decode=False):
        """The native stream front end over `fh`, or None (no read() at all):
        a plain file -> the library reads the descriptor itself (reader threads, pread);
        a fresh GzipFile over a plain file -> the library inflates it itself (zlib in the reader thread);
        anything else with readinto() / read() -> chunks are read here, straight into pinned memory.
        decode: every fill's qualities are decoded on the device (FileStream.quals())."""
        from .index import _fileno, _leave_at
        chunk = self.chunk_bytes(fbufsize)
        f = _fileno(fh)
        if f is not None:
            st = _hip.FileStream(self._context(), f[0], chunk, decode=decode, start=f[1])
            st.on_close = lambda: _leave_at(fh, st)
            return st
        g = _gzip_fileno(fh)
        if g is not None:
            st = _hip.FileStream(self._context(), g[0], chunk, decode=decode, start=g[1], gzip=True)
            st.on_close = lambda: _gzip_leave(fh, st)
            return st
        if getattr(fh, "readinto", None) is None and getattr(fh, "read", None) is None:
            return None
        # (reads are coalesced up to `chunk`, but a live source that comes back short is not waited on beyond fbufsize)
        return _hip.PushStream(self._context(), fh, chunk, decode=decode, min_fill=max(int(fbufsize), 1))

    # -- batched protocol used by this package's readfastq_iter ------------
    def scan_buffer(self, buf, offset, eof):
        table, res = self._context().scan_host(buf, sentinel=False, offset=offset, eof=eof, add=0)
        rows = array('q')
        rows.frombytes(np.ascontiguousarray(table).tobytes())
        return rows, int(res.end_state), int(res.end_offset)


def _gzip_fileno(fh):
    """(fd, start) of the COMPRESSED file under a gzip.GzipFile that has not been read from yet and
    sits on a plain file -- what gzip.open(path) / FORMAT_OPENERS['gz'] return -- else None."""
    import gzip
    import io
    if not isinstance(fh, gzip.GzipFile) or getattr(fh, "mode", None) != gzip.READ:
        return None
    raw = getattr(fh, "fileobj", None)
    base = raw.raw if isinstance(raw, io.BufferedReader) else raw
    if not isinstance(base, io.FileIO):
        return None
    try:
        if fh.tell() != 0:                       # (decompressed position: something was consumed already)
            return None
        return raw.fileno(), raw.tell()
    except (OSError, ValueError, AttributeError):
        return None


def _gzip_leave(fh, st):
    """The library inflated the compressed file itself (pread: the GzipFile object was never read from).  The
    reference's loop leaves `fh` where its last read(fbufsize) ended: exhausted once the stream is through -- the raw
    file is then left at its end, so that a later fh.read() (or a second iterator over the same object) reads as empty
    instead of replaying the whole file --, and behind the fills handed out when the iterator is abandoned early (a
    break out of the loop, an exception in the caller): the GzipFile is moved there (it inflates up to that point
    itself: only an abandoned stream pays for it), and a later fh.read() gives the data that follows."""
    try:
        if st.at_end:
            fh.fileobj.seek(0, 2)
        elif st.consumed > 0:
            fh.seek(st.consumed)
    except (OSError, ValueError, AttributeError, EOFError):
        pass


entrypos = _GpuEntrypos()


class _GpuEntryposFasta:
    """entrypos_fasta(buf, offset, posbuffer) -> status on the MI355X: the FASTA plug-in scanner
    of the reference (fastqandfurious.py:103-143) with the per-call protocol kept; the first call
    on a buffer scans all of it (ffq_scan_fasta_host), the following calls (offset = the previous
    pos[3]) are served from that table.  Like the reference's, it only fills the positions it
    gets to."""

    def __init__(self, device=None):
        self._device = device
        self._blob = None
        self._table = None
        self._row = 0
        self._next_offset = None
        self._term = None

    def __call__(self, buf, offset, posbuffer):
        pos = _writable_view(posbuffer, 8, 'q')
        if pos.nbytes < 32:
            raise ValueError("posbuffer must hold at least 4 positions")
        out = np.frombuffer(pos, dtype=np.int64)
        offset = int(offset)
        cacheable = isinstance(buf, bytes)
        if not (cacheable and self._blob is buf and offset == self._next_offset):
            table, res = _hip.default_context(self._device).scan_fasta_host(buf, offset=offset)
            self._blob = buf if cacheable else None
            self._table = table
            self._row = 0
            self._term = (int(res.last_status), [int(x) for x in res.last_pos])
        if self._row < len(self._table):
            row = self._table[self._row]
            out[:4] = row[:4]
            self._row += 1
            self._next_offset = int(row[3])
            return COMPLETE
        status, last = self._term
        for i in range(4):
            if last[i] >= 0:
                out[i] = last[i]
        self._next_offset = None
        return status


entrypos_fasta = _GpuEntryposFasta()


def arrayadd_b(a, value):
    """Add `value` to every element of the int8 buffer `a`, in place
    (:161-185).  The reference parses `value` as a C short and keeps its low 8
    bits; so does this."""
    if not isinstance(value, int):
        raise TypeError("an integer is required (got type %s)" % type(value).__name__)
    if value < -32768:
        raise OverflowError("signed short integer is less than minimum")
    if value > 32767:
        raise OverflowError("signed short integer is greater than maximum")
    m = _writable_view(a, 1, 'b')
    if m.nbytes:
        _hip.default_context().arrayadd_b(np.frombuffer(m, dtype=np.int8), value)
    return None


def arrayadd_q(a, value):
    """Add `value` to every element of the int64 buffer `a`, in place,
    wrapping modulo 2**64 (:193-217)."""
    if not isinstance(value, int):
        raise TypeError("an integer is required (got type %s)" % type(value).__name__)
    if not (-2**63 <= value < 2**63):
        raise OverflowError("Python int too large to convert to C long")
    m = _writable_view(a, 8, 'q')
    if m.nbytes:
        _hip.default_context().arrayadd_q(np.frombuffer(m, dtype=np.int64), value)
    return None
