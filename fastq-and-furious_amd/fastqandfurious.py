"""Host-side mirror of the reference module `fastqandfurious.fastqandfurious`
for the FASTQ hot path (reference: /root/reference/src/fastqandfurious.py).

Same names, argument order, status constants and exception texts as the
reference, so code written against it runs unchanged:

    from fastqandfurious_amd import fastqandfurious, _fastqandfurious
    for header, sequence, quality in fastqandfurious.readfastq_iter(
            fh, fbufsize, fastqandfurious.entryfunc, _fastqandfurious.entrypos):
        ...

`entrypos` below is the pure-Python plug-in scanner of the reference
(:39-100) -- the CPU-only configuration of BASELINE.json -- written from its
documented behaviour.  `_fastqandfurious.entrypos` is the MI355X scanner; when
readfastq_iter is handed a scanner that exposes `scan_buffer` it parses every
record of a buffer fill with ONE batched GPU call instead of one call per
record, and yields exactly the same entries.

Beyond the hot path: the FASTA plug-in scanner of the reference
(`entrypos_fasta`, `entryfunc_fasta`, :103-143, :174-183) and a working `automagic_open`
(:282-334; the reference's own cannot run: it calls `importlib.importmodule` and ignores its
`openers` argument).
"""
from array import array
from collections import namedtuple
import importlib
import os
import typing

from . import entries as _entries

CHAR_AT: int = ord(b'@')
CHAR_PLUS: int = ord(b'+')
CHAR_NEWLINE: int = ord(b'\n')
BYTES_NEWLINE_AT: bytes = b'\n@'
BYTES_NEWLINE_PLUS: bytes = b'\n+'
CHAR_GT: int = ord(b'>')
BYTES_NEWLINE_GT: bytes = b'\n>'
ARRAY_INIT = array('q', [-1, ] * 6)

Entry = namedtuple('Entry', 'header sequence quality')
EntryType = typing.Tuple[bytes, bytes, typing.Optional[bytes]]

# status codes, reference :19-27
INVALID: int = -1
MISSING_SEQHEADER_BEGIN: int = 0
MISSING_SEQHEADER_END: int = 1
MISSING_SEQ_BEG: int = 2
MISSING_SEQ_END: int = 3
MISSING_QUAL_BEGIN: int = 4
MISSING_QUAL_END: int = 5
COMPLETE: int = 6
MISSING_QUALHEADER_END: int = 7

# end states reported by a batched scanner (include/ffq.h FFQ_END_*)
_END_OK, _END_REFILL, _END_ERR_FINAL_QUAL, _END_ERR_INCOMPLETE, _END_ERR_INVALID = range(5)


def read(fh: typing.BinaryIO, fbufsize: int) -> typing.Tuple[bytes, bool]:
    """One chunk of the stream and whether it was the last one.

    Reference :30-36: end of stream is signalled by a short read."""
    blob = fh.read(fbufsize)
    return (blob, len(blob) < fbufsize)


def entrypos(buf: bytes, offset: int, posbuffer) -> int:
    """Pure-Python scanner: positions of the next FASTQ entry in `buf`.

    Behaviour of the reference's Python `entrypos` (:39-100): searches with
    bytes.find, fills posbuffer[0..5] as far as it gets (earlier content is
    left in place, there is no reset) and returns a status code.
    """
    size = len(buf)
    nl_at = buf.find(BYTES_NEWLINE_AT, offset)
    if nl_at < 0:
        return MISSING_SEQHEADER_BEGIN
    posbuffer[0] = nl_at + 1
    head_end = buf.find(b'\n', nl_at + 2)
    if head_end < 0:
        return MISSING_SEQHEADER_END
    posbuffer[1] = head_end
    seq_beg = head_end + 1
    if seq_beg >= size:
        return MISSING_SEQ_BEG
    posbuffer[2] = seq_beg
    seq_end = buf.find(BYTES_NEWLINE_PLUS, seq_beg)
    if seq_end < 0:
        return MISSING_SEQ_END
    posbuffer[3] = seq_end
    plus_end = buf.find(b'\n', seq_end + 2)
    if plus_end < 0:
        return MISSING_QUALHEADER_END
    plus_line = plus_end - seq_end          # '+' line incl. its newline
    if plus_line - 1 > 1 and plus_line != head_end - nl_at:
        return INVALID                      # '+' line carries text of another length
    qual_beg = plus_end + 1
    if qual_beg >= size:
        return MISSING_QUAL_BEGIN
    posbuffer[4] = qual_beg
    qual_end = qual_beg + (seq_end - seq_beg)
    if qual_end + 2 >= size:
        return MISSING_QUAL_END
    posbuffer[5] = qual_end
    return COMPLETE


def entrypos_fasta(buf: bytes, offset: int, posbuffer) -> int:
    """Plug-in scanner for FASTA (reference :103-143): next "\\n>" from `offset`, end of the
    header line, then the sequence up to the next "\\n>".  Fills posbuffer[0..3] as far as
    it gets and returns MISSING_SEQHEADER_BEGIN / _END, MISSING_SEQ_BEG, MISSING_SEQ_END (the
    buffer ended first: posbuffer[3] is then the end of the buffer, without a trailing
    newline) or COMPLETE."""
    i = buf.find(BYTES_NEWLINE_GT, offset)
    if i < 0:
        return MISSING_SEQHEADER_BEGIN
    posbuffer[0] = i + 1
    j = buf.find(b'\n', i + 2)
    if j < 0:
        return MISSING_SEQHEADER_END
    posbuffer[1] = j
    if j + 1 >= len(buf):
        return MISSING_SEQ_BEG
    posbuffer[2] = j + 1
    k = buf.find(BYTES_NEWLINE_GT, j + 1)
    if k < 0:
        posbuffer[3] = len(buf) - 1 if buf[-1] == CHAR_NEWLINE else len(buf)
        return MISSING_SEQ_END
    posbuffer[3] = k
    return COMPLETE


def entryfunc_fasta(buf: bytes, pos, globaloffset: int):
    """(header, sequence) byte slices of a FASTA entry (reference :174-183)."""
    return (buf[(pos[0] + 1):pos[1]], buf[pos[2]:pos[3]])


def entryfunc_namedtuple(buf: bytes, pos, globaloffset: int) -> Entry:
    """Entry(header, sequence, quality) namedtuple (reference :146-158)."""
    return Entry(buf[(pos[0] + 1):pos[1]], buf[pos[2]:pos[3]], buf[pos[4]:pos[5]])


def entryfunc(buf: bytes, pos, globaloffset: int) -> EntryType:
    """(header, sequence, quality) byte slices (reference :161-171)."""
    return (buf[(pos[0] + 1):pos[1]], buf[pos[2]:pos[3]], buf[pos[4]:pos[5]])


_ENTRYFUNC = entryfunc          # (readfastq_iter's parameter shadows the name)


def entryfunc_phred(buf: bytes, pos, globaloffset: int):
    """(header, sequence, quality) with the quality as an array('b') of Phred scores: the entryfunc the
    reference's user guide writes for this (doc/user-guide.rst:126-141, :206-214; benchmark.py:155-168),

        quality = array('b'); quality.frombytes(buf[pos[4]:pos[5]]); arrayadd_b(quality, -33)

    Called per record (any scanner) it does exactly that, through this package's arrayadd_b -- one
    device round trip per record.  readfastq_iter RECOGNISES it when the scanner is the GPU one: the
    stream front end then decodes the qualities of a whole buffer fill on the device, together with
    the scan (FFQ_F_DECODE_QUAL, k_decode_stream), and the iterator only wraps the decoded bytes --
    the same entries, without a call per record."""
    from . import _fastqandfurious as _C
    quality = array('b')
    quality.frombytes(buf[pos[4]:pos[5]])
    _C.arrayadd_b(quality, -33)
    return (buf[(pos[0] + 1):pos[1]], buf[pos[2]:pos[3]], quality)


class entryfunc_lengthfilter:
    """The length filter of the reference's user guide (doc/user-guide.rst:153-180) as an entryfunc OBJECT:

        LENGTH_THRESHOLD = 25
        def lengthfilter_entryfunc(buf, posarray):
            if posarray[3] - posarray[2] < LENGTH_THRESHOLD:
                return buf[posarray[2]:posarray[3]]
            else:
                return None

    is entryfunc_lengthfilter(25): called per record (any scanner; with or without the third argument the iterator
    passes, :255) it does exactly that -- the sequence of a read SHORTER than the threshold, None for the others; the
    iterator yields one item per record either way.  More generally min_len <= pos[3] - pos[2] <= max_len is kept
    (the length is the byte length of the slice, as in the guide: a wrapped read counts its newlines), and `column`
    says what is built for a kept record: "sequence" (the guide's), "header", "quality", or "entry" -- the (header,
    sequence, quality) tuple of entryfunc.

    readfastq_iter RECOGNISES the object when the scanner is the GPU one: the stream front end filters every buffer
    fill's offset table on the device, gathers the one component of the kept rows there (ffq_stream_set_filter), and
    copies back nothing of a dropped record -- for which the iterator then spends one None in a list instead of a
    scanner call, an entryfunc call and three slices.  Same items, same order.

    yield_dropped=False (an extension: the guide's loop skips the Nones itself, `if sequence is None: # do nothing`):
    the iterator yields the kept records' items only -- a dropped record then never reaches the interpreter at all."""

    def __init__(self, threshold=None, min_len=None, max_len=None, column="sequence", yield_dropped=True):
        if column not in ("sequence", "header", "quality", "entry"):
            raise ValueError("column must be 'sequence', 'header', 'quality' or 'entry'")
        if threshold is not None:
            if max_len is not None:
                raise ValueError("threshold and max_len say the same thing")
            max_len = int(threshold) - 1
        self.min_len = None if min_len is None else int(min_len)
        self.max_len = None if max_len is None else int(max_len)
        self.column = column
        self.yield_dropped = bool(yield_dropped)

    def keeps(self, length):
        return (self.min_len is None or length >= self.min_len) and (self.max_len is None or length <= self.max_len)

    def __call__(self, buf, pos, globaloffset=None):
        if not self.keeps(pos[3] - pos[2]):
            return None
        c = self.column
        if c == "sequence":
            return buf[pos[2]:pos[3]]
        if c == "header":
            return buf[(pos[0] + 1):pos[1]]
        if c == "quality":
            return buf[pos[4]:pos[5]]
        return (buf[(pos[0] + 1):pos[1]], buf[pos[2]:pos[3]], buf[pos[4]:pos[5]])


def entryfunc_abspos(buf: bytes, pos, globaloffset: int):
    """Absolute stream positions: pos[i] += globaloffset, in place; returns
    the same `pos` object (reference :186-195)."""
    for i in range(6):
        pos[i] += globaloffset
    return pos


def _raise_for_end(end_state: int, where: int):
    if end_state == _END_ERR_FINAL_QUAL:
        raise ValueError('Incomplete final quality string at byte')
    if end_state == _END_ERR_INCOMPLETE:
        raise ValueError('Incomplete entry at byte %i' % where)
    if end_state == _END_ERR_INVALID:
        raise ValueError('Entry is invalid at byte %i' % where)
    raise RuntimeError('unknown end state %r' % (end_state,))


_ENTRY_CHUNK = 1024     # rows per native call: the tuples of one chunk are consumed (and their memory
                        # reused) before the next is built -- a whole fill at once is 3 x slower


def _default_entries(buf, rows, shift, cls=None):
    """The default entryfunc (:161-171) over a table: (header, sequence, quality) of every row of
    `rows` (C-contiguous int64, six per record; positions minus `shift` index `buf`); cls=Entry: what
    entryfunc_namedtuple (:146-158) builds."""
    cut = _entries.native().entries
    mv = memoryview(rows).cast('B')
    step = 48 * _ENTRY_CHUNK
    # (one LIST per chunk: the caller yields from the list itself -- a generator level less per record)
    for at in range(0, len(mv), step):
        yield cut(buf, mv[at:at + step], shift, 1, cls)


def _pushes_down(entryfunc):
    """Is this entryfunc the library's own length filter, unchanged?  Only then may the device evaluate it from min_len /
    max_len / column alone; a subclass that overrides __call__ or keeps() is CALLED per record like any other entryfunc --
    on every scanner alike (the same entryfunc must not yield different items depending on the scanner)."""
    if not isinstance(entryfunc, entryfunc_lengthfilter):
        return False
    t = type(entryfunc)
    return t is entryfunc_lengthfilter or (t.__call__ is entryfunc_lengthfilter.__call__ and
                                           getattr(t, "keeps", None) is getattr(entryfunc_lengthfilter, "keeps", None))


def _phred_entries(st, fill, rows, shift):
    """entryfunc_phred over a whole table, from the stream's bulk decode of that fill."""
    qual, qoff = st.quals()
    nat = _entries.native()
    if nat is not None and hasattr(nat, "entries_phred"):
        mv, mo = memoryview(rows).cast('B'), memoryview(qoff).cast('B')
        step = _ENTRY_CHUNK
        for at in range(0, rows.shape[0], step):
            yield nat.entries_phred(fill, mv[48 * at:48 * (at + step)], shift, qual, mo[8 * at:8 * (at + step + 1)], array)
        return
    buf = fill.tobytes()
    qb = qual.tobytes()
    offs = qoff.tolist()
    # (record i's bytes: qual[qoff[i] : qoff[i] + pos5 - pos4] -- packed stream and single-pass segments alike)
    yield [(buf[p0 + 1:p1], buf[p2:p3], array('b', qb[offs[i]:offs[i] + p5 - p4]))
           for i, (p0, p1, p2, p3, p4, p5) in enumerate((rows - shift).tolist())]


def _np_arange(k):
    import numpy as np
    return np.arange(k, dtype=np.int64)


def _filtered_items(st, flt, rows, fill, fill_offset):
    """What readfastq_iter yields for one fill of a filtered stream: n_scanned items, None for the dropped records, the
    filter's component for the kept ones (from the device's gathered column, or cut out of the fill for "entry")."""
    idx, n_scanned, col, off = st.selected()
    nat = _entries.native()
    k = int(idx.shape[0])
    if not flt.yield_dropped:                   # the kept records' items only: as if every record had been kept
        n_scanned, idx = k, _np_arange(k)
    if n_scanned == 0:
        return []
    if k == 0:
        return [None] * n_scanned
    if flt.column == "entry":
        if nat is not None and hasattr(nat, "sparse_entries"):
            return nat.sparse_entries(n_scanned, memoryview(idx).cast('B'), fill, memoryview(rows).cast('B'), fill_offset)
        out = [None] * n_scanned
        buf = fill.tobytes()
        for k, (p0, p1, p2, p3, p4, p5) in zip(idx.tolist(), (rows - fill_offset).tolist()):
            out[k] = (buf[p0 + 1:p1], buf[p2:p3], buf[p4:p5])
        return out
    if nat is not None and hasattr(nat, "sparse"):
        return nat.sparse(n_scanned, memoryview(idx).cast('B'), col, memoryview(off).cast('B'))
    out = [None] * n_scanned
    cb, o = col.tobytes(), off.tolist()
    for j, k in enumerate(idx.tolist()):
        out[k] = cb[o[j]:o[j + 1]]
    return out


def _iter_batched(fh, fbufsize, entryfunc, scan_buffer):
    """readfastq_iter with a batched scanner: one scan per buffer fill.

    scan_buffer(buf, offset, eof) -> (rows, end_state, end_offset) where rows
    is an array('q') of 6*n buffer-relative positions.  The refill, the
    sentinel, globaloffset and the error texts follow the reference loop
    (:241-279) step for step.
    """
    # a scanner may name a number of bytes below which reads are coalesced into k * fbufsize per scan
    # (the entries do not depend on where the stream is cut into fills)
    co = getattr(getattr(scan_buffer, '__self__', None), 'coalesce_bytes', 0) or 0
    if 0 < fbufsize < co:
        fbufsize *= -(-co // fbufsize)
    globaloffset = -1
    offset = 0
    buf, eof = read(fh, fbufsize)
    buf = b'\n' + buf
    while True:
        rows, end_state, end_offset = scan_buffer(buf, offset, eof)
        if (entryfunc is _ENTRYFUNC or entryfunc is entryfunc_namedtuple) and _entries.native() is not None:
            for chunk in _default_entries(buf, rows, 0, None if entryfunc is _ENTRYFUNC else Entry):
                yield from chunk
        elif entryfunc is _ENTRYFUNC:
            it = iter(rows)                      # the default entryfunc inlined (see _iter_stream)
            for p0, p1, p2, p3, p4, p5 in zip(it, it, it, it, it, it):
                yield (buf[p0 + 1:p1], buf[p2:p3], buf[p4:p5])
        elif isinstance(entryfunc, entryfunc_lengthfilter) and not entryfunc.yield_dropped:
            for i in range(0, len(rows), 6):
                e = entryfunc(buf, rows[i:i + 6], globaloffset)
                if e is not None:
                    yield e
        else:
            for i in range(0, len(rows), 6):
                yield entryfunc(buf, rows[i:i + 6], globaloffset)
        offset = end_offset
        if end_state == _END_OK:
            return
        if end_state != _END_REFILL:
            _raise_for_end(end_state, globaloffset + offset)
        globaloffset += offset
        tmp_buf, eof = read(fh, fbufsize)
        buf = buf[offset:] + tmp_buf
        del tmp_buf
        offset = 0


def _iter_stream(st, entryfunc):
    """readfastq_iter over the native stream front end (ffq_stream_*): the library reads the
    file (ahead, into pinned memory), scans every buffer fill and hands back the rows; this loop
    only builds the entries.  `buf` is the fill as bytes and `pos` its buffer-relative positions,
    exactly what the reference's loop passes to entryfunc (:252-255), globaloffset included."""
    try:
        for rows, fill, fill_offset, end_state, err_offset in st:
            if st.filtered:
                # the stream dropped rows on the device (entryfunc_lengthfilter): one item per scanned record all the same
                yield from _filtered_items(st, entryfunc, rows, fill, fill_offset)
            elif rows.shape[0] and entryfunc is entryfunc_phred and st.decode:
                for chunk in _phred_entries(st, fill, rows, fill_offset):
                    yield from chunk
            elif rows.shape[0] and (entryfunc is _ENTRYFUNC or entryfunc is entryfunc_namedtuple) and _entries.native() is not None:
                # the default entryfunc over the whole table, natively (csrc/ffq_entries.c): the slices
                # are cut straight out of the stream's own (pinned) fill -- no bytes copy of the fill,
                # no posbuffer and no interpreter loop per record
                for chunk in _default_entries(fill, rows, fill_offset, None if entryfunc is _ENTRYFUNC else Entry):
                    yield from chunk
            elif rows.shape[0]:
                buf = fill.tobytes()
                rel = array('q')
                rel.frombytes((rows - fill_offset).tobytes())
                if entryfunc is _ENTRYFUNC:
                    # the default entryfunc, inlined over the whole table: the same three slices per
                    # record (:161-171) without a posbuffer object and a call per record (1.7 x the
                    # entries per second; a list of lists from tolist() is slower than either)
                    it = iter(rel)
                    for p0, p1, p2, p3, p4, p5 in zip(it, it, it, it, it, it):
                        yield (buf[p0 + 1:p1], buf[p2:p3], buf[p4:p5])
                elif isinstance(entryfunc, entryfunc_lengthfilter) and not entryfunc.yield_dropped:
                    # (a length filter that is not pushed down -- a subclass with a __call__ / keeps() of its own: called per
                    # record, its dropped records left out as on every other scanner)
                    for i in range(0, len(rel), 6):
                        e = entryfunc(buf, rel[i:i + 6], fill_offset)
                        if e is not None:
                            yield e
                else:
                    for i in range(0, len(rel), 6):
                        yield entryfunc(buf, rel[i:i + 6], fill_offset)
            if end_state != _END_OK and end_state != _END_REFILL:
                _raise_for_end(end_state, err_offset)
    finally:
        st.close()


def readfastq_iter(fh: typing.BinaryIO, fbufsize: int,
                   entryfunc: typing.Callable = entryfunc,
                   entrypos: typing.Callable = entrypos,
                   globaloffset: int = 0) -> typing.Iterator[EntryType]:
    """Iterate through entries in a FASTQ stream (reference :198-279).

    :param fh: anything with a `read(n)` method returning bytes
    :param fbufsize: chunk size of the reads from `fh`
    :param entryfunc: builds the yielded object from (buf, pos, globaloffset)
    :param entrypos: scanner (buf, offset, posbuffer) -> status
    :param globaloffset: accepted and ignored, as in the reference (:242)

    Differences from the reference, both on malformed input only: an INVALID
    entry met after the end of the stream raises 'Entry is invalid at byte'
    (the reference never leaves its loop, :256-270).
    """
    open_stream = getattr(entrypos, 'open_stream', None)
    if open_stream is not None:
        # the native stream front end: a real file, a gzip file, or anything with readinto() / read();
        # with entryfunc_phred the qualities of every fill are decoded on the device
        st = open_stream(fh, fbufsize, entryfunc is entryfunc_phred) if entryfunc is entryfunc_phred else open_stream(fh, fbufsize)
        if st is not None and _pushes_down(entryfunc):
            # push-down: the filter runs on the device, on every fill's table, before anything is copied back
            try:
                st.set_filter(entryfunc.min_len, entryfunc.max_len, None if entryfunc.column == "entry" else entryfunc.column)
            except BaseException:
                st.close()                 # (the native stream, its pinned buffers and its feeder thread; a gzip stream's hook)
                raise
        if st is not None:
            yield from _iter_stream(st, entryfunc)
            return
    scan_buffer = getattr(entrypos, 'scan_buffer', None)
    if scan_buffer is not None:
        yield from _iter_batched(fh, fbufsize, entryfunc, scan_buffer)
        return

    if isinstance(entryfunc, entryfunc_lengthfilter) and not entryfunc.yield_dropped:
        # (the kept records only: the reference's loop below with the guide's `if sequence is None: # do nothing` folded in;
        # the object itself -- a subclass with a __call__ / keeps() of its own included -- with yield_dropped switched on)
        import copy
        keep_all = copy.copy(entryfunc)
        keep_all.yield_dropped = True
        yield from (e for e in readfastq_iter(fh, fbufsize, keep_all, entrypos) if e is not None)
        return
    posbuffer = array('q', [-1, ] * 6)
    globaloffset = -1
    offset = 0
    buf, eof = read(fh, fbufsize)
    buf = b'\n' + buf               # sentinel: the first '@' is then a "\n@" match
    while True:
        status = entrypos(buf, offset, posbuffer)
        if status == COMPLETE:
            offset = posbuffer[5] - 1
            yield entryfunc(buf, posbuffer, globaloffset)
            continue
        if eof:
            if status == MISSING_SEQHEADER_BEGIN:
                return
            if status == MISSING_QUAL_END:
                # last record of a stream: its quality may run to the last byte
                qualend = posbuffer[4] + (posbuffer[3] - posbuffer[2])
                if qualend >= len(buf):
                    raise ValueError('Incomplete final quality string at byte')
                posbuffer[5] = qualend
                yield entryfunc(buf, posbuffer, globaloffset)
                return
            if status == INVALID:
                raise ValueError('Entry is invalid at byte %i' % (globaloffset + offset))
            raise ValueError('Incomplete entry at byte %i' % (globaloffset + offset))
        if status == INVALID:
            raise ValueError('Entry is invalid at byte %i' % (globaloffset + offset))
        globaloffset += offset
        tmp_buf, eof = read(fh, fbufsize)
        buf = buf[offset:] + tmp_buf
        del tmp_buf
        offset = 0


class RangeEntries:
    """What readfastq_iter_range returns: an iterator over ONE rank's entries of a file that `world` ranks read
    together, with what the ranks agreed on -- `record_base` (global ordinal of this rank's first record: entry i of
    this iterator is record record_base + i of the file), `n_records` (this rank's), `total_records` (the file's),
    `bounds` (the byte ranges), `comm` (the step's figures: transport, halo source, repair rounds)."""

    def __init__(self, shard, entryfunc, batch_rows):
        self._sh, self._entryfunc, self._batch = shard, entryfunc, int(batch_rows)
        res = shard.out
        self.record_base, self.total_records = int(res.record_base), int(res.total_records)
        self.n_records = int(res.row_hi - res.row_lo)
        self.bounds = list(shard.bounds)
        self.comm = {"transport": shard.sh.transport(), "halo_source": "file" if res.halo_source else "ranks",
                     "rescan_rounds": int(res.rounds), "regathers": int(res.regathers), "allgather_ms": float(res.allgather_ms)}
        self._gen = self._entries()

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._gen)

    def close(self):
        self._gen.close()
        self._sh.close()         # (a generator that never started has no `finally` to run)

    def _filtered(self, view, mm):
        """entryfunc_lengthfilter over this rank's records with the filter ON THE DEVICE (FileShard.select): one item per
        record -- None for a dropped one, the filter's component for a kept one (/root/reference/doc/user-guide.rst:153-180) --
        or, yield_dropped=False, the kept records' items alone.  Only the kept rows (and, from a resident range, their
        gathered component) cross the link; a dropped record costs the host one pointer in a list."""
        import numpy as np
        sh, flt, nat = self._sh, self._entryfunc, _entries.native()
        k, idx = sh.select(flt.min_len, flt.max_len)
        step = self._batch
        for i0 in range(0, self.n_records, step):                       # windows of ORIGINAL records
            i1 = min(i0 + step, self.n_records)
            ka, kb = int(np.searchsorted(idx, i0)), int(np.searchsorted(idx, i1))
            n_items = i1 - i0
            if kb == ka:
                if flt.yield_dropped:
                    yield from [None] * n_items
                continue
            rows = sh.kept_rows(ka, kb)
            local = np.ascontiguousarray(idx[ka:kb] - i0)
            if not flt.yield_dropped:
                n_items, local = kb - ka, _np_arange(kb - ka)
            a, b = int(rows[0, 0]), int(rows[-1, 5]) + 1
            if flt.column == "entry":
                if nat is not None and hasattr(nat, "sparse_entries"):
                    yield from nat.sparse_entries(n_items, memoryview(local).cast('B'), view[a:b], memoryview(rows).cast('B'), a)
                else:
                    out, buf = [None] * n_items, mm[a:b]
                    for j, (p0, p1, p2, p3, p4, p5) in zip(local.tolist(), (rows - a).tolist()):
                        out[j] = (buf[p0 + 1:p1], buf[p2:p3], buf[p4:p5])
                    yield from out
                continue
            got = sh.kept_column(ka, kb, flt.column, rows)
            if got is None:
                # the range is not resident (slabs; a view that grew): the kept rows' slices out of the file, in one numpy gather
                ca, shf, cb = {"header": (0, 1, 1), "sequence": (2, 0, 3), "quality": (4, 0, 5)}[flt.column]
                beg, lens = rows[:, ca] + shf - a, np.maximum(rows[:, cb] - rows[:, ca] - shf, 0)
                off = np.zeros(kb - ka + 1, dtype=np.int64)
                np.cumsum(lens, out=off[1:])
                src = np.repeat(beg - off[:-1], lens) + np.arange(int(off[-1]), dtype=np.int64)
                col = np.frombuffer(view[a:b], dtype=np.uint8)[src]
            else:
                col, off = got
            if nat is not None and hasattr(nat, "sparse"):
                yield from nat.sparse(n_items, memoryview(local).cast('B'), col, memoryview(off).cast('B'))
            else:
                out, cb_, offs = [None] * n_items, col.tobytes(), off.tolist()
                for j, kk in zip(local.tolist(), range(kb - ka)):
                    out[j] = cb_[offs[kk]:offs[kk + 1]]
                yield from out

    def _entries(self):
        import mmap
        sh, entryfunc = self._sh, self._entryfunc
        mm = None
        try:
            if self.n_records and hasattr(sh, "host_bytes"):
                # (a BGZF shard: the rank's inflated bytes are in host memory, sliced with stream offsets like a map of a plain file)
                from .sharded import _Shifted
                base, arr = sh.host_bytes()
                mm, view = _Shifted(arr, base, as_bytes=True), _Shifted(arr, base)
            elif self.n_records:
                mm = mmap.mmap(sh.fd, 0, access=mmap.ACCESS_READ)       # (the page cache holds the range: it was just read)
                view = memoryview(mm)
            if self.n_records and _pushes_down(entryfunc):
                yield from self._filtered(view, mm)
                return
            drop_none = isinstance(entryfunc, entryfunc_lengthfilter) and not entryfunc.yield_dropped      # (a filter that is not pushed down)
            for i0 in range(0, self.n_records, self._batch):
                i1 = min(i0 + self._batch, self.n_records)
                rows = sh.rows(i0, i1)
                a, b = int(rows[0, 0]), int(rows[-1, 5]) + 1
                if entryfunc is entryfunc_phred and (sh.decoded or sh.slab_bytes) and _entries.native() is not None and hasattr(_entries.native(), "entries_phred"):
                    # the qualities from the step's own decode -- or, over slabs (nothing resident), decoded on the device batch by batch
                    qual, qoff = sh.quals(i0, i1, rows) if sh.decoded else sh.quals_from_file(rows)
                    yield from _entries.native().entries_phred(view[a:b], memoryview(rows).cast('B'), a, qual, memoryview(qoff).cast('B'), array)
                elif (entryfunc is _ENTRYFUNC or entryfunc is entryfunc_namedtuple) and _entries.native() is not None:
                    for chunk in _default_entries(view[a:b], rows, a, None if entryfunc is _ENTRYFUNC else Entry):
                        yield from chunk
                else:
                    # any entryfunc: `buf` holds the batch's bytes, `pos` is relative to it and pos + globaloffset the
                    # absolute file offsets (entryfunc_abspos, :186-195) -- the contract of the reference's loop (:252-255)
                    buf = mm[a:b]
                    rel = array('q')
                    rel.frombytes((rows - a).tobytes())
                    for i in range(0, len(rel), 6):
                        e = entryfunc(buf, rel[i:i + 6], a)
                        if e is not None or not drop_none:
                            yield e
        finally:
            if mm is not None:
                try:
                    view.release()
                    mm.close()
                except BufferError:
                    pass                  # (an entry handed out still views the map: it goes with the last reference)
            sh.close()


def readfastq_iter_range(path, rank: int, world: int, entryfunc: typing.Callable = entryfunc, comm=None, ctx=None,
                         start: int = 0, end: typing.Optional[int] = None, batch_rows: int = 1 << 15,
                         tail_bytes: typing.Optional[int] = None, head_bytes: typing.Optional[int] = None,
                         bounds: typing.Optional[typing.Sequence[int]] = None,
                         slab_bytes: typing.Optional[int] = None, bgzf: typing.Optional[bool] = None,
                         exchange: typing.Optional[typing.Callable] = None) -> RangeEntries:
    """readfastq_iter for ONE FILE read by `world` ranks (one process per GPU): rank `rank` gets the entries whose '@'
    lies in its byte range [S_rank, S_rank+1) of the file, the same objects in the same order the reference's
    iterator (:198-279) yields for them -- the ranks' iterators concatenated ARE readfastq_iter over the whole file.

    Collective: every rank calls it (and reaches its first entry only when all have: the ranges are proven against
    each other in one step, ffq_shard_step_*).  Each rank reads its own range of the file (plus 1 MiB either side)
    into its GPU's memory -- nothing is handed from rank to rank but eight words each -- (a 100 GiB file over 8 GPUs:
    12.5 GiB each); a range that does not fit there -- or slab_bytes= / FFQ_SHARD_SLAB_BYTES -- goes through one device
    buffer slab after slab (sharded.FileShard), the entries are the same.  Errors of the stream (the iterator's three ValueErrors,
    :262, :269, :272) are raised on every rank alike, before any entry is yielded.

    comm: None (world 1; or torch.distributed's default group hands the communicator id round), 128 bytes of
    ffq_shard_unique_id, or a hip.ShardWorld (logical ranks as threads).  entryfunc_phred: the qualities are decoded
    on the device with the scan.  Returns a RangeEntries (iterate it; .record_base is the global ordinal).

    A BGZF file (bgzip's output; bgzf=None: recognised by its first member, True / False: said by the caller) is read by
    ranges too -- the one compressed format that can be: every rank inflates the members that begin in its share of the
    compressed file, the entries are those of the UNCOMPRESSED stream, what readfastq_iter(gzip.open(path), ...) yields
    (sharded.BgzfFileShard; exchange: how the ranks tell each other their sizes when comm is not a transport object and
    torch.distributed is not there).  start / end / bounds / slab_bytes do not apply to it."""
    from . import hip as _hip, sharded as _sharded
    if ctx is None:
        ctx = _hip.default_context()
    if bgzf is None:
        bgzf = _sharded.is_bgzf(path)
        if not bgzf and _sharded.is_gzip(path):
            raise ValueError("readfastq_iter_range: %r is a gzip file but not BGZF -- ordinary gzip members cannot be entered in the "
                             "middle, so the file cannot be read by ranges; readfastq_iter(automagic_open(path), ...) streams it "
                             "(or recompress it with bgzip)" % (path,))
    if bgzf:
        if start or end is not None or bounds is not None or slab_bytes:
            raise ValueError("readfastq_iter_range: start / end / bounds / slab_bytes do not apply to a BGZF file")
        kwz = {}
        if tail_bytes is not None:
            kwz["tail_bytes"] = tail_bytes
        if head_bytes is not None:
            kwz["head_bytes"] = head_bytes
        sh = _sharded.BgzfFileShard(ctx, path, rank, world, comm=comm, exchange=exchange, **kwz)
        try:
            sh.load()
            sh.scan(decode=entryfunc is entryfunc_phred)
        except BaseException:
            sh.close()
            raise
        return RangeEntries(sh, entryfunc, batch_rows)
    kw = {}
    if tail_bytes is not None:
        kw["tail_bytes"] = tail_bytes
    if head_bytes is not None:
        kw["head_bytes"] = head_bytes
    if slab_bytes:
        kw["slab_bytes"] = slab_bytes
    sh = _sharded.FileShard(ctx, path, rank, world, comm=comm, start=start, end=end, bounds=bounds, **kw)
    try:
        sh.load()
        sh.scan(decode=entryfunc is entryfunc_phred and not sh.slab_bytes)      # (over slabs: decoded batch by batch, FileShard.quals_from_file)
    except BaseException:
        sh.close()
        raise
    return RangeEntries(sh, entryfunc, batch_rows)


# extension -> (module name or namespace, opener name, positional arguments after the file name)
FORMAT_OPENERS: typing.Dict[str, typing.Tuple[typing.Union[str, object], str, list]] = {
    'gz': ('gzip', 'open', list()),
    'gzip': ('gzip', 'open', list()),
    'bgz': ('gzip', 'open', list()),        # (bgzip's output is a gzip file: members that carry their length)
    'bz2': ('bz2', 'open', list()),
    'lzma': ('lzma', 'open', list()),
    'xz': ('lzma', 'open', list()),
}


def automagic_open(filename, openers=None) -> typing.BinaryIO:
    """Open a (presumably FASTQ) file, compressed or not, by its extension (reference
    :290-334): `foo/bar.fq.gz` through gzip, `foo/bar.fq` as a plain binary file.  `openers`
    maps extensions to (module name or namespace, function name, extra positional arguments);
    None means FORMAT_OPENERS.  The stream it returns feeds readfastq_iter: decompression
    runs on the host, the scan of every buffer fill on the GPU."""
    if openers is None:
        openers = FORMAT_OPENERS
    parts = str(filename).rsplit(os.path.extsep, maxsplit=1)
    ext = parts[-1] if len(parts) > 1 else None
    try:
        modulename, funcname, args = openers[ext]
    except KeyError:
        modulename, funcname, args = ('io', 'open', ('rb', ))
    module = importlib.import_module(modulename) if isinstance(modulename, str) else modulename
    return getattr(module, funcname)(filename, *args)
