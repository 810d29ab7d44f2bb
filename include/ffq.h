/*
 * ffq.h -- C ABI of libffq_hip.so: the MI355X (gfx950) FASTQ buffer-scan path.
 *
 * This is the drop-in boundary for the hot path of lgautier/fastq-and-furious
 * (reference checkout: /root/reference).  Every entry point names the
 * reference interface it replaces (file:line).  Plain pointers and sizes only;
 * no torch / Python types.  The reference-side binding a maintainer would add
 * is shown in INTEGRATION.md.
 *
 * Coordinates.  The reference scanners work on one `bytes` buffer; the
 * iterator prepends a b'\n' sentinel to the stream (fastqandfurious.py:245)
 * and compensates with globaloffset = -1 (:242).  All scan entry points take
 *   sentinel = 0  the n_bytes ARE the buffer the scanner sees, or
 *   sentinel = 1  the buffer is b'\n' + bytes, without the copy,
 * and work in BUFFER coordinates (index into that buffer).  `add` is added to
 * every emitted position (entryfunc_abspos' globaloffset, :186-195), so
 * sentinel = 1, add = -1 yields absolute file offsets.
 *
 * Errors.  Functions return FFQ_OK (0) or a negative FFQ_E_* code;
 * ffq_last_error() gives the text for the calling thread.  There is no CPU
 * fallback anywhere in this library: without a usable gfx950 device the
 * context cannot be created and every compute call fails.
 */
#ifndef FFQ_H
#define FFQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFQ_ABI_VERSION 6

/* scanner status codes -- identical to the reference's module constants
 * (_fastqandfurious.c:7-15,254-262; fastqandfurious.py:19-27)             */
#define FFQ_INVALID                 (-1)
#define FFQ_POS_HEAD_BEG            0   /* MISSING_SEQHEADER_BEGIN */
#define FFQ_POS_HEAD_END            1   /* MISSING_SEQHEADER_END   */
#define FFQ_POS_SEQ_BEG             2   /* MISSING_SEQ_BEG         */
#define FFQ_POS_SEQ_END             3   /* MISSING_SEQ_END         */
#define FFQ_POS_QUAL_BEG            4   /* MISSING_QUAL_BEGIN      */
#define FFQ_POS_QUAL_END            5   /* MISSING_QUAL_END        */
#define FFQ_COMPLETE                6
#define FFQ_MISSING_QUALHEADER_END  7

/* end states of a record chain = the exits of readfastq_iter's loop
 * (fastqandfurious.py:256-279)                                            */
#define FFQ_END_OK              0   /* eof, stream ended cleanly (:257-258, :264-266) */
#define FFQ_END_REFILL          1   /* !eof, entry incomplete: caller refills (:274-279) */
#define FFQ_END_ERR_FINAL_QUAL  2   /* 'Incomplete final quality string at byte' (:262) */
#define FFQ_END_ERR_INCOMPLETE  3   /* 'Incomplete entry at byte %i' (:269)            */
#define FFQ_END_ERR_INVALID     4   /* 'Entry is invalid at byte %i' (:272; at eof the
                                       reference loops forever, we report this instead) */

/* return codes */
#define FFQ_OK               0
#define FFQ_E_NODEVICE      (-1)
#define FFQ_E_HIP           (-2)
#define FFQ_E_ARG           (-3)
#define FFQ_E_NOMEM         (-4)
#define FFQ_E_TABLE_FULL    (-5)   /* more records than table_cap: n_records holds the need */
#define FFQ_E_INTERNAL      (-6)
#define FFQ_E_TIMEOUT       (-7)   /* a shard step's watchdog: ffq_last_error names the stage and the ranks */

/* flags for the scan calls */
#define FFQ_F_DECODE_QUAL   1u     /* also emit Phred-decoded qualities (value added = qual_add) */
#define FFQ_F_FORCE_SERIAL  2u     /* debugging/tests: use the single-wave chain walker */
#define FFQ_F_FORCE_RANKED  4u     /* debugging/tests: use the list-ranking tier */
#define FFQ_F_FORCE_GENERAL 64u    /* debugging/tests: skip the four-line fast path (the general chain kernels and, through
                                      dense regions, the window walker take plain four-line input too)             */
#define FFQ_F_POLL_RESULT   8u     /* completion is signalled through the result block the last kernel
                                      writes into host-mapped memory (ffq_scan_wait polls a word of it)
                                      instead of through an event recorded behind that kernel: one stream
                                      marker less per scan (~3 us of idle GPU); ms_chain / ms_total are
                                      not measured then.  Ignored with FFQ_F_DECODE_QUAL.              */

#define FFQ_F_SINGLE_PASS  16u     /* with FFQ_F_DECODE_QUAL: the caller accepts the decoded qualities WITH GAPS instead of packed --
                                      the bytes of record i are d_qual[d_qoff[i] : d_qoff[i] + (pos5 - pos4)], contiguous and in
                                      file order, but there may be gaps between records (d_qoff[i + 1] - d_qoff[i] is NOT a
                                      length; n_qual_bytes = where the last record's bytes end).  On plain four-line input the
                                      line-index pass itself then writes them (csrc/ffq_fused.h; res.path 6) and the input is
                                      read ONCE -- HBM traffic 1.0 x the algorithmic bytes instead of 1.5 x -- in one of two
                                      layouts, by the room the caller gives:
                                        qual_cap >= FFQ_SEG_STRIDE per 16 KiB tile of the buffer (rounded up): SEGMENTED, every
                                          tile owns that many bytes of d_qual and packs the quality lines that start in it
                                          there; lines up to 512 bytes behind a tile's end (reads of a few hundred bases);
                                        qual_cap >= FFQ_INPLACE_STRIDE per tile: also IN PLACE, d_qoff[i] = the offset pos4 has
                                          in d_buf -- lines of any length (long reads); taken when the segmented pass refuses
                                          the buffer for its shape, and first from then on (until ffq_ctx_forget).
                                      Records of ANY other layout -- WRAPPED ones, whose quality lines are known only behind
                                      the record chain -- decode in one pass as well with FFQ_INPLACE_STRIDE bytes per tile
                                      (round 6): a scan that starts on the general kernels (a context that has met such input
                                      remembers it) has its index pass write EVERY byte of the buffer decoded at its own
                                      offset, d_qoff[i] = the offset pos4 has in d_buf (res.path | FFQ_PATH_IN_PLACE); an
                                      embedded newline of a wrapped quality comes out as '\n' + qual_add, as the reference's
                                      slice + arrayadd_b give it (_fastqandfurious.c:129, doc/user-guide.rst:126-141).
                                      Less room -- or four-line input the single pass cannot vouch for (a quality line longer
                                      than its read, text in front of the first record), or the first scan of a context that
                                      only finds out on the way that its input is not four-line -- and the two passes run: the
                                      output is packed (a special case of the same contract).                                */
#define FFQ_SEG_STRIDE     8704
#define FFQ_INPLACE_STRIDE 16384
#define FFQ_PATH_IN_PLACE  8       /* ffq_scan_result.path bit: see there */
#define FFQ_F_NO_TIMING    32u     /* with FFQ_F_POLL_RESULT: no timing marks around the line-index kernel either (ms_index is
                                      0 for this scan): the front then holds no stream marker at all.  Where the library itself
                                      can prove that the order does not matter it also dispatches the index kernel without a
                                      barrier, so that it starts while the previous scan's last, one-workgroup kernel is still
                                      running: only when the last thing enqueued on the context's stream is the front of another
                                      scan (a context sharing the stream; anything else the library enqueues there -- copies, the
                                      stream front end's event waits, table utilities -- ends that, and so does handing the stream
                                      out through ffq_ctx_stream), and none of that scan's outputs overlaps the bytes this one
                                      reads.  The caller vouches for nothing.  A host that times every n-th scan loses nothing but
                                      the marks' idle microseconds on the others.                                            */

typedef struct ffq_ctx ffq_ctx;

typedef struct ffq_scan_result {
    int64_t n_records;      /* rows of the chain (all of them, even if > table_cap)      */
    int64_t n_qual_bytes;   /* total decoded quality bytes (FFQ_F_DECODE_QUAL)           */
    int64_t end_offset;     /* buffer coordinate where the last, unsuccessful search
                               started: the iterator's `offset` at exit (:254, :275-277) */
    int64_t last_pos[6];    /* posbuffer as the last scanner call left it (+add applied
                               to entries != -1); for FFQ_END_OK after the final-record
                               rule, the final record's row                              */
    int32_t last_status;    /* status of that last call                                   */
    int32_t end_state;      /* FFQ_END_*                                                  */
    int32_t path;           /* 3 = four-line fast path, 6 = the same with the Phred decode done by
                               the index pass itself (one pass over the input), 0 = general chain
                               kernels, 2 = the same with the dense LDS budget, 5 = list ranking
                               over the "\n@" matches (long records), 1 = serial walker;
                               | FFQ_PATH_IN_PLACE (8, on 0 / 2 / 5 / 1): the index pass of that
                               scan also decoded every byte in place -- records of any layout,
                               one pass (FFQ_F_SINGLE_PASS)                                */
    int32_t retries;        /* internal re-runs (line-index pool growth)                  */
    int64_t n_lines;        /* newline count seen by the line-index kernel                */
    float   ms_index;       /* device time of the line-index kernel (hipEvent)            */
    float   ms_chain;       /* device time of chain summary + resolve + emit kernels      */
    float   ms_decode;      /* device time of the quality decode kernel                   */
    float   ms_total;       /* device time of the whole call                              */
} ffq_scan_result;

/* ---- library / context ------------------------------------------------ */
int         ffq_abi_version(void);
/* Hash (16 hex digits) of the sources this binary was compiled from: csrc/ and include/ of this
 * repository, computed by build.py and baked in at compile time.  The Python host recomputes it
 * from the tree and rebuilds (or refuses to run) when the two differ, so a stale in-tree .so
 * cannot stand in for the sources.                                                             */
const char *ffq_build_id(void);
const char *ffq_last_error(void);
int         ffq_device_count(void);
/* device = HIP ordinal.  Fails (FFQ_E_NODEVICE) if it is not a gfx950 part. */
int         ffq_ctx_create(int device, ffq_ctx **out);
/* A second context on the same HIP streams as `parent` (own scratch): work submitted
 * through either context executes in submission order.  With ffq_scan_submit/_wait this
 * lets a host queue the next batch behind the current one (no idle GPU between batches).
 * Destroy it before the parent.                                                        */
int         ffq_ctx_create_shared(ffq_ctx *parent, ffq_ctx **out);
void        ffq_ctx_destroy(ffq_ctx *ctx);
/* Pre-size the per-context scratch (line index, group summaries) for buffers
 * of up to max_bytes so that no allocation happens inside a timed scan.     */
int         ffq_ctx_reserve(ffq_ctx *ctx, int64_t max_bytes);
/* A context remembers whether its recent input was plain four-line FASTQ and
 * then starts the following scans with the kernels that fit (results are the
 * same either way).  This forgets that history.                             */
void        ffq_ctx_forget(ffq_ctx *ctx);
/* The HIP stream all of this context's work is enqueued on (hipStream_t): a stream of the
 * context's own, created hipStreamNonBlocking -- it does NOT wait for the legacy default stream or
 * for any other stream.  Whatever fills d_buf (or clears the output buffers) on another stream must
 * be complete, or ordered in front of this stream by the caller (an event recorded there and waited
 * for here), before a scan is submitted.                                                           */
void       *ffq_ctx_stream(ffq_ctx *ctx);

/* ---- memory plumbing (so a ctypes host needs nothing else) ------------ */
int ffq_dev_alloc(ffq_ctx *ctx, int64_t bytes, void **dptr);
int ffq_dev_free(ffq_ctx *ctx, void *dptr);
int ffq_pinned_alloc(int64_t bytes, void **hptr);
int ffq_pinned_free(void *hptr);
int ffq_copy_h2d(ffq_ctx *ctx, void *dptr, const void *hptr, int64_t bytes, int async);
int ffq_copy_d2h(ffq_ctx *ctx, void *hptr, const void *dptr, int64_t bytes, int async);
int ffq_sync(ffq_ctx *ctx);

/* ---- the hot path ------------------------------------------------------
 * Record chain over one device-resident buffer.  Replaces the per-record
 * loop `status = entrypos(buf, offset, posbuffer); yield entryfunc(...)` of
 * readfastq_iter (fastqandfurious.py:251-279) with the C extension's scanner
 * semantics (_fastqandfurious.c:25-153) for every record of the buffer at
 * once.  d_buf must be 16-byte aligned.
 *
 *   offset      buffer coordinate where the first "\n@" search starts
 *   eof         nonzero: apply the iterator's end-of-stream rules (:256-270)
 *   d_table     device int64[table_cap][6]: pos0..pos5 per record, + add
 *   d_qual      device int8[qual_cap] or NULL: with FFQ_F_DECODE_QUAL the
 *               bytes buf[pos4:pos5] + qual_add of every record, packed
 *               (arrayadd_b, _fastqandfurious.c:161-185; usage
 *               doc/user-guide.rst:130-141)
 *   d_qoff      device int64[table_cap+1] or NULL: start of record i in d_qual
 * Blocks until the result is known.                                        */
int ffq_scan_device(ffq_ctx *ctx, const uint8_t *d_buf, int64_t n_bytes,
                    int sentinel, int64_t offset, int eof, int64_t add,
                    uint32_t flags, int qual_add,
                    int64_t *d_table, int64_t table_cap,
                    int8_t *d_qual, int64_t qual_cap, int64_t *d_qoff,
                    ffq_scan_result *res);

/* The same in two halves: ffq_scan_submit enqueues the kernels and returns at once;
 * ffq_scan_wait blocks, applies the fallbacks if the input needs them, and fills `res`.
 * One scan may be pending per context.                                                 */
int ffq_scan_submit(ffq_ctx *ctx, const uint8_t *d_buf, int64_t n_bytes,
                    int sentinel, int64_t offset, int eof, int64_t add,
                    uint32_t flags, int qual_add,
                    int64_t *d_table, int64_t table_cap,
                    int8_t *d_qual, int64_t qual_cap, int64_t *d_qoff);
int ffq_scan_wait(ffq_ctx *ctx, ffq_scan_result *res);

/* Same over a host buffer: pinned staging + hipMemcpyAsync in, kernels, rows
 * (and qualities) copied back.  h_table: int64[table_cap][6].              */
int ffq_scan_host(ffq_ctx *ctx, const uint8_t *h_buf, int64_t n_bytes,
                  int sentinel, int64_t offset, int eof, int64_t add,
                  uint32_t flags, int qual_add,
                  int64_t *h_table, int64_t table_cap,
                  int8_t *h_qual, int64_t qual_cap, int64_t *h_qoff,
                  ffq_scan_result *res);

/* One scanner call: entrypos(blob, offset, posbuffer) -> status
 * (_fastqandfurious.c:25-153).  Host buffers; pos receives 6 x int64.       */
int ffq_entrypos(ffq_ctx *ctx, const uint8_t *h_buf, int64_t len, int64_t offset,
                 int64_t *pos, int *status);

/* arrayadd_b(a, value): int8[i] += value in place, wrapping
 * (_fastqandfurious.c:161-185).  _device: a is device memory.               */
int ffq_arrayadd_b_device(ffq_ctx *ctx, int8_t *d_a, int64_t n, int value);
int ffq_arrayadd_b(ffq_ctx *ctx, int8_t *h_a, int64_t n, int value);
/* arrayadd_q(a, value): int64[i] += value in place, wrapping
 * (_fastqandfurious.c:193-217).                                             */
int ffq_arrayadd_q_device(ffq_ctx *ctx, int64_t *d_a, int64_t n, int64_t value);
int ffq_arrayadd_q(ffq_ctx *ctx, int64_t *h_a, int64_t n, int64_t value);

/* First row i of a device offset table (rows sorted by record) with
 * d_table[i][col] >= value; n_rows if none.  Used to cut a shard's rows out of
 * a table that also covers its halo (multi-GPU byte-range sharding).        */
int ffq_table_lower_bound(ffq_ctx *ctx, const int64_t *d_table, int64_t n_rows, int col,
                          int64_t value, int64_t *idx);

/* The rows of a byte-range shard inside the table of its scan: out = {i0, i1, pos0[i0],
 * pos0[i1], pos5[i0 - 1], pos5[i1 - 1]} with i0 / i1 the first rows whose pos0 (column 0) is
 * >= lo / >= hi (n_rows if none; the positions are -1 then), and pos5 of the rows right in
 * front of them (-1 if there is none): the record chain found row i searching from
 * pos5[i - 1] - 1 (fastqandfurious.py:254), which is where a neighbour shard re-enters the
 * chain.  One launch, one host wait.                                          */
int ffq_table_cut(ffq_ctx *ctx, const int64_t *d_table, int64_t n_rows, int64_t lo, int64_t hi,
                  int64_t out[6]);

/* Rows of a device offset table whose sequence length pos3 - pos2 lies in
 * [min_len, max_len], in order, written to d_out (n_rows rows of room; not
 * d_table itself); *n_out = rows kept.  The length filter of the reference's
 * user guide (doc/user-guide.rst:153-180) evaluated on the table, before any
 * per-record object exists: filtering reads is deleting rows (:199-204).    */
int ffq_table_select_seqlen(ffq_ctx *ctx, const int64_t *d_table, int64_t n_rows, int64_t min_len,
                            int64_t max_len, int64_t *d_out, int64_t *n_out);
/* ... with d_idx[i] = the ordinal in d_table of kept row i (n_rows entries of room; NULL: none): what a host
 * that owes one item per ORIGINAL row -- the guide's loop sees None for a dropped record (:166-170) -- puts the
 * kept ones back by.                                                                                       */
int ffq_table_select_seqlen_idx(ffq_ctx *ctx, const int64_t *d_table, int64_t n_rows, int64_t min_len,
                                int64_t max_len, int64_t *d_out, int64_t *d_idx, int64_t *n_out);

/* One component of every row of a device offset table as a packed stream + CSR offsets: byte j of
 * row i's component is d_out[d_off[i] + j] = buf[pos[col_begin] + begin_shift + j] + value_add,
 * up to pos[col_end].  This is what an entryfunc that builds only one part of an entry returns
 * (doc/user-guide.rst:153-180: buf[posarray[2]:posarray[3]]; the header as entryfunc cuts it,
 * fastqandfurious.py:161-171: col_begin 0, begin_shift 1, col_end 1), evaluated for the whole
 * table on the device; with columns 4 / 5 and value_add = -33 it is the Phred decode
 * (doc/user-guide.rst:206-214).  d_buf / n_bytes / sentinel / add: the buffer the rows were
 * scanned from and the `add` of that scan (rows - add are buffer coordinates).  d_off: n_rows + 1
 * entries.  *n_out_bytes = bytes of the stream; FFQ_E_TABLE_FULL if out_cap is smaller (nothing
 * is written past out_cap).  After ffq_table_select_seqlen this is the user guide's length-filter
 * entryfunc in two calls.                                                                     */
int ffq_table_gather_column(ffq_ctx *ctx, const uint8_t *d_buf, int64_t n_bytes, int sentinel, int64_t add,
                            const int64_t *d_table, int64_t n_rows, int col_begin, int begin_shift,
                            int col_end, int value_add, int8_t *d_out, int64_t out_cap, int64_t *d_off,
                            int64_t *n_out_bytes);

/* ---- FASTA (reference: the plug-in scanner entrypos_fasta, fastqandfurious.py:103-143) -------
 * Every COMPLETE entry of a buffer, i.e. the repeated scanner call with offset := pos[3]:
 * rows = pos0 ('>'), pos1 (header end), pos2, pos3 (the "\n" of the next "\n>") + add, -1, -1.
 * res->n_records = COMPLETE entries; res->last_status / last_pos = the last call (the entry
 * the end of the buffer cuts short: MISSING_SEQ_END with pos3 = end of the buffer, or an
 * earlier MISSING_* code; MISSING_SEQHEADER_BEGIN if there was no entry at all);
 * res->end_offset = the offset of that last call.  sentinel: a virtual "\n" in front.  d_buf and
 * d_table must be 16-byte aligned.  FFQ_E_TABLE_FULL (res->n_records = the rows needed): rows
 * [0, table_cap - 1) are complete; pos3 of row table_cap - 1 may be -1 (its end is the next row's start). */
int ffq_scan_fasta_device(ffq_ctx *ctx, const uint8_t *d_buf, int64_t n_bytes, int sentinel,
                          int64_t offset, int64_t add, int64_t *d_table, int64_t table_cap,
                          ffq_scan_result *res);
int ffq_scan_fasta_host(ffq_ctx *ctx, const uint8_t *h_buf, int64_t n_bytes, int sentinel,
                        int64_t offset, int64_t add, int64_t *h_table, int64_t table_cap,
                        ffq_scan_result *res);

/* ---- stream front end (reference: read(), fastqandfurious.py:30-36, and the refill loop of
 * readfastq_iter, :241-279, natively over a file descriptor) ---------------------------------
 * A three-stage pipeline: chunks of fbufsize bytes are read into pinned memory by a pool of
 * helper threads, copied to the device on a copy stream, and scanned; the read of chunk k+2, the
 * copy of chunk k+1 and the scan + row copy of fill k overlap.  ffq_stream_next returns the rows
 * of ONE fill: absolute stream offsets (what entryfunc_abspos yields, :186-195), in pinned memory
 * owned by the stream and valid until the next call, together with the fill's bytes
 * (h_bytes[i] is stream offset bytes_offset + i).  end_state: FFQ_END_REFILL while more
 * follows, FFQ_END_OK with the last fill, FFQ_END_ERR_* (+ err_offset, the byte the reference's
 * ValueError names) when the stream is malformed.  A call that fails leaves the stream failed
 * (every later call fails too).  The descriptor is not closed and, when it can seek, not moved
 * either (pread).                                                                                */
typedef struct ffq_stream ffq_stream;
int  ffq_stream_open(ffq_ctx *ctx, int fd, int64_t fbufsize, ffq_stream **out);
int  ffq_stream_next(ffq_stream *s, const int64_t **h_rows, int64_t *n_rows, int *end_state,
                     int64_t *err_offset, const uint8_t **h_bytes, int64_t *n_bytes,
                     int64_t *bytes_offset);
void ffq_stream_close(ffq_stream *s);
/* The same with options.  flags = FFQ_F_DECODE_QUAL: every fill's qualities are decoded on the
 * device (array('b').frombytes(buf[pos4:pos5]); arrayadd_b(q, qual_add), doc/user-guide.rst:130-141)
 * and ffq_stream_quals hands back the int8 bytes and their offsets (n_rows + 1 entries) of
 * the fill ffq_stream_next has just returned; pinned memory, valid until the next call: record i's
 * bytes are h_qual[h_qoff[i] : h_qoff[i] + pos5(i) - pos4(i)], h_qoff[n_rows] = *n_qual_bytes = where
 * the last record's end.  Packed back to back -- or, flags = FFQ_F_DECODE_QUAL | FFQ_F_SINGLE_PASS,
 * segmented (see FFQ_F_SINGLE_PASS above) whenever the single pass takes the fill (plain four-line
 * records: the fill is read once); ffq_stream_path = ffq_scan_result.path of the last fill (6 then).
 * start: byte of the file the stream begins at (< 0: the descriptor's current position); stream
 * offsets count from there.  ffq_stream_tell: the file position behind the last chunk HANDED OUT by
 * ffq_stream_next (the reader itself runs ahead of that): where a caller that shares the file object
 * should leave it, as the reference's loop does.  A descriptor that cannot seek (a pipe) is read
 * behind poll(): closing the stream never waits for a writer that keeps the pipe open and idle, and a
 * chunk is handed over short -- which is not the end of the stream -- when nothing more has arrived
 * for 50 ms.  A chunk comes in over the link in two halves on two copy streams (two copy engines);
 * a stream that decodes -- half as many bytes go back as come in -- uses one, the engines are
 * shared between the directions (FFQ_STREAM_ONE_COPY_STREAM = 0 / 1 forces either; FFQ_STREAM_PROF=1
 * prints where a stream's time went when it closes).                                               */
int  ffq_stream_open2(ffq_ctx *ctx, int fd, int64_t fbufsize, uint32_t flags, int qual_add, int64_t start,
                      ffq_stream **out);
int  ffq_stream_quals(ffq_stream *s, const int8_t **h_qual, const int64_t **h_qoff, int64_t *n_qual_bytes);
int64_t ffq_stream_tell(ffq_stream *s);
int  ffq_stream_path(ffq_stream *s);
/* Push-down into the stream -- the reference's user guide (doc/user-guide.rst:153-180): an entryfunc that looks at the
 * read's length and builds ONE component of the entries it keeps, `buf[posarray[2]:posarray[3]] if posarray[3] -
 * posarray[2] < LENGTH_THRESHOLD else None`, so that a dropped record costs (nearly) nothing.  ffq_stream_set_filter
 * (any kind of stream, before the first ffq_stream_next; not with FFQ_F_DECODE_QUAL): every fill's table is filtered on
 * the DEVICE before anything is copied back -- ffq_stream_next then returns the rows with min_seq_len <= pos3 - pos2 <=
 * max_seq_len only, in order -- and, column != FFQ_COL_NONE, that component of every kept row is gathered there into a
 * packed stream (header as entryfunc cuts it, fastqandfurious.py:161-171: buf[pos0 + 1 : pos1]; + value_add per byte:
 * -33 on the quality is the Phred decode of the kept records only).  ffq_stream_selected, for the fill just returned:
 * h_index[i] = the ordinal of kept row i among the n_scanned records of the fill (a caller that owes one item per
 * record -- the guide's loop sees None for a dropped one -- puts the kept ones back by it); bytes of kept row i =
 * h_col[h_coloff[i] : h_coloff[i + 1]].  Pinned memory of the stream, valid until the next call.                     */
#define FFQ_COL_NONE     0
#define FFQ_COL_HEADER   1
#define FFQ_COL_SEQUENCE 2
#define FFQ_COL_QUALITY  3
int  ffq_stream_set_filter(ffq_stream *s, int64_t min_seq_len, int64_t max_seq_len, int column, int value_add);
int  ffq_stream_selected(ffq_stream *s, const int64_t **h_index, int64_t *n_scanned, const int8_t **h_col,
                         const int64_t **h_coloff, int64_t *n_col_bytes);
/* The same over a gzip-compressed file (what FORMAT_OPENERS['gz'] / automagic_open hand to
 * readfastq_iter, fastqandfurious.py:282-334): the stream's reader thread inflates (zlib; concatenated
 * members, zero padding behind the last one) straight into the pinned chunk buffers -- decompression
 * is the feeder stage of the pipeline, fbufsize counts DECOMPRESSED bytes, stream offsets are offsets
 * of the decompressed stream.  start: byte of the compressed file the first member begins at.      */
int  ffq_stream_open_gzip(ffq_ctx *ctx, int fd, int64_t fbufsize, uint32_t flags, int qual_add, int64_t start,
                          ffq_stream **out);
/* Members that say how long they are -- the BGZF blocks bgzip writes ("BC" extra field, SAM specification
 * section 4.1) -- are located without inflating anything and inflated side by side by FFQ_GZ_THREADS threads
 * (default: the host's cores, at most 32; 1 = one member at a time), each straight into its place in the
 * chunk, by this build's own decoder (csrc/ffq_pgz.h; zlib for a member it does not take); every member is checked
 * against the length and CRC-32 of its trailer, and a file that is not what
 * its headers promise goes through the one-at-a-time inflate from that member on (same bytes or same error).
 *
 * ffq_gunzip_fd: that reader on its own, no device involved -- the file behind fd (from its current
 * position; a pipe works) inflated into h_dst in calls of `chunk` bytes as the stream's reader thread makes
 * them.  threads <= 0: the default.  Returns the number of bytes written, FFQ_E_TABLE_FULL if the file holds
 * more than cap, FFQ_E_ARG for corrupt input (ffq_last_error says what); *n_parallel_members (optional):
 * how many members were inflated side by side.                                                         */
int64_t ffq_gunzip_fd(int fd, uint8_t *h_dst, int64_t cap, int64_t chunk, int threads, int64_t *n_parallel_members);
/* A member that does NOT say how long it is -- what plain `gzip` writes: one deflate stream -- is inflated by
 * the same threads when the file is a regular one (csrc/ffq_pgz.h): the compressed bytes are cut into chunks
 * (FFQ_PGZ_CHUNK, default 1 MiB; two per thread and batch), every chunk but the first of a batch enters the
 * stream at the first bit that parses as a dynamic-Huffman block header and inflates into 16-bit symbols
 * (a literal, or a reference into the 32 KiB it was not given), a chunk ends at the first accepted block
 * boundary behind its range, the stitch step takes a chunk if it starts at the bit its predecessor ended at,
 * and the symbols become bytes -- and a CRC-32, checked against the member's trailer -- side by side.  zlib
 * keeps the last word: whatever the engine does not take (a decode error, a block longer than a batch, the
 * end of the file inside a block) makes it stop at the last block boundary it committed, and the serial
 * inflate goes on from that bit (same bytes or the same error).  Members of less than FFQ_PGZ_MIN (4 MiB)
 * compressed bytes, pipes and FFQ_GZ_THREADS=1 go through zlib alone.
 * ffq_gunzip_stats: process-wide counters of that engine since the library was loaded --
 * out[0] batches, [1] chunks taken, [2] chunks not taken, [3] times it gave up, [4] members finished.    */
void ffq_gunzip_stats(int64_t out[5]);
/* A byte RANGE of a BGZF file (bgzip's format: gzip members of at most 64 KiB of data that say how long they are) on its
 * own, host only: the members whose FIRST byte lies in [c_lo, c_hi) of the compressed file -- the first one found by
 * its signature and proven by the chain of headers behind it, nothing inflated to find it -- inflated side by side
 * (threads; 0 = as ffq_gunzip_fd) into h_dst.  h_dst = NULL, cap = 0: nothing is inflated, *n_out = the bytes the
 * members' trailers promise (the pass that sizes the buffer and, summed over the ranks that share a file, gives the cut
 * points of the uncompressed stream).  *c_first / *c_end: file offsets of the first member taken and behind the last
 * (rank r's c_end is rank r + 1's c_first when their shares meet).  FFQ_E_ARG with "gzip: " in the text: not BGZF, cut
 * short, or a member that does not inflate to its trailer's length and CRC-32; FFQ_E_TABLE_FULL: more than cap bytes.
 * What it stands for: the reference's loop over gzip.open(...) (fastqandfurious.py:241-279), per rank of a file that
 * several GPUs read together (sharded.BgzfFileShard).                                                              */
int  ffq_bgzf_range(int fd, int64_t c_lo, int64_t c_hi, uint8_t *h_dst, int64_t cap, int threads,
                    int64_t *c_first, int64_t *c_end, int64_t *n_out, int64_t *n_members);
/* The same without a reader thread, for sources only the host can read (any object with a read() /
 * readinto(): BytesIO, bz2, lzma, sockets ...): the host asks where the next chunk goes
 * (ffq_stream_push_buffer: pinned memory, *cap = fbufsize bytes of room), writes up to *cap bytes
 * there itself and says how many it wrote and whether the source is exhausted (ffq_stream_push: the
 * reference's read(), fastqandfurious.py:30-36, with eof decided by the caller); ffq_stream_next then
 * scans that fill.  Up to two chunks may be pushed ahead of the one being consumed.               */
int  ffq_stream_open_push(ffq_ctx *ctx, int64_t fbufsize, uint32_t flags, int qual_add, ffq_stream **out);
int  ffq_stream_push_buffer(ffq_stream *s, uint8_t **dst, int64_t *cap);
int  ffq_stream_push(ffq_stream *s, int64_t n, int eof);

/* Bytes [pos, pos + n_bytes) of the file behind fd into device memory (read(), fastqandfurious.py:30-36, for a range
 * that stays resident): pread in slices by the context's helper threads into pinned slots, each slot over the link in
 * two halves on two copy streams while the next is read.  Returns when the bytes are in d_dst; *n_loaded < n_bytes:
 * the file ended there.  The descriptor's position is not moved.  The helper threads (FFQ_POOL_THREADS, default 16) run
 * on the CPUs next to the GPU (/sys/bus/pci/devices/<bdf>/local_cpulist within the process's own affinity) and the
 * pinned slots are allocated from there: on a two-socket host the other socket costs a fifth of the rate
 * (FFQ_POOL_AFFINITY=0: nothing is bound; FFQ_POOL_DEBUG=1 says what was done).                                   */
int  ffq_load_fd(ffq_ctx *ctx, int fd, int64_t pos, int64_t n_bytes, void *d_dst, int64_t *n_loaded);

/* ---- synthetic FASTQ generators (bench / test inputs) -----
 * Counter-based (splitmix64), so the numpy generator in
 * fastq-and-furious_amd/synth.py produces the same bytes.
 * S-single: record i = "@SYN.%010d/1\n" + 150 bases + "\n+\n" + 150 quals +
 * "\n" = 322 bytes.  Writes records [first, first+count) at d_out.          */
int ffq_synth_single(ffq_ctx *ctx, uint8_t *d_out, int64_t first, int64_t count,
                     uint64_t seed);
/* S-wrapped: lengths 50..300, 80-column wrap, '+' line repeats the header
 * for one record in four.  d_start[i] (int64, count+1 entries, relative to
 * d_out) must hold the exclusive prefix sum of ffq_synth_wrapped_size().     */
int64_t ffq_synth_wrapped_size(int64_t i, uint64_t seed);
int ffq_synth_wrapped(ffq_ctx *ctx, uint8_t *d_out, const int64_t *d_start,
                      int64_t first, int64_t count, uint64_t seed);

/* ---- diagnostics ---------------------------------------------------------
 * Runs the device self-checks (wave scan, newline mask) and returns FFQ_OK.  */
int ffq_selftest(ffq_ctx *ctx);

/* ---- one stream over the GPUs of a node: byte-range shards ------------------------------------
 * Rank r of `world` owns the stream bytes [bounds[r], bounds[r + 1]) and every record whose '@' lies in
 * them; offsets count from bounds[0] (readfastq_iter's globaloffset, /root/reference/src/fastqandfurious.py
 * :198-242).  What the reference does with a record that does not fit its buffer -- keep buf[offset:] and
 * read more (:274-279) -- happens per range edge.  One process per GPU, one ffq_shard per process; the
 * hand-offs are RCCL over xGMI (librccl, loaded at run time):
 *   ffq_shard_unique_id   rank 0 draws the communicator id (ncclGetUniqueId); the host hands the 128 bytes
 *                         to every rank (MPI, torch.distributed, a file ...)
 *   ffq_shard_create      every rank, collectively (ncclCommInitRank, twice: hand-offs and the gather have
 *                         a communicator each)
 *   ffq_shard_halo        how many bytes in front of / behind its range this rank's buffer holds:
 *                         d_ext = [tail | own bytes | head], tail = min(tail_bytes, lo - bounds[0]),
 *                         head = min(head_bytes, bounds[world] - hi); the caller fills the middle
 *   ffq_shard_step_submit one step, queued, no host wait: the halos from the ranks that own them
 *                         (ncclSend / ncclRecv in one group; overlap_handoff != 0: on the shard's hand-off
 *                         stream, beside whatever the scan stream is doing -- the caller vouches that no
 *                         scan in flight reads d_ext, e.g. every lane has a buffer of its own), the scan of
 *                         the whole view, the rows of the range cut out (as ffq_table_cut) and ONE
 *                         ncclAllGather of eight words per rank
 *   ffq_shard_step_wait   the step's result: rows [row_lo, row_hi) of d_table are this rank's records
 *                         (absolute stream offsets), record_base their global ordinal; exit_pos (first
 *                         record start at / behind my right edge) equals the right neighbour's first_pos
 *                         on return -- that, with rank 0's exact start, proves every range.  A look-ahead
 *                         that ends inside the straddling record is grown (d_ext in the result then points
 *                         at the shard's own, larger view), a contradicted entry guess re-entered from the
 *                         left neighbour's exit: `rounds` counts those.  err_state != 0: the stream's error
 *                         (FFQ_END_ERR_*, the byte the reference's ValueError names in err_byte), the same
 *                         on every rank.  FFQ_E_TABLE_FULL (on every rank) if some rank's table is too small
 *                         (scan.n_records = the rows that rank's view holds) or, decoding, its quality buffer
 *                         (scan.n_records = 0, scan.n_qual_bytes = the bytes needed).
 *   ffq_shard_create_lane a second shard of the same rank on another context (ffq_ctx_create_shared: same
 *                         scan stream), on the first one's communicators: steps queued one ahead.
 *   ffq_shard_world_* / ffq_shard_create_local   k logical ranks as THREADS of one process on one GPU, hand-offs
 *                         by device copies (tests, dry runs): every rank's thread makes the same calls.
 * Every rank must make the same sequence of calls (they are collective).
 *
 * A step cannot hang silently (round 6).  WATCHDOG: every wait of a step -- ffq_shard_step_wait, the in-process
 * barrier -- polls with a deadline (FFQ_SHARD_TIMEOUT_S, default 30 s, 0 = none; ffq_shard_set_timeout); when it
 * runs out the call returns FFQ_E_TIMEOUT and ffq_last_error names the STAGE the step is stuck in (hand-off / scan /
 * gather), the transport, the mode and -- as far as this rank can see them -- the ranks whose words have and have not
 * arrived; RCCL's own asynchronous errors (ncclCommGetAsyncError) end the wait at once.  After FFQ_E_TIMEOUT the shard is
 * poisoned: ffq_shard_abort (ncclCommAbort on its communicators, its streams drained) and ffq_shard_destroy are what is
 * left; the host may then build a new one.  SERIAL MODE: the pipelined step drives TWO communicators from THREE streams
 * (hand-off of step i + 1 beside the scan of step i, the gather behind it); the serial step is ONE communicator on ONE
 * stream -- hand-off, scan, words, gather in order on the scan stream (lanes share it) -- the same rows, nothing
 * overlapped: the fallback a host takes once after a watchdog trip (FFQ_SHARD_SERIAL=1 at creation -- the second
 * communicator is then never made --, or ffq_shard_set_serial between steps).  ffq_shard_result.serial / .nranks and
 * ffq_shard_get_info say what ran.                                                                            */
#define FFQ_SHARD_STAGE_NONE    0
#define FFQ_SHARD_STAGE_HANDOFF 1
#define FFQ_SHARD_STAGE_SCAN    2
#define FFQ_SHARD_STAGE_GATHER  3
#define FFQ_SHARD_MAX_INFO_RANKS 64
typedef struct ffq_shard ffq_shard;
typedef struct ffq_shard_world ffq_shard_world;
typedef struct ffq_shard_result {
    ffq_scan_result scan;      /* the local scan of [tail | own | head]                                    */
    int64_t n_rows;            /* rows it wrote to d_table                                                  */
    int64_t row_lo, row_hi;    /* this rank's records                                                       */
    int64_t exit_pos, first_pos;   /* stream offsets; -1: none (the view reaches the end of the stream)     */
    int64_t n_own_records, record_base, total_records;
    int64_t err_byte;
    int32_t err_state;         /* 0, or FFQ_END_ERR_*                                                       */
    int32_t rounds;            /* repair rounds (0: the first scan of every rank stood)                     */
    int32_t regathers;         /* gathers repeated because some rank's scan needed a later tier             */
    int32_t halo_source;       /* 0: the halos were handed off between ranks; 1: read from the file (ffq_shard_load_fd) */
    int64_t handoff_bytes;     /* bytes this rank sent + received in hand-offs                              */
    float   handoff_ms;        /* device time of the halo hand-off (events on the stream it ran on)         */
    float   allgather_ms;      /* device time of the gather(s) of the eight words                           */
    uint8_t *d_ext;            /* the view the rows refer to (the caller's, or a grown one owned by the shard) */
    int64_t tail, head;
    int32_t nranks;            /* ranks of the communicator the words were gathered on (ncclCommCount; == world) */
    int32_t serial;            /* 1: the serial step ran (one communicator, one stream), 0: the pipelined one   */
    int64_t n_slabs;           /* ffq_shard_scan_fd_slabs: slabs the range went through (re-reads included); else 0 */
    int64_t bytes_read;        /* ... and the bytes it read from the file                                        */
} ffq_shard_result;
/* Who is there: filled at creation (RCCL: ncclCommCount of both communicators and ONE all-gather of every rank's PCI
 * bus id, so that a host can assert "N ranks on N distinct GPUs" before it trusts a number).                     */
typedef struct ffq_shard_info {
    int32_t rank, world;
    int32_t nranks_handoff;    /* ncclCommCount of the hand-off communicator (other transports: world)          */
    int32_t nranks_gather;     /* ... of the gather communicator; 0: there is none (created serial)             */
    int32_t serial;            /* the mode the next step runs in                                                */
    int32_t last_stage;        /* FFQ_SHARD_STAGE_*: where the last watchdog trip found the step (0: no trip)   */
    int32_t poisoned;          /* 1: a watchdog trip or an asynchronous RCCL error: only abort / destroy are left */
    int32_t n_bus;             /* entries of bus_id that are filled: min(world, FFQ_SHARD_MAX_INFO_RANKS)       */
    double  timeout_s;         /* the watchdog's deadline (0: none)                                             */
    int64_t bus_id[FFQ_SHARD_MAX_INFO_RANKS];   /* PCI domain << 16 | bus << 8 | device << 3 | function of rank r's GPU; -1: unknown
                                                   (in-process / hosted transports know their own only)        */
} ffq_shard_info;
int  ffq_shard_unique_id(uint8_t *id128);
int  ffq_shard_create(ffq_ctx *ctx, const uint8_t *id128, int rank, int world, const int64_t *bounds,
                      int64_t tail_bytes, int64_t head_bytes, ffq_shard **out);
/* ... with the mode said by the caller instead of the environment (mode: FFQ_SHARD_F_*) */
#define FFQ_SHARD_F_SERIAL 1u
int  ffq_shard_create2(ffq_ctx *ctx, const uint8_t *id128, int rank, int world, const int64_t *bounds,
                       int64_t tail_bytes, int64_t head_bytes, uint32_t mode, ffq_shard **out);
int  ffq_shard_create_lane(ffq_shard *parent, ffq_ctx *ctx, ffq_shard **out);
int  ffq_shard_world_create(int world, ffq_shard_world **out);
void ffq_shard_world_abort(ffq_shard_world *w);
void ffq_shard_world_destroy(ffq_shard_world *w);
int  ffq_shard_create_local(ffq_ctx *ctx, ffq_shard_world *w, int rank, const int64_t *bounds, int64_t tail_bytes,
                            int64_t head_bytes, ffq_shard **out);
void ffq_shard_destroy(ffq_shard *s);
int  ffq_shard_halo(ffq_shard *s, int64_t *tail, int64_t *head);
int  ffq_shard_exchange_halo(ffq_shard *s, uint8_t *d_ext, int overlap);
int  ffq_shard_step_submit(ffq_shard *s, uint8_t *d_ext, int overlap_handoff, uint32_t flags, int qual_add,
                           int64_t *d_table, int64_t table_cap, int8_t *d_qual, int64_t qual_cap, int64_t *d_qoff);
int  ffq_shard_step_wait(ffq_shard *s, ffq_shard_result *out);
const char *ffq_shard_transport(ffq_shard *s);    /* "rccl" / "in-process" / "hosted" */
int  ffq_shard_get_info(ffq_shard *s, ffq_shard_info *out);
int  ffq_shard_set_timeout(ffq_shard *s, double seconds);      /* the watchdog's deadline (every lane of the rank) */
int  ffq_shard_set_serial(ffq_shard *s, int on);               /* between steps, no lane pending; every rank alike */
/* After FFQ_E_TIMEOUT (or at any time): ncclCommAbort on the shard's communicators -- kernels of a collective that
 * waits for a peer leave --, an injected stall released, the shard's streams drained with a deadline of their own.
 * FFQ_OK: drained; FFQ_E_TIMEOUT: something still runs (ffq_shard_destroy then leaks the shard's device memory
 * rather than wait for ever) -- in particular ncclCommAbort ITSELF, which runs on a thread of its own with a deadline
 * (FFQ_SHARD_ABORT_S, default 30 s) here: with a peer whose process is gone RCCL's teardown waits on its sockets for
 * minutes and holds the device's runtime meanwhile (every other HIP call of the process waits behind it); the caller
 * has its error in time and should leave with _exit.  The same wait meets a rank that aborts AFTER its peers have
 * torn their ends down: the ranks' deadlines do not run out at the same instant, so a host lets them MEET over its
 * own control plane first and abort together (sharded.abort_together: a barrier on the process group, then this
 * call; a peer that never reports fails the meeting and nothing is aborted).  Only ffq_shard_destroy may follow. */
int  ffq_shard_abort(ffq_shard *s);
/* diagnostics: the NEXT step of this shard hangs at `stage` (FFQ_SHARD_STAGE_*) for up to `seconds` -- a one-lane
 * kernel on that stage's stream that waits for a host flag (released by ffq_shard_abort / _destroy) -- so that the
 * watchdog and the recovery can be exercised without a broken peer (tests/test_watchdog.py).                   */
int  ffq_shard_inject_stall(ffq_shard *s, int stage, double seconds);
/* A shard of a FILE: bounds[] are file offsets.  ffq_shard_load_fd reads this rank's bytes [lo - tail, hi + head) of fd
 * (pread: the descriptor's position is not moved) into d_ext -- helper threads -> pinned slots -> hipMemcpyAsync on two
 * copy streams -- and returns when they are there; a step submitted over that d_ext then hands off NOTHING between
 * ranks (the halos are the file's own bytes: ffq_shard_result.halo_source = 1), and a look-ahead that has to grow is
 * read from the file by the rank that needs it.  The gather of the eight words, and with it the proof of every range
 * and the global ordinals, is unchanged.  Every rank of the world or none (the ranks must agree on whether a grown
 * look-ahead is an exchange).  fd < 0: detach (steps hand off between ranks again).  This is what replaces the
 * reference's single reader per rank: read() (fastqandfurious.py:30-36), the first fill and its sentinel (:241-245,
 * rank 0's view starts the stream), the carry of an unfinished entry (:274-279, here the look-ahead).           */
int  ffq_shard_load_fd(ffq_shard *s, int fd, uint8_t *d_ext, int64_t *n_bytes);
/* ... and a range that does NOT fit the GPU (a 2 TB file over eight of them): the same step with this rank's view going
 * through ONE device buffer of slab_bytes, slab after slab -- what the reference's loop does for any size of stream: scan
 * the buffer, keep the unfinished entry, read more (fastqandfurious.py:251-279).  Slab k + 1 begins at the byte the search
 * of slab k stopped at (the iterator's `offset`: exact, no guess between slabs), its rows go behind slab k's in d_table
 * (rows [0, n_rows): all of the pass, [row_lo, row_hi) this rank's, absolute file offsets), the bytes are dropped.  At the
 * rank's two edges nothing changes: the entry guessed from the run-in, eight words, one gather, the same decision; a rank
 * whose guess its left neighbour's chain contradicts streams its range again from that neighbour's exit; a look-ahead that
 * must grow is read on; one record longer than the slab doubles the slab.  Collective like a step (every rank calls it, or
 * ffq_shard_step_* over a loaded range: the words are the same).  No FFQ_F_DECODE_QUAL.  FFQ_E_TABLE_FULL (every rank):
 * scan.n_records = an estimate of the rows that rank's view needs.  d_ext of the result is NULL: nothing stays resident. */
int  ffq_shard_scan_fd_slabs(ffq_shard *s, int fd, int64_t slab_bytes, uint32_t flags, int64_t *d_table, int64_t table_cap,
                             ffq_shard_result *out);
/* The same step over HOST memory with the transport supplied by the caller (and, for tests, the scan): the protocol
 * functions are the device step's (csrc/ffq_shard_proto.h), driven synchronously.  For hosts whose ranks cannot talk RCCL
 * -- a multi-process CPU test-suite over gloo (scan = the test's own engine), a functional dry run of several ranks on
 * one GPU (scan = NULL: ffq_scan_host on `ctx`) -- not a CPU fallback: without a scan callback it needs a context like
 * every other compute entry point.
 *   exchange   collective: the same list of pieces on every rank; for a piece with src == rank, ptr is where its b - a
 *              bytes are read from, with dst == rank where they go (NULL on ranks that are neither)
 *   allgather  collective: SH words (8 x int64) of every rank into all[world][8]
 *   scan       ffq_scan_host's contract over h_buf (rows + add into h_table, res filled; FFQ_E_TABLE_FULL with
 *              res->n_records = the need)
 * h_ext = [tail | own bytes | head] with the middle filled (ffq_shard_halo's sizes from the same bounds and byte counts);
 * on return out->d_ext is the view the rows refer to: h_ext, or a larger one the step allocated (a look-ahead had to
 * grow) that the caller releases with ffq_shard_host_free.  Callbacks return 0 or a negative FFQ_E_* code.          */
typedef struct ffq_shard_piece { int32_t src, dst; int64_t a, b; uint8_t *ptr; } ffq_shard_piece;
typedef struct ffq_shard_host_ops {
    void *user;
    int (*scan)(void *user, const uint8_t *h_buf, int64_t n_bytes, int sentinel, int64_t offset, int eof, int64_t add,
                int64_t *h_table, int64_t table_cap, ffq_scan_result *res);
    int (*exchange)(void *user, const ffq_shard_piece *pieces, int n_pieces);
    int (*allgather)(void *user, const int64_t *mine, int64_t *all);
} ffq_shard_host_ops;
/* ... and the DEVICE step (ffq_shard_step_*: buffers in HBM, the scan and the words on the device) over the caller's
 * transport: hand-offs staged through host memory around ops->exchange, the words gathered by ops->allgather (ops->scan is
 * not used; the callbacks and `user` must stay valid while the shard lives).  ffq_shard_transport says "hosted".  For ranks
 * that cannot talk RCCL -- several processes sharing ONE GPU, a group over gloo / MPI; with file-backed shards
 * (ffq_shard_load_fd) the only traffic is the 64 bytes of words per rank.                                            */
int  ffq_shard_create_hosted(ffq_ctx *ctx, const ffq_shard_host_ops *ops, int rank, int world, const int64_t *bounds,
                             int64_t tail_bytes, int64_t head_bytes, ffq_shard **out);
int  ffq_shard_host_step(const ffq_shard_host_ops *ops, ffq_ctx *ctx, int rank, int world, const int64_t *bounds,
                         int64_t tail_bytes, int64_t head_bytes, uint8_t *h_ext, int64_t *h_table, int64_t table_cap,
                         ffq_shard_result *out);
void ffq_shard_host_free(void *p);
/* diagnostics: n bytes from d_src to d_dst through the shard's transport with this rank at both ends */
int  ffq_shard_self_exchange(ffq_shard *s, const uint8_t *d_src, uint8_t *d_dst, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* FFQ_H */
