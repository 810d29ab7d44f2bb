/*
 * ffq_oracle.c -- CPU restatement of the reference FASTQ buffer-scan path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The product path (fastq-and-furious_amd/) never imports, links or executes
 * anything from here.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function below
 * against (a) the reference's own C extension compiled from
 * /root/reference/src/_fastqandfurious.c into oracle/_ref/ (when present) and
 * (b) the committed golden vectors under tests/golden/ that were captured from
 * the reference (tests/golden/make_golden.py).
 *
 * Each function cites the reference file:line it restates.  Where the
 * reference has undefined behaviour the oracle picks a defined result and says
 * so; none of those cases is reachable from well-formed input.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stddef.h>
#include <string.h>

/* status codes: /root/reference/src/_fastqandfurious.c:7-15,
 *               /root/reference/src/fastqandfurious.py:19-27            */
#define FFQ_INVALID                -1
#define FFQ_MISSING_SEQHEADER_BEGIN 0
#define FFQ_MISSING_SEQHEADER_END   1
#define FFQ_MISSING_SEQ_BEG         2
#define FFQ_MISSING_SEQ_END         3
#define FFQ_MISSING_QUAL_BEGIN      4
#define FFQ_MISSING_QUAL_END        5
#define FFQ_COMPLETE                6
#define FFQ_MISSING_QUALHEADER_END  7

/* outcome codes of ffq_oracle_scan (the iterator's end states,
 * /root/reference/src/fastqandfurious.py:256-279)                        */
#define FFQ_END_OK                  0   /* eof reached cleanly                          */
#define FFQ_END_REFILL              1   /* !eof and entry incomplete: caller refills    */
#define FFQ_END_ERR_FINAL_QUAL      2   /* 'Incomplete final quality string at byte'    */
#define FFQ_END_ERR_INCOMPLETE      3   /* 'Incomplete entry at byte %i'                */
#define FFQ_END_ERR_INVALID         4   /* 'Entry is invalid at byte %i'                */
#define FFQ_END_TABLE_FULL          5   /* caller's table capacity exhausted            */

/* first index >= from of the two-byte needle {c0,c1} in b[0,len); -1 if none.
 * Stands for memmem(blob+from, len-from, needle, 2) (_fastqandfurious.c:62,87)
 * and bytes.find(needle, from) (fastqandfurious.py:51,66).                 */
static int64_t find2(const uint8_t *b, int64_t len, int64_t from,
                     uint8_t c0, uint8_t c1)
{
    if (from < 0) from = 0;
    while (from + 1 < len) {
        const uint8_t *p = (const uint8_t *)memchr(b + from, c0, (size_t)(len - 1 - from));
        if (p == NULL) return -1;
        if (p[1] == c1) return (int64_t)(p - b);
        from = (int64_t)(p - b) + 1;
    }
    return -1;
}

/* first index in [from, to) holding c; -1 if none or the range is empty.   */
static int64_t find1(const uint8_t *b, int64_t from, int64_t to, uint8_t c)
{
    if (from < 0) from = 0;
    if (to <= from) return -1;
    const uint8_t *p = (const uint8_t *)memchr(b + from, c, (size_t)(to - from));
    return p ? (int64_t)(p - b) : -1;
}

/* Buffer coordinates: the scanners see a buffer of `len` bytes whose byte at
 * coordinate c is d[c - s].  s = 0: d IS the buffer.  s = 1: the buffer is
 * b'\n' + d (the iterator's sentinel, fastqandfurious.py:245) without the copy;
 * coordinate 0 is then only ever touched by the first "\n@" search.        */
static int64_t F1(const uint8_t *d, int64_t s, int64_t from, int64_t to, uint8_t c)
{
    int64_t r = find1(d, from - s, to - s, c);
    return r < 0 ? -1 : r + s;
}
static int64_t F2(const uint8_t *d, int64_t s, int64_t len, int64_t from,
                  uint8_t c0, uint8_t c1)
{
    if (from > len) return -1;
    if (s && from <= 0) {
        /* the sentinel itself can only match as the '\n' of "\n@" / "\n+" */
        if (len >= 2 && c0 == '\n' && d[0] == c1) return 0;
        from = 1;
    }
    int64_t r = find2(d, len - s, from - s, c0, c1);
    return r < 0 ? -1 : r + s;
}

/*
 * C-extension scanner: /root/reference/src/_fastqandfurious.c:25-153.
 *   - pos[0..5] reset to -1 first (:57-59)
 *   - both memchr calls exclude the LAST byte of the buffer (:70-71, :102-103)
 *   - the "\n+" search starts at seq_beg+1 (:87-88)
 *   - '+'-line LENGTH rule -> INVALID (:109-117)
 *   - the final validity expression (:138-149) can never fail: omitted
 * Defined where the reference is undefined: when the buffer ends right after
 * "\n@" the reference calls memchr with length (size_t)-1 (:70-71); the oracle
 * returns MISSING_SEQHEADER_END.
 */
static int entrypos_c(const uint8_t *d, int64_t s, int64_t len, int64_t offset, int64_t *pos)
{
    for (int i = 0; i < 6; i++) pos[i] = -1;
    int64_t h = F2(d, s, len, offset, '\n', '@');
    if (h < 0) return FFQ_MISSING_SEQHEADER_BEGIN;
    pos[0] = h + 1;
    int64_t he = F1(d, s, pos[0] + 1, len - 1, '\n');
    if (he < 0) return FFQ_MISSING_SEQHEADER_END;
    pos[1] = he;
    if (he + 1 >= len) return FFQ_MISSING_SEQ_BEG;
    pos[2] = he + 1;
    int64_t se = F2(d, s, len, pos[2] + 1, '\n', '+');
    if (se < 0) return FFQ_MISSING_SEQ_END;
    pos[3] = se;
    if (se + 2 >= len) return FFQ_MISSING_QUALHEADER_END;
    int64_t qhe = F1(d, s, se + 2, len - 1, '\n');
    if (qhe < 0) return FFQ_MISSING_QUALHEADER_END;
    if ((qhe - se - 1 > 1) && (qhe - se != he - pos[0] + 1)) return FFQ_INVALID;
    int64_t qb = qhe + 1;
    if (qb >= len) return FFQ_MISSING_QUAL_BEGIN;
    pos[4] = qb;
    int64_t qe = qb + se - he - 1;
    if (qe + 2 >= len) return FFQ_MISSING_QUAL_END;
    pos[5] = qe;
    return FFQ_COMPLETE;
}

/*
 * Pure-Python scanner: /root/reference/src/fastqandfurious.py:39-100.
 * Differs from the C one in: pos[] is NOT reset; both newline searches reach
 * the last byte (:56, :73); the "\n+" search starts at seq_beg (:66) so an
 * empty read parses; a buffer ending right after the '+' line gives
 * MISSING_QUAL_BEGIN (:80-81) where C gives MISSING_QUALHEADER_END.
 */
static int entrypos_py(const uint8_t *d, int64_t s, int64_t len, int64_t offset, int64_t *pos)
{
    int64_t h = F2(d, s, len, offset, '\n', '@');
    if (h < 0) return FFQ_MISSING_SEQHEADER_BEGIN;
    pos[0] = h + 1;
    int64_t he = F1(d, s, h + 2, len, '\n');
    if (he < 0) return FFQ_MISSING_SEQHEADER_END;
    pos[1] = he;
    if (he + 1 >= len) return FFQ_MISSING_SEQ_BEG;
    pos[2] = he + 1;
    int64_t se = F2(d, s, len, he + 1, '\n', '+');
    if (se < 0) return FFQ_MISSING_SEQ_END;
    pos[3] = se;
    int64_t qhe = F1(d, s, se + 2, len, '\n');
    if (qhe < 0) return FFQ_MISSING_QUALHEADER_END;
    if ((qhe - se - 1 > 1) && (qhe - se != he - h)) return FFQ_INVALID;
    int64_t qb = qhe + 1;
    if (qb >= len) return FFQ_MISSING_QUAL_BEGIN;
    pos[4] = qb;
    int64_t qe = qb + se - he - 1;
    if (qe + 2 >= len) return FFQ_MISSING_QUAL_END;
    pos[5] = qe;
    return FFQ_COMPLETE;
}

int ffq_oracle_entrypos_c(const uint8_t *b, int64_t len, int64_t offset, int64_t *pos)
{
    return entrypos_c(b, 0, len, offset, pos);
}

int ffq_oracle_entrypos_py(const uint8_t *b, int64_t len, int64_t offset, int64_t *pos)
{
    return entrypos_py(b, 0, len, offset, pos);
}

/*
 * The record chain of readfastq_iter over ONE buffer:
 * /root/reference/src/fastqandfurious.py:251-279.
 *
 *   d, n_bytes the bytes; sentinel = 0: they ARE the buffer entrypos sees;
 *              sentinel = 1: the buffer is b'\n' + d (:245) and len = n_bytes+1
 *   offset     buffer coordinate where the first search starts (:243, :254)
 *   eof        nonzero when no more data can follow (:256)
 *   variant    0 = C scanner, 1 = Python scanner
 *   add        added to every emitted position (entryfunc_abspos' globaloffset,
 *              :186-195; the iterator starts it at -1, :242)
 *   table      out, cap rows of 6 x int64
 *   out[0]     number of rows written
 *   out[1]     end state (FFQ_END_*)
 *   out[2]     status of the last entrypos call
 *   out[3]     `offset` (buffer coordinate) at the end: refill keeps
 *              buf[offset:] (:277); the messages print globaloffset+offset
 *              (:269, :272)
 *
 * Results of readfastq_iter do not depend on fbufsize on any input it
 * terminates on, so one call with eof=1 over the whole file equals the
 * iterator at every fbufsize (checked in tests/golden/make_golden.py).
 *
 * Deviation (documented): INVALID at eof makes the reference loop forever
 * (:256-270 take no branch); the oracle reports FFQ_END_ERR_INVALID.
 */
void ffq_oracle_scan(const uint8_t *d, int64_t n_bytes, int sentinel, int64_t offset,
                     int eof, int variant, int64_t add,
                     int64_t *table, int64_t cap, int64_t *out)
{
    const int64_t s = sentinel ? 1 : 0;
    const int64_t len = n_bytes + s;
    int64_t n = 0;
    int64_t pos[6] = {-1, -1, -1, -1, -1, -1};
    int status;
    int end;
    for (;;) {
        status = variant ? entrypos_py(d, s, len, offset, pos)
                         : entrypos_c(d, s, len, offset, pos);
        if (status == FFQ_COMPLETE) {
            if (n >= cap) { end = FFQ_END_TABLE_FULL; break; }
            for (int i = 0; i < 6; i++) table[6 * n + i] = pos[i] + add;
            n++;
            offset = pos[5] - 1;
            continue;
        }
        if (eof) {
            if (status == FFQ_MISSING_SEQHEADER_BEGIN) {
                end = FFQ_END_OK;
            } else if (status == FFQ_MISSING_QUAL_END) {
                int64_t qualend = pos[4] + (pos[3] - pos[2]);
                if (qualend >= len) {
                    end = FFQ_END_ERR_FINAL_QUAL;
                } else if (n >= cap) {
                    end = FFQ_END_TABLE_FULL;
                } else {
                    pos[5] = qualend;
                    for (int i = 0; i < 6; i++) table[6 * n + i] = pos[i] + add;
                    n++;
                    end = FFQ_END_OK;
                }
            } else if (status != FFQ_INVALID) {
                end = FFQ_END_ERR_INCOMPLETE;
            } else {
                end = FFQ_END_ERR_INVALID;
            }
        } else if (status == FFQ_INVALID) {
            end = FFQ_END_ERR_INVALID;
        } else {
            end = FFQ_END_REFILL;
        }
        break;
    }
    out[0] = n;
    out[1] = end;
    out[2] = status;
    out[3] = offset;
}

/* /root/reference/src/_fastqandfurious.c:161-185: int8[i] += value, in place,
 * two's-complement wrap.  The reference parses `value` with format 'h' into a
 * signed char (:165-167), i.e. the value is taken modulo 256.              */
void ffq_oracle_arrayadd_b(int8_t *a, int64_t n, int value)
{
    const uint8_t v = (uint8_t)value;
    uint8_t *u = (uint8_t *)a;
    for (int64_t i = 0; i < n; i++) u[i] = (uint8_t)(u[i] + v);
}

/* /root/reference/src/_fastqandfurious.c:193-217: int64[i] += value in place,
 * computed on unsigned long long (wraps).                                  */
void ffq_oracle_arrayadd_q(int64_t *a, int64_t n, int64_t value)
{
    uint64_t *u = (uint64_t *)a;
    for (int64_t i = 0; i < n; i++) u[i] += (uint64_t)value;
}

/* Phred decode of every record's quality span into one packed int8 stream
 * (the batched form of doc/user-guide.rst:130-141: array('b').frombytes(
 * buf[pos4:pos5]) then arrayadd_b(-33)).  qoff[i] = start of record i in out,
 * qoff[n] = total.  Positions are indices into `base` (caller passes the
 * pointer the table's coordinates refer to).                               */
void ffq_oracle_decode_quals(const uint8_t *base, const int64_t *table, int64_t n,
                             int value, int8_t *out, int64_t *qoff)
{
    int64_t w = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t a = table[6 * i + 4], z = table[6 * i + 5];
        qoff[i] = w;
        memcpy(out + w, base + a, (size_t)(z - a));
        ffq_oracle_arrayadd_b(out + w, z - a, value);
        w += z - a;
    }
    qoff[n] = w;
}

/* ---- entryfunc push-down: what an entryfunc that builds only ONE component
 * of an entry yields, for every row of a table, packed: buf[pos[ca] + shift : pos[cb]] (+ value,
 * int8 wrap as arrayadd_b) and CSR offsets.  /root/reference/doc/user-guide.rst:153-180 returns
 * buf[posarray[2]:posarray[3]]; /root/reference/src/fastqandfurious.py:161-171 cuts the header as
 * buf[pos[0] + 1:pos[1]].                                                                         */
void ffq_oracle_gather_column(const uint8_t *base, const int64_t *table, int64_t n, int ca, int shift,
                              int cb, int value, int8_t *out, int64_t *coff)
{
    int64_t w = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t a = table[6 * i + ca] + shift, z = table[6 * i + cb];
        const int64_t len = z > a ? z - a : 0;
        coff[i] = w;
        memcpy(out + w, base + a, (size_t)len);
        ffq_oracle_arrayadd_b(out + w, len, value);
        w += len;
    }
    coff[n] = w;
}

/* The length filter of /root/reference/doc/user-guide.rst:153-180 (`posarray[3] - posarray[2]`
 * against a threshold) evaluated on a table: rows with lo <= pos3 - pos2 <= hi, in order.        */
int64_t ffq_oracle_select_seqlen(const int64_t *table, int64_t n, int64_t lo, int64_t hi, int64_t *out)
{
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t len = table[6 * i + 3] - table[6 * i + 2];
        if (len >= lo && len <= hi) { memcpy(out + 6 * k, table + 6 * i, 48); k++; }
    }
    return k;
}

/* ---- FASTA (widening row) ---------------------------------------------
 * One scanner call: /root/reference/src/fastqandfurious.py:103-143 (entrypos_fasta, Python; the
 * reference has no C scanner for FASTA).  No sentinel arithmetic here: d IS the buffer.        */
int ffq_oracle_entrypos_fasta(const uint8_t *d, int64_t len, int64_t offset, int64_t *pos)
{
    const int64_t i = find2(d, len, offset, '\n', '>');                 /* :118 */
    if (i < 0) return FFQ_MISSING_SEQHEADER_BEGIN;
    pos[0] = i + 1;
    const int64_t j = find1(d, i + 2, len, '\n');                       /* :123 */
    if (j < 0) return FFQ_MISSING_SEQHEADER_END;
    pos[1] = j;
    if (j + 1 >= len) return FFQ_MISSING_SEQ_BEG;                       /* :130 */
    pos[2] = j + 1;
    const int64_t k = find2(d, len, j + 1, '\n', '>');                  /* :133 */
    if (k < 0) {
        pos[3] = (d[len - 1] == '\n') ? len - 1 : len;                  /* :137-140 */
        return FFQ_MISSING_SEQ_END;
    }
    pos[3] = k;
    return FFQ_COMPLETE;
}

/* Every COMPLETE entry of a buffer: repeated calls, each from the "\n>" the previous one ended
 * at (offset := pos[3]).  table rows = pos0..pos3 + add, -1, -1.  out = {n_complete, last
 * status, position of the "\n>" the last call matched (its offset argument if it matched none),
 * 0}; last_pos receives the posbuffer of that last call
 * (+ add where set).                                                                          */
void ffq_oracle_scan_fasta(const uint8_t *d, int64_t len, int64_t offset, int64_t add,
                           int64_t *table, int64_t cap, int64_t *last_pos, int64_t *out)
{
    int64_t n = 0;
    int status;
    int64_t pos[6];
    for (;;) {
        for (int i = 0; i < 6; i++) pos[i] = -1;
        status = ffq_oracle_entrypos_fasta(d, len, offset, pos);
        if (status != FFQ_COMPLETE || n >= cap) break;
        for (int i = 0; i < 4; i++) table[6 * n + i] = pos[i] + add;
        table[6 * n + 4] = -1; table[6 * n + 5] = -1;
        n++;
        offset = pos[3];
    }
    for (int i = 0; i < 6; i++) last_pos[i] = pos[i] >= 0 ? pos[i] + add : -1;
    /* where a caller that refills should carry on from: the "\n>" the last call matched */
    out[0] = n; out[1] = status; out[2] = (pos[0] >= 0) ? pos[0] - 1 : offset; out[3] = 0;
}

