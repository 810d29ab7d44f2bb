// ffq_stream.h -- the stream front end of the path, natively (host code of libffq_hip.so):
//   /root/reference/src/fastqandfurious.py:30-36    read(fh, fbufsize): one read, eof := short read
//   /root/reference/src/fastqandfurious.py:241-279  the refill loop of readfastq_iter: sentinel once,
//                                                   buf = buf[offset:] + next chunk, globaloffset
// over a file descriptor.  Chunks are read into pinned memory; the read of chunk i+1 (a helper
// thread) overlaps the H2D copy, the scan and the D2H copy of fill i.  Each call of
// ffq_stream_next hands back the rows (absolute stream offsets, what entryfunc_abspos yields,
// :186-195) of one buffer fill and the fill's bytes for slicing.
#pragma once
#include <errno.h>
#include <unistd.h>

#include <future>

struct ffq_stream {
    ffq_ctx *c = nullptr;
    int fd = -1;
    bool seekable = false;
    int64_t file_pos = 0;               // next byte to read (pread offset)
    int64_t fbufsize = 0;
    int64_t carry_room = 0;             // bytes in front of every chunk for the carried tail
    uint8_t *hbuf[2] = {nullptr, nullptr};      // pinned: [carry_room | fbufsize]
    uint8_t *dbuf = nullptr;            // device copy of [carry | chunk]
    int64_t dcap = 0;
    int64_t *dtab = nullptr;            // rows on the device / in pinned memory
    int64_t *htab = nullptr;
    int64_t tab_cap = 0;
    uint32_t flags = 0;                 // FFQ_F_DECODE_QUAL: qualities decoded per fill
    int qual_add = -33;
    int8_t *dqual = nullptr, *hqual = nullptr;  // decoded stream of the fill (device / pinned)
    int64_t qual_cap = 0;
    int64_t *dqoff = nullptr, *hqoff = nullptr; // CSR offsets, tab_cap + 1 entries
    int64_t last_nq = 0;
    std::future<int64_t> rd;            // read-ahead into hbuf[cur ^ 1] + carry_room
    bool rd_pending = false;
    int cur = 0;
    int64_t start = 0, len = 0;         // the current fill is hbuf[cur][start, start + len)
    bool fill_eof = false;              // its chunk was a short read
    bool first = true, done = false;
    int64_t globaloffset = -1;          // readfastq_iter :242
};

static int64_t stream_read_full(int fd, uint8_t *dst, int64_t n, int64_t pos, bool seekable)
{
    int64_t got = 0;
    while (got < n) {
        const ssize_t r = seekable ? pread(fd, dst + got, (size_t)(n - got), (off_t)(pos + got))
                                   : read(fd, dst + got, (size_t)(n - got));
        if (r < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        if (r == 0) break;
        got += r;
    }
    return got;
}

// a chunk of a seekable descriptor by several threads (one pread loop saturates at the memcpy
// rate of a single core, ~10 GB/s from the page cache); the result is the bytes read up to the
// first short slice -- the same prefix a single read would have returned
static int64_t stream_read_chunk(int fd, uint8_t *dst, int64_t n, int64_t pos, bool seekable)
{
    const int64_t SL = 2 << 20;
    const int nthr = (int)std::min<int64_t>(8, n / SL);
    if (!seekable || nthr < 2) return stream_read_full(fd, dst, n, pos, seekable);
    const int64_t per = ((n + nthr - 1) / nthr + 4095) & ~(int64_t)4095;
    std::future<int64_t> part[8];
    int used = 0;
    for (int t = 0; t < nthr; t++) {
        const int64_t a = (int64_t)t * per;
        if (a >= n) break;
        const int64_t m = std::min(per, n - a);
        part[t] = std::async(std::launch::async, [=] { return stream_read_full(fd, dst + a, m, pos + a, true); });
        used++;
    }
    int64_t got = 0;
    bool shortread = false, bad = false;
    for (int t = 0; t < used; t++) {
        const int64_t g = part[t].get();
        const int64_t m = std::min(per, n - (int64_t)t * per);
        if (g < 0) bad = true;
        else if (!shortread) { got += g; if (g < m) shortread = true; }
    }
    return bad ? -1 : got;
}

static void stream_free(ffq_stream *s)
{
    if (!s) return;
    if (s->rd_pending) (void)s->rd.get();
    for (int b = 0; b < 2; b++)
        if (s->hbuf[b]) (void)hipHostFree(s->hbuf[b]);
    if (s->htab) (void)hipHostFree(s->htab);
    if (s->hqual) (void)hipHostFree(s->hqual);
    if (s->hqoff) (void)hipHostFree(s->hqoff);
    (void)hipFree(s->dbuf);
    (void)hipFree(s->dtab);
    (void)hipFree(s->dqual);
    (void)hipFree(s->dqoff);
    delete s;
}

static int stream_alloc_tab(ffq_stream *s, int64_t rows)
{
    if (rows <= s->tab_cap) return FFQ_OK;
    if (s->htab) (void)hipHostFree(s->htab);
    (void)hipFree(s->dtab);
    s->htab = nullptr; s->dtab = nullptr; s->tab_cap = 0;
    if (hipMalloc((void **)&s->dtab, (size_t)rows * 48) != hipSuccess ||
        hipHostMalloc((void **)&s->htab, (size_t)rows * 48, hipHostMallocDefault) != hipSuccess)
        return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld rows", (long long)rows);
    s->tab_cap = rows;
    if (s->flags & FFQ_F_DECODE_QUAL) {
        if (s->hqoff) (void)hipHostFree(s->hqoff);
        (void)hipFree(s->dqoff);
        s->hqoff = nullptr; s->dqoff = nullptr;
        if (hipMalloc((void **)&s->dqoff, (size_t)(rows + 1) * 8) != hipSuccess ||
            hipHostMalloc((void **)&s->hqoff, (size_t)(rows + 1) * 8, hipHostMallocDefault) != hipSuccess)
            return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld quality offsets", (long long)rows);
    }
    return FFQ_OK;
}

static int stream_alloc_qual(ffq_stream *s, int64_t bytes)
{
    if (bytes <= s->qual_cap) return FFQ_OK;
    if (s->hqual) (void)hipHostFree(s->hqual);
    (void)hipFree(s->dqual);
    s->hqual = nullptr; s->dqual = nullptr; s->qual_cap = 0;
    if (hipMalloc((void **)&s->dqual, (size_t)bytes) != hipSuccess ||
        hipHostMalloc((void **)&s->hqual, (size_t)bytes, hipHostMallocDefault) != hipSuccess)
        return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld decoded bytes", (long long)bytes);
    s->qual_cap = bytes;
    return FFQ_OK;
}

// (re)allocate the pinned pair with `room` bytes of carry space, keeping the current fill and the
// chunk that has been read ahead (`ahead` bytes at hbuf[cur ^ 1] + carry_room)
static int stream_grow_room(ffq_stream *s, int64_t room, int64_t ahead)
{
    uint8_t *nb[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; b++)
        if (hipHostMalloc((void **)&nb[b], (size_t)(room + s->fbufsize + 16), hipHostMallocDefault) != hipSuccess)
            return fail(FFQ_E_NOMEM, "ffq_stream: no pinned memory for a %lld-byte carry", (long long)room);
    if (s->hbuf[s->cur]) memcpy(nb[s->cur] + room - (s->carry_room - s->start), s->hbuf[s->cur] + s->start, (size_t)s->len);
    if (s->hbuf[s->cur ^ 1] && ahead > 0) memcpy(nb[s->cur ^ 1] + room, s->hbuf[s->cur ^ 1] + s->carry_room, (size_t)ahead);
    s->start = room - (s->carry_room - s->start);
    for (int b = 0; b < 2; b++) {
        if (s->hbuf[b]) (void)hipHostFree(s->hbuf[b]);
        s->hbuf[b] = nb[b];
    }
    s->carry_room = room;
    return FFQ_OK;
}

extern "C" int ffq_stream_open2(ffq_ctx *c, int fd, int64_t fbufsize, uint32_t flags, int qual_add,
                                ffq_stream **out)
{
    if (!c || !out || fd < 0 || fbufsize <= 0) return fail(FFQ_E_ARG, "ffq_stream_open: bad argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    ffq_stream *s = new (std::nothrow) ffq_stream();
    if (!s) return fail(FFQ_E_NOMEM, "out of host memory");
    s->c = c; s->fd = fd; s->fbufsize = fbufsize;
    s->flags = flags & FFQ_F_DECODE_QUAL; s->qual_add = qual_add;
    const off_t at = lseek(fd, 0, SEEK_CUR);
    s->seekable = at != (off_t)-1;
    s->file_pos = s->seekable ? (int64_t)at : 0;
    s->carry_room = std::max<int64_t>(1 << 20, 4096);
    for (int b = 0; b < 2; b++)
        if (hipHostMalloc((void **)&s->hbuf[b], (size_t)(s->carry_room + fbufsize + 16), hipHostMallocDefault) != hipSuccess) {
            stream_free(s);
            return fail(FFQ_E_NOMEM, "ffq_stream_open: no pinned memory for %lld-byte chunks", (long long)fbufsize);
        }
    int rc = stream_alloc_tab(s, fbufsize / 64 + 1024);
    if (rc) { stream_free(s); return rc; }
    *out = s;
    return FFQ_OK;
}

extern "C" int ffq_stream_open(ffq_ctx *c, int fd, int64_t fbufsize, ffq_stream **out)
{
    return ffq_stream_open2(c, fd, fbufsize, 0, 0, out);
}

extern "C" void ffq_stream_close(ffq_stream *s) { stream_free(s); }

// decoded qualities of the fill ffq_stream_next has just returned (streams opened with
// FFQ_F_DECODE_QUAL): int8 stream + CSR offsets (n_rows + 1), pinned, valid until the next call
extern "C" int ffq_stream_quals(ffq_stream *s, const int8_t **h_qual, const int64_t **h_qoff, int64_t *n_qual_bytes)
{
    if (!s || !h_qual || !h_qoff || !n_qual_bytes) return fail(FFQ_E_ARG, "ffq_stream_quals: NULL argument");
    if (!(s->flags & FFQ_F_DECODE_QUAL)) return fail(FFQ_E_ARG, "ffq_stream_quals: the stream was opened without FFQ_F_DECODE_QUAL");
    *h_qual = s->hqual; *h_qoff = s->hqoff; *n_qual_bytes = s->last_nq;
    return FFQ_OK;
}

extern "C" int ffq_stream_next(ffq_stream *s, const int64_t **h_rows, int64_t *n_rows, int *end_state,
                               int64_t *err_offset, const uint8_t **h_bytes, int64_t *n_bytes,
                               int64_t *bytes_offset)
{
    if (!s || !h_rows || !n_rows || !end_state) return fail(FFQ_E_ARG, "ffq_stream_next: NULL argument");
    ffq_ctx *c = s->c;
    HIPCHK(hipSetDevice(c->device));
    *h_rows = s->htab; *n_rows = 0; *end_state = FFQ_END_OK;
    if (err_offset) *err_offset = -1;
    if (h_bytes) *h_bytes = nullptr;
    if (n_bytes) *n_bytes = 0;
    if (bytes_offset) *bytes_offset = 0;
    if (s->done) return FFQ_OK;

    if (s->first) {
        // the first chunk, synchronously
        const int64_t got = stream_read_chunk(s->fd, s->hbuf[0] + s->carry_room, s->fbufsize, s->file_pos, s->seekable);
        if (got < 0) return fail(FFQ_E_ARG, "ffq_stream: read failed: %s", strerror(errno));
        s->file_pos += got;
        // buf = b'\n' + first chunk (:245): the sentinel is a real byte of the buffer, it is carried
        // over a refill like any other (a first record longer than fbufsize needs it again)
        s->hbuf[0][s->carry_room - 1] = (uint8_t)'\n';
        s->cur = 0; s->start = s->carry_room - 1; s->len = got + 1; s->fill_eof = got < s->fbufsize;
    }
    // read ahead while this fill is on the GPU
    if (!s->fill_eof) {
        uint8_t *dst = s->hbuf[s->cur ^ 1] + s->carry_room;
        const int fd = s->fd; const int64_t n = s->fbufsize, pos = s->file_pos; const bool sk = s->seekable;
        s->rd = std::async(std::launch::async, [fd, dst, n, pos, sk] { return stream_read_chunk(fd, dst, n, pos, sk); });
        s->rd_pending = true;
    }
    ffq_scan_result res;
    memset(&res, 0, sizeof res);
    int rc = FFQ_OK;
    {
        if (s->dcap < s->len + 16) {
            (void)hipFree(s->dbuf);
            s->dbuf = nullptr; s->dcap = 0;
            const int64_t want = std::max<int64_t>(s->len + 16, s->carry_room + s->fbufsize + 16);
            if (hipMalloc((void **)&s->dbuf, (size_t)want) != hipSuccess)
                return fail(FFQ_E_NOMEM, "ffq_stream: no device memory for a %lld-byte fill", (long long)want);
            s->dcap = want;
        }
        HIPCHK(hipMemcpyAsync(s->dbuf, s->hbuf[s->cur] + s->start, (size_t)s->len, hipMemcpyHostToDevice, c->stream));
        const bool decode = (s->flags & FFQ_F_DECODE_QUAL) != 0;
        if (decode) {
            int rc2 = stream_alloc_qual(s, s->len / 2 + 64);      // qualities are at most half of the bytes
            if (rc2) return rc2;
        }
        for (int attempt = 0; attempt < 2; attempt++) {
            rc = ffq_scan_device(c, s->dbuf, s->len, 0, 0, s->fill_eof ? 1 : 0, s->globaloffset, s->flags, s->qual_add,
                                 s->dtab, s->tab_cap, decode ? s->dqual : nullptr, decode ? s->qual_cap : 0,
                                 decode ? s->dqoff : nullptr, &res);
            if (rc != FFQ_E_TABLE_FULL) break;
            int rc2 = stream_alloc_tab(s, res.n_records + 1024);
            if (rc2) return rc2;
        }
        if (rc != FFQ_OK) return rc;
        if (res.n_records > 0)
            HIPCHK(hipMemcpyAsync(s->htab, s->dtab, (size_t)res.n_records * 48, hipMemcpyDeviceToHost, c->stream));
        s->last_nq = 0;
        if (decode) {
            s->last_nq = res.n_qual_bytes;
            if (res.n_qual_bytes > 0)
                HIPCHK(hipMemcpyAsync(s->hqual, s->dqual, (size_t)res.n_qual_bytes, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(s->hqoff, s->dqoff, (size_t)(res.n_records + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *h_rows = s->htab;
    *n_rows = res.n_records;
    if (h_bytes) *h_bytes = s->hbuf[s->cur] + s->start;
    if (n_bytes) *n_bytes = s->len;
    // byte i of this fill is stream offset globaloffset + i (the sentinel of the first fill is -1)
    if (bytes_offset) *bytes_offset = s->globaloffset;
    *end_state = res.end_state;
    if (res.end_state != FFQ_END_REFILL) {
        if (res.end_state != FFQ_END_OK && err_offset) *err_offset = s->globaloffset + res.end_offset;
        s->done = true;
        if (s->rd_pending) { (void)s->rd.get(); s->rd_pending = false; }
        s->first = false;
        return FFQ_OK;
    }
    // refill: buf = buf[offset:] + next chunk (:277), globaloffset += offset (:275)
    int64_t got = 0;
    if (s->rd_pending) {
        got = s->rd.get();
        s->rd_pending = false;
        if (got < 0) return fail(FFQ_E_ARG, "ffq_stream: read failed: %s", strerror(errno));
        s->file_pos += got;
    }
    const int64_t ds = res.end_offset;                       // buf[offset:] is carried over
    const int64_t carry = s->len - ds;
    if (carry > s->carry_room) {
        rc = stream_grow_room(s, std::max<int64_t>(2 * s->carry_room, carry + 4096), got);
        if (rc) return rc;
    }
    const int nxt = s->cur ^ 1;
    memcpy(s->hbuf[nxt] + s->carry_room - carry, s->hbuf[s->cur] + s->start + ds, (size_t)carry);
    s->globaloffset += res.end_offset;
    s->cur = nxt;
    s->start = s->carry_room - carry;
    s->len = carry + got;
    s->fill_eof = got < s->fbufsize;
    s->first = false;
    return FFQ_OK;
}
