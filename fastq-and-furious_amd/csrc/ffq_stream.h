// ffq_stream.h -- the stream front end of the path, natively (host code of libffq_hip.so):
//   /root/reference/src/fastqandfurious.py:30-36    read(fh, fbufsize): one read, eof := short read
//   /root/reference/src/fastqandfurious.py:241-279  the refill loop of readfastq_iter: sentinel once,
//                                                   buf = buf[offset:] + next chunk, globaloffset
// over a file descriptor, as a three-stage pipeline:
//
//   feeder thread   chunk k+2: pread by a pool of helper threads into pinned slot memory, then
//                   hipMemcpyAsync of the chunk to the slot's device buffer on a COPY stream
//   copy stream     chunk k+1 on its way over PCIe
//   caller thread   fill k: the carried tail buf[offset:] of fill k-1 is put in front of chunk k
//                   (host memcpy + a small H2D on the scan stream), the scan waits for the
//                   chunk's copy event, rows come back into pinned memory
//
// so the file read, the H2D copy and the scan + D2H of three consecutive chunks overlap (the
// first version overlapped only the read).  A slot is [room | fbufsize] on both sides: the
// chunk always lands at offset `room`, the carry ends there, and the fill is the contiguous
// range [room - carry, room + got).  The scan is given the 16-byte aligned address below the
// fill's first byte and starts its search at the fill's first byte (`offset`), so no byte is
// moved to align anything.  Each call of ffq_stream_next hands back the rows (absolute stream
// offsets, what entryfunc_abspos yields, :186-195) of one buffer fill and the fill's bytes for
// slicing; both stay valid until the next call.
//
// Three kinds of source feed the same slots:
//   a descriptor        the feeder thread: pread in parallel slices (regular files), or read()
//                       behind poll() (pipes: interruptible, and a chunk is handed over short when
//                       nothing more has arrived for a while -- a short chunk is not the end)
//   a gzip descriptor   the feeder thread inflates (zlib; concatenated members) straight into the
//                       pinned slot: decompression is the feeder stage, no interpreter involved
//                       (reference: FORMAT_OPENERS / automagic_open, fastqandfurious.py:282-334)
//   pushed chunks       no feeder thread: the host writes every chunk into the slot's pinned memory
//                       itself (ffq_stream_push_buffer / ffq_stream_push) -- any Python file-like
//                       object (BytesIO, bz2, lzma ...) read with readinto(), no bytes copy
#pragma once
#include <errno.h>
#include <poll.h>
#include <unistd.h>
#include <zlib.h>
#include <memory>
#include <sys/stat.h>

#include "ffq_pgz.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#ifndef FFQ_STREAM_SLOTS
#define FFQ_STREAM_SLOTS 3
#endif
#ifndef FFQ_STREAM_AHEAD
#define FFQ_STREAM_AHEAD 2
#endif
constexpr int STREAM_SLOTS = FFQ_STREAM_SLOTS;

struct StreamSlot {
    uint8_t *h = nullptr;        // pinned  [room | fbufsize (+16)]
    uint8_t *d = nullptr;        // device  same layout
    hipEvent_t copied[2] = {nullptr, nullptr};   // the chunk's H2D copy is through (one half per copy stream)
    int64_t got = 0;             // bytes of the chunk (at offset room)
    bool eof = false;            // the source is exhausted behind this chunk
    int64_t end_pos = 0;         // position of the source behind this chunk (ffq_stream_tell)
    ChunkRead cr;                // its slices while they are being read
};

// everything a stream allocates; parked in the context when a stream closes so that the next
// stream of the same chunk size starts without the pinned allocations (milliseconds each)
struct StreamBufs {
    int64_t fbufsize = 0, room = 0;
    StreamSlot slot[STREAM_SLOTS];
    hipStream_t cs[2] = {nullptr, nullptr};      // copy streams of the chunks: a chunk goes over in two halves, one
                                                 // per stream (two DMA engines: one did 41-45 GB/s of the link's ~55)
    bool one_copy_stream = false;                // ... unless much goes BACK as well (stream_copy_chunk)
    int64_t *dtab = nullptr, *htab = nullptr;
    int64_t tab_cap = 0;
    int8_t *dqual = nullptr, *hqual = nullptr;
    int64_t qual_cap = 0;
    int64_t *dqoff = nullptr, *hqoff = nullptr;
    int64_t qoff_cap = 0;
    // push-down (ffq_stream_set_filter): the kept rows and their ordinals, the gathered column
    int64_t *dsel = nullptr, *didx = nullptr, *hidx = nullptr;
    int64_t sel_cap = 0;
    int8_t *dcol = nullptr, *hcol = nullptr;
    int64_t col_cap = 0;
    int64_t *dcoff = nullptr, *hcoff = nullptr;
    int64_t coff_cap = 0;
    ReadPool *pool = nullptr;    // the context's helper threads (not owned)
};

static void streambufs_free_slots(StreamBufs *b)
{
    for (auto &s : b->slot) {
        if (s.h) (void)hipHostFree(s.h);
        (void)hipFree(s.d);
        s.h = nullptr; s.d = nullptr;
    }
}

static void streambufs_free(StreamBufs *b)
{
    if (!b) return;
    streambufs_free_slots(b);
    for (auto &s : b->slot) for (auto &e : s.copied) if (e) (void)hipEventDestroy(e);
    if (b->htab) (void)hipHostFree(b->htab);
    if (b->hqual) (void)hipHostFree(b->hqual);
    if (b->hqoff) (void)hipHostFree(b->hqoff);
    if (b->hidx) (void)hipHostFree(b->hidx);
    if (b->hcol) (void)hipHostFree(b->hcol);
    if (b->hcoff) (void)hipHostFree(b->hcoff);
    (void)hipFree(b->dsel); (void)hipFree(b->didx); (void)hipFree(b->dcol); (void)hipFree(b->dcoff);
    (void)hipFree(b->dtab); (void)hipFree(b->dqual); (void)hipFree(b->dqoff);
    for (auto &st : b->cs) if (st) (void)hipStreamDestroy(st);
    delete b;
}

// called by ffq_ctx_destroy
static void stream_cache_drop(ffq_ctx *c)
{
    streambufs_free(static_cast<StreamBufs *>(c->stream_cache));
    c->stream_cache = nullptr;
}

static int streambufs_alloc_slots(ffq_ctx *c, StreamBufs *b, int64_t room)
{
    NearGpu near(c);                   // (the pinned slots on the GPU's node: ffq_hip.hip)
    for (auto &s : b->slot) {
        if (hipHostMalloc((void **)&s.h, (size_t)(room + b->fbufsize + 16), hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&s.d, (size_t)(room + b->fbufsize + 16)) != hipSuccess)
            return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld-byte chunks with a %lld-byte carry",
                        (long long)b->fbufsize, (long long)room);
    }
    b->room = room;
    return FFQ_OK;
}

// the H2D copy of a slot's chunk: two halves, two streams, two events
static hipError_t stream_copy_chunk(StreamBufs *b, StreamSlot &sl, int64_t got)
{
    // (one stream for a stream that DECODES: two copy streams in and the copies back on a third share the copy engines --
    // 42.9 GB/s in + 28.7 back against 54.7 + 36.6 with one stream in, tools/link_duplex.py; the stream front end with the
    // decode 34-36 -> 43-45 GB/s, without it 54 -> 47-50, so a plain stream keeps its two)
    const bool one = b->one_copy_stream;
    const int64_t half = one ? got : ((got / 2) + 4095) & ~(int64_t)4095;
    hipError_t e = hipSuccess;
    for (int h = 0; h < 2 && e == hipSuccess; h++) {
        const int64_t a = h ? std::min(half, got) : 0, z = h ? got : std::min(half, got);
        hipStream_t st = b->cs[one ? 0 : h];
        if (z > a) e = hipMemcpyAsync(sl.d + b->room + a, sl.h + b->room + a, (size_t)(z - a), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(sl.copied[h], st);
    }
    return e;
}

constexpr int SRC_FD = 0, SRC_GZIP = 1, SRC_PUSH = 2;
constexpr int64_t GZ_IN = 4 << 20;      // compressed bytes held at a time

// ---- BGZF members inflated side by side ------------------------------------------------------------
// A gzip member says how long it is only if its writer put that into the header: bgzip does (the "BC"
// extra field of the BGZF format, SAM specification section 4.1: BSIZE = length of the member - 1, at
// most 64 KiB of data per member, the uncompressed length in the member's last four bytes).  Members of
// such a file are found without inflating anything and inflated independently, each straight into its
// place in the pinned chunk.  Anything else -- and any member that does not hold what its header and
// trailer promise -- goes through the one-member-at-a-time inflate below, which has the last word.
struct GzJob {
    const uint8_t *src;     // raw deflate data of the member
    uint32_t clen;
    uint8_t *dst;
    uint32_t isize, crc;    // from the member's trailer
};

struct GzPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    const GzJob *jobs = nullptr;
    int njobs = 0, active = 0;
    std::atomic<int> next{0}, failed{0};
    uint64_t gen = 0;
    bool quit = false;

    static bool inflate_one(z_stream *z, const GzJob &j)
    {
        uint8_t none[8];
        if (inflateReset(z) != Z_OK) return false;
        z->next_in = const_cast<Bytef *>(j.src);
        z->avail_in = j.clen;
        z->next_out = j.isize ? j.dst : none;
        z->avail_out = j.isize ? j.isize : (uInt)sizeof none;
        const int r = inflate(z, Z_FINISH);
        if (r != Z_STREAM_END || z->avail_in != 0 || z->total_out != j.isize) return false;
        return (uint32_t)crc32(crc32(0L, Z_NULL, 0), j.dst, j.isize) == j.crc;
    }
    // The same member through this build's own decoder (ffq_pgz.h; about twice zlib's rate on FASTQ), into a scratch
    // buffer of the thread (a match is copied a word at a time: the bytes behind a member's end belong to the next
    // member, which another thread is writing); whatever it does not like is zlib's (inflate_one).
    struct Scratch { pgz::Inflater<uint8_t> inf; std::vector<uint8_t> buf; };
    static bool inflate_fast(const GzJob &j)
    {
        thread_local std::unique_ptr<Scratch> sc;
        try {
            if (!sc) sc.reset(new Scratch());
            const int64_t cap = (int64_t)j.isize + pgz::OUT_SLACK + 8;
            if ((int64_t)sc->buf.size() < cap) sc->buf.resize((size_t)cap);
        } catch (const std::bad_alloc &) { return false; }
        pgz::Inflater<uint8_t> &f = sc->inf;
        f.win_valid = 0;
        f.limit_bit = INT64_MAX;
        f.set_out(sc->buf.data(), 0, (int64_t)j.isize + pgz::OUT_SLACK + 8);
        f.start(j.src, j.clen, 0);
        if (f.run() != pgz::R_FINAL || f.b_out != (int64_t)j.isize || ((f.b_bit + 7) >> 3) != (int64_t)j.clen) return false;
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), sc->buf.data(), j.isize) != j.crc) return false;
        memcpy(j.dst, sc->buf.data(), j.isize);
        return true;
    }
    void work(z_stream *z)
    {
        for (;;) {
            const int j = next.fetch_add(1);
            if (j >= njobs) break;
            if (!inflate_fast(jobs[j]) && !inflate_one(z, jobs[j])) failed.store(1);
        }
    }
    void worker()
    {
        z_stream z;
        memset(&z, 0, sizeof z);
        const bool ok = inflateInit2(&z, -15) == Z_OK;      // raw deflate: header and trailer are read by the parser
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv_go.wait(lk, [&] { return quit || gen != seen; });
                if (quit) break;
                seen = gen;
            }
            if (ok) work(&z);
            std::lock_guard<std::mutex> lk(m);
            if (--active == 0) cv_done.notify_one();
        }
        if (ok) (void)inflateEnd(&z);
    }
    bool start(int nthreads)
    {
        try {
            for (int i = 0; i < nthreads; i++) th.emplace_back(&GzPool::worker, this);
        } catch (...) {}
        return !th.empty();
    }
    // all jobs, by the pool's threads and the caller's (own: the caller's raw-deflate state); false: a member
    // did not inflate to what its trailer says
    bool run(const GzJob *js, int n, z_stream *own)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            jobs = js; njobs = n; next.store(0); failed.store(0);
            active = (int)th.size();
            gen++;
        }
        cv_go.notify_all();
        work(own);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return active == 0; });
        return failed.load() == 0;
    }
    ~GzPool()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
};

// threads that inflate (FFQ_GZ_THREADS; default: this process's share of the host's cores -- cores / LOCAL_WORLD_SIZE
// when a launcher runs one rank per GPU --, at most 32): 1 = no pool
static int gz_threads_default()
{
    const char *e = getenv("FFQ_GZ_THREADS");
    if (e && atoi(e) > 0) return std::min(atoi(e), 256);
    unsigned hw = std::max<unsigned>(std::thread::hardware_concurrency(), 1u);
    const char *lw = getenv("LOCAL_WORLD_SIZE");
    if (lw && atoi(lw) > 1) hw = std::max<unsigned>(hw / (unsigned)atoi(lw), 1u);
    return (int)std::min<unsigned>(hw, 32u);
}

struct ffq_stream {
    ffq_ctx *c = nullptr;
    StreamBufs *b = nullptr;
    int fd = -1;
    int src = SRC_FD;
    bool seekable = false;
    // ---- gzip source (feeder thread only) ----
    z_stream zs;
    bool z_init = false, z_member = false, z_in_eof = false;
    uint8_t *zin = nullptr;
    int64_t zin_len = 0, zin_pos = 0, z_filepos = 0;
    std::string z_msg;
    int gz_threads = 1;                 // > 1: BGZF members are inflated side by side
    bool bgzf_ok = true;                // cleared when a member did not hold what it promised: one at a time from there
    GzPool *gz_pool = nullptr;
    z_stream zraw;                      // the reader thread's own raw-deflate state (it takes jobs too)
    bool zraw_init = false;
    std::vector<GzJob> gz_jobs;
    int64_t bgzf_members = 0;           // members inflated by the pool (statistics)
    // one member inflated by several threads (ffq_pgz.h: a regular file that is not BGZF), and zlib going on from
    // the block boundary the engine gave up at
    pgz::Engine *pgz = nullptr;
    bool pgz_ok = true, pgz_active = false;
    int pgz_small = 0;
    int64_t pgz_member_off = 0, gz_file_size = -1;
    bool z_raw = false;                 // zs is a raw deflate stream: the member's trailer is checked here
    uint32_t z_crc = 0;
    uint64_t z_isize = 0;
    int64_t handed_pos = 0;             // position of the source behind the last chunk handed out
    uint32_t flags = 0;                 // FFQ_F_DECODE_QUAL: qualities decoded per fill
    int qual_add = -33;
    // ---- feeder <-> caller (under m) ----
    std::thread feeder;
    std::mutex m;
    std::condition_variable cv;
    int64_t produced = 0;               // chunks [0, produced) are read and their copy is enqueued
    int64_t released = 0;               // chunks [0, released) are consumed: their slots may be refilled
    int64_t file_pos = 0;               // next byte to read
    bool stop = false, feeder_done = false, pause_req = false, paused = false;
    int feeder_rc = FFQ_OK;
    std::string feeder_msg;
    // ---- caller only ----
    int64_t cur = -1;                   // chunk of the fill handed out last
    int64_t fill_start = 0, fill_len = 0;   // that fill is slot.h[fill_start, fill_start + fill_len)
    int64_t carry_from = 0;             // fill-relative offset the next fill starts with (buf[offset:], :277)
    bool done = false, failed = false;
    int64_t globaloffset = -1;          // readfastq_iter :242
    int64_t last_nq = 0;
    int last_path = -1;                 // ffq_scan_result.path of the last fill's scan
    // push-down: rows by sequence length, one component of the kept rows (ffq_stream_set_filter)
    bool filter_on = false;
    int64_t f_min = 0, f_max = 0;
    int f_col = 0, f_add = 0;
    int64_t last_scanned = 0, last_kept = 0, last_col_bytes = 0;
    // FFQ_STREAM_PROF=1: where the time of a stream goes (printed when it closes)
    bool prof = false;
    double t_read = 0, t_slot = 0, t_feed = 0, t_scan = 0, t_rows = 0, t_copy = 0;
};

static inline double stream_now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One chunk from a descriptor that cannot seek (a pipe, a socket): read() behind poll(), so that a
// request to stop or to park (stream_grow_room) is seen within 50 ms even when the writer keeps the
// pipe open and idle.  Returns the bytes read; *eof: read() returned 0.  A chunk is handed over SHORT
// (not eof) when something was read and then nothing arrived for 50 ms, or a pause was asked for:
// records reach the caller as they come in.  -2: asked to stop / park before anything was read.
// -1: error (errno).
static int64_t stream_fd_read(ffq_stream *s, uint8_t *dst, int64_t n, bool *eof)
{
    int64_t got = 0;
    *eof = false;
    while (got < n) {
        bool stop, pause;
        { std::lock_guard<std::mutex> lk(s->m); stop = s->stop; pause = s->pause_req; }
        if (stop || pause) return got > 0 ? got : -2;
        struct pollfd pf = {s->fd, POLLIN, 0};
        const int pr = poll(&pf, 1, 50);
        if (pr < 0) { if (errno == EINTR) continue; return -1; }
        if (pr == 0) { if (got > 0) return got; continue; }
        const ssize_t r = read(s->fd, dst + got, (size_t)(n - got));
        if (r < 0) { if (errno == EINTR || errno == EAGAIN) continue; return -1; }
        if (r == 0) { *eof = true; break; }
        got += r;
    }
    return got;
}

// One chunk of the DECOMPRESSED stream of a gzip file (RFC 1952; members may be concatenated, as
// bgzip and `cat a.gz b.gz` produce them; zero padding behind the last member is ignored).  Fills
// dst completely unless the stream ends (*eof).  -1: error, s->z_msg says what.
// More compressed bytes behind what is left in zin (moved to the front).  1: something was added, 0: nothing
// (end of the file, or zin is full), -1: error (z_msg), -2: asked to stop / park with nothing read.
static int gz_more_input(ffq_stream *s)
{
    const int64_t rem = s->zin_len - s->zin_pos;
    if (s->z_in_eof || rem >= GZ_IN) return 0;
    if (rem > 0 && s->zin_pos > 0) memmove(s->zin, s->zin + s->zin_pos, (size_t)rem);
    s->zin_pos = 0; s->zin_len = rem;
    int64_t r;
    bool in_eof = false;
    if (s->seekable) { r = ReadPool::read_full(s->fd, s->zin + rem, GZ_IN - rem, s->z_filepos, true); in_eof = r >= 0 && r < GZ_IN - rem; }
    else {
        // a pipe: behind poll(), like the uncompressed case (a stop request is seen within 50 ms)
        r = stream_fd_read(s, s->zin + rem, GZ_IN - rem, &in_eof);
        if (r == -2) return -2;
    }
    if (r < 0) { s->z_msg = std::string("read failed: ") + strerror(errno); return -1; }
    s->z_filepos += r;
    s->zin_len += r;
    if (in_eof) s->z_in_eof = true;
    return r > 0 ? 1 : 0;
}

// Length of the BGZF member at p; 0: not one; -1: its header is not all there yet.  A BGZF member: gzip magic,
// deflate, FLG = FEXTRA alone, a "BC" subfield of two bytes.  *xlen: length of the extra field.
static int64_t bgzf_member_len(const uint8_t *p, int64_t avail, int *xlen)
{
    if (avail >= 4 && !(p[0] == 0x1f && p[1] == 0x8b && p[2] == 8 && p[3] == 4)) return 0;
    if (avail < 12) return -1;
    const int xl = p[10] | (p[11] << 8);
    if (avail < 12 + xl) return -1;
    for (int q = 0; q + 4 <= xl;) {
        const uint8_t *f = p + 12 + q;
        const int sl = f[2] | (f[3] << 8);
        if (f[0] == 'B' && f[1] == 'C' && sl == 2 && q + 6 <= xl) {
            const int64_t total = (int64_t)(f[4] | (f[5] << 8)) + 1;
            *xlen = xl;
            return total >= xl + 20 ? total : 0;
        }
        q += 4 + sl;
    }
    return 0;
}

// At a member boundary: as many whole BGZF members as zin holds and dst has room for, inflated side by side.
// Returns the bytes written (zin_pos is behind those members then), 0 if there is nothing to do here (not
// BGZF, a member that is not all there at the end of the file, no room for even one: the serial inflate
// takes it from the same place), -1 / -2 as gz_more_input.
static int64_t gz_bgzf_batch(ffq_stream *s, uint8_t *dst, int64_t room)
{
    int xl = 0;
    for (;;) {        // the first member whole in zin
        const int64_t avail = s->zin_len - s->zin_pos;
        const int64_t total = bgzf_member_len(s->zin + s->zin_pos, avail, &xl);
        if (total == 0) return 0;
        if (total > 0 && total <= avail) break;
        const int r = gz_more_input(s);
        if (r <= 0) return r;
    }
    std::vector<GzJob> &jobs = s->gz_jobs;
    jobs.clear();
    int64_t p = s->zin_pos, out = 0;
    while (p < s->zin_len) {
        const int64_t total = bgzf_member_len(s->zin + p, s->zin_len - p, &xl);
        if (total <= 0 || total > s->zin_len - p) break;
        const uint8_t *e = s->zin + p + total;
        const uint32_t crc = (uint32_t)e[-8] | ((uint32_t)e[-7] << 8) | ((uint32_t)e[-6] << 16) | ((uint32_t)e[-5] << 24);
        const uint32_t isz = (uint32_t)e[-4] | ((uint32_t)e[-3] << 8) | ((uint32_t)e[-2] << 16) | ((uint32_t)e[-1] << 24);
        if ((int64_t)isz > room - out) break;
        jobs.push_back(GzJob{s->zin + p + 12 + xl, (uint32_t)(total - xl - 20), dst + out, isz, crc});
        out += isz;
        p += total;
    }
    if (jobs.size() < 2) return 0;            // (one member: the serial inflate is as good)
    if (!s->gz_pool) {
        s->gz_pool = new (std::nothrow) GzPool();
        if (!s->gz_pool || !s->gz_pool->start(s->gz_threads - 1)) { delete s->gz_pool; s->gz_pool = nullptr; s->bgzf_ok = false; return 0; }
    }
    if (!s->zraw_init) {
        memset(&s->zraw, 0, sizeof s->zraw);
        if (inflateInit2(&s->zraw, -15) != Z_OK) { s->bgzf_ok = false; return 0; }
        s->zraw_init = true;
    }
    if (!s->gz_pool->run(jobs.data(), (int)jobs.size(), &s->zraw)) { s->bgzf_ok = false; return 0; }
    s->bgzf_members += (int64_t)jobs.size();
    s->zin_pos = p;
    return out;
}

// The compressed input goes on at file offset `off` (what zin held is dropped).
static void gz_reposition(ffq_stream *s, int64_t off)
{
    s->z_filepos = off;
    s->zin_len = s->zin_pos = 0;
    s->z_in_eof = s->gz_file_size >= 0 && off >= s->gz_file_size;
}

// At a member boundary of a regular file: the member is handed to the several-thread engine if enough of the file
// lies behind it to be worth a batch (FFQ_PGZ_MIN compressed bytes, default 4 MiB).  false: zlib takes the member.
static bool gz_pgz_begin(ffq_stream *s)
{
    if (s->gz_file_size < 0) {
        struct stat st;
        if (fstat(s->fd, &st) != 0 || !S_ISREG(st.st_mode)) { s->pgz_ok = false; return false; }
        s->gz_file_size = (int64_t)st.st_size;
    }
    int xl = 0;
    if (bgzf_member_len(s->zin + s->zin_pos, s->zin_len - s->zin_pos, &xl) != 0) return false;      // (a BGZF member: 64 KiB at most)
    const int64_t member_off = s->z_filepos - (s->zin_len - s->zin_pos);
    if (s->gz_file_size - member_off < pgz::env_i64("FFQ_PGZ_MIN", 4 << 20)) return false;
    if (!s->pgz) {
        s->pgz = new (std::nothrow) pgz::Engine();
        if (!s->pgz || !s->pgz->init(s->fd, s->gz_threads, s->gz_file_size)) { delete s->pgz; s->pgz = nullptr; s->pgz_ok = false; return false; }
    }
    if (!s->pgz->begin(member_off)) return false;
    s->pgz_member_off = member_off;
    s->pgz_active = true;
    return true;
}

// The engine gave up at a block boundary: zlib goes on from that bit with the window the engine holds, as a raw
// deflate stream; the CRC-32 and the length run on here and are compared with the trailer (gz_raw_trailer).
static bool gz_handoff(ffq_stream *s)
{
    pgz::Engine &e = *s->pgz;
    if (inflateReset2(&s->zs, -15) != Z_OK) { s->z_msg = "inflateReset failed"; return false; }
    if (e.win_valid > 0 && inflateSetDictionary(&s->zs, e.window + pgz::WSIZE - e.win_valid, (uInt)e.win_valid) != Z_OK) {
        s->z_msg = "inflateSetDictionary failed";
        return false;
    }
    int64_t byte = e.pos_bit >> 3;
    const int r = (int)(e.pos_bit & 7);
    if (r) {
        uint8_t b = 0;
        if (e.pread_full(&b, 1, byte) != 1) { s->z_msg = "compressed file ended before the end-of-stream marker was reached"; return false; }
        if (inflatePrime(&s->zs, 8 - r, b >> r) != Z_OK) { s->z_msg = "inflatePrime failed"; return false; }
        byte++;
    }
    gz_reposition(s, byte);
    s->z_raw = true; s->z_crc = e.crc; s->z_isize = e.isize;
    s->z_member = true;
    return true;
}

static bool gz_raw_trailer(ffq_stream *s)
{
    while (s->zin_len - s->zin_pos < 8) {
        if (s->z_in_eof) { s->z_msg = "compressed file ended before the end-of-stream marker was reached"; return false; }
        if (gz_more_input(s) < 0) return false;
    }
    const uint8_t *t = s->zin + s->zin_pos;
    const uint32_t fcrc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    const uint32_t flen = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    s->zin_pos += 8;
    s->z_raw = false;
    if (fcrc != s->z_crc) { s->z_msg = "incorrect data check"; return false; }
    if (flen != (uint32_t)s->z_isize) { s->z_msg = "incorrect length check"; return false; }
    return true;
}

static int64_t stream_gz_read(ffq_stream *s, uint8_t *dst, int64_t n, bool *eof)
{
    int64_t got = 0;
    *eof = false;
    while (got < n) {
        { std::lock_guard<std::mutex> lk(s->m); if (s->stop) break; }
        if (s->pgz_active) {
            got += s->pgz->read(dst + got, n - got);
            if (s->pgz->pending()) continue;                           // (the chunk is full)
            if (s->pgz->failed) { s->z_msg = s->pgz->msg; return -1; }
            if (s->pgz->member_done) {
                s->pgz_active = false;
                if (s->pgz->end_off - s->pgz_member_off < (2 << 20) && ++s->pgz_small >= 4) s->pgz_ok = false;   // (a file of small members)
                gz_reposition(s, s->pgz->end_off);
            } else if (s->pgz->gave_up) {
                s->pgz_active = false;
                if (!gz_handoff(s)) return -1;
            }
            continue;
        }
        if (s->zin_pos == s->zin_len && !s->z_in_eof) {
            const int r = gz_more_input(s);
            if (r == -2) break;                                  // asked to stop / park with nothing read
            if (r < 0) return -1;
        }
        if (!s->z_member) {
            // between members: zero padding, then the next member or the end of the file
            while (s->zin_pos < s->zin_len && s->zin[s->zin_pos] == 0) s->zin_pos++;
            if (s->zin_pos == s->zin_len) {
                if (s->z_in_eof) { *eof = true; break; }
                continue;
            }
            if (s->gz_threads > 1 && s->bgzf_ok) {
                const int64_t r = gz_bgzf_batch(s, dst + got, n - got);
                if (r == -2) break;
                if (r < 0) return -1;
                if (r > 0) { got += r; continue; }
            }
            if (s->gz_threads > 1 && s->seekable && s->pgz_ok && gz_pgz_begin(s)) continue;
            if (inflateReset2(&s->zs, 15 + 16) != Z_OK) { s->z_msg = "inflateReset failed"; return -1; }
            s->z_member = true;
        } else if (s->zin_pos == s->zin_len && s->z_in_eof) {
            s->z_msg = "compressed file ended before the end-of-stream marker was reached";
            return -1;
        }
        s->zs.next_in = s->zin + s->zin_pos;
        s->zs.avail_in = (uInt)(s->zin_len - s->zin_pos);
        s->zs.next_out = dst + got;
        s->zs.avail_out = (uInt)std::min<int64_t>(n - got, 1 << 30);
        const uInt out0 = s->zs.avail_out;
        const int zr = inflate(&s->zs, Z_NO_FLUSH);
        s->zin_pos = s->zin_len - (int64_t)s->zs.avail_in;
        if (s->z_raw) {
            const int64_t made = (int64_t)(out0 - s->zs.avail_out);
            s->z_crc = (uint32_t)crc32(s->z_crc, dst + got, (uInt)made);
            s->z_isize += (uint64_t)made;
        }
        got += (int64_t)(out0 - s->zs.avail_out);
        if (zr == Z_STREAM_END) {
            if (s->z_raw && !gz_raw_trailer(s)) return -1;
            s->z_member = false;
        }
        else if (zr != Z_OK && zr != Z_BUF_ERROR) {
            s->z_msg = s->zs.msg ? s->zs.msg : (zr == Z_DATA_ERROR ? "not a gzip file / corrupt data" : "inflate failed");
            return -1;
        }
    }
    if (got == n && !*eof && s->zin_pos == s->zin_len && s->z_in_eof && !s->z_member) *eof = true;   // ended exactly at the chunk's end
    return got;
}

// The feeder: queues the slices of chunk e as soon as slot e % STREAM_SLOTS is free (up to two
// chunks being read at a time), and completes chunks in order: once the last slice of chunk p is
// in, its H2D copy goes onto the copy stream and the caller may take it.
static void stream_feeder(ffq_stream *s)
{
    (void)hipSetDevice(s->c->device);
    StreamBufs *b = s->b;
    const int64_t pos0 = s->file_pos;
    int64_t e = 0;                        // chunks [0, e) have their slices queued (seekable descriptors)
    for (int64_t p = 0;; p++) {           // chunk to complete next
        const double tw0 = s->prof ? stream_now() : 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(s->m);
                for (;;) {
                    if (s->stop) break;
                    if (s->pause_req && e == p) {          // the caller reallocates the slots: park here (nothing in flight)
                        s->paused = true;
                        s->cv.notify_all();
                        s->cv.wait(lk);
                        continue;
                    }
                    s->paused = false;
                    if (e > p || ((s->seekable && s->src == SRC_FD) ? e : p) - s->released < STREAM_SLOTS) break;   // a read to wait for, or a free slot
                    s->cv.wait(lk);
                }
                if (s->stop) {
                    lk.unlock();
                    for (int64_t k = p; k < e; k++) b->pool->wait(&b->slot[k % STREAM_SLOTS].cr);   // reads in flight
                    lk.lock();
                    s->feeder_done = true;
                    s->cv.notify_all();
                    return;
                }
                if (s->seekable && s->src == SRC_FD && !s->pause_req && e - s->released < STREAM_SLOTS && e - p < FFQ_STREAM_AHEAD) {
                    StreamSlot &sq = b->slot[e % STREAM_SLOTS];
                    lk.unlock();
                    b->pool->enqueue(s->fd, sq.h + b->room, b->fbufsize, pos0 + e * b->fbufsize, &sq.cr);
                    e++;
                    continue;
                }
            }
            break;
        }
        StreamSlot &sl = b->slot[p % STREAM_SLOTS];
        const double tr0 = s->prof ? stream_now() : 0;
        int64_t got;
        bool src_eof = false;
        if (s->seekable && s->src == SRC_FD) { b->pool->wait(&sl.cr); got = sl.cr.total(); src_eof = got < b->fbufsize; }
        else if (s->src == SRC_GZIP) { got = stream_gz_read(s, sl.h + b->room, b->fbufsize, &src_eof); e = p + 1; }
        else {
            got = stream_fd_read(s, sl.h + b->room, b->fbufsize, &src_eof);
            if (got == -2) { p--; continue; }      // stop / pause asked for with nothing read yet: back to the parking loop
            e = p + 1;
        }
        if (s->prof) { const double t = stream_now(); s->t_slot += tr0 - tw0; s->t_read += t - tr0; }
        int rc = FFQ_OK;
        std::string msg;
        if (got < 0) {
            rc = FFQ_E_ARG;
            msg = s->src == SRC_GZIP ? std::string("ffq_stream: gzip: ") + s->z_msg : std::string("ffq_stream: read failed: ") + strerror(errno);
        } else {
            sl.got = got;
            sl.eof = src_eof;
            const hipError_t er = stream_copy_chunk(b, sl, got);
            if (er != hipSuccess) { rc = FFQ_E_HIP; msg = std::string("ffq_stream: chunk copy failed: ") + hipGetErrorString(er); }
        }
        if (rc || sl.eof)
            for (int64_t k = p + 1; k < e; k++) b->pool->wait(&b->slot[k % STREAM_SLOTS].cr);    // reads past the end
        {
            std::lock_guard<std::mutex> lk(s->m);
            if (rc) { s->feeder_rc = rc; s->feeder_msg = msg; s->feeder_done = true; }
            else {
                s->file_pos = s->src != SRC_GZIP ? s->file_pos + got :
                              s->pgz_active ? (s->pgz->pos_bit >> 3) : s->z_filepos - (s->zin_len - s->zin_pos);      // (inside a member the engine inflates: the block boundary it has committed)
                sl.end_pos = s->file_pos;
                s->produced = p + 1;
                if (sl.eof) s->feeder_done = true;
            }
        }
        s->cv.notify_all();
        if (rc || sl.eof) return;
    }
}

static void stream_stop_feeder(ffq_stream *s)
{
    if (!s->feeder.joinable()) return;
    { std::lock_guard<std::mutex> lk(s->m); s->stop = true; }
    s->cv.notify_all();
    s->feeder.join();
}

static void stream_free(ffq_stream *s)
{
    if (!s) return;
    stream_stop_feeder(s);
    if (s->prof)
        fprintf(stderr, "[ffq stream] %lld fills: reader %.3f ms reading, %.3f ms waiting for a slot; caller %.3f ms waiting "
                        "for the reader, %.3f ms carry + scan (of which %.3f ms waiting for the chunk's copy), %.3f ms rows back\n",
                (long long)(s->cur + 1), s->t_read * 1e3, s->t_slot * 1e3, s->t_feed * 1e3, s->t_scan * 1e3, s->t_copy * 1e3,
                s->t_rows * 1e3);
    if (s->z_init) (void)inflateEnd(&s->zs);
    delete s->gz_pool;
    delete s->pgz;
    if (s->zraw_init) (void)inflateEnd(&s->zraw);
    free(s->zin);
    if (s->b) {
        for (auto &st : s->b->cs) if (st) (void)hipStreamSynchronize(st);
        if (s->c->stream) (void)hipStreamSynchronize(s->c->stream);
        // park the buffers in the context for the next stream
        if (!s->c->stream_cache) s->c->stream_cache = s->b;
        else streambufs_free(s->b);
    }
    delete s;
}

static int stream_alloc_tab(ffq_stream *s, int64_t rows)
{
    StreamBufs *b = s->b;
    NearGpu near(s->c);
    if (rows > b->tab_cap) {
        if (b->htab) (void)hipHostFree(b->htab);
        (void)hipFree(b->dtab);
        b->htab = nullptr; b->dtab = nullptr; b->tab_cap = 0;
        if (hipMalloc((void **)&b->dtab, (size_t)rows * 48) != hipSuccess ||
            hipHostMalloc((void **)&b->htab, (size_t)rows * 48, hipHostMallocDefault) != hipSuccess)
            return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld rows", (long long)rows);
        b->tab_cap = rows;
    }
    if ((s->flags & FFQ_F_DECODE_QUAL) && b->qoff_cap < b->tab_cap + 1) {
        if (b->hqoff) (void)hipHostFree(b->hqoff);
        (void)hipFree(b->dqoff);
        b->hqoff = nullptr; b->dqoff = nullptr; b->qoff_cap = 0;
        if (hipMalloc((void **)&b->dqoff, (size_t)(b->tab_cap + 1) * 8) != hipSuccess ||
            hipHostMalloc((void **)&b->hqoff, (size_t)(b->tab_cap + 1) * 8, hipHostMallocDefault) != hipSuccess)
            return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld quality offsets", (long long)b->tab_cap);
        b->qoff_cap = b->tab_cap + 1;
    }
    return FFQ_OK;
}

// device / pinned buffers of the push-down for a fill of up to `rows` rows (and `bytes` of column)
static int stream_alloc_sel(ffq_stream *s, int64_t rows, int64_t bytes)
{
    StreamBufs *b = s->b;
    if (rows > b->sel_cap) {
        if (b->hidx) (void)hipHostFree(b->hidx);
        (void)hipFree(b->dsel); (void)hipFree(b->didx);
        b->hidx = nullptr; b->dsel = nullptr; b->didx = nullptr; b->sel_cap = 0;
        if (hipMalloc((void **)&b->dsel, (size_t)rows * 48) != hipSuccess || hipMalloc((void **)&b->didx, (size_t)rows * 8) != hipSuccess ||
            hipHostMalloc((void **)&b->hidx, (size_t)rows * 8, hipHostMallocDefault) != hipSuccess)
            return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld selected rows", (long long)rows);
        b->sel_cap = rows;
    }
    if (s->f_col && rows + 1 > b->coff_cap) {
        if (b->hcoff) (void)hipHostFree(b->hcoff);
        (void)hipFree(b->dcoff);
        b->hcoff = nullptr; b->dcoff = nullptr; b->coff_cap = 0;
        if (hipMalloc((void **)&b->dcoff, (size_t)(rows + 1) * 8) != hipSuccess ||
            hipHostMalloc((void **)&b->hcoff, (size_t)(rows + 1) * 8, hipHostMallocDefault) != hipSuccess)
            return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld column offsets", (long long)rows);
        b->coff_cap = rows + 1;
    }
    if (s->f_col && bytes > b->col_cap) {
        if (b->hcol) (void)hipHostFree(b->hcol);
        (void)hipFree(b->dcol);
        b->hcol = nullptr; b->dcol = nullptr; b->col_cap = 0;
        if (hipMalloc((void **)&b->dcol, (size_t)bytes) != hipSuccess || hipHostMalloc((void **)&b->hcol, (size_t)bytes, hipHostMallocDefault) != hipSuccess)
            return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld column bytes", (long long)bytes);
        b->col_cap = bytes;
    }
    return FFQ_OK;
}

static int stream_alloc_qual(ffq_stream *s, int64_t bytes)
{
    StreamBufs *b = s->b;
    if (bytes <= b->qual_cap) return FFQ_OK;
    if (b->hqual) (void)hipHostFree(b->hqual);
    (void)hipFree(b->dqual);
    b->hqual = nullptr; b->dqual = nullptr; b->qual_cap = 0;
    if (hipMalloc((void **)&b->dqual, (size_t)bytes) != hipSuccess ||
        hipHostMalloc((void **)&b->hqual, (size_t)bytes, hipHostMallocDefault) != hipSuccess)
        return fail(FFQ_E_NOMEM, "ffq_stream: no memory for %lld decoded bytes", (long long)bytes);
    b->qual_cap = bytes;
    return FFQ_OK;
}

// A carry that does not fit in front of a chunk (a record longer than `room`): every slot is
// reallocated with more room.  The feeder is parked first; chunks it has read ahead are moved
// and their copy is enqueued again.  Called at the START of ffq_stream_next, before any pointer
// of the new fill is handed out -- the previous fill's pointers expire with this call anyway.
static int stream_grow_room(ffq_stream *s, int64_t need)
{
    StreamBufs *b = s->b;
    if (s->src != SRC_PUSH) {
        std::unique_lock<std::mutex> lk(s->m);
        s->pause_req = true;
        s->cv.notify_all();
        s->cv.wait(lk, [&] { return s->paused || s->feeder_done; });
    }
    int rc = FFQ_OK;
    hipError_t e = hipStreamSynchronize(b->cs[0]);
    if (e == hipSuccess) e = hipStreamSynchronize(b->cs[1]);
    if (e != hipSuccess) rc = fail(FFQ_E_HIP, "ffq_stream: %s", hipGetErrorString(e));
    const int64_t room = (std::max<int64_t>(2 * b->room, need + 4096) + 4095) & ~(int64_t)4095;
    StreamSlot old[STREAM_SLOTS];
    for (int i = 0; i < STREAM_SLOTS; i++) { old[i] = b->slot[i]; b->slot[i].h = nullptr; b->slot[i].d = nullptr; }
    const int64_t old_room = b->room;
    if (!rc) rc = streambufs_alloc_slots(s->c, b, room);
    if (!rc) {
        // chunks [released, produced) are alive: the one being carried from (cur, all of its fill)
        // and the ones read ahead (their chunk bytes)
        for (int64_t k = std::max<int64_t>(s->released, 0); k < s->produced && !rc; k++) {
            StreamSlot &n = b->slot[k % STREAM_SLOTS];
            const StreamSlot &o = old[k % STREAM_SLOTS];
            if (k == s->cur) {
                memcpy(n.h + room - (old_room - s->fill_start), o.h + s->fill_start, (size_t)s->fill_len);
                s->fill_start = room - (old_room - s->fill_start);
            } else {
                memcpy(n.h + room, o.h + old_room, (size_t)o.got);
                e = stream_copy_chunk(b, n, o.got);
                if (e != hipSuccess) rc = fail(FFQ_E_HIP, "ffq_stream: %s", hipGetErrorString(e));
            }
        }
    }
    for (auto &o : old) { if (o.h) (void)hipHostFree(o.h); (void)hipFree(o.d); }
    {
        std::lock_guard<std::mutex> lk(s->m);
        s->pause_req = false;
    }
    s->cv.notify_all();
    return rc;
}

static int stream_open_impl(ffq_ctx *c, int src, int fd, int64_t fbufsize, uint32_t flags, int qual_add, int64_t start,
                            ffq_stream **out)
{
    if (!c || !out || (src != SRC_PUSH && fd < 0) || fbufsize <= 0) return fail(FFQ_E_ARG, "ffq_stream_open: bad argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    ffq_stream *s = new (std::nothrow) ffq_stream();
    if (!s) return fail(FFQ_E_NOMEM, "out of host memory");
    s->c = c; s->fd = fd; s->src = src;
    // (FFQ_F_SINGLE_PASS: the fill's qualities come SEGMENTED -- ffq_stream_quals -- from the pass that builds the index)
    s->flags = flags & (FFQ_F_DECODE_QUAL | ((flags & FFQ_F_DECODE_QUAL) ? FFQ_F_SINGLE_PASS : 0u)); s->qual_add = qual_add;
    s->prof = getenv("FFQ_STREAM_PROF") != nullptr;
    if (src == SRC_PUSH) s->feeder_done = true;          // there is no reader thread: chunks are pushed
    else {
        // where to read from: `start` (pread; the descriptor's own position is left alone), or the
        // descriptor's current position if start < 0; a descriptor that cannot seek is read in order
        const off_t at = lseek(fd, 0, SEEK_CUR);
        s->seekable = at != (off_t)-1;
        s->file_pos = s->seekable ? (start >= 0 ? start : (int64_t)at) : 0;
        s->handed_pos = s->file_pos;
    }
    if (src == SRC_GZIP) {
        memset(&s->zs, 0, sizeof s->zs);
        s->zin = static_cast<uint8_t *>(malloc((size_t)GZ_IN + 16));      // (+16: the decoders read whole words)
        if (!s->zin || inflateInit2(&s->zs, 15 + 16) != Z_OK) {      // 16: gzip wrapper (header, CRC-32, length)
            free(s->zin); delete s;
            return fail(FFQ_E_NOMEM, "ffq_stream_open: zlib could not be initialised");
        }
        s->z_init = true;
        s->z_filepos = s->file_pos;
        s->gz_threads = gz_threads_default();
    }
    StreamBufs *b = static_cast<StreamBufs *>(c->stream_cache);
    c->stream_cache = nullptr;
    if (b && b->fbufsize != fbufsize) { streambufs_free(b); b = nullptr; }
    int rc = FFQ_OK;
    if (!b) {
        b = new (std::nothrow) StreamBufs();
        if (!b) { delete s; return fail(FFQ_E_NOMEM, "out of host memory"); }
        b->fbufsize = fbufsize;
        hipError_t e = hipSuccess;
        for (auto &st : b->cs)
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (auto &sl : b->slot)
            for (auto &ev : sl.copied)
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) rc = fail(FFQ_E_HIP, "ffq_stream_open: %s", hipGetErrorString(e));
        if (!rc) rc = streambufs_alloc_slots(c, b, 1 << 20);
    }
    if (!rc && !(b->pool = ctx_pool(c))) rc = fail(FFQ_E_NOMEM, "out of host memory");
    {
        const char *e1 = getenv("FFQ_STREAM_ONE_COPY_STREAM");                 // (measurements: 0 / 1 whatever the stream does)
        b->one_copy_stream = e1 ? atoi(e1) != 0 : (flags & FFQ_F_DECODE_QUAL) != 0;
    }
    s->b = b;
    if (!rc) rc = stream_alloc_tab(s, fbufsize / 64 + 1024);
    if (rc) { stream_free(s); return rc; }
    if (src != SRC_PUSH) s->feeder = std::thread(stream_feeder, s);
    *out = s;
    return FFQ_OK;
}

extern "C" int ffq_stream_open2(ffq_ctx *c, int fd, int64_t fbufsize, uint32_t flags, int qual_add, int64_t start,
                                ffq_stream **out)
{
    return stream_open_impl(c, SRC_FD, fd, fbufsize, flags, qual_add, start, out);
}

extern "C" int ffq_stream_open(ffq_ctx *c, int fd, int64_t fbufsize, ffq_stream **out)
{
    return stream_open_impl(c, SRC_FD, fd, fbufsize, 0, 0, -1, out);
}

extern "C" int ffq_stream_open_gzip(ffq_ctx *c, int fd, int64_t fbufsize, uint32_t flags, int qual_add, int64_t start,
                                    ffq_stream **out)
{
    return stream_open_impl(c, SRC_GZIP, fd, fbufsize, flags, qual_add, start, out);
}

// The gzip reader on its own: the file behind fd (from its current position; a pipe works), inflated into
// host memory chunk by chunk as the stream's reader thread does it.  Returns the bytes written, or FFQ_E_*.
extern "C" int64_t ffq_gunzip_fd(int fd, uint8_t *h_dst, int64_t cap, int64_t chunk, int threads, int64_t *n_parallel_members)
{
    if (fd < 0 || (!h_dst && cap > 0) || cap < 0 || chunk <= 0) return fail(FFQ_E_ARG, "ffq_gunzip_fd: bad argument");
    ffq_stream *s = new (std::nothrow) ffq_stream();
    if (!s) return fail(FFQ_E_NOMEM, "out of host memory");
    s->fd = fd; s->src = SRC_GZIP;
    const off_t at = lseek(fd, 0, SEEK_CUR);
    s->seekable = at != (off_t)-1;
    s->z_filepos = s->seekable ? (int64_t)at : 0;
    memset(&s->zs, 0, sizeof s->zs);
    s->zin = static_cast<uint8_t *>(malloc((size_t)GZ_IN + 16));      // (+16: the decoders read whole words)
    int64_t rc = FFQ_OK, got = 0;
    if (!s->zin || inflateInit2(&s->zs, 15 + 16) != Z_OK) rc = fail(FFQ_E_NOMEM, "ffq_gunzip_fd: zlib could not be initialised");
    else {
        s->z_init = true;
        s->gz_threads = threads > 0 ? std::min(threads, 256) : gz_threads_default();
        bool eof = false;
        uint8_t one[1];
        while (!eof) {
            const bool full = got == cap;
            const int64_t r = stream_gz_read(s, full ? one : h_dst + got, full ? 1 : std::min(chunk, cap - got), &eof);
            if (r < 0) { rc = fail(FFQ_E_ARG, "ffq_gunzip_fd: gzip: %s", s->z_msg.c_str()); break; }
            if (full && r > 0) { rc = fail(FFQ_E_TABLE_FULL, "ffq_gunzip_fd: the file inflates to more than %lld bytes", (long long)cap); break; }
            got += full ? 0 : r;
        }
    }
    if (n_parallel_members) *n_parallel_members = s->bgzf_members;
    if (s->z_init) (void)inflateEnd(&s->zs);
    delete s->gz_pool;
    delete s->pgz;
    if (s->zraw_init) (void)inflateEnd(&s->zraw);
    free(s->zin);
    delete s;
    return rc ? rc : got;
}

extern "C" void ffq_gunzip_stats(int64_t out[5])
{
    pgz::Stats &st = pgz::stats();
    out[0] = st.batches.load(); out[1] = st.chunks.load(); out[2] = st.rejected.load(); out[3] = st.giveups.load(); out[4] = st.members.load();
}

extern "C" int ffq_stream_open_push(ffq_ctx *c, int64_t fbufsize, uint32_t flags, int qual_add, ffq_stream **out)
{
    return stream_open_impl(c, SRC_PUSH, -1, fbufsize, flags, qual_add, -1, out);
}

// push mode: where the next chunk's bytes go (pinned memory, fbufsize bytes of room) ...
extern "C" int ffq_stream_push_buffer(ffq_stream *s, uint8_t **dst, int64_t *cap)
{
    if (!s || !dst || !cap || s->src != SRC_PUSH) return fail(FFQ_E_ARG, "ffq_stream_push_buffer: not a push stream");
    if (s->failed || s->done) return fail(FFQ_E_ARG, "ffq_stream_push_buffer: the stream has %s", s->failed ? "failed" : "ended");
    if (s->produced - s->released >= STREAM_SLOTS) return fail(FFQ_E_ARG, "ffq_stream_push_buffer: every slot holds an unconsumed chunk");
    *dst = s->b->slot[s->produced % STREAM_SLOTS].h + s->b->room;
    *cap = s->b->fbufsize;
    return FFQ_OK;
}

// ... and: n bytes are there now; eof: nothing follows.  The chunk's copy to the device is enqueued.
extern "C" int ffq_stream_push(ffq_stream *s, int64_t n, int eof)
{
    if (!s || s->src != SRC_PUSH) return fail(FFQ_E_ARG, "ffq_stream_push: not a push stream");
    if (n < 0 || n > s->b->fbufsize) return fail(FFQ_E_ARG, "ffq_stream_push: %lld bytes do not fit a chunk of %lld", (long long)n, (long long)s->b->fbufsize);
    if (s->failed || s->done) return fail(FFQ_E_ARG, "ffq_stream_push: the stream has %s", s->failed ? "failed" : "ended");
    if (s->produced - s->released >= STREAM_SLOTS) return fail(FFQ_E_ARG, "ffq_stream_push: every slot holds an unconsumed chunk");
    HIPCHK(hipSetDevice(s->c->device));
    StreamSlot &sl = s->b->slot[s->produced % STREAM_SLOTS];
    sl.got = n; sl.eof = eof != 0;
    s->file_pos += n;
    sl.end_pos = s->file_pos;
    const hipError_t e = stream_copy_chunk(s->b, sl, n);
    if (e != hipSuccess) { s->failed = true; return fail(FFQ_E_HIP, "ffq_stream_push: chunk copy failed: %s", hipGetErrorString(e)); }
    s->produced++;
    return FFQ_OK;
}

extern "C" void ffq_stream_close(ffq_stream *s) { stream_free(s); }

extern "C" int64_t ffq_stream_tell(ffq_stream *s)
{
    if (!s) return -1;
    // behind the last chunk HANDED OUT (the reader runs up to two chunks ahead of that): where the
    // reference's loop would have left the file after the same fills
    return s->handed_pos;
}

// ffq_scan_result.path of the last fill's scan (6: index and decoded qualities in one pass)
extern "C" int ffq_stream_path(ffq_stream *s) { return s ? s->last_path : -1; }

// decoded qualities of the fill ffq_stream_next has just returned (streams opened with
// FFQ_F_DECODE_QUAL): int8 bytes + offsets (n_rows + 1), pinned, valid until the next call.  Record i's bytes are
// h_qual[h_qoff[i] : h_qoff[i] + pos5(i) - pos4(i)] -- packed back to back (then h_qoff[i + 1] is where they end), or,
// for a stream opened with FFQ_F_SINGLE_PASS whose fill the single pass took, with gaps between the index tiles'
// segments (include/ffq.h, FFQ_F_SINGLE_PASS); h_qoff[n_rows] = *n_qual_bytes = where the last record's bytes end.
extern "C" int ffq_stream_quals(ffq_stream *s, const int8_t **h_qual, const int64_t **h_qoff, int64_t *n_qual_bytes)
{
    if (!s || !h_qual || !h_qoff || !n_qual_bytes) return fail(FFQ_E_ARG, "ffq_stream_quals: NULL argument");
    if (!(s->flags & FFQ_F_DECODE_QUAL)) return fail(FFQ_E_ARG, "ffq_stream_quals: the stream was opened without FFQ_F_DECODE_QUAL");
    *h_qual = s->b->hqual; *h_qoff = s->b->hqoff; *n_qual_bytes = s->last_nq;
    return FFQ_OK;
}

// Push-down into the stream (the reference's user guide, doc/user-guide.rst:153-180: an entryfunc that looks at the read's
// length and builds one component of the entries it keeps): from the next fill on, ffq_stream_next hands back only the rows
// with min_seq_len <= pos3 - pos2 <= max_seq_len, in order, and -- column != 0 -- that component of every kept row as a
// packed stream (+ value_add: -33 on the quality is the Phred decode of the kept records only).
extern "C" int ffq_stream_set_filter(ffq_stream *s, int64_t min_seq_len, int64_t max_seq_len, int column, int value_add)
{
    if (!s) return fail(FFQ_E_ARG, "ffq_stream_set_filter: NULL stream");
    if (column < 0 || column > 3) return fail(FFQ_E_ARG, "ffq_stream_set_filter: column is FFQ_COL_NONE / _HEADER / _SEQUENCE / _QUALITY");
    if (s->flags & FFQ_F_DECODE_QUAL) return fail(FFQ_E_ARG, "ffq_stream_set_filter: the stream decodes every record's qualities (FFQ_F_DECODE_QUAL); "
                                                              "a filtered stream gathers the kept records' (column = FFQ_COL_QUALITY, value_add)");
    s->filter_on = true;
    s->f_min = min_seq_len; s->f_max = max_seq_len; s->f_col = column; s->f_add = value_add;
    return FFQ_OK;
}

// What the filter did with the fill ffq_stream_next has just returned: h_index[i] = ordinal, among the n_scanned records of
// the fill, of kept row i (the caller that owes its own caller one item per record puts the kept ones back by it); the
// gathered column: bytes of kept row i = h_col[h_coloff[i] : h_coloff[i + 1]].  Pinned memory, valid until the next call.
extern "C" int ffq_stream_selected(ffq_stream *s, const int64_t **h_index, int64_t *n_scanned, const int8_t **h_col,
                                   const int64_t **h_coloff, int64_t *n_col_bytes)
{
    if (!s || !h_index || !n_scanned) return fail(FFQ_E_ARG, "ffq_stream_selected: NULL argument");
    if (!s->filter_on) return fail(FFQ_E_ARG, "ffq_stream_selected: the stream has no filter (ffq_stream_set_filter)");
    *h_index = s->b->hidx; *n_scanned = s->last_scanned;
    if (h_col) *h_col = s->f_col ? s->b->hcol : nullptr;
    if (h_coloff) *h_coloff = s->f_col ? s->b->hcoff : nullptr;
    if (n_col_bytes) *n_col_bytes = s->last_col_bytes;
    return FFQ_OK;
}

extern "C" int ffq_stream_next(ffq_stream *s, const int64_t **h_rows, int64_t *n_rows, int *end_state,
                               int64_t *err_offset, const uint8_t **h_bytes, int64_t *n_bytes,
                               int64_t *bytes_offset)
{
    if (!s || !h_rows || !n_rows || !end_state) return fail(FFQ_E_ARG, "ffq_stream_next: NULL argument");
    ffq_ctx *c = s->c;
    StreamBufs *b = s->b;
    HIPCHK(hipSetDevice(c->device));
    *h_rows = b->htab; *n_rows = 0; *end_state = FFQ_END_OK;
    if (err_offset) *err_offset = -1;
    if (h_bytes) *h_bytes = nullptr;
    if (n_bytes) *n_bytes = 0;
    if (bytes_offset) *bytes_offset = 0;
    if (s->failed) return fail(FFQ_E_ARG, "ffq_stream_next: the stream has failed");
    if (s->done) return FFQ_OK;
    s->failed = true;                    // until this call gets through: an error leaves no half-advanced state behind

    // ---- chunk k: wait for the feeder -------------------------------------------------------
    const int64_t k = s->cur + 1;
    const double tp0 = s->prof ? stream_now() : 0;
    {
        std::unique_lock<std::mutex> lk(s->m);
        s->cv.wait(lk, [&] { return s->produced > k || s->feeder_done; });
        if (s->produced <= k) {
            if (s->feeder_rc) return fail(s->feeder_rc, "%s", s->feeder_msg.c_str());
            if (s->src == SRC_PUSH) { s->failed = false; return fail(FFQ_E_ARG, "ffq_stream_next: no chunk has been pushed"); }
            return fail(FFQ_E_INTERNAL, "ffq_stream: the reader stopped early");
        }
    }
    const double tp1 = s->prof ? stream_now() : 0;
    // ---- buf = buf[offset:] + chunk (:277); the first fill: b'\n' + chunk (:245) -----------------
    int64_t carry;
    if (k == 0) carry = 1;
    else {
        carry = s->fill_len - s->carry_from;
        if (carry > b->room) {
            int rc = stream_grow_room(s, carry);
            if (rc) return rc;
        }
    }
    StreamSlot &sl = b->slot[k % STREAM_SLOTS];
    const int64_t room = b->room;
    if (k == 0) sl.h[room - 1] = (uint8_t)'\n';
    else {
        const StreamSlot &pv = b->slot[s->cur % STREAM_SLOTS];
        memcpy(sl.h + room - carry, pv.h + s->fill_start + s->carry_from, (size_t)carry);
        // the previous fill's slot goes back to the feeder (the caller's pointers into it expired
        // with this call)
        { std::lock_guard<std::mutex> lk(s->m); s->released = k; }
        s->cv.notify_all();
    }
    const int64_t start = room - carry, len = carry + sl.got;
    const bool fill_eof = sl.eof;
    mark_other(c);          // (a copy and two event waits go onto the scan stream in front of the scan)
    HIPCHK(hipMemcpyAsync(sl.d + start, sl.h + start, (size_t)carry, hipMemcpyHostToDevice, c->stream));
    if (s->prof) {          // (profiling only: the wait for the chunk's copy on its own)
        const double t = stream_now();
        HIPCHK(hipEventSynchronize(sl.copied[0]));
        HIPCHK(hipEventSynchronize(sl.copied[1]));
        s->t_copy += stream_now() - t;
    }
    HIPCHK(hipStreamWaitEvent(c->stream, sl.copied[0], 0));
    HIPCHK(hipStreamWaitEvent(c->stream, sl.copied[1], 0));

    // ---- scan: from the aligned address below the fill, searching from the fill's first byte ----
    const int64_t mis = start & 15;
    const bool decode = (s->flags & FFQ_F_DECODE_QUAL) != 0;
    if (decode) {
        // qualities are at most half of the bytes; the segmented layout of the single pass owns FFQ_SEG_STRIDE bytes per tile
        int64_t need = len / 2 + 64;
        if (s->flags & FFQ_F_SINGLE_PASS) need = std::max<int64_t>(need, ((len + mis + 16383) >> 14) * (int64_t)FFQ_SEG_STRIDE);
        int rc2 = stream_alloc_qual(s, need);
        if (rc2) return rc2;
    }
    ffq_scan_result res;
    memset(&res, 0, sizeof res);
    int rc = FFQ_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
        rc = ffq_scan_device(c, sl.d + start - mis, len + mis, 0, mis, fill_eof ? 1 : 0, s->globaloffset - mis, s->flags,
                             s->qual_add, b->dtab, b->tab_cap, decode ? b->dqual : nullptr, decode ? b->qual_cap : 0,
                             decode ? b->dqoff : nullptr, &res);
        if (rc != FFQ_E_TABLE_FULL) break;
        int rc2 = stream_alloc_tab(s, res.n_records + 1024);
        if (rc2) return rc2;
    }
    if (rc != FFQ_OK) return rc;
    const double tp2 = s->prof ? stream_now() : 0;
    int64_t n_out = res.n_records;
    s->last_scanned = res.n_records; s->last_kept = res.n_records; s->last_col_bytes = 0;
    if (s->filter_on) {
        // ---- push-down: the fill's table is filtered (and one component of the kept rows gathered) on the DEVICE, before
        // anything is copied back: a dropped record costs the host nothing (doc/user-guide.rst:153-180) -----------------
        n_out = 0;
        if (res.n_records > 0) {
            int rc2 = stream_alloc_sel(s, res.n_records, len + 64);
            if (rc2) return rc2;
            rc2 = table_select(c, b->dtab, res.n_records, s->f_min, s->f_max, b->dsel, b->didx, &n_out);
            if (rc2) return rc2;
        }
        s->last_kept = n_out;
        if (n_out > 0) {
            HIPCHK(hipMemcpyAsync(b->htab, b->dsel, (size_t)n_out * 48, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(b->hidx, b->didx, (size_t)n_out * 8, hipMemcpyDeviceToHost, c->stream));
        }
        if (s->f_col && res.n_records > 0) {
            static const int COLS[4][3] = {{0, 0, 0}, {0, 1, 1}, {2, 0, 3}, {4, 0, 5}};       // header: buf[pos0 + 1 : pos1] (:161-171)
            int64_t nb = 0;
            // rows are stream offsets: buffer coordinate = row - add, with add = globaloffset - mis as the scan was given
            int rc2 = ffq_table_gather_column(c, sl.d + start - mis, len + mis, 0, s->globaloffset - mis, b->dsel, n_out, COLS[s->f_col][0],
                                              COLS[s->f_col][1], COLS[s->f_col][2], s->f_add, b->dcol, b->col_cap, b->dcoff, &nb);
            if (rc2) return rc2;
            s->last_col_bytes = nb;
            if (nb > 0) HIPCHK(hipMemcpyAsync(b->hcol, b->dcol, (size_t)nb, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(b->hcoff, b->dcoff, (size_t)(n_out + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        }
    } else if (res.n_records > 0)
        HIPCHK(hipMemcpyAsync(b->htab, b->dtab, (size_t)res.n_records * 48, hipMemcpyDeviceToHost, c->stream));
    s->last_nq = 0;
    s->last_path = res.path;
    if (decode) {
        s->last_nq = res.n_qual_bytes;
        if (res.n_qual_bytes > 0)
            HIPCHK(hipMemcpyAsync(b->hqual, b->dqual, (size_t)res.n_qual_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(b->hqoff, b->dqoff, (size_t)(res.n_records + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (s->prof) { const double t = stream_now(); s->t_feed += tp1 - tp0; s->t_scan += tp2 - tp1; s->t_rows += t - tp2; }

    s->cur = k;
    s->handed_pos = sl.end_pos;
    s->fill_start = start; s->fill_len = len;
    *h_rows = b->htab;
    *n_rows = n_out;
    if (h_bytes) *h_bytes = sl.h + start;
    if (n_bytes) *n_bytes = len;
    // byte i of this fill is stream offset globaloffset + i (the sentinel of the first fill is -1)
    if (bytes_offset) *bytes_offset = s->globaloffset;
    *end_state = res.end_state;
    const int64_t ds = res.end_offset - mis;             // the iterator's `offset` at exit, fill-relative
    s->failed = false;
    if (res.end_state != FFQ_END_REFILL) {
        if (res.end_state != FFQ_END_OK && err_offset) *err_offset = s->globaloffset + ds;
        s->done = true;
        stream_stop_feeder(s);
        return FFQ_OK;
    }
    // refill at the next call: buf = buf[offset:] + next chunk (:277), globaloffset += offset (:275)
    s->carry_from = ds;
    s->globaloffset += ds;
    return FFQ_OK;
}
