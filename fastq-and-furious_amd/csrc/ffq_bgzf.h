// ffq_bgzf.h -- a byte RANGE of a BGZF file inflated on its own (host code of libffq_hip.so; included by ffq_hip.hip
// behind ffq_stream.h, whose member parser and inflate pool it uses).
//
// A BGZF file (what bgzip writes; SAM specification section 4.1) is a series of gzip members of at most 64 KiB of data
// each, every one saying in its header how long it is (BSIZE) and in its trailer what it inflates to (ISIZE): the
// natural compressed shard.  k ranks read ONE such file together the way they read a plain one (sharded.FileShard):
// rank r takes the members whose FIRST byte lies in its share [c_lo, c_hi) of the compressed file -- the first one is
// found by its signature and proven by the chain of headers behind it, nothing is inflated to find it --, learns from
// the trailers how many bytes they hold, the ranks add those up (one exchange of a number per rank) and have the cut
// points of the UNCOMPRESSED stream; then every rank inflates its members side by side (GzPool) and the ordinary
// sharded step runs over the inflated ranges, halos handed over between the ranks.  Offsets in every row are offsets of
// the uncompressed stream, as for the reference's loop over gzip.open(...) (/root/reference/src/fastqandfurious.py:241-279;
// doc/user-guide.rst opens its files that way).
#pragma once

namespace bgzf_range {

constexpr int64_t WINDOW = 32 << 20;          // compressed bytes read at a time
constexpr int CHAIN = 4;                      // headers that must line up behind a candidate member start

static int64_t pread_full(int fd, uint8_t *dst, int64_t n, int64_t pos)
{
    int64_t got = 0;
    while (got < n) {
        const ssize_t r = pread(fd, dst + got, (size_t)(n - got), (off_t)(pos + got));
        if (r < 0) { if (errno == EINTR) continue; return -1; }
        if (r == 0) break;
        got += r;
    }
    return got;
}

// the member at file offset p: its total length, or 0 (not a BGZF member / not all of its header there)
static int64_t member_at(int fd, int64_t p, int64_t size)
{
    uint8_t h[512];
    const int64_t n = pread_full(fd, h, std::min<int64_t>((int64_t)sizeof h, size - p), p);
    if (n < 18) return 0;
    int xl = 0;
    const int64_t total = bgzf_member_len(h, n, &xl);
    return total > 0 && p + total <= size ? total : 0;
}

// is p the start of a member?  CHAIN headers line up behind it (or the file ends exactly behind one of them)
static bool chain_ok(int fd, int64_t p, int64_t size)
{
    for (int i = 0; i < CHAIN; i++) {
        if (p == size) return i > 0;
        const int64_t t = member_at(fd, p, size);
        if (t <= 0) return false;
        p += t;
    }
    return true;
}

// the first member start at or behind `from`; `size` if there is none; -1: bytes that are no BGZF where a member must be
static int64_t first_member(int fd, int64_t from, int64_t size)
{
    if (from >= size) return size;
    if (from == 0) return chain_ok(fd, 0, size) ? 0 : -1;
    // a member is at most 64 KiB long: one starts within that distance of any byte of the file
    const int64_t span = std::min<int64_t>(size - from, 65536 + 4);
    std::vector<uint8_t> w((size_t)span);
    if (pread_full(fd, w.data(), span, from) != span) return -1;
    for (int64_t i = 0; i + 4 <= span; i++)
        if (w[(size_t)i] == 0x1f && w[(size_t)i + 1] == 0x8b && w[(size_t)i + 2] == 8 && w[(size_t)i + 3] == 4 && chain_ok(fd, from + i, size))
            return from + i;
    return size - from <= 65536 ? size : -1;          // (the tail of the last member: nothing starts here)
}

}  // namespace bgzf_range

namespace bgzf_range {

// the members of a range as they are met, and their inflation side by side
struct Walk {
    uint8_t *h_dst;
    int64_t cap;
    int64_t out = 0, members = 0;
    std::vector<GzJob> jobs;
    GzPool *pool = nullptr;
    z_stream own;
    bool own_init = false;
    int nthreads;

    Walk(uint8_t *dst, int64_t cap_, int threads) : h_dst(dst), cap(cap_), nthreads(threads > 0 ? std::min(threads, 256) : gz_threads_default())
    {
        memset(&own, 0, sizeof own);
    }
    ~Walk()
    {
        delete pool;
        if (own_init) (void)inflateEnd(&own);
    }
    // the member of `avail` readable bytes at mem = byte `at` of the file (size bytes long): its length through *total, or an error
    int take(const uint8_t *mem, int64_t avail, int64_t at, int64_t size, int64_t *total_out)
    {
        int xl = 0;
        const int64_t total = bgzf_member_len(mem, avail, &xl);
        *total_out = total;
        if (total == 0) return fail(FFQ_E_ARG, "ffq_bgzf_range: gzip: what follows byte %lld is not a BGZF member", (long long)at);
        if (total < 0 || total > avail) {
            if (at + std::max<int64_t>(total, 18) > size)
                return fail(FFQ_E_ARG, "ffq_bgzf_range: gzip: compressed file ended before the end-of-stream marker was reached (the BGZF member at byte %lld is cut short)", (long long)at);
            return 1;                       // (all there in the file, not in what the caller holds of it)
        }
        const uint8_t *e = mem + total;
        const uint32_t crc = (uint32_t)e[-8] | ((uint32_t)e[-7] << 8) | ((uint32_t)e[-6] << 16) | ((uint32_t)e[-5] << 24);
        const uint32_t isz = (uint32_t)e[-4] | ((uint32_t)e[-3] << 8) | ((uint32_t)e[-2] << 16) | ((uint32_t)e[-1] << 24);
        if (isz > 65536) return fail(FFQ_E_ARG, "ffq_bgzf_range: gzip: the member at byte %lld says it holds %u bytes (a BGZF member holds at most 65536)", (long long)at, isz);
        if (h_dst) {
            if (out + (int64_t)isz > cap) return fail(FFQ_E_TABLE_FULL, "ffq_bgzf_range: the range inflates to more than %lld bytes", (long long)cap);
            try { jobs.push_back(GzJob{mem + 12 + xl, (uint32_t)(total - xl - 20), h_dst + out, isz, crc}); }
            catch (const std::bad_alloc &) { return fail(FFQ_E_NOMEM, "out of host memory"); }
        }
        out += isz;
        members++;
        return FFQ_OK;
    }
    // the jobs gathered so far (their compressed bytes still where take() saw them), by the pool's threads and this one
    int inflate(int64_t from, int64_t to)
    {
        if (!h_dst || jobs.empty()) { jobs.clear(); return FFQ_OK; }
        if (!own_init) {
            if (inflateInit2(&own, -15) != Z_OK) return fail(FFQ_E_NOMEM, "ffq_bgzf_range: zlib could not be initialised");
            own_init = true;
        }
        if (!pool && nthreads > 1 && jobs.size() > 1) {
            pool = new (std::nothrow) GzPool();
            if (pool && !pool->start(nthreads - 1)) { delete pool; pool = nullptr; }
        }
        bool ok = true;
        for (size_t j0 = 0; j0 < jobs.size() && ok; j0 += (size_t)1 << 20) {              // (in batches the pool's counter can hold)
            const int nj = (int)std::min<size_t>((size_t)1 << 20, jobs.size() - j0);
            if (pool) ok = pool->run(jobs.data() + j0, nj, &own);
            else for (int j = 0; j < nj && ok; j++) ok = GzPool::inflate_fast(jobs[j0 + (size_t)j]) || GzPool::inflate_one(&own, jobs[j0 + (size_t)j]);
        }
        jobs.clear();
        if (!ok) return fail(FFQ_E_ARG, "ffq_bgzf_range: gzip: a member between bytes %lld and %lld does not inflate to what its trailer says (length or CRC-32)", (long long)from, (long long)to);
        return FFQ_OK;
    }
};

}  // namespace bgzf_range

extern "C" int ffq_bgzf_range(int fd, int64_t c_lo, int64_t c_hi, uint8_t *h_dst, int64_t cap, int threads,
                              int64_t *c_first, int64_t *c_end, int64_t *n_out, int64_t *n_members)
{
    using namespace bgzf_range;
    if (fd < 0 || c_lo < 0 || c_hi < c_lo || cap < 0 || (cap > 0 && !h_dst)) return fail(FFQ_E_ARG, "ffq_bgzf_range: bad argument");
    struct stat sb;
    if (fstat(fd, &sb) != 0) return fail(FFQ_E_ARG, "ffq_bgzf_range: fstat: %s", strerror(errno));
    const int64_t size = (int64_t)sb.st_size;
    c_hi = std::min(c_hi, size);
    int64_t p = first_member(fd, std::min(c_lo, size), size);
    if (p < 0) return fail(FFQ_E_ARG, "ffq_bgzf_range: gzip: no BGZF member at or behind byte %lld (not a BGZF file, or one that is cut short)", (long long)c_lo);
    if (c_first) *c_first = p;
    Walk w(h_dst, cap, threads);
    int rc = FFQ_OK;
    bool mapped = false;
    // ---- a file that can be mapped: the members are walked and inflated where the page cache holds them -- no copy of the
    // compressed bytes, no read in front of every batch (the windows of the loop below cost 30 ms of a 65 ms inflate per 256 MB
    // of FASTQ, one thread reading while the others wait), ONE batch of jobs over the whole range ---------------------------
    if (p < c_hi && !getenv("FFQ_BGZF_NO_MMAP")) {              // (the variable: tests of the loop below)
        const int64_t page = (int64_t)sysconf(_SC_PAGESIZE);
        const int64_t map_lo = p & ~(page - 1), map_hi = std::min(size, c_hi + 65536 + 64);
        void *mp = mmap(nullptr, (size_t)(map_hi - map_lo), PROT_READ, MAP_PRIVATE, fd, (off_t)map_lo);
        if (mp != MAP_FAILED) {
            mapped = true;
            const uint8_t *m = static_cast<const uint8_t *>(mp) - map_lo;          // m[x] = byte x of the file
            const int64_t p0 = p;
            while (p < c_hi && !rc) {
                int64_t total = 0;
                rc = w.take(m + p, map_hi - p, p, size, &total);
                if (rc == 1) rc = fail(FFQ_E_INTERNAL, "ffq_bgzf_range: the member at byte %lld reaches past the mapped range", (long long)p);
                if (!rc) p += total;
            }
            if (!rc) rc = w.inflate(p0, p);
            (void)munmap(mp, (size_t)(map_hi - map_lo));
        }
    }
    // ---- what cannot be mapped: windows of compressed bytes read in front of each batch -------------------------------------
    std::vector<uint8_t> win;
    while (!mapped && p < c_hi && !rc) {
        // one window from member start p: its whole members, those that start in front of c_hi
        const int64_t n = std::min<int64_t>(WINDOW + 65536, size - p);
        try { win.resize((size_t)n + 16); } catch (const std::bad_alloc &) { rc = fail(FFQ_E_NOMEM, "out of host memory"); break; }
        if (pread_full(fd, win.data(), n, p) != n) { rc = fail(FFQ_E_ARG, "ffq_bgzf_range: read failed at byte %lld: %s", (long long)p, strerror(errno)); break; }
        int64_t q = 0;
        while (p + q < c_hi && q < WINDOW && !rc) {
            int64_t total = 0;
            rc = w.take(win.data() + q, n - q, p + q, size, &total);
            if (rc == 1) { rc = q > 0 ? FFQ_OK : fail(FFQ_E_INTERNAL, "ffq_bgzf_range: no progress at byte %lld", (long long)p); break; }      // (the next window begins with it)
            if (!rc) q += total;
        }
        if (!rc) rc = w.inflate(p, p + q);
        p += q;
    }
    if (rc) return rc;
    if (c_end) *c_end = p;
    if (n_out) *n_out = w.out;
    if (n_members) *n_members = w.members;
    return FFQ_OK;
}
