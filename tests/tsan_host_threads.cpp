// tsan_host_threads.cpp -- the library's HOST threads under ThreadSanitizer (tests/test_tsan.py builds and runs it; no
// device is touched).  Linked against a -fsanitize=thread build of libffq_hip.so (hipcc instruments the host code, the
// device code is left alone) and of the test oracle (oracle/ffq_oracle.c: the scan of the shard mode).
//
//   gunzip FILE THREADS CHUNK      ffq_gunzip_fd: the stream front end's gzip reader -- BGZF members side by side (GzPool),
//                                  ONE plain member by several threads (ffq_pgz.h: chunk hand-overs, stitch, CRC pieces),
//                                  zlib taking over; prints "bytes crc32" (the test compares with Python's gzip)
//   pool FILE                      the context's helper threads (ffq_pool.h): slices of preads and of host copies of
//                                  consecutive chunks through ONE queue; compared with a plain read
//   shards FILE WORLD TAIL HEAD    WORLD threads, each a rank of ffq_shard_host_step over its byte range of FILE: exchange and
//                                  gather through ffq_shard_world's breakable barrier (csrc/ffq_shard_proto.h), the settle
//                                  rounds of the protocol; prints the rows' checksum
//   abort FILE WORLD               the same with rank 1's scan failing: it breaks the barrier, every other rank comes back
//   stall FILE WORLD SECONDS       the same with rank 1 never arriving: the barrier's DEADLINE (the watchdog of the in-process
//                                  world) breaks it for everybody, every other rank comes back with FFQ_E_TIMEOUT within
//                                  SECONDS and the world names rank 1 as the absent one
//   race X                         a deliberate data race: the test checks that the detector reports it
#include "../include/ffq.h"
#include "../fastq-and-furious_amd/csrc/ffq_pool.h"
#include "../fastq-and-furious_amd/csrc/ffq_shard_proto.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" void ffq_oracle_scan(const uint8_t *d, int64_t n_bytes, int sentinel, int64_t offset, int eof, int variant, int64_t add,
                                int64_t *table, int64_t cap, int64_t *out);
extern "C" int ffq_oracle_entrypos_c(const uint8_t *b, int64_t len, int64_t offset, int64_t *pos);

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

static int mode_gunzip(const char *path, int threads, int64_t chunk)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { perror(path); return 2; }
    const int64_t cap = 1ll << 30;
    std::vector<uint8_t> out((size_t)(256 << 20));
    int64_t npar = 0;
    const int64_t n = ffq_gunzip_fd(fd, out.data(), (int64_t)out.size() < cap ? (int64_t)out.size() : cap, chunk, threads, &npar);
    close(fd);
    if (n < 0) { printf("error %lld: %s\n", (long long)n, ffq_last_error()); return 1; }
    int64_t st[5];
    ffq_gunzip_stats(st);
    printf("%lld %08lx %lld %lld %lld\n", (long long)n, (unsigned long)crc32(crc32(0L, Z_NULL, 0), out.data(), (uInt)n), (long long)npar,
           (long long)st[1], (long long)st[3]);            // bytes, CRC-32, BGZF members side by side, engine chunks taken, times it gave up
    return 0;
}

static int mode_pool(const char *path)
{
    const std::vector<uint8_t> want = slurp(path);
    const int fd = open(path, O_RDONLY);
    ReadPool pool;
    pool.start(6);
    const int64_t n = (int64_t)want.size(), CH = 3 << 20;
    std::vector<uint8_t> a((size_t)n + 64), b((size_t)n + 64);
    ChunkRead cr[3], cc[3];
    bool used[3] = {false, false, false};
    int64_t got = 0;
    // chunks of preads three at a time through one queue, each chunk copied on (enqueue_copy) while the next are read
    const int64_t nch = (n + CH - 1) / CH;
    for (int64_t k = 0, e = 0; k < nch; k++) {
        for (; e < nch && e - k < 3; e++) {
            if (used[e % 3]) pool.wait(&cc[e % 3]);
            pool.enqueue(fd, a.data() + e * CH, std::min(CH, n - e * CH), e * CH, &cr[e % 3]);
        }
        pool.wait(&cr[k % 3]);
        const int64_t m = cr[k % 3].total();
        if (m != std::min(CH, n - k * CH)) { printf("short read %lld\n", (long long)m); return 1; }
        pool.enqueue_copy(b.data() + k * CH, a.data() + k * CH, m, &cc[k % 3]);
        used[k % 3] = true;
        got += m;
    }
    for (int i = 0; i < 3; i++) if (used[i]) pool.wait(&cc[i]);
    close(fd);
    const bool ok = got == n && memcmp(b.data(), want.data(), (size_t)n) == 0;
    printf("%s %lld\n", ok ? "equal" : "DIFFERENT", (long long)n);
    return ok ? 0 : 1;
}

// ---- k ranks as threads over the breakable barrier ------------------------------------------------------------------------
struct Rank {
    ffq_shard_world *W;
    int rank, world;
    bool fail_scan;
    std::vector<const ffq_shard_piece *> *boards;      // what each rank parked for the others
    double deadline = 0;                               // seconds a rank waits at the barrier (0: for ever)
    // through / FFQ_E_TIMEOUT (somebody's wait ran out: the world says who was missing) / FFQ_E_INTERNAL (another rank failed)
    int meet() const
    {
        const int r = W->wait_for(rank, deadline);
        return r > 0 ? FFQ_OK : (r < 0 || W->timed_out) ? FFQ_E_TIMEOUT : FFQ_E_INTERNAL;
    }
};

static int cb_scan(void *u, const uint8_t *buf, int64_t n, int sentinel, int64_t offset, int eof, int64_t add, int64_t *table, int64_t cap,
                   ffq_scan_result *res)
{
    Rank *r = static_cast<Rank *>(u);
    if (r->fail_scan) return FFQ_E_INTERNAL;
    int64_t out[4];
    ffq_oracle_scan(buf, n, sentinel, offset, eof, 0, add, table, cap, out);
    res->n_records = out[0]; res->end_state = (int32_t)out[1]; res->last_status = (int32_t)out[2]; res->end_offset = out[3];
    if (res->end_state == 5) return FFQ_E_TABLE_FULL;                   // (the oracle's own code for a table that is full)
    for (int i = 0; i < 6; i++) res->last_pos[i] = -1;
    if (res->end_state != FFQ_END_OK) {
        std::vector<uint8_t> tmp;
        const uint8_t *b = buf;
        int64_t len = n;
        if (sentinel) { tmp.resize((size_t)n + 1); tmp[0] = '\n'; memcpy(tmp.data() + 1, buf, (size_t)n); b = tmp.data(); len = n + 1; }
        int64_t pos[6] = {-1, -1, -1, -1, -1, -1};
        (void)ffq_oracle_entrypos_c(b, len, res->end_offset, pos);
        for (int i = 0; i < 6; i++) res->last_pos[i] = pos[i] >= 0 ? pos[i] + add : -1;
    }
    return FFQ_OK;
}

static int cb_exchange(void *u, const ffq_shard_piece *ps, int n)
{
    Rank *r = static_cast<Rank *>(u);
    (*r->boards)[(size_t)r->rank] = ps;
    int rc = r->meet();
    if (rc) return rc;
    for (int i = 0; i < n; i++)
        if (ps[i].dst == r->rank) memcpy(ps[i].ptr, (*r->boards)[(size_t)ps[i].src][i].ptr, (size_t)(ps[i].b - ps[i].a));   // (the same list on every rank)
    return r->meet();          // the sources must stay as they are until read
}

static int cb_gather(void *u, const int64_t *mine, int64_t *all)
{
    Rank *r = static_cast<Rank *>(u);
    memcpy(&r->W->slots[(size_t)r->rank * 8], mine, 64);
    int rc = r->meet();
    if (rc) return rc;
    memcpy(all, r->W->slots.data(), (size_t)r->world * 64);
    return r->meet();
}

static int mode_shards(const char *path, int world, int64_t tail_bytes, int64_t head_bytes, bool with_abort, double stall_deadline = 0)
{
    const std::vector<uint8_t> data = slurp(path);
    const int64_t total = (int64_t)data.size();
    std::vector<int64_t> B((size_t)world + 1);
    for (int r = 0; r < world; r++) B[(size_t)r] = (r * total / world) / 16 * 16;
    B[(size_t)world] = total;
    ffq_shard_world W;
    W.world = world;
    W.slots.assign((size_t)world * 8, 0);
    std::vector<const ffq_shard_piece *> boards((size_t)world, nullptr);
    std::vector<uint64_t> sums((size_t)world, 0);
    std::vector<int64_t> counts((size_t)world, 0), bases((size_t)world, 0);
    std::vector<int> rcs((size_t)world, 0), errs((size_t)world, 0), rounds((size_t)world, 0);
    std::vector<int64_t> errb((size_t)world, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++)
        th.emplace_back([&, r] {
            if (stall_deadline > 0 && r == 1) return;                  // (the rank that never arrives)
            Rank me{&W, r, world, with_abort && r == 1, &boards, stall_deadline};
            ffq_shard_host_ops ops{&me, cb_scan, cb_exchange, cb_gather};
            const int64_t lo = B[(size_t)r], hi = B[(size_t)r + 1];
            const int64_t tail = std::min(tail_bytes, lo - B[0]), head = std::min(head_bytes, total - hi);
            std::vector<uint8_t> ext((size_t)(tail + (hi - lo) + head) + 64, 0);
            memcpy(ext.data() + tail, data.data() + lo, (size_t)(hi - lo));
            const int64_t cap = total / 20 + 64;
            std::vector<int64_t> table((size_t)cap * 6);
            ffq_shard_result out;
            const int rc = ffq_shard_host_step(&ops, nullptr, r, world, B.data(), tail_bytes, head_bytes, ext.data(), table.data(), cap, &out);
            rcs[(size_t)r] = rc;
            if (rc) { W.abort(); return; }
            errs[(size_t)r] = out.err_state; errb[(size_t)r] = out.err_byte; rounds[(size_t)r] = out.rounds;
            counts[(size_t)r] = out.row_hi - out.row_lo; bases[(size_t)r] = out.record_base;
            uint64_t s = 0;
            for (int64_t i = out.row_lo * 6; i < out.row_hi * 6; i++) s = s * 1000003u + (uint64_t)table[(size_t)i];
            sums[(size_t)r] = s;
            if (out.d_ext != ext.data()) ffq_shard_host_free(out.d_ext);
        });
    for (auto &t : th) t.join();
    if (stall_deadline > 0) {
        int timed = 0;
        for (int r = 0; r < world; r++) timed += rcs[(size_t)r] == FFQ_E_TIMEOUT;
        printf("stall: %d of %d ranks came back with FFQ_E_TIMEOUT; absent: %s\n", timed, world - 1, W.absent_list().c_str());
        return (timed == world - 1 && W.absent_list() == "1") ? 0 : 1;
    }
    if (with_abort) {
        int failed = 0;
        for (int r = 0; r < world; r++) failed += rcs[(size_t)r] != 0;
        printf("abort: %d of %d ranks came back with an error\n", failed, world);
        return failed == world ? 0 : 1;
    }
    for (int r = 0; r < world; r++) if (rcs[(size_t)r]) { printf("rank %d: rc %d\n", r, rcs[(size_t)r]); return 1; }
    // the whole stream by ONE scan: the ranks' rows, concatenated, must be its rows
    std::vector<int64_t> table((size_t)(total / 20 + 64) * 6);
    int64_t o[4];
    ffq_oracle_scan(data.data(), total, 1, 0, 1, 0, -1, table.data(), total / 20 + 64, o);
    int64_t n = 0;
    bool ok = true;
    if (errs[0]) {
        for (int r = 1; r < world; r++) ok = ok && errs[(size_t)r] == errs[0] && errb[(size_t)r] == errb[0];
        ok = ok && errs[0] == (int)o[1] && errb[0] == o[3] - 1;
        printf("%s stream error %d at byte %lld\n", ok ? "equal" : "DIFFERENT", errs[0], (long long)errb[0]);
        return ok ? 0 : 1;
    }
    for (int r = 0; r < world; r++) {
        uint64_t s = 0;
        for (int64_t i = n * 6; i < (n + counts[(size_t)r]) * 6; i++) s = s * 1000003u + (uint64_t)table[(size_t)i];
        ok = ok && s == sums[(size_t)r] && bases[(size_t)r] == n;
        n += counts[(size_t)r];
    }
    ok = ok && n == o[0] && o[1] == FFQ_END_OK;
    int maxr = 0;
    for (int r = 0; r < world; r++) maxr = std::max(maxr, rounds[(size_t)r]);
    printf("%s %lld records over %d ranks, %d repair rounds\n", ok ? "equal" : "DIFFERENT", (long long)n, world, maxr);
    return ok ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: tsan_host_threads gunzip|pool|shards|abort FILE ...\n"); return 2; }
    const std::string mode = argv[1];
    if (mode == "gunzip") return mode_gunzip(argv[2], argc > 3 ? atoi(argv[3]) : 4, argc > 4 ? atoll(argv[4]) : (4 << 20));
    if (mode == "pool") return mode_pool(argv[2]);
    if (mode == "shards") return mode_shards(argv[2], argc > 3 ? atoi(argv[3]) : 3, argc > 4 ? atoll(argv[4]) : (1 << 20), argc > 5 ? atoll(argv[5]) : (1 << 20), false);
    if (mode == "race") {           // (is the detector awake?  two threads, one plain int)
        int x = 0;
        std::thread a([&] { for (int i = 0; i < 100000; i++) x++; }), b([&] { for (int i = 0; i < 100000; i++) x++; });
        a.join(); b.join();
        printf("%d\n", x);
        return 0;
    }
    if (mode == "abort") return mode_shards(argv[2], argc > 3 ? atoi(argv[3]) : 3, 1 << 20, 1 << 20, true);
    if (mode == "stall") return mode_shards(argv[2], argc > 3 ? atoi(argv[3]) : 3, 1 << 20, 1 << 20, false, argc > 4 ? atof(argv[4]) : 1.0);
    return 2;
}
