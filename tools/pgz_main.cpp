// Development harness of csrc/ffq_pgz.h (the several-thread inflate of one gzip member), host only:
//   g++ -O2 -std=c++17 -pthread -o /tmp/pgz_main tools/pgz_main.cpp -lz
//   /tmp/pgz_main file.gz [threads] [check]
// inflates the file member by member with the engine, prints bytes, CRC-32, seconds and the engine's counters;
// "check": the same through zlib's gzread, byte for byte.
#include "../fastq-and-furious_amd/csrc/ffq_pgz.h"

#include <chrono>
#include <cstdio>
#include <fcntl.h>
#include <sys/stat.h>

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: pgz_main file.gz [threads] [check]\n"); return 2; }
    const int threads = argc > 2 ? atoi(argv[2]) : 4;
    const bool check = argc > 3;
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror("open"); return 2; }
    struct stat st;
    fstat(fd, &st);
    ffq::pgz::Engine e;
    if (!e.init(fd, threads, st.st_size)) { fprintf(stderr, "init failed\n"); return 2; }
    std::vector<uint8_t> out;
    const int64_t piece = 64 << 20;
    std::vector<uint8_t> buf((size_t)piece);
    int64_t off = 0, total = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (off < st.st_size) {
        uint8_t z;
        if (pread(fd, &z, 1, off) == 1 && z == 0) { off++; continue; }
        if (!e.begin(off)) { fprintf(stderr, "not a gzip member at %lld\n", (long long)off); return 1; }
        for (;;) {
            const int64_t r = e.read(buf.data(), piece);
            total += r;
            if (check) out.insert(out.end(), buf.begin(), buf.begin() + r);
            if (e.pending()) continue;
            if (e.failed) { fprintf(stderr, "failed: %s\n", e.msg.c_str()); return 1; }
            if (e.gave_up) { fprintf(stderr, "gave up at bit %lld (window %lld)\n", (long long)e.pos_bit, (long long)e.win_valid); return 3; }
            if (e.member_done) break;
        }
        off = e.end_off;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    auto &s = ffq::pgz::stats();
    printf("%lld bytes  %.3f s  %.3f GB/s  batches %lld chunks %lld rejected %lld members %lld\n", (long long)total, dt, total / dt / 1e9,
           (long long)s.batches.load(), (long long)s.chunks.load(), (long long)s.rejected.load(), (long long)s.members.load());
    printf("ms: read %.1f stage1 %.1f (exact %.1f markers %.1f of which find %.1f, summed over threads) stitch %.1f stage2 %.1f\n", s.ns_read / 1e6, s.ns_stage1 / 1e6,
           s.ns_exact / 1e6, s.ns_markers / 1e6, s.ns_find / 1e6, s.ns_stitch / 1e6, s.ns_stage2 / 1e6);
    if (check) {
        gzFile g = gzopen(argv[1], "rb");
        gzbuffer(g, 1 << 20);
        std::vector<uint8_t> ref;
        const auto t1 = std::chrono::steady_clock::now();
        for (;;) {
            const int r = gzread(g, buf.data(), (unsigned)piece);
            if (r <= 0) break;
            ref.insert(ref.end(), buf.begin(), buf.begin() + r);
        }
        const double dz = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        gzclose(g);
        printf("zlib: %lld bytes %.3f s %.3f GB/s -- %s\n", (long long)ref.size(), dz, ref.size() / dz / 1e9, ref == out ? "EQUAL" : "DIFFERENT");
        return ref == out ? 0 : 1;
    }
    return 0;
}
