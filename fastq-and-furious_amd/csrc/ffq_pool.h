// ffq_pool.h -- helper threads of a context (host code of libffq_hip.so): slices of file reads
// (the stream front end, ffq_stream.h) and of host-to-host copies (the staging of the host-buffer
// entry points, ffq_scan_host) run on them in parallel.
#pragma once
#include <errno.h>
#include <sched.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

// ---- helper threads: a chunk of a seekable descriptor is read in slices -------------------------
// (one pread loop saturates at the copy rate of a single core, ~10 GB/s from the page cache).
// The slices of consecutive chunks go through ONE queue: the helpers never meet at a per-chunk
// barrier, a chunk is complete when its last slice is (ChunkRead::left).
struct ChunkRead {
    int64_t got[64];
    int64_t want[64];
    int nsl = 0;
    int left = 0;                // slices still being read (under ReadPool::m)
    // the bytes read up to the first short slice -- the same prefix a single read would return
    int64_t total() const
    {
        int64_t t = 0;
        for (int i = 0; i < nsl; i++) {
            if (got[i] < 0) return -1;
            t += got[i];
            if (got[i] < want[i]) break;
        }
        return t;
    }
};

struct ReadPool {
    struct Job { int fd; const uint8_t *src; uint8_t *dst; int64_t n, pos; ChunkRead *cr; int idx; };   // src: a copy, else a pread
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::deque<Job> q;
    bool stop = false;

    static int64_t read_full(int fd, uint8_t *dst, int64_t n, int64_t pos, bool seekable)
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
    // The CPUs next to a PCI device ("0000:8b:00.0": /sys/bus/pci/devices/<bdf>/local_cpulist), of those this process may
    // run on.  The boxes here have two sockets with four GPUs each: helpers that run on the OTHER socket write the pinned
    // slots (which the runtime puts on the GPU's node) across the socket link while the DMA engine reads them -- the file
    // loader then runs at 37-41 GB/s instead of 51 (tools/load_threads.py, profiles/r06_probes/loader_numa.txt).
    static bool local_cpus(const char *bdf, cpu_set_t *out)
    {
        CPU_ZERO(out);
        char path[160], buf[1024];
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
        FILE *f = fopen(path, "r");
        if (!f) return false;
        const bool ok = fgets(buf, sizeof buf, f) != nullptr;
        fclose(f);
        if (!ok) return false;
        cpu_set_t mine;
        CPU_ZERO(&mine);
        if (sched_getaffinity(0, sizeof mine, &mine) != 0) return false;
        int n = 0;
        for (const char *p = buf; *p && *p != '\n';) {                     // "0-63,128-191"
            char *e = nullptr;
            long a = strtol(p, &e, 10), b = a;
            if (e == p) break;
            p = e;
            if (*p == '-') { b = strtol(p + 1, &e, 10); if (e == p + 1) break; p = e; }
            for (long c = a; c <= b && c < CPU_SETSIZE; c++)
                if (c >= 0 && CPU_ISSET((int)c, &mine)) { CPU_SET((int)c, out); n++; }
            if (*p == ',') p++;
        }
        return n >= 2;
    }
    void start(int n, const cpu_set_t *cpus = nullptr)
    {
        cpu_set_t set;
        const bool bind = cpus != nullptr;
        if (bind) set = *cpus;
        for (int i = 0; i < n; i++)
            th.emplace_back([this, bind, set] {
                if (bind) (void)sched_setaffinity(0, sizeof set, &set);    // (this thread only; a refusal changes nothing)
                for (;;) {
                    Job j;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_job.wait(lk, [this] { return stop || !q.empty(); });
                        if (q.empty()) return;
                        j = q.front(); q.pop_front();
                    }
                    int64_t g = j.n;
                    if (j.src) memcpy(j.dst, j.src, (size_t)j.n);
                    else g = read_full(j.fd, j.dst, j.n, j.pos, true);
                    {
                        std::lock_guard<std::mutex> lk(m);
                        j.cr->got[j.idx] = g;
                        if (--j.cr->left == 0) cv_done.notify_all();
                    }
                }
            });
    }
    ~ReadPool()
    {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_job.notify_all();
        for (auto &t : th) t.join();
    }
    // queue the slices of one chunk of a seekable descriptor (returns at once)
    void enqueue(int fd, uint8_t *dst, int64_t n, int64_t pos, ChunkRead *cr)
    {
        const int64_t SL = 1 << 20;
        const int nsl = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)th.size(), n / SL, (int64_t)64}));
        const int64_t per = ((n + nsl - 1) / nsl + 4095) & ~(int64_t)4095;
        {
            std::lock_guard<std::mutex> lk(m);
            cr->nsl = 0;
            for (int t = 0; t < nsl; t++) {
                const int64_t a = (int64_t)t * per;
                if (a >= n) break;
                cr->want[t] = std::min(per, n - a);
                cr->got[t] = 0;
                q.push_back(Job{fd, nullptr, dst + a, cr->want[t], pos + a, cr, t});
                cr->nsl = t + 1;
            }
            cr->left = cr->nsl;
        }
        cv_job.notify_all();
    }
    // queue a host-to-host copy in slices (pageable <-> pinned staging of the host-buffer entry points)
    void enqueue_copy(uint8_t *dst, const uint8_t *src, int64_t n, ChunkRead *cr)
    {
        const int64_t SL = 512 << 10;
        if (n < 2 * SL) {                    // not worth waking anybody
            memcpy(dst, src, (size_t)n);
            std::lock_guard<std::mutex> lk(m);
            cr->nsl = 0; cr->left = 0;
            return;
        }
        const int nsl = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)th.size(), n / SL, (int64_t)64}));
        const int64_t per = ((n + nsl - 1) / nsl + 4095) & ~(int64_t)4095;
        {
            std::lock_guard<std::mutex> lk(m);
            cr->nsl = 0;
            for (int t = 0; t < nsl; t++) {
                const int64_t a = (int64_t)t * per;
                if (a >= n) break;
                cr->want[t] = std::min(per, n - a);
                cr->got[t] = 0;
                q.push_back(Job{-1, src + a, dst + a, cr->want[t], 0, cr, t});
                cr->nsl = t + 1;
            }
            cr->left = cr->nsl;
        }
        cv_job.notify_all();
    }
    void wait(ChunkRead *cr)
    {
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [cr] { return cr->left == 0; });
    }
};

