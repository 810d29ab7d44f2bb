// ffq_hip.hip -- C ABI of libffq_hip.so (see include/ffq.h).
// Host side of the MI355X FASTQ buffer-scan path: context, scratch sizing,
// kernel sequencing on one HIP stream, pinned staging for host buffers.
// No CPU fallback: every compute entry point needs a live gfx950 device.
#include "../../include/ffq.h"
#ifdef FFQ_PROBES
#include "../../include/ffq_probe.h"
#endif
#include "ffq_kernels.h"
#include "ffq_fasta.h"
#include "ffq_pool.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

using namespace ffq;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                     \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return fail(FFQ_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                             \
    } while (0)

struct ScanArgs {
    const uint8_t *d_buf; int64_t n_bytes; int s; int64_t offset; int eof; int64_t add; uint32_t flags;
    int qual_add; int64_t *d_table; int64_t table_cap; int8_t *d_qual; int64_t qual_cap; int64_t *d_qoff;
};

struct ScanState {
    bool active = false;
    ScanArgs a{};
    int retries = 0;
    int repairs = 0;          // repair passes of the general kernels (reported with retries)
    bool dense_cfg = false, fast4_failed = false;
    bool probe4 = false;      // the front carries the fast path's kernels as a probe (see ffq_ctx::fast4_skip)
    bool untimed = false;     // FFQ_F_NO_TIMING: no marks around the index kernel, which may start beside the previous front's last kernel

    bool dense4 = false;      // the fast path's row kernel is the DENSE instantiation (it met a dense tile on this buffer, or the context remembers one)
    bool fused = false;       // the front is the single-pass index + decode kernel (ffq_fused.h)
    bool no_fused = false;    // ... which did not stand on this buffer: the two-pass kernels take it
    bool in_place = false;    // the single pass of this front writes in place (k_scan_ident) rather than segments (k_scan_seg)
    bool seg_refused = false; // the segmented single pass met a shape it cannot take (a long line): the in-place one is next
    unsigned long long poll_seq = 0;   // FFQ_F_POLL_RESULT: the front ends in a publisher that writes this number; no end event
    bool wide = false;        // the index kernel of this scan also wrote EVERY byte decoded in place (k_scan_lines<.., WIDE>): the general
                              //   path's decode in one pass -- no tier of this scan runs a decode kernel, qoff[i] = pos4's offset in the buffer
    bool go_ranked = false;   // the front is the index kernel only: the list-ranking tier follows at the wait
    bool index_done = false;  // the line index of this buffer is built (a later tier re-uses it)
    int stage = 0;            // what the pending front consisted of: 1 fast four-line path, 2 general path
    int64_t ntiles = 0;
    int ngroups = 0;
};

// What the library knows about the tail of a scan stream (kept in the context that owns the stream; contexts that share it
// point there).  FFQ_F_NO_TIMING lets a front start its index kernel without a barrier in front -- beside the previous scan's
// last, one-workgroup kernel -- and the library does that only when IT can vouch for the order not mattering: the last thing
// enqueued on the stream is the front of another scan (nothing else -- no copy, no wait for an event, no other kernel -- has gone
// onto the stream since, and the stream's handle was never given out), and none of that scan's outputs overlaps the bytes this
// one reads.
struct StreamTail {
    bool is_scan = false;                // the last operation enqueued on the stream is the end of a scan front
    bool exposed = false;                // ffq_ctx_stream() handed the stream out: others may enqueue on it unseen
    const void *out_p[3] = {nullptr, nullptr, nullptr};      // that scan's outputs (table, qualities, quality offsets)
    size_t out_n[3] = {0, 0, 0};
};

struct ffq_ctx {
    ScanState pend;                      // the scan enqueued by ffq_scan_submit, if any
    ffq_ctx *owner = nullptr;            // the context whose stream this one uses (itself unless created shared)
    StreamTail tail;                     // (in the owner)
    bool owns_streams = true;            // false: streams borrowed from another context
    double watchdog_s = 0;               // > 0: the waits INSIDE a scan (its marks, the stream in front of a scratch that grows) poll and give
                                         //   up after this long with FFQ_E_TIMEOUT -- set by a shard step around its scan (ffq_shard.h): a scan
                                         //   stream that waits for a hand-off, or carries a gather, whose peer never comes must not hold the host
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool fast4_remember = false;         // the recent input was not plain four-line: scans start with the general kernels
    int fast4_skip = 0;                  //   ... and this many of them do so without asking again; the one after is a PROBE
                                         //   scan: general kernels as before, the fast path's kernels in front of them just
                                         //   to see whether they would have stood (DevRes::fast4_hint) -- no host round trip,
                                         //   no second front, whichever way it comes out
    int ranked_skip = 0;                 // scans left that go straight to the list-ranking tier (long records)
    RankBufs rk = {};                    // its scratch (ffq_ranked.h), grow-only
    int64_t rk_cap_tiles = 0, rk_cap_c = 0;
    int dense_skip = 0;                  // scans left that start with the dense configuration of them
    bool lite_ran = false;               // the last general front used the lean kernel
    int lite_skip = 0;                   // scans left whose general path runs k_chain_wave over all groups (the lean kernel declined too many)
    bool dense4_remember = false;        // four-line input with dense tiles (reads of a dozen bases): the fast path starts with k_rows4<., true>
    // single-pass index + decode (ffq_fused.h): per-tile phases, verdict; grow-only
    uint8_t *fz_qphase = nullptr;
    int64_t fz_tiles_cap = 0;
    uint32_t *fz_bad = nullptr;
    int fused_skip = 0, fused_backoff = 15;   // scans left that do not try it (it failed: odd records), and the next count
    bool fz_in_place = false;                 // the last single pass that stood wrote in place (long lines): start with that one
    bool decode_timed = false;           // ev[6] marks the start of the decode kernel of the pending front
    // scratch, grow-only
    int64_t cap_tiles = 0;
    uint16_t *ent = nullptr;
    uint32_t *cnt = nullptr;
    unsigned long long *ovf = nullptr;
    uint16_t *pool = nullptr;
    unsigned long long pool_cap = 0;
    int64_t cap_groups = 0;
    ChainBufs cb = {};
    int64_t stage_cap = 0;        // StageRec entries allocated
    int64_t dstage_chunks = 0;    // chunks of the walked groups' stage (cb.dstage, DCHUNK records each), grow-only
    unsigned long long *prof_d = nullptr;
    long long *sbbase = nullptr;
    TileQ *tileq = nullptr;              // fast path + decode: records / quality bytes per tile
    unsigned int *sbq = nullptr;         //   quality bytes per 64 tiles
    long long *sbqbase = nullptr;        //   their exclusive scan
    Fast4Hdr *hdr4 = nullptr;
    TermInfo4 *tinfo4 = nullptr;
    Ctl *ctl = nullptr;
    LineIndex *d_L = nullptr;          // device copy of the LineIndex (out-of-line device functions)
    LineIndex *h_L = nullptr;          // pinned source of that copy
    DevRes *dres = nullptr;
    int64_t *qdir = nullptr;           // directory of the decoded-quality stream (qdir_mark)
    int64_t qdir_cap = 0;
    int64_t *p4s = nullptr;            // pos4 of every record, compact: what the decode reads instead of the 48-byte rows
    int64_t p4s_cap = 0;
    uint32_t *qrel = nullptr;          // tile-relative quality offsets of the fast path (p4s_cap entries)
    // pinned mirrors
    Ctl *h_ctl = nullptr;               // host-mapped pinned: written by the publishing kernel (Pub)
    DevRes *h_res = nullptr;
    Ctl *hm_ctl = nullptr;              // device addresses of the two
    DevRes *hm_res = nullptr;
    unsigned long long *h_seq = nullptr, *hm_seq = nullptr;   // completion word of FFQ_F_POLL_RESULT (host-mapped) and its device address
    unsigned long long seq = 0;         // last number handed out
    unsigned long long pub_seq = 0;     // what the publisher being enqueued writes (0: nothing)
    bool ctl_clean = false;             // the control block is zero (creation, or a publisher ran last)
    bool ctl_was_clean = false;         //   ... as it was when the front being enqueued began (no memset in front of it)
    int64_t *d_word = nullptr;          // 2 scratch words for the small table queries
    int64_t *h_word = nullptr;          //   and their pinned mirror
    int64_t *d_cut = nullptr, *h_cut = nullptr;    // ffq_table_cut: 6 words
    FaHdr *fa_hdr = nullptr;            // FASTA scan: starts, the last start
    // staging for the host-buffer entry points
    uint8_t *stage_d = nullptr;
    int64_t stage_d_cap = 0;
    uint8_t *stage_h = nullptr;
    int64_t stage_h_cap = 0;
    int64_t *tab_d = nullptr;
    int64_t tab_d_cap = 0;
    int8_t *qual_d = nullptr;
    int64_t qual_d_cap = 0;
    int64_t *qoff_d = nullptr;
    int64_t qoff_d_cap = 0;
    unsigned int *sel_cnt = nullptr;   // row selection: kept rows per workgroup, their scan
    int64_t sel_cnt_cap = 0;
    long long *sel_base = nullptr;
    int64_t sel_base_cap = 0;
    int64_t *tab_h = nullptr;      // pinned bounce for rows
    int64_t tab_h_cap = 0;
    long long *col_sum = nullptr;  // column selection: bytes per block of rows, their scan
    int64_t col_sum_cap = 0;
    long long *scan_bs = nullptr;  // block sums of the two-level scan (launch_scan_i64v)
    int64_t scan_bs_cap = 0;
    DevRes *scan_res = nullptr;    // (the middle level's result block: nobody reads it)
    DevRes *col_res = nullptr;     //   its result block (row count, total bytes)
    // ffq_scan_host: pageable memory goes through three pinned staging slots, copied in by the
    // helper threads and out over two copy streams (and back the same way)
    hipEvent_t stage_ev[3][2] = {};
    hipStream_t stage_cs[2] = {nullptr, nullptr};
    ChunkRead stage_cr[3];
    ReadPool *helpers = nullptr;   // helper threads (ffq_pool.h), started on first use
    cpu_set_t near_cpus;           // the CPUs next to the GPU (ctx_near), near_ok: known
    int near_ok = -1;
    void *stream_cache = nullptr;  // buffers of the last closed ffq_stream (ffq_stream.h), reused by the next one
};

extern "C" int ffq_abi_version(void) { return FFQ_ABI_VERSION; }
// hash of the sources this binary was compiled from (csrc/*, include/*.h), baked in by build.py
#ifndef FFQ_BUILD_ID
#define FFQ_BUILD_ID "unknown"
#endif
#ifdef FFQ_PROBES
static const char g_build_tag[] = "FFQ_BUILD_ID=" FFQ_BUILD_ID "+probes";      // (build.py reads the tag out of the file)
#else
static const char g_build_tag[] = "FFQ_BUILD_ID=" FFQ_BUILD_ID;
#endif
extern "C" const char *ffq_build_id(void) { return g_build_tag + 13; }
extern "C" const char *ffq_last_error(void) { return g_err.c_str(); }

extern "C" int ffq_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int ctx_create_impl(int device, ffq_ctx *share, ffq_ctx **out);
extern "C" void ffq_ctx_destroy(ffq_ctx *c);
static void stream_cache_drop(ffq_ctx *c);

extern "C" int ffq_ctx_create(int device, ffq_ctx **out) { return ctx_create_impl(device, nullptr, out); }

static int ctx_create_impl(int device, ffq_ctx *share, ffq_ctx **out)
{
    if (!out) return fail(FFQ_E_ARG, "ffq_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(FFQ_E_NODEVICE, "no HIP device visible: the FASTQ scan path needs an MI355X (gfx950); there is no CPU fallback");
    if (device < 0 || device >= n) return fail(FFQ_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FFQ_E_NODEVICE, "device %d is %s; libffq_hip is built for gfx950 only", device, prop.gcnArchName);
    HIPCHK(hipSetDevice(device));
    ffq_ctx *c = new (std::nothrow) ffq_ctx();
    if (!c) return fail(FFQ_E_NOMEM, "out of host memory");
    c->device = device;
    hipError_t e = hipSuccess;
    c->owner = share ? share->owner : c;
    if (share) {
        // same streams as `share`: scans of the two contexts execute in submission order
        c->stream = share->stream; c->owns_streams = false;
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    // timing marks need no system-scope release (an L2 write-back per record); only ev[3] / ev[2],
    // which the host waits on before it reads the published result block, keep the default
    for (int i = 0; i < 7 && e == hipSuccess; i++)
        e = (i == 2 || i == 3) ? hipEventCreate(&c->ev[i])
                               : hipEventCreateWithFlags(&c->ev[i], hipEventDisableSystemFence);
    if (e == hipSuccess) e = hipMalloc((void **)&c->ctl, sizeof(Ctl));
    if (e == hipSuccess) e = hipMalloc((void **)&c->dres, sizeof(DevRes));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_L, sizeof(LineIndex));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_word, 16);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_word, 16, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_cut, 48);
    if (e == hipSuccess) e = hipMalloc((void **)&c->fa_hdr, sizeof(FaHdr));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_cut, 48, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_L, sizeof(LineIndex), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_ctl, sizeof(Ctl), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_res, sizeof(DevRes), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_seq, 64, hipHostMallocMapped);
    if (e == hipSuccess) { *c->h_seq = 0; e = hipHostGetDevicePointer((void **)&c->hm_seq, c->h_seq, 0); }
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->hm_ctl, c->h_ctl, 0);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&c->hm_res, c->h_res, 0);
    if (e != hipSuccess) {
        ffq_ctx_destroy(c);
        return fail(FFQ_E_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return FFQ_OK;
}

static void free_chain(ffq_ctx *c)
{
    (void)hipFree(c->cb.y); (void)hipFree(c->cb.exit); (void)hipFree(c->cb.cnt); (void)hipFree(c->cb.flags);
    (void)hipFree(c->cb.lines); (void)hipFree(c->cb.qb); (void)hipFree(c->cb.term); (void)hipFree(c->cb.stage);
    (void)hipFree(c->cb.rloc); (void)hipFree(c->cb.qloc); (void)hipFree(c->cb.part); (void)hipFree(c->cb.mins);
    (void)hipFree(c->cb.force); (void)hipFree(c->cb.dlist); (void)hipFree(c->cb.ilist);
    (void)hipFree(c->sbbase); (void)hipFree(c->tinfo4);
    (void)hipFree(c->tileq); (void)hipFree(c->sbq); (void)hipFree(c->sbqbase);
    c->sbbase = nullptr; c->tinfo4 = nullptr;
    c->tileq = nullptr; c->sbq = nullptr; c->sbqbase = nullptr;
    {
        // (the walked groups' stage does not depend on the tile count: it stays)
        StageRec *ds = c->cb.dstage; const int32_t dc = c->cb.dchunks;
        c->cb = ChainBufs{};
        c->cb.dstage = ds; c->cb.dchunks = dc;
    }
    c->stage_cap = 0;
    c->cap_groups = 0;
}

extern "C" int ffq_ctx_create_shared(ffq_ctx *parent, ffq_ctx **out)
{
    if (!parent || !out) return fail(FFQ_E_ARG, "ffq_ctx_create_shared: NULL argument");
    return ctx_create_impl(parent->device, parent, out);
}

extern "C" void ffq_ctx_destroy(ffq_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    stream_cache_drop(c);
    for (auto &r : c->stage_ev) for (auto &e : r) if (e) (void)hipEventDestroy(e);
    for (auto &st : c->stage_cs) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    delete c->helpers;
    (void)hipFree(c->ent); (void)hipFree(c->cnt); (void)hipFree(c->ovf); (void)hipFree(c->pool);
    (void)hipFree(c->rk.tbase); (void)hipFree(c->rk.cand); (void)hipFree(c->rk.rec); (void)hipFree(c->rk.succ);
    (void)hipFree(c->rk.S[0]); (void)hipFree(c->rk.S[1]); (void)hipFree(c->rk.C[0]); (void)hipFree(c->rk.C[1]);
    (void)hipFree(c->rk.D); (void)hipFree(c->rk.root);
    free_chain(c);
    (void)hipFree(c->cb.dstage);
    (void)hipFree(c->fz_qphase); (void)hipFree(c->fz_bad);
    (void)hipFree(c->ctl); (void)hipFree(c->dres); (void)hipFree(c->d_L); (void)hipFree(c->hdr4);
    if (c->h_L) (void)hipHostFree(c->h_L);
    (void)hipFree(c->qdir); (void)hipFree(c->p4s); (void)hipFree(c->qrel);
    (void)hipFree(c->sel_cnt); (void)hipFree(c->sel_base); (void)hipFree(c->col_sum); (void)hipFree(c->scan_bs); (void)hipFree(c->scan_res); (void)hipFree(c->col_res);
    (void)hipFree(c->stage_d); (void)hipFree(c->tab_d); (void)hipFree(c->qual_d); (void)hipFree(c->qoff_d);
    if (c->h_word) (void)hipHostFree(c->h_word);
    if (c->h_cut) (void)hipHostFree(c->h_cut);
    (void)hipFree(c->d_word); (void)hipFree(c->d_cut); (void)hipFree(c->fa_hdr);
    if (c->h_seq) (void)hipHostFree(c->h_seq);
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->stage_h) (void)hipHostFree(c->stage_h);
    if (c->tab_h) (void)hipHostFree(c->tab_h);
    for (int i = 0; i < 7; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream && c->owns_streams) (void)hipStreamDestroy(c->stream);
    delete c;
}

// something other than a scan front goes onto the context's stream (every entry point that enqueues there says so)
static inline void mark_other(ffq_ctx *c) { if (c && c->owner) c->owner->tail.is_scan = false; }

// The waits of the scan path, with the context's watchdog (ffq_ctx::watchdog_s; 0: the plain blocking calls).  A scan is
// through in milliseconds: the first two are spun, the rest slept in 100 us pieces.
static int ctx_wait(ffq_ctx *c, hipEvent_t ev, hipStream_t st)
{
    if (c->watchdog_s <= 0) {
        if (ev) HIPCHK(hipEventSynchronize(ev)); else HIPCHK(hipStreamSynchronize(st));
        return FFQ_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1;; spins++) {
        const hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(st);
        if (e == hipSuccess) return FFQ_OK;
        if (e != hipErrorNotReady) return fail(FFQ_E_HIP, "%s failed: %s", ev ? "hipEventQuery" : "hipStreamQuery", hipGetErrorString(e));
        if ((spins & 31) == 0) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt > c->watchdog_s) return fail(FFQ_E_TIMEOUT, "scan: the stream made no progress within %.1f s", dt);
            if (dt > 2e-3) usleep(100);
        }
        __builtin_ia32_pause();
    }
}
#define CTX_WAIT_EVENT(c, ev) do { const int rc__ = ctx_wait((c), (ev), nullptr); if (rc__) return rc__; } while (0)
#define CTX_SYNC(c) do { const int rc__ = ctx_wait((c), nullptr, (c)->stream); if (rc__) return rc__; } while (0)

extern "C" void *ffq_ctx_stream(ffq_ctx *c)
{
    if (!c) return nullptr;
    c->owner->tail.exposed = true;       // (whoever holds the handle may enqueue on the stream without the library seeing it)
    c->owner->tail.is_scan = false;
    return (void *)c->stream;
}

// The CPUs next to the context's GPU (sysfs: /sys/bus/pci/devices/<bdf>/local_cpulist), looked up once.
// FFQ_POOL_AFFINITY=0: nothing is bound.
static bool ctx_near(ffq_ctx *c)
{
    if (c->near_ok < 0) {
        char bdf[32] = {0};
        const char *a = getenv("FFQ_POOL_AFFINITY");
        bool ok = !(a && atoi(a) == 0) && hipDeviceGetPCIBusId(bdf, (int)sizeof bdf - 1, c->device) == hipSuccess;
        for (char *p = bdf; *p; p++) *p = (char)tolower((unsigned char)*p);      // (sysfs spells the address in lower case)
        ok = ok && ReadPool::local_cpus(bdf, &c->near_cpus);
        c->near_ok = ok ? 1 : 0;
        if (getenv("FFQ_POOL_DEBUG")) fprintf(stderr, "[ffq pool] device %d at %s: %s (%d CPUs)\n", c->device, bdf, ok ? "helpers and staging memory next to the GPU" : "not bound", ok ? CPU_COUNT(&c->near_cpus) : 0);
    }
    return c->near_ok == 1;
}

// The calling thread next to the GPU for as long as this lives: pinned staging memory is allocated (and touched) from there,
// so that its pages lie on the GPU's node whatever the runtime's own policy is (on the two-socket boxes here the loader ran
// at 46 instead of 52 GB/s in one process of six with the helpers bound but the slots allocated from wherever the caller ran).
struct NearGpu {
    cpu_set_t saved;
    bool bound = false;
    explicit NearGpu(ffq_ctx *c)
    {
        if (!ctx_near(c) || sched_getaffinity(0, sizeof saved, &saved) != 0) return;
        bound = sched_setaffinity(0, sizeof c->near_cpus, &c->near_cpus) == 0;
    }
    ~NearGpu() { if (bound) (void)sched_setaffinity(0, sizeof saved, &saved); }
    NearGpu(const NearGpu &) = delete;
    NearGpu &operator=(const NearGpu &) = delete;
};

static ReadPool *ctx_pool(ffq_ctx *c)
{
    if (!c->helpers) {
        c->helpers = new (std::nothrow) ReadPool();
        if (c->helpers) {
            const unsigned hw = std::thread::hardware_concurrency();
            const char *e = getenv("FFQ_POOL_THREADS");              // (measurements; default: up to 16)
            const unsigned want = (e && atoi(e) > 0) ? (unsigned)std::min(atoi(e), 64) : 16u;
            // the helpers run on the CPUs next to the GPU (FFQ_POOL_AFFINITY=0: wherever the scheduler puts them)
            const bool bind = ctx_near(c);
            c->helpers->start((int)std::min<unsigned>(want, hw > 2 ? hw - 2 : 1), bind ? &c->near_cpus : nullptr);
        }
    }
    return c->helpers;
}

static int64_t tiles_for(int64_t n) { return (n + TILE - 1) >> TILE_SHIFT; }
static int64_t groups_for(int64_t ntiles) { return (ntiles + OWN_T - 1) / OWN_T; }
constexpr int PER_FAST = 6, EMAX_FAST = 2048, WPB_FAST = 2;   // k_chain_wave, usual line/record density
constexpr int PER_DENSE = 15, EMAX_DENSE = NTW * SLOT + 8, WPB_DENSE = 1;   // short records / short lines
constexpr int NMAX_FAST = PER_FAST * 64, NMAX_DENSE = PER_DENSE * 64;
constexpr int WPB_LITE = 2;                                    // k_chain_lite (ffq_lite.h): groups per workgroup

static int reserve_tiles(ffq_ctx *c, int64_t ntiles)
{
    if (ntiles <= c->cap_tiles) return FFQ_OK;
    CTX_SYNC(c);
    (void)hipFree(c->ent); (void)hipFree(c->cnt); (void)hipFree(c->ovf);
    c->ent = nullptr; c->cnt = nullptr; c->ovf = nullptr;
    free_chain(c);
    c->cap_tiles = 0;
    const int64_t ng = groups_for(ntiles);
    const int64_t nblk = (ng + RES_BLOCK - 1) / RES_BLOCK;
    HIPCHK(hipMalloc((void **)&c->ent, (size_t)ntiles * SLOT * sizeof(uint16_t)));
    HIPCHK(hipMalloc((void **)&c->cnt, (size_t)ntiles * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **)&c->ovf, (size_t)ntiles * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&c->cb.y, (size_t)ng * 8));
    HIPCHK(hipMalloc((void **)&c->cb.exit, (size_t)ng * 8));
    HIPCHK(hipMalloc((void **)&c->cb.cnt, (size_t)ng * 4));
    HIPCHK(hipMalloc((void **)&c->cb.flags, (size_t)(2 * ng + 8) * 4));      // (flags | sbase | dhead | dcnt: one fill per scan, chain_bufs)
    HIPCHK(hipMalloc((void **)&c->cb.dlist, (size_t)ng * 4));
    HIPCHK(hipMalloc((void **)&c->cb.ilist, (size_t)ng * 4));
    HIPCHK(hipMalloc((void **)&c->cb.lines, (size_t)ng * 4));
    HIPCHK(hipMalloc((void **)&c->cb.qb, (size_t)ng * 8));
    HIPCHK(hipMalloc((void **)&c->cb.term, (size_t)ng * sizeof(GroupTerm)));
    HIPCHK(hipMalloc((void **)&c->cb.rloc, (size_t)ng * 8));
    HIPCHK(hipMalloc((void **)&c->cb.qloc, (size_t)ng * 8));
    HIPCHK(hipMalloc((void **)&c->cb.part, (size_t)nblk * 4 * 8));
    HIPCHK(hipMalloc((void **)&c->cb.mins, 16));
    HIPCHK(hipMalloc((void **)&c->cb.force, (size_t)ng * 8));
    {
        const int64_t nsb = (ntiles + SB_TILES - 1) / SB_TILES;
        HIPCHK(hipMalloc((void **)&c->sbbase, (size_t)nsb * sizeof(long long)));
        HIPCHK(hipMalloc((void **)&c->tinfo4, (size_t)ntiles * sizeof(TermInfo4)));
        HIPCHK(hipMalloc((void **)&c->tileq, (size_t)ntiles * sizeof(TileQ)));
        HIPCHK(hipMalloc((void **)&c->sbq, (size_t)nsb * sizeof(unsigned int)));
        HIPCHK(hipMalloc((void **)&c->sbqbase, (size_t)nsb * sizeof(long long)));
        if (!c->hdr4) HIPCHK(hipMalloc((void **)&c->hdr4, sizeof(Fast4Hdr)));
    }
    c->cap_tiles = ntiles;
    c->cap_groups = ng;
    return FFQ_OK;
}

// the walked groups' stage (k_dense_walk): `chunks` chunks of DCHUNK records
static int reserve_dstage(ffq_ctx *c, int64_t chunks)
{
    if (chunks <= c->dstage_chunks) return FFQ_OK;
    CTX_SYNC(c);
    (void)hipFree(c->cb.dstage);
    c->cb.dstage = nullptr; c->dstage_chunks = 0; c->cb.dchunks = 0;
    hipError_t e = hipMalloc((void **)&c->cb.dstage, (size_t)chunks * DCHUNK * sizeof(StageRec));
    if (e != hipSuccess) return fail(FFQ_E_NOMEM, "hipMalloc(walked groups' stage, %lld chunks) failed: %s", (long long)chunks, hipGetErrorString(e));
    c->dstage_chunks = chunks;
    c->cb.dchunks = (int32_t)chunks;
    return FFQ_OK;
}

static int reserve_stage(ffq_ctx *c, int64_t ng, int nmax)
{
    const int64_t need = ng * nmax;
    if (need <= c->stage_cap) return FFQ_OK;
    CTX_SYNC(c);
    (void)hipFree(c->cb.stage);
    c->cb.stage = nullptr; c->stage_cap = 0;
    hipError_t e = hipMalloc((void **)&c->cb.stage, (size_t)need * sizeof(StageRec));
    if (e != hipSuccess) return fail(FFQ_E_NOMEM, "hipMalloc(stage) failed: %s", hipGetErrorString(e));
    c->stage_cap = need;
    return FFQ_OK;
}

// blocks of the decoded-quality stream for this scan (upper bound: half of the bytes are
// qualities at most, and no more than the caller's buffer holds)
static int64_t qdir_blocks(int64_t n_bytes, int64_t qual_cap)
{
    return (std::min<int64_t>(qual_cap, n_bytes / 2 + 16) + DQ_BLK - 1) / DQ_BLK + 1;
}

static int reserve_qdir(ffq_ctx *c, int64_t blocks)
{
    if (blocks <= c->qdir_cap) return FFQ_OK;
    CTX_SYNC(c);
    (void)hipFree(c->qdir);
    c->qdir = nullptr; c->qdir_cap = 0;
    hipError_t e = hipMalloc((void **)&c->qdir, (size_t)blocks * sizeof(int64_t));
    if (e != hipSuccess) return fail(FFQ_E_NOMEM, "hipMalloc(qdir) failed: %s", hipGetErrorString(e));
    c->qdir_cap = blocks;
    return FFQ_OK;
}

// one int64 per possible record of this scan (a record starts with "\n@" and has three more
// newlines: no more than a quarter of the bytes), capped by the caller's table
static int64_t p4s_need(int64_t n_bytes, int64_t table_cap) { return std::min<int64_t>(table_cap, n_bytes / 4 + 2); }
static int reserve_p4s(ffq_ctx *c, int64_t entries)
{
    if (entries <= c->p4s_cap) return FFQ_OK;
    CTX_SYNC(c);
    (void)hipFree(c->p4s); (void)hipFree(c->qrel);
    c->p4s = nullptr; c->qrel = nullptr; c->p4s_cap = 0;
    hipError_t e = hipMalloc((void **)&c->p4s, (size_t)entries * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->qrel, (size_t)entries * sizeof(uint32_t));
    if (e != hipSuccess) return fail(FFQ_E_NOMEM, "hipMalloc(p4s) failed: %s", hipGetErrorString(e));
    c->p4s_cap = entries;
    return FFQ_OK;
}

// the smallest pool: every region holds one full tile of newlines
constexpr unsigned long long POOL_MIN = (unsigned long long)POOL_NB * TILE;

static int reserve_pool(ffq_ctx *c, unsigned long long entries)
{
    if (entries <= c->pool_cap) return FFQ_OK;
    CTX_SYNC(c);
    (void)hipFree(c->pool);
    c->pool = nullptr; c->pool_cap = 0;
    HIPCHK(hipMalloc((void **)&c->pool, (size_t)entries * sizeof(uint16_t)));
    c->pool_cap = entries;
    return FFQ_OK;
}

extern "C" void ffq_ctx_forget(ffq_ctx *c)
{
    if (c) { c->fast4_remember = false; c->fast4_skip = 0; c->dense_skip = 0; c->lite_skip = 0; c->dense4_remember = false; c->ranked_skip = 0; c->fused_skip = 0; c->fused_backoff = 15; c->fz_in_place = false; }
}

extern "C" int ffq_ctx_reserve(ffq_ctx *c, int64_t max_bytes)
{
    if (!c || max_bytes < 0) return fail(FFQ_E_ARG, "ffq_ctx_reserve: bad argument");
    HIPCHK(hipSetDevice(c->device));
    const int64_t nt = std::max<int64_t>(tiles_for(max_bytes), 1);
    int rc = reserve_tiles(c, nt);
    if (rc) return rc;
    rc = reserve_stage(c, groups_for(nt), NMAX_FAST);
    if (rc) return rc;
    return reserve_pool(c, POOL_MIN);
}

// ---- memory plumbing -----------------------------------------------------
extern "C" int ffq_dev_alloc(ffq_ctx *c, int64_t bytes, void **dptr)
{
    if (!c || !dptr || bytes < 0) return fail(FFQ_E_ARG, "ffq_dev_alloc: bad argument");
    HIPCHK(hipSetDevice(c->device));
    *dptr = nullptr;
    hipError_t e = hipMalloc(dptr, (size_t)std::max<int64_t>(bytes, 16));
    if (e != hipSuccess) return fail(FFQ_E_NOMEM, "hipMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e));
    return FFQ_OK;
}
extern "C" int ffq_dev_free(ffq_ctx *c, void *dptr)
{
    if (!c) return fail(FFQ_E_ARG, "ffq_dev_free: ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipFree(dptr));
    return FFQ_OK;
}
extern "C" int ffq_pinned_alloc(int64_t bytes, void **hptr)
{
    if (!hptr || bytes < 0) return fail(FFQ_E_ARG, "ffq_pinned_alloc: bad argument");
    *hptr = nullptr;
    hipError_t e = hipHostMalloc(hptr, (size_t)std::max<int64_t>(bytes, 16), hipHostMallocDefault);
    if (e != hipSuccess) return fail(FFQ_E_NOMEM, "hipHostMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e));
    return FFQ_OK;
}
extern "C" int ffq_pinned_free(void *hptr)
{
    HIPCHK(hipHostFree(hptr));
    return FFQ_OK;
}
extern "C" int ffq_copy_h2d(ffq_ctx *c, void *dptr, const void *hptr, int64_t bytes, int async)
{
    mark_other(c);
    if (!c || bytes < 0) return fail(FFQ_E_ARG, "ffq_copy_h2d: bad argument");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dptr, hptr, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    if (!async) HIPCHK(hipStreamSynchronize(c->stream));
    return FFQ_OK;
}
extern "C" int ffq_copy_d2h(ffq_ctx *c, void *hptr, const void *dptr, int64_t bytes, int async)
{
    mark_other(c);
    if (!c || bytes < 0) return fail(FFQ_E_ARG, "ffq_copy_d2h: bad argument");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpyAsync(hptr, dptr, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    if (!async) HIPCHK(hipStreamSynchronize(c->stream));
    return FFQ_OK;
}
extern "C" int ffq_sync(ffq_ctx *c)
{
    if (!c) return fail(FFQ_E_ARG, "ffq_sync: ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FFQ_OK;
}

// ---- the hot path ----------------------------------------------------------
static void fill_result(ffq_scan_result *res, const DevRes &r, int path, int retries)
{
    res->n_records = r.n_records;
    res->n_qual_bytes = r.n_qual_bytes;
    res->end_offset = r.end_offset;
    for (int i = 0; i < 6; i++) res->last_pos[i] = r.last_pos[i];
    res->last_status = r.last_status;
    res->end_state = r.end_state;
    res->path = path;
    res->retries = retries;
    res->n_lines = r.n_lines;
}

// ---- one scan = front (enqueue) + finish (sync, fallbacks) ---------------------------------
// ffq_scan_submit enqueues the front and returns; ffq_scan_wait finishes.  ffq_scan_device is
// submit + wait.  Two contexts that share their streams (ffq_ctx_create_shared) let a host
// keep the next batch's kernels queued behind the current one's: no idle GPU between steps.
static LineIndex make_index(ffq_ctx *c, const ScanArgs &a, int64_t ntiles)
{
    LineIndex L;
    L.d = a.d_buf; L.n = a.n_bytes; L.s = a.s; L.ntiles = (int32_t)ntiles; L.ready = (int32_t)ntiles; L.pad_ = 0;
    L.ent = c->ent; L.cnt = c->cnt; L.ovf = c->ovf; L.pool = c->pool; L.pool_cap = c->pool_cap;
    return L;
}

// The line-index launch: one workgroup per whole tile; the ragged last tile (if any) rides along
// as workgroup 0's second tile.  A buffer shorter than a tile is one workgroup.
static void launch_scan_lines(ffq_ctx *c, hipStream_t st, const uint8_t *d_buf, int64_t n_bytes, int64_t ntiles,
                              const LineIndex &L, uint32_t at_char, int ablate = 0, bool any_order = false,
                              int8_t *wout = nullptr, int wadd = 0)
{
    const int64_t nfull = n_bytes >> TILE_SHIFT;
    const int ragged = ntiles > nfull ? (int)nfull : -1;
    int8_t *const no_out = nullptr;
    if (wout) {
        // the WIDE instantiation (ffq_kernels.h): the index AND every byte decoded in place, for the decode of records of any
        // layout in one pass (FFQ_F_DECODE_QUAL | FFQ_F_SINGLE_PASS on the general path)
        if (nfull > 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_lines<true, 8, true>), dim3((unsigned)nfull), dim3(256), 0, st, d_buf,
                               n_bytes, c->ent, c->cnt, c->ovf, c->pool, c->pool_cap, c->ctl, 0, ablate, L, c->d_L,
                               at_char, ragged, wout, (uint32_t)wadd);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_lines<false, 8, true>), dim3(1), dim3(256), 0, st, d_buf,
                               n_bytes, c->ent, c->cnt, c->ovf, c->pool, c->pool_cap, c->ctl, 0, ablate, L, c->d_L,
                               at_char, 0, wout, (uint32_t)wadd);
        return;
    }
    if (nfull > 0 && any_order)
        // No barrier in front of this dispatch: it may start while the kernel queued before it (the previous
        // scan's last, one-workgroup kernel -- another context's, on the same stream) is still running.  The
        // index kernel reads the caller's bytes and writes this context's own scratch, nothing the previous
        // scan touches; the kernels behind it are ordinary launches and wait for everything in front of them.
        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_lines<true, 8>), dim3((unsigned)nfull), dim3(256), 0, st, nullptr, nullptr,
                              hipExtAnyOrderLaunch, d_buf, n_bytes, c->ent, c->cnt, c->ovf, c->pool, c->pool_cap, c->ctl, 0,
                              ablate, L, c->d_L, at_char, ragged, no_out, 0u);
    else if (nfull > 0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_lines<true, 8>), dim3((unsigned)nfull), dim3(256), 0, st, d_buf,
                           n_bytes, c->ent, c->cnt, c->ovf, c->pool, c->pool_cap, c->ctl, 0, ablate, L, c->d_L,
                           at_char, ragged, no_out, 0u);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_lines<false, 8>), dim3(1), dim3(256), 0, st, d_buf,
                           n_bytes, c->ent, c->cnt, c->ovf, c->pool, c->pool_cap, c->ctl, 0, ablate, L, c->d_L,
                           at_char, 0, no_out, 0u);
}

// Phred decode of the finished table: the grid covers the largest possible quality stream;
// workgroups past the real end return at once
static void enqueue_decode(ffq_ctx *c, const ScanArgs &a, hipStream_t st, bool timed = false)
{
    if (timed) { (void)hipEventRecord(c->ev[6], st); c->decode_timed = true; }
    const int64_t nblk = qdir_blocks(a.n_bytes, a.qual_cap);
    static const int ablate = (PROBES && getenv("FFQ_DQ_ABLATE")) ? atoi(getenv("FFQ_DQ_ABLATE")) : 0;
    hipLaunchKernelGGL(k_decode_stream, dim3((unsigned)nblk), dim3(256), 0, st, a.d_buf, a.n_bytes, a.s,
                       (const int64_t *)c->p4s, (const int64_t *)a.d_qoff, (const int64_t *)c->qdir,
                       (const DevRes *)c->dres, std::min<int64_t>(a.table_cap, c->p4s_cap), a.add, a.qual_add, a.d_qual,
                       a.qual_cap, ablate);
}

// ---- the single-pass index + decode front (ffq_fused.h) ---------------------------------------------
static LineIndex make_index(ffq_ctx *c, const ScanArgs &a, int64_t ntiles);

static_assert(SG_STRIDE == FFQ_SEG_STRIDE, "include/ffq.h and csrc/ffq_fused.h disagree on the segment stride");

static int enqueue_fused_index(ffq_ctx *c, const ScanArgs &a, int64_t ntiles, bool in_place)
{
    if (ntiles > c->fz_tiles_cap) {
        CTX_SYNC(c);
        (void)hipFree(c->fz_qphase);
        c->fz_qphase = nullptr; c->fz_tiles_cap = 0;
        if (hipMalloc((void **)&c->fz_qphase, (size_t)ntiles) != hipSuccess) return fail(FFQ_E_NOMEM, "hipMalloc(fused scratch) failed");
        c->fz_tiles_cap = ntiles;
    }
    if (!c->fz_bad && hipMalloc((void **)&c->fz_bad, 16) != hipSuccess) return fail(FFQ_E_NOMEM, "hipMalloc failed");
    hipStream_t sA = c->stream;
    HIPCHK(hipMemsetAsync(c->fz_bad, 0, 16, sA));
    SegArgs fa{};
    fa.d = a.d_buf; fa.n = a.n_bytes; fa.s = a.s; fa.ntiles = (int32_t)ntiles;
    fa.ent = c->ent; fa.cnt = c->cnt;
    fa.qphase = c->fz_qphase; fa.bad = c->fz_bad;
    fa.out = a.d_qual; fa.out_cap = a.qual_cap; fa.qadd = a.qual_add; fa.at_char = (uint32_t)'@';
    fa.Lval = make_index(c, a, ntiles); fa.d_L = c->d_L;      // (the device copy of the index descriptor, as k_scan_lines leaves it)
    HIPCHK(hipEventRecord(c->ev[0], sA));
    if (in_place) hipLaunchKernelGGL(k_scan_ident, dim3((unsigned)ntiles), dim3(256), 0, sA, fa);
    else hipLaunchKernelGGL(k_scan_seg, dim3((unsigned)ntiles), dim3(256), 0, sA, fa);
    HIPCHK(hipEventRecord(c->ev[1], sA));
    return FFQ_OK;
}

static Pub make_pub(ffq_ctx *c) { return Pub{c->ctl, c->hm_ctl, c->hm_res, c->pub_seq ? c->hm_seq : nullptr, c->pub_seq}; }
static Pub no_pub(ffq_ctx *c) { return Pub{c->ctl, nullptr, nullptr, nullptr, 0}; }

// FFQ_F_POLL_RESULT: wait for the publisher of a front by polling the host-mapped completion word
// (every event record on the stream is a few microseconds of idle GPU; a front that ends in a
// publisher needs none to say that it is through)
static int poll_seq(ffq_ctx *c, unsigned long long want)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1;; spins++) {
        if (__atomic_load_n(c->h_seq, __ATOMIC_ACQUIRE) >= want) return FFQ_OK;
        if ((spins & 0x7FFF) == 0) {
            if (c->watchdog_s > 0) {
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (dt > c->watchdog_s) return fail(FFQ_E_TIMEOUT, "scan: the result block was not published within %.1f s", dt);
                if (dt > 2e-3) usleep(100);
            }
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipSuccess) {           // the stream has drained: the publisher has run, or never will
                if (__atomic_load_n(c->h_seq, __ATOMIC_ACQUIRE) >= want) return FFQ_OK;
                return fail(FFQ_E_INTERNAL, "the result block was not published");
            }
            if (e != hipErrorNotReady) return fail(FFQ_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
        }
        __builtin_ia32_pause();
    }
}

// verification + scan of the group counts, rows, result block (publishes) [-> decode]
static int enqueue_resolve(ffq_ctx *c, const ScanArgs &a, const ChainBufs &cb, bool timed, bool mins_set = false)
{
    const bool decode = (a.flags & FFQ_F_DECODE_QUAL) != 0;
    const bool wide = c->pend.wide;          // (the qualities are there already, in place: k_expand writes qoff[i] = pos4's buffer offset)
    int64_t *qoff = decode ? a.d_qoff : nullptr;
    hipStream_t sA = c->stream;
    const int ngroups = cb.ng;
    const int nblk = (ngroups + RES_BLOCK - 1) / RES_BLOCK;
    if (!mins_set) HIPCHK(hipMemsetAsync(cb.mins, 0x7F, 16, sA));       // (the list kernel of the lean path sets them itself)
    hipLaunchKernelGGL(k_resolve_a, dim3(nblk), dim3(RES_BLOCK), 0, sA, cb);
    hipLaunchKernelGGL(k_resolve_b, dim3(1), dim3(1024), 0, sA, cb, nblk, a.eof, a.offset, a.add, c->dres);
    hipLaunchKernelGGL(k_expand, dim3(ngroups), dim3(64), 0, sA, cb, (const DevRes *)c->dres, a.add, a.d_table,
                       a.table_cap, qoff, wide ? (int64_t *)nullptr : c->qdir, c->qdir_cap, (qoff && !wide) ? c->p4s : (int64_t *)nullptr,
                       (qoff && !wide) ? std::min<int64_t>(a.table_cap, c->p4s_cap) : (int64_t)0, a.s, wide ? 1 : 0);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, sA, c->dres, a.d_table, a.table_cap, a.add, a.offset, qoff,
                       make_pub(c), wide ? a.s : -1);
    c->ctl_clean = true;
    if (decode && !wide) enqueue_decode(c, a, sA, timed);
    return FFQ_OK;
}

// the chain scratch of a scan of `ngroups` groups: the walked groups' chunk numbers and the chunk counter lie right behind the
// group flags (one fill zeroes all three at the start of the chain stage; sbase: chunk + 1, 0 = none)
static ChainBufs chain_bufs(ffq_ctx *c, int ngroups, int nmax)
{
    ChainBufs cb = c->cb;
    cb.ng = ngroups;
    cb.nmax = nmax;
    cb.prof = nullptr;
    cb.sbase = reinterpret_cast<int32_t *>(cb.flags + ngroups);
    cb.dhead = cb.flags + 2 * (size_t)ngroups;
    cb.dcnt = cb.dhead + 1;
    cb.icnt = cb.dhead + 2;
    return cb;
}

// repair pass: the groups whose entry guess the verification rejected are re-run from their
// predecessor's exit (k_repair_mark), then everything is verified again
static int enqueue_repair(ffq_ctx *c, const ScanArgs &a, const LineIndex &L, bool dense_cfg, int ngroups)
{
    ChainBufs cb = chain_bufs(c, ngroups, dense_cfg ? NMAX_DENSE : NMAX_FAST);
    hipStream_t sA = c->stream;
    hipLaunchKernelGGL(k_repair_mark, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, sA, cb);
    if (!dense_cfg)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_wave<PER_FAST, EMAX_FAST, WPB_FAST, false>),
                           dim3((ngroups + WPB_FAST - 1) / WPB_FAST), dim3(WPB_FAST * 64), 0, sA, L,
                           (const LineIndex *)c->d_L, a.offset, a.eof, cb, 0, ngroups, 2, 0);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_wave<PER_DENSE, EMAX_DENSE, WPB_DENSE, true>),
                           dim3((ngroups + WPB_DENSE - 1) / WPB_DENSE), dim3(WPB_DENSE * 64), 0, sA, L,
                           (const LineIndex *)c->d_L, a.offset, a.eof, cb, 0, ngroups, 2, 0);
    // the groups that do not fit the kernel above (dense tiles) and whose entry is known: walked
    hipLaunchKernelGGL(k_dense_walk, dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, sA, L, cb, a.offset, a.eof, 0, dense_cfg ? 1 : 0, c->ctl, 0);
    return enqueue_resolve(c, a, cb, false);
}

// general path: chain summaries -> resolve -> expand -> finalize (publishes) [-> decode]
static int enqueue_general(ffq_ctx *c, const ScanArgs &a, const LineIndex &L, bool dense_cfg, int ngroups,
                           bool timed = false)
{
    const int nmax = dense_cfg ? NMAX_DENSE : NMAX_FAST;
    int rc = reserve_stage(c, ngroups, nmax);
    if (rc) return rc;
    rc = reserve_dstage(c, 64);             // (8 MiB to start with; a scan that walks more groups asks for more: ERR_DSTAGE)
    if (rc) return rc;
    ChainBufs cb = chain_bufs(c, ngroups, nmax);
    hipStream_t sA = c->stream;
    const char *abl = PROBES ? getenv("FFQ_ABLATE") : nullptr;
    const int ablate = abl ? atoi(abl) : 0;
    if (PROBES && getenv("FFQ_PROF")) {
        if (!c->prof_d) HIPCHK(hipMalloc((void **)&c->prof_d, 128));
        HIPCHK(hipMemsetAsync(c->prof_d, 0, 128, sA));
        cb.prof = c->prof_d;
    }
    HIPCHK(hipMemsetAsync(cb.flags, 0, (2 * (size_t)ngroups + 3) * 4, sA));      // flags, and: no group has a chunk of the walked groups' stage yet
    static const bool no_lite = getenv("FFQ_NO_LITE") != nullptr;
    bool lite = !dense_cfg && ablate == 0 && (!cb.prof || (PROBES && getenv("FFQ_PROF_LITE"))) && !no_lite;
    // (a context whose recent input the lean kernel mostly declined -- tiles of more than LT_E lines that still fit the usual
    // window: lines of 43-48 bytes -- runs k_chain_wave directly for a while: the list kernel is the slower way to run many groups)
    if (lite && c->lite_skip > 0) { c->lite_skip--; lite = false; }
    c->lite_ran = lite;
    if (lite) {
        // ordinary groups by the lean kernel; what it declines (flag bit 3) goes to k_chain_wave right behind.  The groups it
        // cannot take by their place -- the first one (sentinel, search offset), the last ones (the buffer's end) -- are
        // k_chain_wave's from the start: two small launches IN FRONT of the lean kernel's, which is dispatched without a barrier
        // so that their one-wave latency (30 us) runs under it (all three read the same finished line index and write different
        // groups' summaries)
        const int gl = lite_first_end_group(a.n_bytes, L.ntiles, ngroups);
        // (round 5: those groups go onto the lean kernel's list of declined groups and are run by the list kernel behind
        // it with whatever else was declined -- one launch of one-wave latency instead of three, and none in FRONT of the lean
        // kernel, where round 4's ordinary launch of the first group held the whole GPU for 29 us)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_lite<WPB_LITE>), dim3((ngroups + WPB_LITE - 1) / WPB_LITE), dim3(WPB_LITE * 64), 0, sA,
                           L, a.offset, cb, ngroups, gl, (PROBES && getenv("FFQ_LITE_ABLATE")) ? atoi(getenv("FFQ_LITE_ABLATE")) : 0);
    }
    if (lite)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_wave_list<PER_FAST, EMAX_FAST, WPB_FAST, false>),
                           dim3(std::min((ngroups + WPB_FAST - 1) / WPB_FAST, 4096)), dim3(WPB_FAST * 64), 0, sA, L,
                           (const LineIndex *)c->d_L, a.offset, a.eof, cb);
    else if (!dense_cfg)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_wave<PER_FAST, EMAX_FAST, WPB_FAST, false>),
                           dim3((ngroups + WPB_FAST - 1) / WPB_FAST), dim3(WPB_FAST * 64), 0, sA, L,
                           (const LineIndex *)c->d_L, a.offset, a.eof, cb, 0, ngroups, 0, ablate);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_wave<PER_DENSE, EMAX_DENSE, WPB_DENSE, true>),
                           dim3((ngroups + WPB_DENSE - 1) / WPB_DENSE), dim3(WPB_DENSE * 64), 0, sA, L,
                           (const LineIndex *)c->d_L, a.offset, a.eof, cb, 0, ngroups, 0, ablate);
    // the groups the kernel above declined (dense tiles), each from a guessed entry
    if (ablate == 0)
        hipLaunchKernelGGL(k_dense_walk, dim3((unsigned)std::min((ngroups + 3) / 4, 1024)), dim3(256), 0, sA, L, cb, a.offset, a.eof, 1,
                           dense_cfg ? 1 : 0, c->ctl, 1);
    return enqueue_resolve(c, a, cb, timed, lite);
}

// front of a scan: everything on ONE in-order stream (the context's), no host synchronisation
// and no copy or fill between the kernels:
//     k_scan_lines -> (k_sbscan -> k_rows4 -> k_finalize4)  or  (general kernels) [-> decode]
// The ragged last tile of the buffer rides along in the index kernel's launch (workgroup 0's
// second tile).  The last result-writing kernel publishes the result block into host-mapped
// memory and zeroes the control block for the next scan; ev[3] follows the last kernel.  A
// second context on the same stream queues its front right behind: the GPU never idles.
// the stream now ends in a scan front of `c` (StreamTail)
static void note_scan_tail(ffq_ctx *c, const ScanArgs &a)
{
    StreamTail &tl = c->owner->tail;
    tl.is_scan = true;
    const bool decode = (a.flags & FFQ_F_DECODE_QUAL) != 0;
    tl.out_p[0] = a.d_table; tl.out_n[0] = (size_t)a.table_cap * 48;
    tl.out_p[1] = decode ? a.d_qual : nullptr; tl.out_n[1] = decode ? (size_t)a.qual_cap : 0;
    tl.out_p[2] = decode ? a.d_qoff : nullptr; tl.out_n[2] = decode ? ((size_t)a.table_cap + 1) * 8 : 0;
}

static int enqueue_front(ffq_ctx *c, ScanState &st)
{
    const ScanArgs &a = st.a;
    const bool serial = (a.flags & FFQ_F_FORCE_SERIAL) != 0;
    const bool decode = (a.flags & FFQ_F_DECODE_QUAL) != 0;
    const char *abl = PROBES ? getenv("FFQ_ABLATE") : nullptr;
    const int ablate = abl ? atoi(abl) : 0;
    const int k1abl = (PROBES && getenv("FFQ_K1_ABLATE")) ? atoi(getenv("FFQ_K1_ABLATE")) : 0;
    hipStream_t sA = c->stream;
    const int64_t ntiles = st.ntiles;
    const int nsb = (int)((ntiles + SB_TILES - 1) / SB_TILES);
    // the four-line fast path (ffq_rows4.h) is tried first unless it already failed on this buffer
    // (nor while the context remembers that its recent input was not four-line)
    if (c->ranked_skip > 0 && !serial && !st.index_done && ablate == 0) { c->ranked_skip--; st.go_ranked = true; }
    if ((a.flags & FFQ_F_FORCE_RANKED) && !serial) st.go_ranked = true;
    st.probe4 = false;
    if (c->fast4_remember && !st.fast4_failed) {
        st.fast4_failed = true;
        if (c->fast4_skip > 0) c->fast4_skip--;
        else st.probe4 = !serial && !st.go_ranked && !st.index_done && ablate == 0;
    }
    if (c->dense_skip > 0 && !st.dense_cfg && !st.index_done) { c->dense_skip--; st.dense_cfg = true; }
    const bool try_fast4 = !serial && !st.go_ranked && !st.dense_cfg && !st.fast4_failed && ablate == 0 &&
                           !(a.flags & FFQ_F_FORCE_GENERAL) && getenv("FFQ_NO_FAST4") == nullptr;
    const LineIndex L = make_index(c, a, ntiles);
    c->decode_timed = false;
    // without the decode every front ends in a publisher: it can say "done" itself
    st.poll_seq = ((a.flags & FFQ_F_POLL_RESULT) && !decode) ? ++c->seq : 0;
    c->pub_seq = st.poll_seq;
    c->ctl_was_clean = c->ctl_clean;
    if (!c->ctl_clean) HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(Ctl), sA));    // first scan, or an abandoned front
    c->ctl_clean = false;

    // ---- four-line input with the decode: index AND decoded stream in one pass over the bytes ---------
    st.fused = false;
    // (two layouts, both "record i at d_qual + d_qoff[i]", include/ffq.h: SEGMENTED -- SG_STRIDE bytes of the caller's buffer
    // per tile, lines up to SG_EXT bytes behind a tile's end -- and IN PLACE -- TILE bytes per tile, lines of any length, a
    // few per cent slower on short reads (more bytes written); both write whole 16-byte pieces: an unaligned d_qual gets the
    // packed stream.  The segmented one first; once it has refused a buffer for its shape the in-place one, which the
    // context then remembers.)
    st.in_place = (c->fz_in_place || st.seg_refused) && a.qual_cap >= ntiles * (int64_t)TILE;
    if (decode && (a.flags & FFQ_F_SINGLE_PASS) && try_fast4 && !st.index_done && !st.no_fused && a.offset < 16 &&
        (st.in_place || (!st.seg_refused && a.qual_cap >= ntiles * (int64_t)SG_STRIDE)) && (reinterpret_cast<uintptr_t>(a.d_qual) & 15) == 0) {
        if (c->fused_skip > 0) c->fused_skip--;
        else st.fused = true;
    }
    if (st.fused) {
        int rc = enqueue_fused_index(c, a, ntiles, st.in_place);
        if (rc) return rc;
        const unsigned int *presum = nullptr;
        if (nsb > 2048) {
            hipLaunchKernelGGL(k_sum64, dim3((unsigned)((nsb + 3) / 4)), dim3(256), 0, sA, (const uint32_t *)c->cnt, 1,
                               (int64_t)ntiles, c->sbq, nsb);
            presum = c->sbq;
        }
        hipLaunchKernelGGL(k_sbscan, dim3(1), dim3(1024), 0, sA, L, nsb, c->sbbase, a.offset, c->hdr4, presum);
        auto rows4 = st.in_place ? k_rows4<2, false> : k_rows4<1, false>;
        hipLaunchKernelGGL(rows4, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, sA, L,
                           (const long long *)c->sbbase, a.eof, a.add, c->hdr4, c->tinfo4, a.d_table, a.table_cap,
                           (uint32_t *)nullptr, c->tileq, (int64_t *)nullptr, a.table_cap,
                           (int64_t)(st.in_place ? TILE : SG_STRIDE), (const uint8_t *)c->fz_qphase, a.d_qoff);
        hipLaunchKernelGGL(k_finalize4, dim3(1), dim3(64), 0, sA, L, c->hdr4, (const TermInfo4 *)c->tinfo4, a.eof,
                           a.offset, a.add, (const int64_t *)a.d_table, a.table_cap, c->dres, no_pub(c), (const uint32_t *)c->fz_bad);
        hipLaunchKernelGGL(k_qtotal4, dim3(1), dim3(1), 0, sA, c->dres, (const int64_t *)a.d_table, a.table_cap,
                           a.d_qoff, make_pub(c));
        c->ctl_clean = true;
        st.stage = 1;
        c->pub_seq = 0;
        if (!st.poll_seq) HIPCHK(hipEventRecord(c->ev[3], sA));
        HIPCHK(hipGetLastError());
        note_scan_tail(c, a);
        return FFQ_OK;
    }

    // ---- line index --------------------------------------------------------------------
    // (FFQ_F_NO_TIMING with the polled completion: no stream marker anywhere in the front -- each is a barrier
    // packet with a few microseconds of idle GPU around it)
    st.untimed = (a.flags & FFQ_F_NO_TIMING) && st.poll_seq && !st.index_done;
    // ... and the index kernel without a barrier in front of it only where the library itself can vouch for it (StreamTail)
    bool any_order = st.untimed && c->ctl_was_clean;
    {
        const StreamTail &tl = c->owner->tail;
        if (!tl.is_scan || tl.exposed) any_order = false;
        const uintptr_t b0 = reinterpret_cast<uintptr_t>(a.d_buf), b1 = b0 + (size_t)a.n_bytes;
        for (int i = 0; i < 3 && any_order; i++) {
            const uintptr_t o0 = reinterpret_cast<uintptr_t>(tl.out_p[i]), o1 = o0 + tl.out_n[i];
            if (tl.out_p[i] && o0 < b1 && b0 < o1) any_order = false;
        }
    }
    // The decode of records of ANY layout in one pass (round 6): a scan that starts on the general path (wrapped records:
    // the context remembers them; FFQ_F_FORCE_GENERAL; the ranking / one-wave tiers) with FFQ_F_SINGLE_PASS and
    // FFQ_INPLACE_STRIDE bytes of quality buffer per tile has its index kernel write EVERY byte decoded at the offset it has
    // in the buffer (k_scan_lines<.., WIDE>): no tier of the scan then runs a decode kernel -- qoff[i] = pos4's offset --
    // and the input is read once.  (The four-line fast path has its own single passes, above; a scan that comes here from a
    // refused fast path already has its index and takes the packed decode, as before.)
    static const bool no_wide = getenv("FFQ_NO_WIDE") != nullptr;
    if (!st.index_done)
        st.wide = decode && (a.flags & FFQ_F_SINGLE_PASS) && !try_fast4 && ablate == 0 && !no_wide &&
                  a.qual_cap >= ntiles * (int64_t)TILE && (reinterpret_cast<uintptr_t>(a.d_qual) & 15) == 0;
    if (!st.untimed) HIPCHK(hipEventRecord(c->ev[0], sA));
    if (!st.index_done)          // (a later tier of the same scan: the index is there already)
        launch_scan_lines(c, sA, a.d_buf, a.n_bytes, ntiles, L, (uint32_t)'@', k1abl, any_order && !st.wide,
                          st.wide ? a.d_qual : (int8_t *)nullptr, a.qual_add);
    if (!st.untimed) HIPCHK(hipEventRecord(c->ev[1], sA));

    if (try_fast4) {
        // ---- plain four-line records: rows straight from newline ordinals, then validated -----
        const unsigned int *presum = nullptr;
        if (nsb > 2048) {
            // a large buffer: the per-superblock sums by many workgroups, the scan kernel only scans
            hipLaunchKernelGGL(k_sum64, dim3((unsigned)((nsb + 3) / 4)), dim3(256), 0, sA, (const uint32_t *)c->cnt, 1,
                               (int64_t)ntiles, c->sbq, nsb);
            presum = c->sbq;
        }
        hipLaunchKernelGGL(k_sbscan, dim3(1), dim3(1024), 0, sA, L, nsb, c->sbbase, a.offset, c->hdr4, presum);
        if (decode) HIPCHK(hipMemsetAsync(c->tileq, 0, (size_t)ntiles * sizeof(TileQ), sA));
        if (c->dense4_remember) st.dense4 = true;
        auto rows4 = st.dense4 ? k_rows4<false, true> : k_rows4<false, false>;
        hipLaunchKernelGGL(rows4, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, sA, L,
                           (const long long *)c->sbbase, a.eof, a.add, c->hdr4, c->tinfo4, a.d_table, a.table_cap,
                           decode ? c->qrel : (uint32_t *)nullptr, c->tileq, decode ? c->p4s : (int64_t *)nullptr,
                           decode ? std::min<int64_t>(a.table_cap, c->p4s_cap) : (int64_t)0,
                           (int64_t)0, (const uint8_t *)nullptr, (int64_t *)nullptr);
        hipLaunchKernelGGL(k_finalize4, dim3(1), dim3(64), 0, sA, L, c->hdr4, (const TermInfo4 *)c->tinfo4, a.eof,
                           a.offset, a.add, (const int64_t *)a.d_table, a.table_cap, c->dres,
                           decode ? no_pub(c) : make_pub(c));
        if (decode) {
            // quality offsets: superblock sums + scan, per-tile fix-up (+ stream directory), total
            // (publishes); then the decode itself.  All of it is skipped on the device if the
            // fast path is rejected.
            hipLaunchKernelGGL(k_sum64, dim3((unsigned)((nsb + 3) / 4)), dim3(256), 0, sA,
                               reinterpret_cast<const uint32_t *>(c->tileq) + 3, 4, (int64_t)ntiles, c->sbq, nsb);
            hipLaunchKernelGGL(k_qscan4, dim3(1), dim3(1024), 0, sA, (const unsigned int *)c->sbq, nsb, c->sbqbase);
            hipLaunchKernelGGL(k_qfix4, dim3((unsigned)((ntiles + 7) / 8)), dim3(256), 0, sA, (int)ntiles,
                               (const Fast4Hdr *)c->hdr4, (const TileQ *)c->tileq, (const long long *)c->sbqbase,
                               (const uint32_t *)c->qrel, a.d_qoff, std::min<int64_t>(a.table_cap, c->p4s_cap), c->qdir,
                               c->qdir_cap);
            hipLaunchKernelGGL(k_qtotal4, dim3(1), dim3(1), 0, sA, c->dres, (const int64_t *)a.d_table, a.table_cap,
                               a.d_qoff, make_pub(c));
            enqueue_decode(c, a, sA, true);
        }
        c->ctl_clean = true;
        st.stage = 1;
    } else {
        if (st.probe4) {
            // has the input turned plain four-line again?  The fast path's kernels (rows into the caller's
            // table, which the general kernels then write again; no decode tail), their verdict left in
            // DevRes::fast4_hint for the publisher of the general path to carry out
            const unsigned int *presum = nullptr;
            if (nsb > 2048) {
                hipLaunchKernelGGL(k_sum64, dim3((unsigned)((nsb + 3) / 4)), dim3(256), 0, sA, (const uint32_t *)c->cnt, 1,
                                   (int64_t)ntiles, c->sbq, nsb);
                presum = c->sbq;
            }
            hipLaunchKernelGGL(k_sbscan, dim3(1), dim3(1024), 0, sA, L, nsb, c->sbbase, a.offset, c->hdr4, presum);
            auto rows4 = c->dense4_remember ? k_rows4<false, true> : k_rows4<false, false>;
            hipLaunchKernelGGL(rows4, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, sA, L,
                               (const long long *)c->sbbase, a.eof, a.add, c->hdr4, c->tinfo4, a.d_table, a.table_cap,
                               (uint32_t *)nullptr, c->tileq, (int64_t *)nullptr, (int64_t)0,
                               (int64_t)0, (const uint8_t *)nullptr, (int64_t *)nullptr);
            hipLaunchKernelGGL(k_finalize4, dim3(1), dim3(64), 0, sA, L, c->hdr4, (const TermInfo4 *)c->tinfo4, a.eof,
                               a.offset, a.add, (const int64_t *)a.d_table, a.table_cap, c->dres, no_pub(c));
        }
        if (!serial && !st.go_ranked) {
            int rc = enqueue_general(c, a, L, st.dense_cfg, st.ngroups, true);
            if (rc) return rc;
        } else {
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, sA, c->dres, make_pub(c), 1);
            c->ctl_clean = true;
        }
        st.stage = 2;
    }
    // ev[3]: the last kernel of this front is through (its result block is in host memory)
    c->pub_seq = 0;                       // (publishers of later tiers, enqueued at the wait, signal with events)
    if (!st.poll_seq) HIPCHK(hipEventRecord(c->ev[3], sA));
    HIPCHK(hipGetLastError());
    note_scan_tail(c, a);
    return FFQ_OK;
}

template <class T>
static int grow_dev(ffq_ctx *c, T **p, int64_t *cap, int64_t need);

// in-place exclusive scan of nv int64 values on stream st, total into res (k_scan_i64v's contract): one workgroup for short
// arrays, two levels for long ones
static int launch_scan_i64v(ffq_ctx *c, hipStream_t st, long long *v, int64_t nv, int64_t n_rows, DevRes *res)
{
    if (nv <= 2 * 32768) {
        hipLaunchKernelGGL(k_scan_i64v, dim3(1), dim3(1024), 0, st, v, nv, n_rows, res);
        return FFQ_OK;
    }
    const int64_t nb = (nv + SCAN_BLK - 1) / SCAN_BLK;
    int rc = grow_dev(c, &c->scan_bs, &c->scan_bs_cap, nb);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_blksum, dim3((unsigned)nb), dim3(256), 0, st, (const long long *)v, nv, c->scan_bs);
    hipLaunchKernelGGL(k_scan_i64v, dim3(1), dim3(1024), 0, st, c->scan_bs, nb, n_rows, res);
    hipLaunchKernelGGL(k_scan_blkapply, dim3((unsigned)nb), dim3(256), 0, st, v, nv, (const long long *)c->scan_bs);
    return FFQ_OK;
}

// the same for u32 counts -> int64 offsets + their total (k_scan_i64's contract)
static int launch_scan_u32(ffq_ctx *c, hipStream_t st, const unsigned int *v, int64_t nv, long long *base, long long *total)
{
    if (nv <= 32768) {
        hipLaunchKernelGGL(k_scan_i64, dim3(1), dim3(1024), 0, st, v, nv, base, total);
        return FFQ_OK;
    }
    const int64_t nb = (nv + SCAN_BLK - 1) / SCAN_BLK;
    int rc = grow_dev(c, &c->scan_bs, &c->scan_bs_cap, nb);
    if (rc) return rc;
    if (!c->scan_res && hipMalloc((void **)&c->scan_res, sizeof(DevRes)) != hipSuccess) return fail(FFQ_E_NOMEM, "hipMalloc failed");
    hipLaunchKernelGGL(k_scan_blksum_u32, dim3((unsigned)nb), dim3(256), 0, st, v, nv, c->scan_bs);
    hipLaunchKernelGGL(k_scan_i64v, dim3(1), dim3(1024), 0, st, c->scan_bs, nb, (int64_t)0, c->scan_res);
    hipLaunchKernelGGL(k_scan_blkapply_u32, dim3((unsigned)nb), dim3(256), 0, st, v, nv, (const long long *)c->scan_bs, base, total);
    return FFQ_OK;
}

// ---- the list-ranking tier (ffq_ranked.h): exact on any input, cost per "\n@" match ----------------
// Returns FFQ_OK with the result published, 1 if the tier cannot run here (too many candidates for
// 32-bit ranks, no memory: the caller then takes the one-wave walker), 2 if only_if_sparse is set
// and the buffer has more than one candidate per 512 bytes.
static int run_ranked(ffq_ctx *c, const ScanArgs &a, const LineIndex &L, int64_t ntiles, bool only_if_sparse)
{
    hipStream_t sA = c->stream;
    RankBufs &R = c->rk;
    if (ntiles > c->rk_cap_tiles) {
        (void)hipFree(R.tbase); R.tbase = nullptr; c->rk_cap_tiles = 0;
        if (hipMalloc((void **)&R.tbase, (size_t)ntiles * sizeof(long long)) != hipSuccess) return 1;
        c->rk_cap_tiles = ntiles;
    }
    if (!R.root && hipMalloc((void **)&R.root, 16) != hipSuccess) return 1;
    if (!c->col_res && hipMalloc((void **)&c->col_res, sizeof(DevRes)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_rk_count, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, sA, L, R.tbase);
    if (launch_scan_i64v(c, sA, R.tbase, ntiles, (int64_t)0, c->col_res)) return 1;
    HIPCHK(hipMemcpyAsync(c->h_word, &c->col_res->n_qual_bytes, sizeof(int64_t), hipMemcpyDeviceToHost, sA));
    HIPCHK(hipGetLastError());
    CTX_SYNC(c);
    const int64_t nc = c->h_word[0];
    if (nc >= 0x7FFFFFF0ll) return 1;
    // a context that remembers long records meets short ones again: back to the group kernels
    if (only_if_sparse && nc * 512 > a.n_bytes) return 2;
    if (nc > c->rk_cap_c) {
        (void)hipFree(R.cand); (void)hipFree(R.rec); (void)hipFree(R.succ); (void)hipFree(R.D);
        for (int i = 0; i < 2; i++) { (void)hipFree(R.S[i]); (void)hipFree(R.C[i]); R.S[i] = R.C[i] = nullptr; }
        R.cand = nullptr; R.rec = nullptr; R.succ = nullptr; R.D = nullptr; c->rk_cap_c = 0;
        const size_t n = (size_t)nc + (nc >> 3) + 1024;
        bool ok = hipMalloc((void **)&R.cand, n * sizeof(H)) == hipSuccess && hipMalloc((void **)&R.rec, n * sizeof(RankRec)) == hipSuccess &&
                  hipMalloc((void **)&R.succ, n * 4) == hipSuccess && hipMalloc((void **)&R.D, n * 4) == hipSuccess;
        for (int i = 0; i < 2 && ok; i++)
            ok = hipMalloc((void **)&R.S[i], n * 4) == hipSuccess && hipMalloc((void **)&R.C[i], n * 4) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); return 1; }
        c->rk_cap_c = (int64_t)n;
    }
    R.nc = nc;
    const unsigned gthr = (unsigned)std::max<int64_t>((nc + 255) / 256, 1);
    if (nc > 0) {
        hipLaunchKernelGGL(k_rk_list, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, sA, L, R);
        hipLaunchKernelGGL(k_rk_succ, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, sA, L, R, a.eof);
    }
    hipLaunchKernelGGL(k_rk_root, dim3(1), dim3(1024), 0, sA, L, R, a.offset, c->dres);
    int rounds = 0;
    while (((int64_t)1 << rounds) <= nc) rounds++;
    int cur = 0;
    for (int k = 0; k < rounds; k++, cur ^= 1)
        hipLaunchKernelGGL(k_rk_round, dim3(gthr), dim3(256), 0, sA, nc, (const uint32_t *)R.S[cur], (const uint32_t *)R.C[cur],
                           R.S[cur ^ 1], R.C[cur ^ 1], R.D, k);
    hipLaunchKernelGGL(k_rk_emit, dim3(gthr), dim3(256), 0, sA, L, R, a.eof, a.offset, a.add, a.d_table, a.table_cap, c->dres);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, sA, c->dres, (const int64_t *)a.d_table, a.table_cap, a.add, a.offset,
                       (int64_t *)nullptr, make_pub(c), -1);
    c->ctl_clean = true;
    HIPCHK(hipGetLastError());
    return FFQ_OK;
}

// quality offsets of a finished table (n rows known to the host) + the decode: the column kernels
// with columns 4 / 5 (what the fast path and k_expand do on the way, for the tiers that do not)
static int enqueue_offsets_and_decode(ffq_ctx *c, const ScanArgs &a, int64_t n_rows)
{
    hipStream_t sA = c->stream;
    if (n_rows <= 0 || n_rows > a.table_cap) {
        if (n_rows == 0) HIPCHK(hipMemsetAsync(a.d_qoff, 0, sizeof(int64_t), sA));     // qoff[0] = 0
        return FFQ_OK;
    }
    const int64_t nblk = (n_rows + 255) / 256;
    int rc = grow_dev(c, &c->col_sum, &c->col_sum_cap, nblk);
    if (rc) return rc;
    hipLaunchKernelGGL(k_col_sum, dim3((unsigned)nblk), dim3(256), 0, sA, (const int64_t *)a.d_table, n_rows, 4, 0, 5, c->col_sum);
    // (the scan's totals go into the scan's own result block: the decode kernel and the host read them there)
    if ((rc = launch_scan_i64v(c, sA, c->col_sum, nblk, n_rows, c->dres))) return rc;
    hipLaunchKernelGGL(k_col_offsets, dim3((unsigned)nblk), dim3(256), 0, sA, (const int64_t *)a.d_table, n_rows, 4, 0, 5,
                       (const long long *)c->col_sum, (const DevRes *)c->dres, a.d_qoff, c->p4s, c->qdir, c->qdir_cap);
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, sA, c->dres, make_pub(c), 0);
    enqueue_decode(c, a, sA);
    return FFQ_OK;
}

// ... and for a scan whose index pass decoded everything in place (ScanState::wide): only the offsets, qoff[i] = the
// offset pos4 has in the buffer, from the finished table
static int enqueue_offsets_in_place(ffq_ctx *c, const ScanArgs &a, int64_t n_rows)
{
    (void)n_rows;
    hipLaunchKernelGGL(k_qoff_in_place, dim3(256), dim3(256), 0, c->stream, c->dres, (const int64_t *)a.d_table, a.table_cap, a.add, a.s,
                       a.d_qoff, make_pub(c));
    return FFQ_OK;
}

static int scan_finish(ffq_ctx *c, ScanState &st, ffq_scan_result *res)
{
    const ScanArgs &a = st.a;
    const bool serial = (a.flags & FFQ_F_FORCE_SERIAL) != 0;
    const bool decode = (a.flags & FFQ_F_DECODE_QUAL) != 0;
    int64_t *qoff = decode ? a.d_qoff : nullptr;
    hipStream_t sA = c->stream;
    bool front_done = true;           // the first front was enqueued by the caller (submit)
    for (;;) {
        if (!front_done) {
            int rc = enqueue_front(c, st);
            if (rc) return rc;
        }
        front_done = false;
        // wait for THIS scan's front only: the stream may already hold the next scan of a
        // context that shares it
        if (st.poll_seq) {
            int rc = poll_seq(c, st.poll_seq);
            if (rc) return rc;
            // (the GPU is past this mark; the wait only lets the runtime note it before the mark is read)
            if (!st.untimed) CTX_WAIT_EVENT(c, c->ev[1]);
        } else CTX_WAIT_EVENT(c, c->ev[3]);
        const LineIndex L = make_index(c, a, st.ntiles);

        if (c->h_ctl->err & ERR_POOL) {
            // dense tiles did not fit the overflow pool: size it for what was asked and re-run
            if (st.retries >= 2) return fail(FFQ_E_INTERNAL, "line-index pool overflow persists");
            int rc = reserve_pool(c, std::max<unsigned long long>(c->h_ctl->pool_head + (c->h_ctl->pool_head >> 4), POOL_MIN));
            if (rc) return rc;
            st.retries++;
            st.index_done = false;
            continue;
        }
        if (c->h_ctl->err & ERR_INTERNAL) return fail(FFQ_E_INTERNAL, "chain kernel invariant failed");
        // more groups were walked (dense regions, ffq_dense.h) than their stage has chunks for: size it for what was
        // asked and run the chain kernels again, from the index that is there
        auto dstage_short = [&]() -> int {
            if (!(c->h_ctl->err & ERR_DSTAGE)) return 0;
            if (st.retries >= 4) return fail(FFQ_E_INTERNAL, "the walked groups' stage stays too small");
            uint32_t asked = 0;
            HIPCHK(hipMemcpy(&asked, c->cb.flags + 2 * (size_t)st.ngroups, sizeof asked, hipMemcpyDeviceToHost));
            int rc = reserve_dstage(c, std::max<int64_t>((int64_t)asked + (asked >> 3) + 8, 2 * c->dstage_chunks));
            if (rc) return rc;
            st.retries++;
            st.index_done = true;
            return 1;
        };
        float ms = 0;
        if (!st.untimed) HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        if (!st.index_done) res->ms_index = ms;
        st.untimed = false;
        st.index_done = true;
        res->ms_decode = 0;
        if (!st.poll_seq) {                // (a polled front has no end mark: its tail is not timed)
            HIPCHK(hipEventElapsedTime(&ms, c->ev[1], c->ev[3])); res->ms_chain += ms;
            if (c->decode_timed) {
                // the decode kernel is the tail of the front: split it off the chain time
                HIPCHK(hipEventElapsedTime(&ms, c->ev[6], c->ev[3]));
                res->ms_decode = ms; res->ms_chain -= ms;
            }
            HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[3])); res->ms_total += ms;
        }

        const bool tiers = !serial && !st.go_ranked;       // the group kernels ran: their fallbacks apply
        if (st.probe4) {
            // the probe's verdict: the next scan starts with the fast path again, or 15 more do not ask
            st.probe4 = false;
            if (c->h_res->fast4_hint == 1) { c->fast4_remember = false; c->fast4_skip = 0; }
            else c->fast4_skip = 15;
        }
        if (st.stage == 1) {
            if (!c->h_res->fallback) {
                fill_result(res, *c->h_res, st.fused ? 6 : 3, st.retries);
                if (st.fused) { c->fused_backoff = 15; c->fz_in_place = st.in_place; }
                break;
            }
            if (st.fused) {
                // the single pass did not stand (lines longer than a tile, a quality line that is not as
                // long as its sequence line, text in front of the first record, not four-line input at
                // all ...): the two-pass kernels, from the index it built if that is whole
                if (PROBES && getenv("FFQ_DEBUG")) {
                    Fast4Hdr hh;
                    HIPCHK(hipMemcpy(&hh, c->hdr4, sizeof hh, hipMemcpyDeviceToHost));
                    fprintf(stderr, "[ffq debug] single pass (%s) refused: fused_bad %d attempt %d j0 %lld irr_min %llu term_min %llu\n",
                            st.in_place ? "in place" : "segmented", c->h_res->fused_bad, hh.attempt, hh.j0, hh.irr_min, hh.term_min);
                }
                if (!st.in_place && !st.seg_refused && (c->h_res->fused_bad & (int32_t)FZ_BAD_SHAPE) &&
                    !(c->h_res->fused_bad & (int32_t)FZ_BAD_INDEX) && a.qual_cap >= st.ntiles * (int64_t)TILE) {
                    // a shape the segments cannot take (a line longer than SG_EXT behind a tile, no S P pair in a tile
                    // of a few long lines) and the caller's buffer has room for the in-place layout: that pass next
                    st.seg_refused = true;
                    st.fused = false;
                    st.index_done = false;          // (that pass builds the index itself)
                    st.retries++;
                    continue;
                }
                c->fz_in_place = false;
                st.no_fused = true;
                st.fused = false;
                c->fused_skip = c->fused_backoff;
                c->fused_backoff = std::min(4 * c->fused_backoff + 3, 1023);
                st.index_done = !(c->h_res->fused_bad & (int32_t)FZ_BAD_INDEX);
                st.retries++;
                continue;
            }
            if (PROBES && getenv("FFQ_DEBUG") && !st.fused) {
                Fast4Hdr hh;
                HIPCHK(hipMemcpy(&hh, c->hdr4, sizeof hh, hipMemcpyDeviceToHost));
                fprintf(stderr, "[ffq debug] fast path refused: dense4 %d attempt %d dense_seen %d j0 %lld irr_min %llu term_min %llu (k %llu tile %llu)\n",
                        (int)st.dense4, hh.attempt, hh.dense_seen, hh.j0, hh.irr_min, hh.term_min, hh.term_min >> 24, hh.term_min & 0xFFFFFF);
            }
            if (c->h_res->fast4_dense && !st.dense4 && !st.fused && !st.no_fused) {
                // the row kernel met a DENSE tile (lines under 16 bytes on average: reads of a dozen bases): its DENSE
                // instantiation takes those -- the same front again from the index that is there, and the context's next
                // scans start with it
                st.dense4 = true;
                c->dense4_remember = true;
                st.retries++;
                continue;
            }
            // not plain four-line input: the general kernels, from the same line index.  The next
            // scans of this context skip the attempt (and the host round trip it costs here).
            st.fast4_failed = true;
            c->fast4_remember = true;
            c->fast4_skip = 15;
            HIPCHK(hipEventRecord(c->ev[4], sA));
            int rc = enqueue_general(c, a, L, st.dense_cfg, st.ngroups);
            if (rc) return rc;
            HIPCHK(hipEventRecord(c->ev[2], sA));
            HIPCHK(hipGetLastError());
            CTX_WAIT_EVENT(c, c->ev[2]);
            HIPCHK(hipEventElapsedTime(&ms, c->ev[4], c->ev[2]));
            res->ms_chain += ms; res->ms_total += ms;
            if (c->h_ctl->err & ERR_INTERNAL) return fail(FFQ_E_INTERNAL, "chain kernel invariant failed");
        }
        if (PROBES && tiers && getenv("FFQ_PROF") && c->prof_d) {
            unsigned long long hp[16];
            HIPCHK(hipMemcpy(hp, c->prof_d, 128, hipMemcpyDeviceToHost));
            if (getenv("FFQ_PROF_LITE"))
                fprintf(stderr, "[ffq prof] k_chain_lite declined of %d groups: first/last/offset %llu, tile over %d entries / dense look-ahead %llu, no node / too many %llu, "
                        "a node it does not take on the chain %llu, stage full %llu; runs walked %llu, successors found entry by entry %llu\n", st.ngroups, hp[8], LT_E, hp[9], hp[10], hp[11], hp[12], hp[14], hp[13]);
            if (hp[6])
                fprintf(stderr, "[ffq prof] k_chain_wave per-wave cycles: load %.0f lds %.0f nodes %.0f scan %.0f member %.0f summary %.0f (waves %llu)\n",
                        (double)hp[0] / hp[6], (double)hp[1] / hp[6], (double)hp[2] / hp[6], (double)hp[3] / hp[6],
                        (double)hp[4] / hp[6], (double)hp[5] / hp[6], hp[6]);
            if (hp[6])
                fprintf(stderr, "[ffq prof] generic-path nodes per wave %.2f, serial generic rounds per wave %.2f\n",
                        (double)(hp[7] & 0xFFFFFFFFull) / hp[6], (double)(hp[7] >> 32) / hp[6]);
        }
        if (PROBES && tiers && getenv("FFQ_DEBUG")) {
            const int ng = std::min(st.ngroups, 24);
            std::vector<int64_t> y(ng), ex(ng);
            std::vector<uint32_t> cn(ng), fl(ng);
            int32_t mins[2];
            HIPCHK(hipMemcpy(y.data(), c->cb.y, ng * 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(ex.data(), c->cb.exit, ng * 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(cn.data(), c->cb.cnt, ng * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(fl.data(), c->cb.flags, ng * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(mins, c->cb.mins, 8, hipMemcpyDeviceToHost));
            fprintf(stderr, "[ffq debug] dense=%d fallback=%d ngroups=%d term=%d bad=%d\n", (int)st.dense_cfg,
                    c->h_res->fallback, st.ngroups, mins[0], mins[1]);
            for (int g = 0; g < ng; g++)
                fprintf(stderr, "[ffq debug]  g=%d y=%lld exit=%lld cnt=%u flags=%u\n", g, (long long)y[g],
                        (long long)ex[g], cn[g], fl[g]);
            // ... and the first group the verification rejected, with its neighbours
            fprintf(stderr, "[ffq debug] n_bad=%d bad_irregular=%d declined=%d\n", c->h_res->n_bad, c->h_res->bad_irregular, c->h_res->n_declined);
            if (mins[1] >= 0 && mins[1] < st.ngroups) {
                const int g0 = std::max(mins[1] - 1, 0), g1 = std::min(mins[1] + 2, st.ngroups);
                for (int g = g0; g < g1; g++) {
                    int64_t yy, ee; uint32_t cc, ff;
                    HIPCHK(hipMemcpy(&yy, c->cb.y + g, 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(&ee, c->cb.exit + g, 8, hipMemcpyDeviceToHost));
                    HIPCHK(hipMemcpy(&cc, c->cb.cnt + g, 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(&ff, c->cb.flags + g, 4, hipMemcpyDeviceToHost));
                    fprintf(stderr, "[ffq debug]  g=%d y=%lld exit=%lld cnt=%u flags=%u\n", g, (long long)yy, (long long)ee, cc, ff);
                }
            }
        }
        int path = 0;
        if (PROBES && tiers && getenv("FFQ_ABLATE") && atoi(getenv("FFQ_ABLATE")) != 0) {
            // diagnostics build of the pipeline: results are meaningless, only timings count
            fill_result(res, *c->h_res, 0, st.retries);
            return FFQ_OK;
        }
        // a guess the verification rejected is repaired, not escalated: the rejected groups are
        // re-run from their predecessor's exit and everything is verified again.  Every round
        // makes the first rejected group exact, so the first bad group moves forward; a round
        // that does not move it (a group that does not fit the kernel at all) ends the repairs.
        static const bool no_ranked = getenv("FFQ_NO_RANKED") != nullptr;
        if (tiers && !(PROBES && getenv("FFQ_ABLATE") && atoi(getenv("FFQ_ABLATE")) != 0) && !getenv("FFQ_NO_REPAIR")) {
            int prev_bad = -1;
            const int first_bad = c->h_res->bad_group;
            // (a group that does not FIT the kernel -- bad_irregular -- is walked by k_group_walk in
            // the same passes; if that cannot take it either, the first bad group does not move)
            // With list ranking behind them the repair passes only mend sporadic wrong guesses (two
            // rounds); records that span whole groups correct one another a few groups per round,
            // and list ranking is the tool for that.
            // (groups that do not FIT -- dense tiles -- are walked one repair pass after their
            // predecessor: those passes go on as before)
            for (int round = 0; round < 16 && c->h_res->fallback && c->h_res->bad_group > prev_bad &&
                                c->h_res->bad_group < st.ngroups && !(c->h_ctl->err & ERR_DSTAGE); round++) {
                if (!no_ranked && round >= 2 && !c->h_res->bad_irregular) break;
                // a guess in eight did not stand and the records are long (2.5 KB and more on average:
                // wrapped reads from ~1.2 kb): repair passes would mend them a pass at a time; list ranking
                // needs no guess and, at one wave per "\n@" match, costs less than two such passes there
                // (tools/shape_sweep_wrapped.py: 1.5-3 kb reads 0.75-0.46 -> TB/s figures in DESIGN.md)
                if (!no_ranked && !c->h_res->bad_irregular && (int64_t)c->h_res->n_bad * 8 > (int64_t)st.ngroups &&
                    c->h_res->approx_records * 2560 < a.n_bytes) break;
                if (round >= 4 && !c->h_res->bad_irregular &&
                    c->h_res->bad_group - first_bad < round * std::max(st.ngroups / 64, 1)) break;
                prev_bad = c->h_res->bad_group;
                HIPCHK(hipEventRecord(c->ev[4], sA));
                int rc = enqueue_repair(c, a, L, st.dense_cfg, st.ngroups);
                if (rc) return rc;
                HIPCHK(hipEventRecord(c->ev[2], sA));
                HIPCHK(hipGetLastError());
                CTX_WAIT_EVENT(c, c->ev[2]);
                HIPCHK(hipEventElapsedTime(&ms, c->ev[4], c->ev[2]));
                res->ms_chain += ms; res->ms_total += ms;
                st.repairs++;
                if (c->h_ctl->err & ERR_INTERNAL) return fail(FFQ_E_INTERNAL, "chain kernel invariant failed");
            }
        }
        if (tiers) {
            const int ds = dstage_short();
            if (ds < 0) return ds;
            if (ds > 0) { st.fast4_failed = true; continue; }
        }
        if (tiers && c->h_res->fallback && !st.dense_cfg && c->h_res->bad_irregular) {
            // second tier: the same kernels with the LDS budget for short lines / short records
            // (only a group that does not FIT is helped by it; a guess that stays wrong is not)
            st.dense_cfg = true;
            c->dense_skip = 15;          // and the next scans of this context start there
            continue;
        }
        if (st.dense_cfg) path = 2;
        bool walk = serial || (tiers && c->h_res->fallback);
        if (!serial && !no_ranked && (st.go_ranked || c->h_res->fallback)) {
            // the group kernels could not prove a chain (records long against a group): list ranking
            HIPCHK(hipEventRecord(c->ev[4], sA));
            int rr = run_ranked(c, a, L, st.ntiles, st.go_ranked && !(a.flags & FFQ_F_FORCE_RANKED));
            if (rr < 0) return rr;
            if (rr == 2) {
                c->ranked_skip = 0;
                c->fast4_skip = 0;           // (the input has changed character: what is remembered of it is void)
                c->fast4_remember = false;
                st.go_ranked = false;
                st.fast4_failed = false;
                continue;                    // the usual tiers, from the line index that is there
            }
            walk = rr > 0;
            if (rr == 0) {
                HIPCHK(hipEventRecord(c->ev[2], sA));
                CTX_WAIT_EVENT(c, c->ev[2]);
                if (decode && !c->h_res->fallback) {
                    int rc = st.wide ? enqueue_offsets_in_place(c, a, c->h_res->n_records) : enqueue_offsets_and_decode(c, a, c->h_res->n_records);
                    if (rc) return rc;
                    HIPCHK(hipEventRecord(c->ev[2], sA));
                    HIPCHK(hipGetLastError());
                    CTX_WAIT_EVENT(c, c->ev[2]);
                }
                HIPCHK(hipEventElapsedTime(&ms, c->ev[4], c->ev[2]));
                res->ms_chain += ms; res->ms_total += ms;
                path = 5;
                if (!(a.flags & FFQ_F_FORCE_RANKED)) c->ranked_skip = 15;     // and the next scans of this context start there
            }
        } else if (st.go_ranked) walk = true;
        if (walk) {
            path = 1;
            HIPCHK(hipEventRecord(c->ev[4], sA));
            int64_t *const sq = st.wide ? (int64_t *)nullptr : qoff;       // (wide: the offsets from the finished table, below)
            hipLaunchKernelGGL(k_chain_serial, dim3(1), dim3(64), 0, sA, L, a.offset, a.eof, a.add, a.d_table,
                               a.table_cap, sq, c->qdir, c->qdir_cap, sq ? c->p4s : (int64_t *)nullptr,
                               sq ? std::min<int64_t>(a.table_cap, c->p4s_cap) : (int64_t)0, c->dres);
            hipLaunchKernelGGL(k_finalize_serial, dim3(1), dim3(64), 0, sA, c->dres, a.table_cap, sq, (decode && st.wide) ? no_pub(c) : make_pub(c));
            c->ctl_clean = true;
            if (decode && st.wide) hipLaunchKernelGGL(k_qoff_in_place, dim3(256), dim3(256), 0, sA, c->dres, (const int64_t *)a.d_table, a.table_cap, a.add, a.s, a.d_qoff, make_pub(c));
            else if (decode) enqueue_decode(c, a, sA);
            HIPCHK(hipEventRecord(c->ev[2], sA));
            HIPCHK(hipGetLastError());
            CTX_WAIT_EVENT(c, c->ev[2]);
            HIPCHK(hipEventElapsedTime(&ms, c->ev[4], c->ev[2]));
            res->ms_chain += ms;
            res->ms_total += ms;
        }
        if (tiers && c->lite_ran && (int64_t)c->h_res->n_declined * 4 > (int64_t)st.ngroups) c->lite_skip = 15;
        fill_result(res, *c->h_res, path | ((decode && st.wide) ? FFQ_PATH_IN_PLACE : 0), st.retries + st.repairs);
        break;
    }
    if (res->n_records > a.table_cap)
        return fail(FFQ_E_TABLE_FULL, "table holds %lld rows, the buffer has %lld records", (long long)a.table_cap,
                    (long long)res->n_records);
    if (decode && res->n_qual_bytes > a.qual_cap)
        return fail(FFQ_E_TABLE_FULL, "quality buffer holds %lld bytes, %lld needed", (long long)a.qual_cap,
                    (long long)res->n_qual_bytes);
    return FFQ_OK;
}

// the scratch a scan of n_bytes needs (grow-only: a second call with the same sizes does nothing)
static int scan_reserve(ffq_ctx *c, int64_t n_bytes, uint32_t flags, int64_t table_cap, int64_t qual_cap)
{
    const int64_t ntiles = tiles_for(n_bytes);
    if (ntiles == 0 || ntiles > 0x7FFFFFF0) return FFQ_OK;          // (nothing to enqueue / refused by the submit)
    int rc = reserve_tiles(c, ntiles);
    if (!rc) rc = reserve_pool(c, POOL_MIN);
    if (!rc && (flags & FFQ_F_DECODE_QUAL)) rc = reserve_qdir(c, qdir_blocks(n_bytes, qual_cap));
    if (!rc && (flags & FFQ_F_DECODE_QUAL)) rc = reserve_p4s(c, p4s_need(n_bytes, table_cap));
    return rc;
}

extern "C" int ffq_scan_submit(ffq_ctx *c, const uint8_t *d_buf, int64_t n_bytes, int sentinel, int64_t offset,
                               int eof, int64_t add, uint32_t flags, int qual_add, int64_t *d_table,
                               int64_t table_cap, int8_t *d_qual, int64_t qual_cap, int64_t *d_qoff)
{
    if (!c) return fail(FFQ_E_ARG, "ffq_scan_submit: ctx is NULL");
    if (c->pend.active) return fail(FFQ_E_ARG, "ffq_scan_submit: a scan is already pending on this context");
    if (n_bytes < 0 || offset < 0 || table_cap < 0) return fail(FFQ_E_ARG, "ffq_scan: negative size");
    if (n_bytes > 0 && !d_buf) return fail(FFQ_E_ARG, "ffq_scan: d_buf is NULL");
    if ((reinterpret_cast<uintptr_t>(d_buf) & 15) != 0) return fail(FFQ_E_ARG, "ffq_scan: d_buf must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(d_table) & 15) != 0) return fail(FFQ_E_ARG, "ffq_scan: d_table must be 16-byte aligned");
    if (table_cap > 0 && !d_table) return fail(FFQ_E_ARG, "ffq_scan: d_table is NULL");
    const bool decode = (flags & FFQ_F_DECODE_QUAL) != 0;
    if (decode && (!d_qual || !d_qoff)) return fail(FFQ_E_ARG, "ffq_scan: FFQ_F_DECODE_QUAL needs d_qual and d_qoff");
    HIPCHK(hipSetDevice(c->device));
    ScanState &st = c->pend;
    st = ScanState{};
    st.a = ScanArgs{d_buf, n_bytes, sentinel ? 1 : 0, offset, eof, add, flags, qual_add, d_table, table_cap,
                    d_qual, qual_cap, d_qoff};
    st.ntiles = tiles_for(n_bytes);
    st.active = true;
    if (st.ntiles == 0) return FFQ_OK;                   // nothing to enqueue: ffq_scan_wait fills the result
    if (st.ntiles > 0x7FFFFFF0) { st.active = false; return fail(FFQ_E_ARG, "buffer too large"); }
    int rc = scan_reserve(c, n_bytes, flags, table_cap, qual_cap);
    if (!rc) {
        st.ngroups = (int)groups_for(st.ntiles);
        rc = enqueue_front(c, st);
    }
    if (rc) st.active = false;
    return rc;
}

extern "C" int ffq_scan_wait(ffq_ctx *c, ffq_scan_result *res)
{
    if (!c || !res) return fail(FFQ_E_ARG, "ffq_scan_wait: ctx/res is NULL");
    if (!c->pend.active) return fail(FFQ_E_ARG, "ffq_scan_wait: no scan is pending on this context");
    memset(res, 0, sizeof *res);
    HIPCHK(hipSetDevice(c->device));
    ScanState &st = c->pend;
    st.active = false;
    if (st.ntiles == 0) {
        // empty data: with a sentinel the buffer is "\n", no "\n@" can match
        res->end_state = st.a.eof ? FFQ_END_OK : FFQ_END_REFILL;
        res->last_status = FFQ_POS_HEAD_BEG;
        res->end_offset = st.a.offset;
        for (int i = 0; i < 6; i++) res->last_pos[i] = -1;
        if ((st.a.flags & FFQ_F_DECODE_QUAL) != 0) {
            HIPCHK(hipMemsetAsync(st.a.d_qoff, 0, sizeof(int64_t), c->stream));
            CTX_SYNC(c);
        }
        return FFQ_OK;
    }
    return scan_finish(c, st, res);
}

extern "C" int ffq_scan_device(ffq_ctx *c, const uint8_t *d_buf, int64_t n_bytes, int sentinel,
                               int64_t offset, int eof, int64_t add, uint32_t flags, int qual_add,
                               int64_t *d_table, int64_t table_cap, int8_t *d_qual,
                               int64_t qual_cap, int64_t *d_qoff, ffq_scan_result *res)
{
    if (!c || !res) return fail(FFQ_E_ARG, "ffq_scan_device: ctx/res is NULL");
    int rc = ffq_scan_submit(c, d_buf, n_bytes, sentinel, offset, eof, add, flags, qual_add, d_table, table_cap,
                             d_qual, qual_cap, d_qoff);
    if (rc) return rc;
    return ffq_scan_wait(c, res);
}

template <class T>
static int grow_dev(ffq_ctx *c, T **p, int64_t *cap, int64_t need)
{
    if (need <= *cap) return FFQ_OK;
    CTX_SYNC(c);
    (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    const int64_t want = std::max<int64_t>(need, 1 << 16);
    hipError_t e = hipMalloc((void **)p, (size_t)want * sizeof(T));
    if (e != hipSuccess) return fail(FFQ_E_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    *cap = want;
    return FFQ_OK;
}

// ---- staging of the host-buffer entry points --------------------------------------------------
constexpr int64_t STAGE_CH = 8 << 20;

static int stage_setup(ffq_ctx *c, int64_t ch = STAGE_CH)
{
    if (c->stage_h_cap < 3 * ch) {
        if (c->stage_h) (void)hipHostFree(c->stage_h);
        c->stage_h = nullptr; c->stage_h_cap = 0;
        NearGpu near(c);
        hipError_t e = hipHostMalloc((void **)&c->stage_h, (size_t)(3 * ch), hipHostMallocDefault);
        if (e != hipSuccess) return fail(FFQ_E_NOMEM, "hipHostMalloc failed: %s", hipGetErrorString(e));
        c->stage_h_cap = 3 * ch;
    }
    for (auto &st : c->stage_cs)
        if (!st) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto &r : c->stage_ev)
        for (auto &e : r)
            if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!ctx_pool(c)) return fail(FFQ_E_NOMEM, "out of host memory");
    return FFQ_OK;
}

// device -> pageable host memory: chunks over the link into the pinned slots (after whatever the
// scan stream holds), out of them by the helper threads; the copy out of chunk k runs while chunk
// k+1 is on the link
static int stage_d2h(ffq_ctx *c, void *h_dst, const void *d_src, int64_t bytes)
{
    if (bytes <= 0) return FFQ_OK;
    int rc = stage_setup(c);
    if (rc) return rc;
    uint8_t *dst = static_cast<uint8_t *>(h_dst);
    const uint8_t *src = static_cast<const uint8_t *>(d_src);
    const int64_t nch = (bytes + STAGE_CH - 1) / STAGE_CH;
    bool busy[3] = {false, false, false};               // a host copy out of the slot is in flight
    for (int64_t k = 0; k <= nch; k++) {
        if (k < nch) {
            const int b = (int)(k % 3);
            if (busy[b]) { c->helpers->wait(&c->stage_cr[b]); busy[b] = false; }
            const int64_t at = k * STAGE_CH, m = std::min<int64_t>(STAGE_CH, bytes - at);
            HIPCHK(hipMemcpyAsync(c->stage_h + b * STAGE_CH, src + at, (size_t)m, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipEventRecord(c->stage_ev[b][0], c->stream));
        }
        if (k > 0) {
            const int b = (int)((k - 1) % 3);
            const int64_t at = (k - 1) * STAGE_CH, m = std::min<int64_t>(STAGE_CH, bytes - at);
            HIPCHK(hipEventSynchronize(c->stage_ev[b][0]));
            c->helpers->enqueue_copy(dst + at, c->stage_h + b * STAGE_CH, m, &c->stage_cr[b]);
            busy[b] = true;
        }
    }
    for (int b = 0; b < 3; b++)
        if (busy[b]) c->helpers->wait(&c->stage_cr[b]);
    return FFQ_OK;
}

// ---- a byte range of a file into device memory ----------------------------------------------------------------------
// (/root/reference/src/fastqandfurious.py:30-36 `read(fh, fbufsize)`, for a range that stays resident: a shard of a
// file, ffq_shard.h)  pread in slices by the helper threads into three pinned slots of 32 MiB, each slot over the link
// in two halves on the two copy streams while the next one is read; the context's stream waits for the last copies.
constexpr int64_t FSTAGE_CH = 32 << 20;

static int stage_fd2d(ffq_ctx *c, uint8_t *d_dst, int fd, int64_t pos, int64_t n, int64_t *got)
{
    *got = 0;
    if (n <= 0) return FFQ_OK;
    int rc = stage_setup(c, FSTAGE_CH);
    if (rc) return rc;
    mark_other(c);
    // the copies start behind whatever the scan stream holds (a scan in flight may still read d_dst)
    hipEvent_t front = nullptr;
    HIPCHK(hipEventCreateWithFlags(&front, hipEventDisableTiming));
    hipError_t e = hipEventRecord(front, c->stream);
    for (auto &st : c->stage_cs) if (e == hipSuccess) e = hipStreamWaitEvent(st, front, 0);
    (void)hipEventDestroy(front);
    if (e != hipSuccess) return fail(FFQ_E_HIP, "ffq_load_fd: %s", hipGetErrorString(e));
    bool used[3] = {false, false, false};
    const int64_t nch = (n + FSTAGE_CH - 1) / FSTAGE_CH;
    int64_t k_enq = 0, k = 0;
    static const int ablate = (PROBES && getenv("FFQ_LOAD_ABLATE")) ? atoi(getenv("FFQ_LOAD_ABLATE")) : 0;     // 1: no copies, 2: no reads, 4: one copy stream
    bool ended = false;
    for (; k < nch && !ended && !rc; k++) {
        for (; k_enq < nch && k_enq - k < 3; k_enq++) {
            const int b = (int)(k_enq % 3);
            if (used[b]) { (void)hipEventSynchronize(c->stage_ev[b][0]); (void)hipEventSynchronize(c->stage_ev[b][1]); }
            const int64_t at = k_enq * FSTAGE_CH;
            if (ablate & 2) { std::lock_guard<std::mutex> lk(c->helpers->m); ChunkRead &cr = c->stage_cr[b]; cr.nsl = 1; cr.left = 0; cr.got[0] = cr.want[0] = std::min<int64_t>(FSTAGE_CH, n - at); }
            else c->helpers->enqueue(fd, c->stage_h + b * FSTAGE_CH, std::min<int64_t>(FSTAGE_CH, n - at), pos + at, &c->stage_cr[b]);
        }
        const int b = (int)(k % 3);
        const int64_t at = k * FSTAGE_CH, want = std::min<int64_t>(FSTAGE_CH, n - at);
        c->helpers->wait(&c->stage_cr[b]);
        const int64_t m = c->stage_cr[b].total();
        if (m < 0) { rc = fail(FFQ_E_ARG, "ffq_load_fd: read failed at byte %lld: %s", (long long)(pos + at), strerror(errno)); break; }
        if (m < want) ended = true;                       // the file ends here
        const int64_t half = (ablate & 4) ? m : ((m / 2) + 4095) & ~(int64_t)4095;
        for (int h = 0; h < 2 && !rc; h++) {
            const int64_t a = h ? std::min(half, m) : 0, z = h ? m : std::min(half, m);
            if (z > a && !(ablate & 1)) e = hipMemcpyAsync(d_dst + at + a, c->stage_h + b * FSTAGE_CH + a, (size_t)(z - a), hipMemcpyHostToDevice, c->stage_cs[h]);
            if (e == hipSuccess) e = hipEventRecord(c->stage_ev[b][h], c->stage_cs[h]);
            if (e != hipSuccess) rc = fail(FFQ_E_HIP, "ffq_load_fd: chunk copy failed: %s", hipGetErrorString(e));
        }
        used[b] = true;
        *got += m;
    }
    // reads queued past the end of the file (or past an error) still write into the slots
    for (int64_t j = k; j < k_enq; j++) c->helpers->wait(&c->stage_cr[j % 3]);
    // the slots belong to the next caller when this returns: the copies out of them are waited for here (the bytes
    // are then in d_dst for whatever is enqueued next, on any stream)
    for (int b = 0; b < 3; b++)
        if (used[b])
            for (int h = 0; h < 2; h++)
                if ((e = hipEventSynchronize(c->stage_ev[b][h])) != hipSuccess && !rc) rc = fail(FFQ_E_HIP, "ffq_load_fd: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int ffq_load_fd(ffq_ctx *c, int fd, int64_t pos, int64_t n_bytes, void *d_dst, int64_t *n_loaded)
{
    if (!c || fd < 0 || pos < 0 || n_bytes < 0 || (n_bytes > 0 && !d_dst)) return fail(FFQ_E_ARG, "ffq_load_fd: bad argument");
    HIPCHK(hipSetDevice(c->device));
    int64_t got = 0;
    const int rc = stage_fd2d(c, static_cast<uint8_t *>(d_dst), fd, pos, n_bytes, &got);
    if (n_loaded) *n_loaded = got;
    return rc;
}

extern "C" int ffq_scan_host(ffq_ctx *c, const uint8_t *h_buf, int64_t n_bytes, int sentinel,
                             int64_t offset, int eof, int64_t add, uint32_t flags, int qual_add,
                             int64_t *h_table, int64_t table_cap, int8_t *h_qual, int64_t qual_cap,
                             int64_t *h_qoff, ffq_scan_result *res)
{
    mark_other(c);
    if (!c || !res) return fail(FFQ_E_ARG, "ffq_scan_host: ctx/res is NULL");
    if (n_bytes < 0 || table_cap < 0 || qual_cap < 0) return fail(FFQ_E_ARG, "ffq_scan_host: negative size");
    if (n_bytes > 0 && !h_buf) return fail(FFQ_E_ARG, "ffq_scan_host: h_buf is NULL");
    const bool decode = (flags & FFQ_F_DECODE_QUAL) != 0;
    if (decode && (!h_qual || !h_qoff)) return fail(FFQ_E_ARG, "ffq_scan_host: FFQ_F_DECODE_QUAL needs h_qual and h_qoff");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = grow_dev(c, &c->stage_d, &c->stage_d_cap, n_bytes + 16))) return rc;
    if ((rc = grow_dev(c, &c->tab_d, &c->tab_d_cap, std::max<int64_t>(table_cap, 1) * 6))) return rc;
    if (decode) {
        if ((rc = grow_dev(c, &c->qual_d, &c->qual_d_cap, std::max<int64_t>(qual_cap, 16)))) return rc;
        if ((rc = grow_dev(c, &c->qoff_d, &c->qoff_d_cap, table_cap + 1))) return rc;
    }
    // pageable -> device: three pinned staging slots, filled by the helper threads (slices of a
    // chunk in parallel), emptied over two copy streams (half a chunk each): the host copy of chunk
    // k+1 runs while chunk k is on the link
    rc = stage_setup(c);
    if (rc) return rc;
    bool used[3] = {false, false, false};
    const int64_t nch = (n_bytes + STAGE_CH - 1) / STAGE_CH;
    for (int64_t k_enq = 0, k = 0; k < nch; k++) {
        // keep the helper threads fed: the host copies of up to three chunks are queued at a time
        // (a slot is free again once the link copy out of it, three chunks back, is through)
        for (; k_enq < nch && k_enq - k < 3; k_enq++) {
            const int b = (int)(k_enq % 3);
            if (used[b]) { HIPCHK(hipEventSynchronize(c->stage_ev[b][0])); HIPCHK(hipEventSynchronize(c->stage_ev[b][1])); }
            const int64_t at = k_enq * STAGE_CH;
            c->helpers->enqueue_copy(c->stage_h + b * STAGE_CH, h_buf + at, std::min<int64_t>(STAGE_CH, n_bytes - at), &c->stage_cr[b]);
        }
        const int b = (int)(k % 3);
        const int64_t at = k * STAGE_CH, m = std::min<int64_t>(STAGE_CH, n_bytes - at);
        c->helpers->wait(&c->stage_cr[b]);
        const int64_t half = ((m / 2) + 4095) & ~(int64_t)4095;
        for (int h = 0; h < 2; h++) {
            const int64_t a = h ? std::min(half, m) : 0, z = h ? m : std::min(half, m);
            if (z > a)
                HIPCHK(hipMemcpyAsync(c->stage_d + at + a, c->stage_h + b * STAGE_CH + a, (size_t)(z - a), hipMemcpyHostToDevice,
                                      c->stage_cs[h]));
            HIPCHK(hipEventRecord(c->stage_ev[b][h], c->stage_cs[h]));
        }
        used[b] = true;
    }
    // the scan stream waits for the copies (the last event of each copy stream covers the earlier ones)
    for (int b = 0; b < 3; b++)
        if (used[b]) { HIPCHK(hipStreamWaitEvent(c->stream, c->stage_ev[b][0], 0)); HIPCHK(hipStreamWaitEvent(c->stream, c->stage_ev[b][1], 0)); }

    rc = ffq_scan_device(c, c->stage_d, n_bytes, sentinel, offset, eof, add, flags, qual_add, c->tab_d,
                         table_cap, decode ? c->qual_d : nullptr, qual_cap, decode ? c->qoff_d : nullptr, res);
    if (rc != FFQ_OK && rc != FFQ_E_TABLE_FULL) return rc;
    const int64_t rows = std::min<int64_t>(res->n_records, table_cap);
    int rc2 = stage_d2h(c, h_table, c->tab_d, rows * 48);
    if (!rc2 && decode) {
        rc2 = stage_d2h(c, h_qual, c->qual_d, std::min<int64_t>(res->n_qual_bytes, qual_cap));
        if (!rc2 && res->n_records <= table_cap) rc2 = stage_d2h(c, h_qoff, c->qoff_d, (rows + 1) * 8);
    }
    return rc2 ? rc2 : rc;
}

extern "C" int ffq_entrypos(ffq_ctx *c, const uint8_t *h_buf, int64_t len, int64_t offset, int64_t *pos,
                            int *status)
{
    mark_other(c);
    if (!c || !pos || !status) return fail(FFQ_E_ARG, "ffq_entrypos: NULL argument");
    int64_t row[6];
    ffq_scan_result r;
    int rc = ffq_scan_host(c, h_buf, len, 0, offset, 0, 0, 0, 0, row, 1, nullptr, 0, nullptr, &r);
    if (rc != FFQ_OK && rc != FFQ_E_TABLE_FULL) return rc;
    if (r.n_records >= 1) {
        for (int i = 0; i < 6; i++) pos[i] = row[i];
        *status = FFQ_COMPLETE;
    } else {
        for (int i = 0; i < 6; i++) pos[i] = r.last_pos[i];
        *status = r.last_status;
    }
    g_err.clear();
    return FFQ_OK;
}

extern "C" int ffq_table_lower_bound(ffq_ctx *c, const int64_t *d_table, int64_t n_rows, int col,
                                     int64_t value, int64_t *idx)
{
    mark_other(c);
    if (!c || !idx || n_rows < 0 || col < 0 || col > 5 || (n_rows > 0 && !d_table))
        return fail(FFQ_E_ARG, "ffq_table_lower_bound: bad argument");
    HIPCHK(hipSetDevice(c->device));
    int64_t *slot = c->h_word, *dslot = c->d_word;
    hipLaunchKernelGGL(k_table_lower_bound, dim3(1), dim3(64), 0, c->stream, d_table, n_rows, col, value, dslot);
    HIPCHK(hipMemcpyAsync(slot, dslot, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    *idx = *slot;
    return FFQ_OK;
}

// ---- FASTA (ffq_fasta.h) --------------------------------------------------------------------
extern "C" int ffq_scan_fasta_device(ffq_ctx *c, const uint8_t *d_buf, int64_t n_bytes, int sentinel,
                                     int64_t offset, int64_t add, int64_t *d_table, int64_t table_cap,
                                     ffq_scan_result *res)
{
    mark_other(c);
    if (!c || !res) return fail(FFQ_E_ARG, "ffq_scan_fasta: ctx/res is NULL");
    if (n_bytes < 0 || offset < 0 || table_cap < 0) return fail(FFQ_E_ARG, "ffq_scan_fasta: negative size");
    if (n_bytes > 0 && !d_buf) return fail(FFQ_E_ARG, "ffq_scan_fasta: d_buf is NULL");
    if ((reinterpret_cast<uintptr_t>(d_buf) & 15) != 0) return fail(FFQ_E_ARG, "ffq_scan_fasta: d_buf must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(d_table) & 15) != 0) return fail(FFQ_E_ARG, "ffq_scan_fasta: d_table must be 16-byte aligned");
    if (table_cap > 0 && !d_table) return fail(FFQ_E_ARG, "ffq_scan_fasta: d_table is NULL");
    if (c->pend.active) return fail(FFQ_E_ARG, "ffq_scan_fasta: a scan is pending on this context");
    memset(res, 0, sizeof *res);
    HIPCHK(hipSetDevice(c->device));
    const int64_t ntiles = tiles_for(n_bytes);
    if (ntiles == 0) {
        // nothing but (at most) the sentinel: no "\n>" can match
        res->last_status = FFQ_POS_HEAD_BEG;
        res->end_offset = offset;
        for (int i = 0; i < 6; i++) res->last_pos[i] = -1;
        res->path = 4;
        return FFQ_OK;
    }
    if (ntiles > 0x7FFFFFF0) return fail(FFQ_E_ARG, "buffer too large");
    int rc = reserve_tiles(c, ntiles);
    if (!rc) rc = reserve_pool(c, POOL_MIN);
    if (!rc) rc = grow_dev(c, &c->sel_cnt, &c->sel_cnt_cap, ntiles);
    if (!rc) rc = grow_dev(c, &c->sel_base, &c->sel_base_cap, ntiles);
    if (rc) return rc;
    hipStream_t sA = c->stream;
    ScanArgs a{};
    a.d_buf = d_buf; a.n_bytes = n_bytes; a.s = sentinel ? 1 : 0;
    int attempt = 0;
    for (;; attempt++) {
        const LineIndex L = make_index(c, a, ntiles);
        if (!c->ctl_clean) HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(Ctl), sA));
        c->ctl_clean = false;
        HIPCHK(hipEventRecord(c->ev[0], sA));
        launch_scan_lines(c, sA, d_buf, n_bytes, ntiles, L, (uint32_t)'>');
        HIPCHK(hipEventRecord(c->ev[1], sA));
        const unsigned tb = (unsigned)((ntiles + 3) / 4);
        hipLaunchKernelGGL(k_fa_count, dim3(tb), dim3(256), 0, sA, L, offset, c->sel_cnt);
        { const int rs = launch_scan_u32(c, sA, (const unsigned int *)c->sel_cnt, ntiles, c->sel_base, (long long *)c->d_word); if (rs) return rs; }
        hipLaunchKernelGGL(k_fa_rows, dim3(tb), dim3(256), 0, sA, L, offset, add, (const long long *)c->sel_base,
                           (const long long *)c->d_word, d_table, table_cap, c->fa_hdr, (const unsigned int *)c->sel_cnt);
        hipLaunchKernelGGL(k_fa_fix, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, sA, L, offset, add,
                           (const FaHdr *)c->fa_hdr, (const unsigned int *)c->sel_cnt, (const long long *)c->sel_base, d_table,
                           table_cap, c->dres, make_pub(c));
        c->ctl_clean = true;
        HIPCHK(hipEventRecord(c->ev[3], sA));
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventSynchronize(c->ev[3]));
        if (c->h_ctl->err & ERR_POOL) {
            if (attempt >= 2) return fail(FFQ_E_INTERNAL, "line-index pool overflow persists");
            rc = reserve_pool(c, std::max<unsigned long long>(c->h_ctl->pool_head + (c->h_ctl->pool_head >> 4), POOL_MIN));
            if (rc) return rc;
            continue;
        }
        break;
    }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); res->ms_index = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[1], c->ev[3])); res->ms_chain = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[3])); res->ms_total = ms;
    fill_result(res, *c->h_res, 4, attempt);
    if (res->n_records > table_cap)
        return fail(FFQ_E_TABLE_FULL, "table holds %lld rows, the buffer has %lld FASTA entries", (long long)table_cap,
                    (long long)res->n_records);
    return FFQ_OK;
}

extern "C" int ffq_scan_fasta_host(ffq_ctx *c, const uint8_t *h_buf, int64_t n_bytes, int sentinel, int64_t offset,
                                   int64_t add, int64_t *h_table, int64_t table_cap, ffq_scan_result *res)
{
    mark_other(c);
    if (!c || !res) return fail(FFQ_E_ARG, "ffq_scan_fasta_host: ctx/res is NULL");
    if (n_bytes < 0 || table_cap < 0) return fail(FFQ_E_ARG, "ffq_scan_fasta_host: negative size");
    if (n_bytes > 0 && !h_buf) return fail(FFQ_E_ARG, "ffq_scan_fasta_host: h_buf is NULL");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = grow_dev(c, &c->stage_d, &c->stage_d_cap, n_bytes + 16))) return rc;
    if ((rc = grow_dev(c, &c->tab_d, &c->tab_d_cap, std::max<int64_t>(table_cap, 1) * 6))) return rc;
    if (n_bytes > 0) HIPCHK(hipMemcpyAsync(c->stage_d, h_buf, (size_t)n_bytes, hipMemcpyHostToDevice, c->stream));
    rc = ffq_scan_fasta_device(c, c->stage_d, n_bytes, sentinel, offset, add, c->tab_d, table_cap, res);
    if (rc != FFQ_OK && rc != FFQ_E_TABLE_FULL) return rc;
    const int64_t rows = std::min<int64_t>(res->n_records, table_cap);
    if (rows > 0) HIPCHK(hipMemcpy(h_table, c->tab_d, (size_t)rows * 48, hipMemcpyDeviceToHost));
    return rc;
}

extern "C" int ffq_table_cut(ffq_ctx *c, const int64_t *d_table, int64_t n_rows, int64_t lo, int64_t hi,
                             int64_t out[6])
{
    mark_other(c);
    if (!c || !out || n_rows < 0 || (n_rows > 0 && !d_table)) return fail(FFQ_E_ARG, "ffq_table_cut: bad argument");
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_table_cut, dim3(1), dim3(64), 0, c->stream, d_table, n_rows, lo, hi, c->d_cut);
    HIPCHK(hipMemcpyAsync(c->h_cut, c->d_cut, 6 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 6; i++) out[i] = c->h_cut[i];
    return FFQ_OK;
}

static int table_select(ffq_ctx *c, const int64_t *d_table, int64_t n_rows, int64_t min_len, int64_t max_len, int64_t *d_out,
                        int64_t *d_idx, int64_t *n_out);

extern "C" int ffq_table_select_seqlen(ffq_ctx *c, const int64_t *d_table, int64_t n_rows, int64_t min_len,
                                       int64_t max_len, int64_t *d_out, int64_t *n_out)
{
    return table_select(c, d_table, n_rows, min_len, max_len, d_out, nullptr, n_out);
}

extern "C" int ffq_table_select_seqlen_idx(ffq_ctx *c, const int64_t *d_table, int64_t n_rows, int64_t min_len,
                                           int64_t max_len, int64_t *d_out, int64_t *d_idx, int64_t *n_out)
{
    return table_select(c, d_table, n_rows, min_len, max_len, d_out, d_idx, n_out);
}

// (d_idx, optional: the ordinal of every kept row -- the stream front end's push-down, ffq_stream_set_filter)
static int table_select(ffq_ctx *c, const int64_t *d_table, int64_t n_rows, int64_t min_len, int64_t max_len, int64_t *d_out,
                        int64_t *d_idx, int64_t *n_out)
{
    mark_other(c);
    if (!c || !n_out || n_rows < 0 || (n_rows > 0 && (!d_table || !d_out)))
        return fail(FFQ_E_ARG, "ffq_table_select_seqlen: bad argument");
    if (d_table == d_out && n_rows > 0) return fail(FFQ_E_ARG, "ffq_table_select_seqlen: d_out must not be d_table");
    if (((reinterpret_cast<uintptr_t>(d_table) | reinterpret_cast<uintptr_t>(d_out)) & 15) != 0)
        return fail(FFQ_E_ARG, "ffq_table_select_seqlen: tables must be 16-byte aligned");
    HIPCHK(hipSetDevice(c->device));
    *n_out = 0;
    if (n_rows == 0) return FFQ_OK;
    const int64_t nblk = (n_rows + 255) / 256;
    int rc = grow_dev(c, &c->sel_cnt, &c->sel_cnt_cap, nblk);
    if (!rc) rc = grow_dev(c, &c->sel_base, &c->sel_base_cap, nblk);
    if (rc) return rc;
    hipStream_t st = c->stream;
    hipLaunchKernelGGL(k_sel_count, dim3((unsigned)nblk), dim3(256), 0, st, d_table, n_rows, min_len, max_len,
                       c->sel_cnt);
    if ((rc = launch_scan_u32(c, st, (const unsigned int *)c->sel_cnt, nblk, c->sel_base, (long long *)c->d_word))) return rc;
    hipLaunchKernelGGL(k_sel_scatter, dim3((unsigned)nblk), dim3(256), 0, st, d_table, n_rows, min_len, max_len,
                       (const long long *)c->sel_base, d_out, d_idx);
    HIPCHK(hipMemcpyAsync(c->h_word, c->d_word, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    *n_out = c->h_word[0];
    return FFQ_OK;
}

// ---- column selection ----------------------------------------------------------------------
extern "C" int ffq_table_gather_column(ffq_ctx *c, const uint8_t *d_buf, int64_t n_bytes, int sentinel, int64_t add,
                                       const int64_t *d_table, int64_t n_rows, int col_begin, int begin_shift,
                                       int col_end, int value_add, int8_t *d_out, int64_t out_cap, int64_t *d_off,
                                       int64_t *n_out_bytes)
{
    mark_other(c);
    if (!c || !n_out_bytes || n_rows < 0 || n_bytes < 0 || out_cap < 0 || !d_off || (n_rows > 0 && !d_table))
        return fail(FFQ_E_ARG, "ffq_table_gather_column: bad argument");
    if (col_begin < 0 || col_begin > 5 || col_end < 0 || col_end > 5)
        return fail(FFQ_E_ARG, "ffq_table_gather_column: columns are 0..5");
    if (c->pend.active) return fail(FFQ_E_ARG, "ffq_table_gather_column: a scan is pending on this context");
    HIPCHK(hipSetDevice(c->device));
    *n_out_bytes = 0;
    hipStream_t st = c->stream;
    if (n_rows == 0) {
        HIPCHK(hipMemsetAsync(d_off, 0, sizeof(int64_t), st));
        HIPCHK(hipStreamSynchronize(st));
        return FFQ_OK;
    }
    const int64_t nblk = (n_rows + 255) / 256;
    int rc = grow_dev(c, &c->col_sum, &c->col_sum_cap, nblk);
    if (!rc && !c->col_res) {
        hipError_t e = hipMalloc((void **)&c->col_res, sizeof(DevRes));
        if (e != hipSuccess) rc = fail(FFQ_E_NOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    // the copy kernel's scratch: a start per row, the directory of the output stream
    const int64_t nqb = (out_cap + DQ_BLK - 1) / DQ_BLK + 1;
    if (!rc) rc = reserve_qdir(c, nqb);
    if (!rc) rc = reserve_p4s(c, n_rows);
    if (rc) return rc;
    hipLaunchKernelGGL(k_col_sum, dim3((unsigned)nblk), dim3(256), 0, st, d_table, n_rows, col_begin, begin_shift, col_end,
                       c->col_sum);
    if ((rc = launch_scan_i64v(c, st, c->col_sum, nblk, n_rows, c->col_res))) return rc;
    hipLaunchKernelGGL(k_col_offsets, dim3((unsigned)nblk), dim3(256), 0, st, d_table, n_rows, col_begin, begin_shift,
                       col_end, (const long long *)c->col_sum, (const DevRes *)c->col_res, d_off, c->p4s, c->qdir,
                       c->qdir_cap);
    if (out_cap > 0)
        hipLaunchKernelGGL(k_decode_stream, dim3((unsigned)nqb), dim3(256), 0, st, d_buf, n_bytes, sentinel ? 1 : 0,
                           (const int64_t *)c->p4s, (const int64_t *)d_off, (const int64_t *)c->qdir,
                           (const DevRes *)c->col_res, n_rows, add, value_add, d_out, out_cap, 0);
    HIPCHK(hipMemcpyAsync(c->h_word, &c->col_res->n_qual_bytes, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    *n_out_bytes = c->h_word[0];
    if (*n_out_bytes > out_cap)
        return fail(FFQ_E_TABLE_FULL, "output holds %lld bytes, the column has %lld", (long long)out_cap,
                    (long long)*n_out_bytes);
    return FFQ_OK;
}

// ---- arrayadd ----------------------------------------------------------------
extern "C" int ffq_arrayadd_b_device(ffq_ctx *c, int8_t *d_a, int64_t n, int value)
{
    mark_other(c);
    if (!c || n < 0 || (n > 0 && !d_a)) return fail(FFQ_E_ARG, "ffq_arrayadd_b_device: bad argument");
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) return FFQ_OK;
    const int64_t nv = (n + 15) / 16;
    const unsigned grid = (unsigned)std::min<int64_t>((nv + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(k_arrayadd_b, dim3(std::max(grid, 1u)), dim3(256), 0, c->stream, (uint8_t *)d_a, n,
                       (uint32_t)(uint8_t)value);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return FFQ_OK;
}

extern "C" int ffq_arrayadd_q_device(ffq_ctx *c, int64_t *d_a, int64_t n, int64_t value)
{
    mark_other(c);
    if (!c || n < 0 || (n > 0 && !d_a)) return fail(FFQ_E_ARG, "ffq_arrayadd_q_device: bad argument");
    if ((reinterpret_cast<uintptr_t>(d_a) & 7) != 0) return fail(FFQ_E_ARG, "ffq_arrayadd_q_device: unaligned");
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) return FFQ_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((n / 2 + 255) / 256 + 1, 256 * 8);
    hipLaunchKernelGGL(k_arrayadd_q, dim3(grid), dim3(256), 0, c->stream, (unsigned long long *)d_a, n,
                       (unsigned long long)value);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return FFQ_OK;
}

extern "C" int ffq_arrayadd_b(ffq_ctx *c, int8_t *h_a, int64_t n, int value)
{
    mark_other(c);
    if (!c || n < 0 || (n > 0 && !h_a)) return fail(FFQ_E_ARG, "ffq_arrayadd_b: bad argument");
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) return FFQ_OK;
    int rc = grow_dev(c, &c->stage_d, &c->stage_d_cap, n + 16);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->stage_d, h_a, (size_t)n, hipMemcpyHostToDevice, c->stream));
    rc = ffq_arrayadd_b_device(c, (int8_t *)c->stage_d, n, value);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_a, c->stage_d, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FFQ_OK;
}

extern "C" int ffq_arrayadd_q(ffq_ctx *c, int64_t *h_a, int64_t n, int64_t value)
{
    mark_other(c);
    if (!c || n < 0 || (n > 0 && !h_a)) return fail(FFQ_E_ARG, "ffq_arrayadd_q: bad argument");
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) return FFQ_OK;
    int rc = grow_dev(c, &c->stage_d, &c->stage_d_cap, n * 8 + 16);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->stage_d, h_a, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    rc = ffq_arrayadd_q_device(c, (int64_t *)c->stage_d, n, value);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_a, c->stage_d, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return FFQ_OK;
}

// ---- synthetic inputs ------------------------------------------------------------
extern "C" int ffq_synth_single(ffq_ctx *c, uint8_t *d_out, int64_t first, int64_t count, uint64_t seed)
{
    mark_other(c);
    if (!c || count < 0 || (count > 0 && !d_out)) return fail(FFQ_E_ARG, "ffq_synth_single: bad argument");
    HIPCHK(hipSetDevice(c->device));
    if (count == 0) return FFQ_OK;
    hipLaunchKernelGGL(k_synth_single, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, d_out,
                       first, count, seed);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return FFQ_OK;
}

extern "C" int64_t ffq_synth_wrapped_size(int64_t i, uint64_t seed) { return synth_wrapped_size(i, seed); }

extern "C" int ffq_synth_wrapped(ffq_ctx *c, uint8_t *d_out, const int64_t *d_start, int64_t first,
                                 int64_t count, uint64_t seed)
{
    mark_other(c);
    if (!c || count < 0 || (count > 0 && (!d_out || !d_start))) return fail(FFQ_E_ARG, "ffq_synth_wrapped: bad argument");
    HIPCHK(hipSetDevice(c->device));
    if (count == 0) return FFQ_OK;
    hipLaunchKernelGGL(k_synth_wrapped, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, d_out,
                       d_start, first, count, seed);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return FFQ_OK;
}

#ifdef FFQ_PROBES
// ---- probes: only in the instrumented build, libffq_probe.so (include/ffq_probe.h; tools/) ----------
extern "C" int ffq_read_probe(ffq_ctx *c, const uint8_t *d_buf, int64_t n_bytes, int mode, int reps, float *ms_avg)
{
    mark_other(c);
    if (!c || !d_buf || !ms_avg || n_bytes < TILE || reps < 1) return fail(FFQ_E_ARG, "ffq_read_probe: bad argument");
    HIPCHK(hipSetDevice(c->device));
    const int64_t ntiles = n_bytes >> TILE_SHIFT;
    uint32_t *sink = reinterpret_cast<uint32_t *>(c->ctl);
    ScanArgs a{};
    a.d_buf = d_buf; a.n_bytes = ntiles << TILE_SHIFT; a.s = 1;
    if (mode == 2) {
        // the index kernel alone, back to back: its steady-state time without the rest of a step
        int rc = reserve_tiles(c, ntiles);
        if (rc) return rc;
        rc = reserve_pool(c, POOL_MIN);
        if (rc) return rc;
    }
    const LineIndex L = make_index(c, a, ntiles);
    if (mode >= 200 && mode < 220) {
        // the persistent streaming loop with a lagged two-level prefix (k_pipe_probe): lag = mode - 200
        int rc = reserve_tiles(c, ntiles);
        if (rc) return rc;
        rc = reserve_pool(c, POOL_MIN);
        if (rc) return rc;
        const bool big = mode >= 210;                  // 72 KiB of LDS per workgroup, two per CU
        const int G = big ? 512 : 1024, lag = big ? mode - 210 : mode - 200;
        const int64_t niter = (ntiles + G - 1) / G;
        PipeArgs pa{};
        pa.d = d_buf; pa.ntiles = ntiles; pa.lag = lag;
        HIPCHK(hipMalloc((void **)&pa.descA, (size_t)niter * G * 8));
        HIPCHK(hipMalloc((void **)&pa.descG, (size_t)niter * (G / PP_GROUP) * 8));
        HIPCHK(hipMalloc((void **)&pa.prefix, (size_t)niter * G * 8));
        HIPCHK(hipMalloc((void **)&pa.err, 8));
        // the tile counts to check the prefixes against
        launch_scan_lines(c, c->stream, a.d_buf, a.n_bytes, ntiles, L, (uint32_t)'@', 0);
        float sum = 0;
        for (int r = 0; r < reps + 2; r++) {
            HIPCHK(hipMemsetAsync(pa.descA, 0, (size_t)niter * G * 8, c->stream));
            HIPCHK(hipMemsetAsync(pa.descG, 0, (size_t)niter * (G / PP_GROUP) * 8, c->stream));
            HIPCHK(hipMemsetAsync(pa.err, 0, 8, c->stream));
            HIPCHK(hipEventRecord(c->ev[0], c->stream));
            if (big) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_probe<72>), dim3(G), dim3(256), 0, c->stream, pa, sink);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_probe<36>), dim3(G), dim3(256), 0, c->stream, pa, sink);
            HIPCHK(hipEventRecord(c->ev[1], c->stream));
            HIPCHK(hipEventSynchronize(c->ev[1]));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            if (r >= 2) sum += ms;
        }
        *ms_avg = sum / reps;
        int bad = 0;
        uint32_t herr2[2] = {0, 0};
        HIPCHK(hipMemcpy(herr2, pa.err, 8, hipMemcpyDeviceToHost));
        const uint32_t herr = herr2[0];
        if (lag) {
            std::vector<uint32_t> hc((size_t)ntiles);
            std::vector<long long> hp((size_t)ntiles);
            HIPCHK(hipMemcpy(hc.data(), c->cnt, (size_t)ntiles * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(hp.data(), pa.prefix, (size_t)ntiles * 8, hipMemcpyDeviceToHost));
            long long run = 0;
            for (int64_t t = 0; t < ntiles; t++) { if (hp[(size_t)t] != run) bad++; run += hc[(size_t)t]; }
        }
        (void)hipFree(pa.descA); (void)hipFree(pa.descG); (void)hipFree(pa.prefix); (void)hipFree(pa.err);
        HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(Ctl), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (herr || bad) return fail(FFQ_E_INTERNAL, "pipe probe: %d wrong prefixes, poll gave up: %u", bad, herr);
        return FFQ_OK;
    }
    if (mode == 7 || mode >= 100) {
        // the index kernel with a decoupled look-back over the tile counts riding along (a probe:
        // what a single-pass design would pay for its prefix sums on this part)
        int rc = reserve_tiles(c, ntiles);
        if (rc) return rc;
        rc = reserve_pool(c, POOL_MIN);
        if (rc) return rc;
        float sum = 0;
        for (int r = 0; r < reps + 2; r++) {
            HIPCHK(hipMemsetAsync(c->ovf, 0, (size_t)ntiles * 8, c->stream));
            HIPCHK(hipEventRecord(c->ev[0], c->stream));
            launch_scan_lines(c, c->stream, a.d_buf, a.n_bytes, ntiles, L, (uint32_t)'@', mode == 7 ? 8 : mode);
            HIPCHK(hipEventRecord(c->ev[1], c->stream));
            HIPCHK(hipEventSynchronize(c->ev[1]));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            if (r >= 2) sum += ms;
        }
        *ms_avg = sum / reps;
        std::vector<uint32_t> hc((size_t)ntiles);
        HIPCHK(hipMemcpy(hc.data(), c->cnt, (size_t)ntiles * 4, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> hd((size_t)ntiles);
        HIPCHK(hipMemcpy(hd.data(), c->ovf, (size_t)ntiles * 8, hipMemcpyDeviceToHost));
        const int variant = mode >= 100 ? ((mode - 100) >> 5) & 7 : 0;
        if (mode >= 100) {
            // rounds / retries of the look-backs, as the inclusive descriptors carry them
            const int64_t nd = variant == 3 ? ntiles >> 2 : ntiles;
            double rounds = 0, retries = 0;
            for (int64_t t = 1; t < nd; t++) { rounds += (double)((hd[(size_t)t] >> 40) & 0xFF); retries += (double)((hd[(size_t)t] >> 48) & 0x3FFF); }
            fprintf(stderr, "[ffq probe] mode %d: %.2f rounds, %.2f retries per look-back\n", mode, rounds / std::max<int64_t>(nd - 1, 1),
                    retries / std::max<int64_t>(nd - 1, 1));
        }
        if (variant == 0 || variant == 4) {
            const unsigned long long VM = mode >= 100 ? (1ull << 40) - 1ull : (1ull << 62) - 1ull;
            unsigned long long tot = 0, totm = 0;
            for (int64_t t = 0; t < ntiles; t++) { tot += hc[(size_t)t]; if (t <= ntiles / 2) totm += hc[(size_t)t]; }
            const unsigned long long last = hd[(size_t)(ntiles - 1)], mid = hd[(size_t)(ntiles / 2)];
            if ((last & VM) != tot || (mid & VM) != totm || (last >> 62) != 2)
                return fail(FFQ_E_INTERNAL, "look-back probe: prefix %llu / %llu, expected %llu / %llu", (last & VM), (mid & VM), tot, totm);
        }
        HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(Ctl), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return FFQ_OK;
    }
    if (mode == 3 || mode == 4 || mode == 8) {
        // (mode 4: the variant for a buffer whose last tile is ragged; mode 8: non-temporal entry stores)
        if (mode == 4) a.n_bytes -= 5;
        // mode 2's kernel, every launch between its own pair of events (as a scan times it)
        int rc = reserve_tiles(c, ntiles);
        if (rc) return rc;
        rc = reserve_pool(c, POOL_MIN);
        if (rc) return rc;
        float sum = 0;
        for (int r = 0; r < reps + 2; r++) {
            HIPCHK(hipEventRecord(c->ev[0], c->stream));
            launch_scan_lines(c, c->stream, a.d_buf, a.n_bytes, ntiles, L, (uint32_t)'@', mode == 8 ? 9 : 0);
            HIPCHK(hipEventRecord(c->ev[1], c->stream));
            HIPCHK(hipEventSynchronize(c->ev[1]));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            if (r >= 2) sum += ms;
        }
        *ms_avg = sum / reps;
        HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(Ctl), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return FFQ_OK;
    }
    for (int r = 0; r < reps + 2; r++) {
        if (r == 2) HIPCHK(hipEventRecord(c->ev[0], c->stream));
        if (mode == 2)
            launch_scan_lines(c, c->stream, a.d_buf, a.n_bytes, ntiles, L, (uint32_t)'@');
        else if (mode == 6)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_read_probe<2>), dim3((unsigned)ntiles), dim3(256), 0, c->stream, d_buf, ntiles, sink);
        else if (mode == 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_read_probe<0>), dim3((unsigned)ntiles), dim3(256), 0, c->stream, d_buf, ntiles, sink);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_read_probe<1>), dim3(256 * 8), dim3(256), 0, c->stream, d_buf, ntiles, sink);
    }
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    *ms_avg = ms / reps;
    if (mode == 2) { HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(Ctl), c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
    return FFQ_OK;
}

#endif  // FFQ_PROBES

// ---- diagnostics ---------------------------------------------------------------------
extern "C" int ffq_selftest(ffq_ctx *c)
{
    mark_other(c);
    if (!c) return fail(FFQ_E_ARG, "ffq_selftest: ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    const int NB = 256 * 16;
    uint8_t host[NB];
    uint64_t x = 12345;
    for (int i = 0; i < NB; i++) {
        x = splitmix64(x);
        const unsigned r = (unsigned)(x & 7);
        host[i] = (r == 0) ? '\n' : (r == 1) ? 0x0B : (r == 2) ? 0x8A : (r == 3) ? 0x00 : (uint8_t)(x >> 8);
    }
    uint8_t *d = nullptr;
    uint32_t *bad = nullptr;
    HIPCHK(hipMalloc((void **)&d, NB));
    HIPCHK(hipMalloc((void **)&bad, 4));
    HIPCHK(hipMemcpyAsync(d, host, NB, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(bad, 0, 4, c->stream));
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(256), 0, c->stream, d, bad);
    uint32_t hb = 0;
    HIPCHK(hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(d);
    (void)hipFree(bad);
    if (hb) return fail(FFQ_E_INTERNAL, "device self-test: %u mismatches (wave scan / newline mask)", hb);
    return FFQ_OK;
}

#include "ffq_stream.h"
#include "ffq_bgzf.h"
#include "ffq_shard.h"
#include "ffq_shard_host.h"
