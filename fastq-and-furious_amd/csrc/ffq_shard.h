// ffq_shard.h -- byte-range sharding of ONE FASTQ stream over the GPUs of a node, behind the C ABI (ffq_shard_*,
// include/ffq.h).  Included at the end of ffq_hip.hip.
//
// The reference has nothing like it (a single-threaded generator); what is sharded is the record chain of
// readfastq_iter (/root/reference/src/fastqandfurious.py:251-279): rank r owns the stream bytes [S_r, S_r+1) and every
// record whose '@' lies in them.  What the reference does with a record that does not fit its buffer -- keep buf[offset:]
// and read more (:274-279) -- happens here per range edge.  One step of a rank:
//   1. halo hand-off: ncclSend / ncclRecv inside ONE group, on a stream of its own beside the previous step's scan -- the
//      tail_bytes in front of the range (run-in) and the head_bytes behind it (look-ahead), from whichever ranks own them;
//   2. one ordinary scan of [tail | own | head] (ffq_scan_submit), the scan stream waiting for the hand-off's end event;
//   3. k_shard_words, queued behind the scan: the rows with S_r <= pos0 < S_r+1 (two lower bounds over the table, as
//      ffq_table_cut) and the EIGHT WORDS the ranks compare -- exit (first record start at / behind my right edge as my
//      chain sees it), first (first record start in my range), own count, look-ahead wanted, look-ahead had, error,
//      error byte, where the search that found the exit started;
//   4. ONE ncclAllGather of those words (a communicator of its own, on a stream of its own that waits for them) and one copy to pinned memory:
//      the host reads them when it waits for the step -- nothing else comes back before that.
// exit[r] must equal first[r + 1]; rank 0's start is exact, so that proves every range by induction, and the counts give
// global record ordinals.  A rank whose look-ahead ends inside the record that straddles its edge asks for more (served by
// whoever owns the bytes) and scans again; a rank whose guessed entry its left neighbour's chain contradicts scans again
// from that neighbour's exit; each round settles the first unsettled rank.  Stream errors (the iterator's three
// ValueErrors) are reported by every rank alike, once the failing rank's entry is proven.
//
// This is the C++ restatement of the protocol fastq-and-furious_amd/sharded.py runs over torch.distributed (kept there for
// the CPU tests over gloo); the transports here are RCCL (librccl, resolved at run time: the library loads without it) and
// an in-process one (k logical ranks as threads of one process on one GPU: tests, single-GPU dry runs).
#pragma once
#include "ffq_shard_proto.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <atomic>
#include <functional>
#include <memory>
#include <thread>

namespace ffq {

// ---- step 3 on the device: rows of my range + the hand-off words ----------------------------------------------------
// out[0..7] the words (ffq_shard_proto.h: sh_words_from), out[8] row_lo, out[9] row_hi, out[10] rows in the table
__global__ __launch_bounds__(64) void k_shard_words(const DevRes *__restrict__ res, const int64_t *__restrict__ table,
                                                    int64_t table_cap, int64_t qual_cap, ShView v, int64_t offset, int64_t head_bytes,
                                                    int64_t *__restrict__ out, int64_t *__restrict__ host_out)
{
    const int lane = threadIdx.x;
    const int64_t n = res->n_records;
    int64_t w[SH_WORDS] = {SH_UNKNOWN, SH_UNKNOWN, 0, 0, v.head, 0, 0, 0};
    int64_t i0 = 0, i1 = 0, nrows = 0;
    if (res->fallback) w[5] = SH_NOT_READY;
    else if (n > table_cap) { w[5] = SH_ERR_TABLE_FULL; w[6] = n; }
    else if (qual_cap >= 0 && res->n_qual_bytes > qual_cap) { w[5] = SH_ERR_QUAL_FULL; w[6] = res->n_qual_bytes; }      // (decoding: qual_cap >= 0)
    else {
        nrows = n;
        int64_t cut[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {       // (the two lower bounds of k_table_cut: 64 probes per round trip)
            const int64_t value = q ? sh_hi_bound(v) : sh_lo_bound(v);
            int64_t a = 0, b = n;
            while (b - a > 0) {
                const int64_t stride = (b - a + 63) / 64;
                const int64_t i = a + (int64_t)lane * stride;
                const bool below = (i < b) && (table[i * 6] < value);
                const int k = __popcll(__ballot(below));
                if (k == 0) { b = a; break; }
                const int64_t last = a + (int64_t)(k - 1) * stride;
                a = last + 1;
                b = min(b, last + stride);
            }
            cut[q] = a;
        }
        i0 = cut[0]; i1 = cut[1];
        ShScanFacts f;
        f.n = n; f.i0 = i0; f.i1 = i1;
        f.p_i0 = (i0 < n) ? table[i0 * 6] : -1;
        f.p_i1 = (i1 < n) ? table[i1 * 6] : -1;
        f.q1 = (i1 > 0) ? table[(i1 - 1) * 6 + 5] : -1;
        f.end_state = res->end_state; f.last_status = res->last_status;
        f.last_pos0 = res->last_pos[0]; f.end_offset = res->end_offset;
        sh_words_from(v, f, offset, head_bytes, w);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < SH_WORDS; i++) out[i] = w[i];
        out[8] = i0; out[9] = i1; out[10] = nrows;
        // (the same into host-mapped memory: this rank's own copy, no copy packet on the scan stream; visible to the host
        // once the gather behind this kernel is through)
#pragma unroll
        for (int i = 0; i < SH_WORDS; i++) host_out[i] = w[i];
        host_out[8] = i0; host_out[9] = i1; host_out[10] = nrows;
        __threadfence_system();
    }
}

// diagnostics (ffq_shard_inject_stall): one lane waits for a host flag -- or for `budget` ticks of the 100 MHz wall clock,
// whichever comes first: it cannot outlive its budget --, so that a step hangs at a chosen stage without a broken peer
__global__ __launch_bounds__(64) void k_shard_stall(const int *flag, unsigned long long budget)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < budget)
        __builtin_amdgcn_s_sleep(127);
}

// ---- transports ---------------------------------------------------------------------------------------------------
typedef std::function<uint8_t *(int64_t, int64_t)> ShPtrFn;

struct ShTransport {
    int rank = 0, world = 1;
    bool serial = false;                   // ONE communicator on ONE stream (the fallback mode; include/ffq.h)
    bool poisoned = false, aborted = false;
    double timeout_s = sh_default_timeout();
    std::vector<int64_t> bus;              // PCI bus id of every rank's GPU (-1: unknown)
    virtual ~ShTransport() {}
    virtual int nranks(int /*which: 0 hand-off, 1 gather*/) const { return world; }
    virtual const char *async_error() { return nullptr; }           // RCCL: ncclCommGetAsyncError of either communicator
    virtual void abort() { aborted = true; }                        // RCCL: ncclCommAbort
    // every piece of the plan this rank sends or receives, enqueued on `st`
    virtual int exchange(const std::vector<ShPiece> &plan, const ShPtrFn &provide, const ShPtrFn &accept, hipStream_t st) = 0;
    // the words of every rank: d_mine (device, SH_WORDS int64) -> h_all (pinned, world * SH_WORDS); enqueue on `st`, then finish
    virtual int gather_enqueue(const int64_t *d_mine, int64_t *d_all, int64_t *h_all, hipStream_t st) = 0;
    virtual int gather_finish(int64_t *h_all, hipEvent_t done) = 0;      // `done`: recorded behind gather_enqueue's work
    virtual const char *name() const = 0;
};

// librccl, resolved at run time (a process that has PyTorch's copy loaded gets that one)
struct RcclApi {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
};

static RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("FFQ_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (int pass = 0; pass < 2 && !api.h; pass++)
            for (const char *nm : names) {
                if (!nm || !*nm) continue;
                api.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));      // first: a copy the process already holds
                if (api.h) break;
            }
        if (!api.h) return;
#define FFQ_RCCL_SYM(f) api.f = reinterpret_cast<decltype(api.f)>(dlsym(api.h, "nccl" #f))
        FFQ_RCCL_SYM(GetUniqueId); FFQ_RCCL_SYM(CommInitRank); FFQ_RCCL_SYM(CommDestroy); FFQ_RCCL_SYM(GetErrorString);
        FFQ_RCCL_SYM(AllGather); FFQ_RCCL_SYM(Broadcast); FFQ_RCCL_SYM(Send); FFQ_RCCL_SYM(Recv); FFQ_RCCL_SYM(GroupStart); FFQ_RCCL_SYM(GroupEnd);
        FFQ_RCCL_SYM(CommCount); FFQ_RCCL_SYM(CommAbort); FFQ_RCCL_SYM(CommGetAsyncError);
#undef FFQ_RCCL_SYM
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.Broadcast || !api.Send || !api.Recv ||
            !api.GroupStart || !api.GroupEnd || !api.CommCount || !api.CommAbort)
            api.h = nullptr;
    });
    return api.h ? &api : nullptr;
}

#define RCCLCHK(expr)                                                                                           \
    do {                                                                                                        \
        ncclResult_t r__ = (expr);                                                                              \
        if (r__ != ncclSuccess)                                                                                 \
            return fail(FFQ_E_HIP, "%s failed: %s", #expr, A->GetErrorString ? A->GetErrorString(r__) : "rccl error"); \
    } while (0)

// PCI domain << 16 | bus << 8 | device << 3 (| function 0) of a HIP device: what tells two ranks on ONE GPU from two GPUs
static int64_t sh_bus_id(int device)
{
    int dom = 0, bus = 0, dev = 0;
    if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) != hipSuccess ||
        hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess ||
        hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) != hipSuccess) return -1;
    return ((int64_t)dom << 16) | ((int64_t)(bus & 0xFF) << 8) | ((int64_t)(dev & 0x1F) << 3);
}

// The watchdog's wait: polls `ev` until it is through, `seconds` have passed (1) or the transport reports an
// asynchronous error (2); seconds <= 0: an ordinary wait.  A step is through in milliseconds, so the first two are
// spun and the rest slept in 100 us pieces.
static int sh_wait_event(hipEvent_t ev, double seconds, ShTransport *tr, double *waited = nullptr)
{
    if (seconds <= 0) { HIPCHK(hipEventSynchronize(ev)); return 0; }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1;; spins++) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail(FFQ_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
        if ((spins & 31) == 0) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited) *waited = dt;
            if (tr && dt > 1e-3 && tr->async_error()) return 2;
            if (dt > seconds) return 1;
            if (dt > 2e-3) usleep(100);
        }
        __builtin_ia32_pause();
    }
}

// RCCL over xGMI.  Pipelined mode: the hand-offs and the gather on communicators of their own (one per stream: the
// hand-off of step i + 1 runs beside the scan of step i, the gather of step i behind that scan).  Serial mode: cx alone,
// everything on the scan stream.
struct ShRccl : ShTransport {
    RcclApi *A = nullptr;
    ncclComm_t cx = nullptr, cg = nullptr;
    bool owner = true;                     // (a shard created beside another one borrows its communicators)
    int n_cx = 0, n_cg = 0;
    char async_text[160] = {0};
    ~ShRccl() override
    {
        if (!owner || !A) return;
        // (a communicator a collective is stuck on is aborted, not destroyed: ncclCommDestroy would wait for it)
        if (cx) { if (poisoned) A->CommAbort(cx); else A->CommDestroy(cx); }
        if (cg) { if (poisoned) A->CommAbort(cg); else A->CommDestroy(cg); }
    }
    int nranks(int which) const override { return which ? n_cg : n_cx; }
    const char *async_error() override
    {
        if (!A || !A->CommGetAsyncError) return nullptr;
        ncclComm_t cs[2] = {cx, cg};
        for (ncclComm_t c : cs) {
            ncclResult_t r = ncclSuccess;
            if (c && A->CommGetAsyncError(c, &r) == ncclSuccess && r != ncclSuccess && r != ncclInProgress) {
                snprintf(async_text, sizeof async_text, "%s", A->GetErrorString ? A->GetErrorString(r) : "rccl error");
                return async_text;
            }
        }
        return nullptr;
    }
    // ncclCommAbort raises the communicator's abort flag first -- the kernels of a collective that waits for a peer leave --
    // and then tears the communicator down, which with a peer that is GONE (its process dead: the proxy thread sits in a socket
    // call) does not come back for minutes (tests/test_multigpu.py: a peer that dies).  So the abort runs on a thread of its
    // own and this call waits for it with a deadline; what is still tearing down after that is left to that thread.
    int device = 0;
    bool abort_stuck = false;
    void abort() override
    {
        if (aborted) return;
        aborted = true;
        if (!(owner && A)) return;
        ncclComm_t a = cx, b = cg;
        cx = cg = nullptr;
        if (!a && !b) return;
        auto done = std::make_shared<std::atomic<int>>(0);
        RcclApi *api = A;
        const int dev = device;
        std::thread([api, a, b, done, dev] {
            (void)hipSetDevice(dev);
            if (a) api->CommAbort(a);
            if (b) api->CommAbort(b);
            done->store(1);
        }).detach();
        // (how long: seconds when the peers are there -- 8 ranks tearing down at once over sockets took more than 10 s now and
        // then --, never while a peer's process is gone; FFQ_SHARD_ABORT_S, default 30)
        static const double abort_s = getenv("FFQ_SHARD_ABORT_S") ? std::max(1.0, atof(getenv("FFQ_SHARD_ABORT_S"))) : 30.0;
        const auto t0 = std::chrono::steady_clock::now();
        while (!done->load()) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > abort_s) { abort_stuck = true; return; }
            usleep(1000);
        }
    }
    // a collective of the set-up, waited for with the watchdog's deadline: the FIRST thing that talks to the peers must
    // not be able to hang either
    int settle(hipStream_t st, const char *what)
    {
        hipEvent_t ev = nullptr;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, st);
        int w = e == hipSuccess ? sh_wait_event(ev, timeout_s, this) : fail(FFQ_E_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
        if (w <= 0) (void)hipEventDestroy(ev);            // (left alone while something may still refer to it)
        if (w == 0) return FFQ_OK;
        if (w < 0) return w;
        poisoned = true;
        if (w == 2) return fail(FFQ_E_HIP, "ffq_shard_create: rank %d of %d: RCCL reports an asynchronous error during %s: %s", rank, world, what, async_text);
        return fail(FFQ_E_TIMEOUT, "ffq_shard_create: rank %d of %d: %s did not complete within %.1f s (FFQ_SHARD_TIMEOUT_S): a peer is missing or the fabric does not carry the communicator",
                    rank, world, what, timeout_s);
    }
    int init(const uint8_t *id128, int rank_, int world_, int device, hipStream_t st, bool serial_)
    {
        A = rccl_api();
        if (!A) return fail(FFQ_E_NODEVICE, "ffq_shard: librccl could not be loaded (FFQ_RCCL_LIB names another copy)");
        rank = rank_; world = world_; serial = serial_; this->device = device;
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        RCCLCHK(A->CommInitRank(&cx, world, id, rank));
        RCCLCHK(A->CommCount(cx, &n_cx));
        // who is there: every rank's PCI bus id, ONE all-gather on the new communicator (and its first collective)
        bus.assign((size_t)world, -1);
        int64_t *d_b = nullptr;
        HIPCHK(hipMalloc((void **)&d_b, (size_t)(world + 1) * 8 + sizeof(ncclUniqueId)));
        const int64_t mine = sh_bus_id(device);
        HIPCHK(hipMemcpyAsync(d_b + world, &mine, 8, hipMemcpyHostToDevice, st));
        RCCLCHK(A->AllGather(d_b + world, d_b, 1, ncclInt64, cx, st));
        HIPCHK(hipMemcpyAsync(bus.data(), d_b, (size_t)world * 8, hipMemcpyDeviceToHost, st));
        int rc = settle(st, "the first all-gather (the ranks' bus ids)");
        if (rc) return rc;                  // (d_b is left to the process: something may still write it)
        if (!serial) {
            // the second communicator: rank 0 draws its id and sends it round over the first
            ncclUniqueId id2;
            if (rank == 0) RCCLCHK(A->GetUniqueId(&id2));
            uint8_t *d_id = reinterpret_cast<uint8_t *>(d_b + world + 1);
            if (rank == 0) HIPCHK(hipMemcpyAsync(d_id, &id2, sizeof id2, hipMemcpyHostToDevice, st));
            RCCLCHK(A->Broadcast(d_id, d_id, sizeof id2, ncclUint8, 0, cx, st));
            HIPCHK(hipMemcpyAsync(&id2, d_id, sizeof id2, hipMemcpyDeviceToHost, st));
            if ((rc = settle(st, "the broadcast of the second communicator's id"))) return rc;
            RCCLCHK(A->CommInitRank(&cg, world, id2, rank));
            RCCLCHK(A->CommCount(cg, &n_cg));
        }
        (void)hipFree(d_b);
        if (n_cx != world || (!serial && n_cg != world))
            return fail(FFQ_E_INTERNAL, "ffq_shard_create: the communicators count %d / %d ranks, the world has %d", n_cx, n_cg, world);
        return FFQ_OK;
    }
    int exchange(const std::vector<ShPiece> &plan, const ShPtrFn &provide, const ShPtrFn &accept, hipStream_t st) override
    {
        bool any = false;
        for (const ShPiece &p : plan) any = any || p.src == rank || p.dst == rank;
        if (!any) return FFQ_OK;
        RCCLCHK(A->GroupStart());
        for (const ShPiece &p : plan) {
            if (p.src == rank) RCCLCHK(A->Send(provide(p.a, p.b), (size_t)(p.b - p.a), ncclUint8, p.dst, cx, st));
            if (p.dst == rank) RCCLCHK(A->Recv(accept(p.a, p.b), (size_t)(p.b - p.a), ncclUint8, p.src, cx, st));
        }
        RCCLCHK(A->GroupEnd());
        return FFQ_OK;
    }
    int gather_enqueue(const int64_t *d_mine, int64_t *d_all, int64_t *h_all, hipStream_t st) override
    {
        RCCLCHK(A->AllGather(d_mine, d_all, SH_WORDS, ncclInt64, serial ? cx : cg, st));
        HIPCHK(hipMemcpyAsync(h_all, d_all, (size_t)world * SH_WORDS * 8, hipMemcpyDeviceToHost, st));
        return FFQ_OK;
    }
    // (the step waits for the gather's end event itself, with the watchdog: nothing left to do here)
    int gather_finish(int64_t *, hipEvent_t) override { return FFQ_OK; }
    const char *name() const override { return "rccl"; }
};

// k logical ranks as threads of ONE process (ranges of one resident buffer on one GPU): hand-offs by device copies
struct ShLocal : ShTransport {
    ffq_shard_world *W = nullptr;
    int last_stage = FFQ_SHARD_STAGE_NONE;
    // the in-process world's barrier with the watchdog's deadline: whoever runs out of time breaks it for everybody
    int meet(int stage)
    {
        const int r = W->wait_for(rank, timeout_s);
        if (r > 0) return FFQ_OK;
        if (r < 0 || W->timed_out) {
            poisoned = true; last_stage = stage;
            return fail(FFQ_E_TIMEOUT, "ffq_shard: rank %d of %d: no progress within %.1f s at stage '%s' (transport in-process, %s step): rank(s) %s did not arrive",
                        rank, world, timeout_s, sh_stage_name(stage), serial ? "serial" : "pipelined", W->absent_list().c_str());
        }
        return fail(FFQ_E_INTERNAL, "ffq_shard: another logical rank failed");
    }
    int exchange(const std::vector<ShPiece> &plan, const ShPtrFn &provide, const ShPtrFn &accept, hipStream_t st) override
    {
        W->providers[rank] = &provide;
        int rc = meet(FFQ_SHARD_STAGE_HANDOFF);
        if (rc) return rc;
        hipError_t e = hipSuccess;
        for (const ShPiece &p : plan)
            if (p.dst == rank && e == hipSuccess)
                e = hipMemcpyAsync(accept(p.a, p.b), (*static_cast<const ShPtrFn *>(W->providers[p.src]))(p.a, p.b), (size_t)(p.b - p.a), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);            // the sources must stay as they are until read
        if ((rc = meet(FFQ_SHARD_STAGE_HANDOFF))) return rc;
        if (e != hipSuccess) return fail(FFQ_E_HIP, "ffq_shard: hand-off copy failed: %s", hipGetErrorString(e));
        return FFQ_OK;
    }
    int gather_enqueue(const int64_t *d_mine, int64_t *, int64_t *h_all, hipStream_t st) override
    {
        HIPCHK(hipMemcpyAsync(h_all + (size_t)rank * SH_WORDS, d_mine, SH_WORDS * 8, hipMemcpyDeviceToHost, st));
        return FFQ_OK;
    }
    int gather_finish(int64_t *h_all, hipEvent_t) override            // (this rank's words are on the host: the step has waited for them)
    {
        memcpy(&W->slots[(size_t)rank * SH_WORDS], h_all + (size_t)rank * SH_WORDS, SH_WORDS * 8);
        int rc = meet(FFQ_SHARD_STAGE_GATHER);
        if (rc) return rc;
        memcpy(h_all, W->slots.data(), (size_t)world * SH_WORDS * 8);
        return meet(FFQ_SHARD_STAGE_GATHER);
    }
    const char *name() const override { return "in-process"; }
};

// The caller's own transport under the DEVICE step (ffq_shard_create_hosted): ranks that cannot talk RCCL -- several
// processes on ONE GPU (a dry run; RCCL refuses two ranks per device), a group over gloo or MPI.  Hand-offs are staged
// through host memory (device -> host, the caller's exchange, host -> device), the words gathered by the caller's
// allgather once this rank's are on the host.  Functional, not fast: file-backed shards hand off nothing and gather 64 bytes.
// (The watchdog covers the device side of a step; how long a callback may block is the caller's transport's business --
// a gloo group has its own timeout.)
struct ShHosted : ShTransport {
    ffq_shard_host_ops ops{};
    int exchange(const std::vector<ShPiece> &plan, const ShPtrFn &provide, const ShPtrFn &accept, hipStream_t st) override
    {
        std::vector<ffq_shard_piece> ps(plan.size());
        std::vector<std::vector<uint8_t>> stage(plan.size());
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < plan.size(); i++) {
            const ShPiece &p = plan[i];
            ps[i].src = p.src; ps[i].dst = p.dst; ps[i].a = p.a; ps[i].b = p.b; ps[i].ptr = nullptr;
            if (p.src != rank && p.dst != rank) continue;
            stage[i].resize((size_t)(p.b - p.a));
            ps[i].ptr = stage[i].data();
            if (p.src == rank && e == hipSuccess) e = hipMemcpyAsync(stage[i].data(), provide(p.a, p.b), (size_t)(p.b - p.a), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail(FFQ_E_HIP, "ffq_shard: hand-off staging failed: %s", hipGetErrorString(e));
        const int r = ops.exchange(ops.user, ps.data(), (int)ps.size());
        if (r) return fail(r < 0 ? r : FFQ_E_INTERNAL, "ffq_shard: the exchange callback failed (%d)", r);
        for (size_t i = 0; i < plan.size(); i++)
            if (plan[i].dst == rank && e == hipSuccess)
                e = hipMemcpyAsync(accept(plan[i].a, plan[i].b), stage[i].data(), stage[i].size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);            // (the staging buffers go away with this call)
        if (e != hipSuccess) return fail(FFQ_E_HIP, "ffq_shard: hand-off staging failed: %s", hipGetErrorString(e));
        return FFQ_OK;
    }
    int gather_enqueue(const int64_t *d_mine, int64_t *, int64_t *h_all, hipStream_t st) override
    {
        HIPCHK(hipMemcpyAsync(h_all + (size_t)rank * SH_WORDS, d_mine, SH_WORDS * 8, hipMemcpyDeviceToHost, st));
        return FFQ_OK;
    }
    int gather_finish(int64_t *h_all, hipEvent_t) override
    {
        int64_t mine[SH_WORDS];
        memcpy(mine, h_all + (size_t)rank * SH_WORDS, sizeof mine);
        const int r = ops.allgather(ops.user, mine, h_all);
        return r ? fail(r < 0 ? r : FFQ_E_INTERNAL, "ffq_shard: the gather callback failed (%d)", r) : FFQ_OK;
    }
    const char *name() const override { return "hosted"; }
};

}  // namespace ffq

using namespace ffq;

struct ffq_shard {
    ffq_ctx *c = nullptr;
    ShTransport *tr = nullptr;
    bool owns_tr = true, owns_streams = true;
    int rank = 0, world = 1;
    std::vector<int64_t> B;                // S_0 .. S_world
    int64_t tail_bytes = 0, head_bytes = 0;
    int64_t lo = 0, hi = 0, total = 0, origin = 0;
    hipStream_t comm = nullptr;            // the hand-off stream
    hipStream_t gstream = nullptr;         // the gather stream
    hipEvent_t ev_x[2] = {nullptr, nullptr}, ev_g[2] = {nullptr, nullptr}, ev_w = nullptr;
    int64_t *d_words = nullptr;            // [16]: the words, row_lo, row_hi, rows
    int64_t *d_all = nullptr;              // [world * SH_WORDS]
    int64_t *h_all = nullptr, *h_own = nullptr;      // pinned
    int64_t *hm_own = nullptr;             // h_own as the device sees it (host-mapped)
    uint8_t *grown = nullptr;              // a view with more look-ahead than the caller's buffer has room for
    int64_t grown_cap = 0;
    // the pending step
    bool pending = false, handoff_timed = false;
    ShView v{};
    uint8_t *ext = nullptr;
    int64_t start = -1;                    // stream offset the first search of the last local scan started at (-1: the view's start)
    uint32_t flags = 0;
    int qual_add = 0;
    int64_t *d_table = nullptr, table_cap = 0, *d_qoff = nullptr, qual_cap = 0;
    int8_t *d_qual = nullptr;
    int64_t handoff_bytes = 0;
    // a range of a FILE (ffq_shard_load_fd): the view's bytes -- halos included -- were read from fd into file_ext; a
    // step over that buffer hands off nothing, and a look-ahead that must grow is read from the file too
    int fd = -1;
    uint8_t *file_ext = nullptr;
    bool from_file = false;                // the pending step runs over file_ext
    // the watchdog (include/ffq.h): where the last trip found the step; a stall to inject into the next one
    int last_stage = FFQ_SHARD_STAGE_NONE;
    int stall_stage = FFQ_SHARD_STAGE_NONE;
    double stall_s = 0;
    int *h_stall = nullptr, *hm_stall = nullptr;     // the flag an injected stall waits for (host-mapped)
    uint8_t *slab = nullptr;               // ffq_shard_scan_fd_slabs: the one device buffer a range that does not fit goes through
    int64_t slab_cap = 0;
    std::vector<uint8_t *> graveyard;      // grown views that were replaced: freed with the shard (hipFree waits for the device)
    hipStream_t diag_st = nullptr;         // what a watchdog trip reads the gather's buffer with (made with the shard: nothing may be
    hipEvent_t diag_ev = nullptr;          //   allocated or freed while a kernel is stuck -- those calls wait for the device)
    int64_t *h_diag = nullptr;
    bool leaked = false;                   // an abort could not drain the streams: destroy frees nothing on the device
    // a failure of THIS rank's own scan inside a step: the peers are told through the words before it is returned
    int local_fail = 0;
    std::string local_msg;
};

// the streams of a step: the pipelined step's own three, or -- serial -- the scan stream for everything
static hipStream_t sh_xs(const ffq_shard *s, bool overlap) { return (overlap && !s->tr->serial) ? s->comm : s->c->stream; }
static hipStream_t sh_gs(const ffq_shard *s) { return s->tr->serial ? s->c->stream : s->gstream; }

static bool sh_debug() { static const bool on = getenv("FFQ_SHARD_DEBUG") != nullptr; return on; }

static ShView sh_view(const ffq_shard *s, int64_t tail, int64_t head) { return sh_make_view(s->lo, s->hi, s->total, s->origin, tail, head); }

static int shard_alloc(ffq_shard *s, ffq_shard *parent = nullptr)
{
    HIPCHK(hipSetDevice(s->c->device));
    if (parent) {
        // (a lane: one communicator is driven from ONE stream -- the hand-off and gather streams are the parent's)
        s->comm = parent->comm; s->gstream = parent->gstream; s->owns_streams = false;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&s->comm, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&s->gstream, hipStreamNonBlocking));
    }
    for (auto &e : s->ev_x) HIPCHK(hipEventCreate(&e));
    for (auto &e : s->ev_g) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventCreateWithFlags(&s->ev_w, hipEventDisableTiming));
    HIPCHK(hipMalloc((void **)&s->d_words, 16 * 8));
    HIPCHK(hipMalloc((void **)&s->d_all, (size_t)s->world * SH_WORDS * 8));
    HIPCHK(hipHostMalloc((void **)&s->h_all, (size_t)s->world * SH_WORDS * 8, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&s->h_own, 16 * 8, hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void **)&s->hm_own, s->h_own, 0));
    HIPCHK(hipStreamCreateWithFlags(&s->diag_st, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&s->diag_ev, hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void **)&s->h_diag, (size_t)s->world * SH_WORDS * 8, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&s->h_stall, 64, hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void **)&s->hm_stall, s->h_stall, 0));
    *s->h_stall = 0;
    return FFQ_OK;
}

static bool sh_serial_env() { const char *e = getenv("FFQ_SHARD_SERIAL"); return e && *e && *e != '0'; }

// every stream of the shard idle within `seconds`?  (after an abort: kernels of a collective that waited for a peer leave)
static bool shard_drained(ffq_shard *s, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t sts[3] = {s->owns_streams ? s->comm : nullptr, s->owns_streams ? s->gstream : nullptr, s->c->stream};
    for (;;) {
        bool busy = false;
        for (hipStream_t st : sts) if (st && hipStreamQuery(st) == hipErrorNotReady) busy = true;
        if (!busy) return true;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
        usleep(200);
    }
}

extern "C" int ffq_shard_abort(ffq_shard *s)
{
    if (!s) return fail(FFQ_E_ARG, "ffq_shard_abort: NULL shard");
    (void)hipSetDevice(s->c->device);
    if (s->h_stall) __atomic_store_n(s->h_stall, 1, __ATOMIC_RELEASE);
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (sh_debug()) fprintf(stderr, "[ffq shard %d/%d] abort: communicators ...\n", s->rank, s->world);
    if (s->tr) { s->tr->poisoned = true; s->tr->abort(); }
    if (sh_debug()) fprintf(stderr, "[ffq shard %d/%d] abort: communicators done after %.2f s; streams ...\n", s->rank, s->world, since());
    s->pending = false;
    s->c->pend.active = false;              // (the scan of the abandoned step: its result is never read)
    if (ShRccl *r = dynamic_cast<ShRccl *>(s->tr)) {
        if (r->abort_stuck) {
            // ncclCommAbort itself has not come back (a peer whose PROCESS is gone: RCCL's teardown waits on its sockets for
            // minutes) and holds the device's runtime meanwhile -- any HIP call of this thread, a stream query included, would
            // wait behind it (seen with a real dead peer, tests/test_multigpu.py).  The caller has its error; nothing more of
            // this shard is touched, and a host that wants to leave should leave with _exit.
            s->leaked = true;
            return fail(FFQ_E_TIMEOUT, "ffq_shard_abort: rank %d of %d: ncclCommAbort has not come back in time (FFQ_SHARD_ABORT_S; a peer's process is gone?); it goes on "
                                       "on a thread of its own and holds the device meanwhile -- the shard is leaked, leave with _exit", s->rank, s->world);
        }
    }
    mark_other(s->c);
    const bool drained = shard_drained(s, 30.0);
    if (sh_debug()) fprintf(stderr, "[ffq shard %d/%d] abort: streams %s after %.2f s\n", s->rank, s->world, drained ? "drained" : "NOT drained", since());
    if (drained) return FFQ_OK;
    s->leaked = true;
    return fail(FFQ_E_TIMEOUT, "ffq_shard_abort: rank %d of %d: the shard's streams did not drain within 30 s of the abort", s->rank, s->world);
}

static int shard_common(ffq_ctx *c, int rank, int world, const int64_t *bounds, int64_t tail_bytes, int64_t head_bytes, ffq_shard **out)
{
    if (!c || !bounds || !out) return fail(FFQ_E_ARG, "ffq_shard_create: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(FFQ_E_ARG, "ffq_shard_create: rank %d of %d", rank, world);
    // (the byte in front of the range must be in view: a record that starts at the range's first byte is found through
    // the "\n" before it)
    if (tail_bytes < 1 || head_bytes < 1) return fail(FFQ_E_ARG, "ffq_shard_create: tail_bytes and head_bytes must be at least 1");
    for (int r = 0; r < world; r++)
        if (bounds[r] > bounds[r + 1]) return fail(FFQ_E_ARG, "ffq_shard_create: bounds must not decrease");
    ffq_shard *s = new (std::nothrow) ffq_shard();
    if (!s) return fail(FFQ_E_NOMEM, "out of host memory");
    s->c = c; s->rank = rank; s->world = world;
    s->B.assign(bounds, bounds + world + 1);
    s->tail_bytes = tail_bytes; s->head_bytes = head_bytes;
    s->lo = bounds[rank]; s->hi = bounds[rank + 1]; s->total = bounds[world]; s->origin = bounds[0];
    *out = s;
    return FFQ_OK;
}

extern "C" void ffq_shard_destroy(ffq_shard *s)
{
    if (!s) return;
    (void)hipSetDevice(s->c->device);
    if (s->h_stall) __atomic_store_n(s->h_stall, 1, __ATOMIC_RELEASE);
    if (s->tr && s->tr->poisoned && !s->leaked) (void)ffq_shard_abort(s);      // (a step that never came back: do not wait for it here)
    if (s->leaked) {
        // something still runs on the shard's streams and nothing will stop it: the device memory those kernels may touch,
        // the streams and the events stay (hipFree would wait for the device); the host objects go
        if (s->owns_tr) delete s->tr;
        delete s;
        return;
    }
    if (s->comm) { (void)hipStreamSynchronize(s->comm); }
    if (s->gstream) { (void)hipStreamSynchronize(s->gstream); }
    (void)hipStreamSynchronize(s->c->stream);
    if (s->owns_tr) delete s->tr;
    if (s->comm && s->owns_streams) (void)hipStreamDestroy(s->comm);
    if (s->gstream && s->owns_streams) (void)hipStreamDestroy(s->gstream);
    if (s->ev_w) (void)hipEventDestroy(s->ev_w);
    for (auto e : s->ev_x) if (e) (void)hipEventDestroy(e);
    for (auto e : s->ev_g) if (e) (void)hipEventDestroy(e);
    (void)hipFree(s->d_words); (void)hipFree(s->d_all); (void)hipFree(s->grown); (void)hipFree(s->slab);
    for (uint8_t *g : s->graveyard) (void)hipFree(g);
    if (s->h_all) (void)hipHostFree(s->h_all);
    if (s->h_own) (void)hipHostFree(s->h_own);
    if (s->h_stall) (void)hipHostFree(s->h_stall);
    if (s->h_diag) (void)hipHostFree(s->h_diag);
    if (s->diag_ev) (void)hipEventDestroy(s->diag_ev);
    if (s->diag_st) (void)hipStreamDestroy(s->diag_st);
    delete s;
}

extern "C" int ffq_shard_unique_id(uint8_t *id128)
{
    if (!id128) return fail(FFQ_E_ARG, "ffq_shard_unique_id: NULL argument");
    RcclApi *A = rccl_api();
    if (!A) return fail(FFQ_E_NODEVICE, "ffq_shard: librccl could not be loaded (FFQ_RCCL_LIB names another copy)");
    ncclUniqueId id;
    RCCLCHK(A->GetUniqueId(&id));
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof id);
    return FFQ_OK;
}

extern "C" int ffq_shard_create2(ffq_ctx *c, const uint8_t *id128, int rank, int world, const int64_t *bounds,
                                 int64_t tail_bytes, int64_t head_bytes, uint32_t mode, ffq_shard **out)
{
    if (!id128) return fail(FFQ_E_ARG, "ffq_shard_create: NULL unique id");
    int rc = shard_common(c, rank, world, bounds, tail_bytes, head_bytes, out);
    if (rc) return rc;
    ffq_shard *s = *out;
    *out = nullptr;
    rc = shard_alloc(s);
    if (!rc) {
        ShRccl *t = new (std::nothrow) ShRccl();
        s->tr = t;
        // (serial: the one-communicator step from the start -- the second communicator is never made, the set-up's own
        // collectives run on the scan stream)
        const bool serial = (mode & FFQ_SHARD_F_SERIAL) != 0;
        rc = t ? t->init(id128, rank, world, c->device, serial ? c->stream : s->comm, serial) : fail(FFQ_E_NOMEM, "out of host memory");
    }
    if (rc) { ffq_shard_destroy(s); return rc; }
    *out = s;
    return FFQ_OK;
}

extern "C" int ffq_shard_create(ffq_ctx *c, const uint8_t *id128, int rank, int world, const int64_t *bounds,
                                int64_t tail_bytes, int64_t head_bytes, ffq_shard **out)
{
    return ffq_shard_create2(c, id128, rank, world, bounds, tail_bytes, head_bytes, sh_serial_env() ? FFQ_SHARD_F_SERIAL : 0u, out);
}

// a second shard object of the same rank on another context (own scratch, own buffers: steps queued one ahead), on the
// first one's communicators
extern "C" int ffq_shard_create_lane(ffq_shard *parent, ffq_ctx *c, ffq_shard **out)
{
    if (!parent || !c || !out) return fail(FFQ_E_ARG, "ffq_shard_create_lane: NULL argument");
    int rc = shard_common(c, parent->rank, parent->world, parent->B.data(), parent->tail_bytes, parent->head_bytes, out);
    if (rc) return rc;
    ffq_shard *s = *out;
    *out = nullptr;
    rc = shard_alloc(s, parent);
    if (rc) { ffq_shard_destroy(s); return rc; }
    s->tr = parent->tr; s->owns_tr = false;
    *out = s;
    return FFQ_OK;
}

extern "C" int ffq_shard_world_create(int world, ffq_shard_world **out)
{
    if (!out || world < 1) return fail(FFQ_E_ARG, "ffq_shard_world_create: bad argument");
    ffq_shard_world *w = new (std::nothrow) ffq_shard_world();
    if (!w) return fail(FFQ_E_NOMEM, "out of host memory");
    w->world = world;
    w->providers.assign(world, nullptr);
    w->slots.assign((size_t)world * SH_WORDS, 0);
    *out = w;
    return FFQ_OK;
}
extern "C" void ffq_shard_world_abort(ffq_shard_world *w) { if (w) w->abort(); }
extern "C" void ffq_shard_world_destroy(ffq_shard_world *w) { delete w; }

extern "C" int ffq_shard_create_local(ffq_ctx *c, ffq_shard_world *w, int rank, const int64_t *bounds, int64_t tail_bytes,
                                      int64_t head_bytes, ffq_shard **out)
{
    if (!w) return fail(FFQ_E_ARG, "ffq_shard_create_local: NULL world");
    int rc = shard_common(c, rank, w->world, bounds, tail_bytes, head_bytes, out);
    if (rc) return rc;
    ffq_shard *s = *out;
    *out = nullptr;
    rc = shard_alloc(s);
    if (!rc) {
        ShLocal *t = new (std::nothrow) ShLocal();
        if (t) { t->W = w; t->rank = rank; t->world = w->world; t->serial = sh_serial_env(); t->bus.assign((size_t)w->world, -1); t->bus[(size_t)rank] = sh_bus_id(c->device); }
        s->tr = t;
        if (!t) rc = fail(FFQ_E_NOMEM, "out of host memory");
    }
    if (rc) { ffq_shard_destroy(s); return rc; }
    *out = s;
    return FFQ_OK;
}

extern "C" int ffq_shard_create_hosted(ffq_ctx *c, const ffq_shard_host_ops *ops, int rank, int world, const int64_t *bounds,
                                       int64_t tail_bytes, int64_t head_bytes, ffq_shard **out)
{
    if (!ops || !ops->allgather || (world > 1 && !ops->exchange)) return fail(FFQ_E_ARG, "ffq_shard_create_hosted: NULL callback");
    int rc = shard_common(c, rank, world, bounds, tail_bytes, head_bytes, out);
    if (rc) return rc;
    ffq_shard *s = *out;
    *out = nullptr;
    rc = shard_alloc(s);
    if (!rc) {
        ShHosted *t = new (std::nothrow) ShHosted();
        if (t) { t->ops = *ops; t->rank = rank; t->world = world; t->serial = sh_serial_env(); t->bus.assign((size_t)world, -1); t->bus[(size_t)rank] = sh_bus_id(c->device); }
        s->tr = t;
        if (!t) rc = fail(FFQ_E_NOMEM, "out of host memory");
    }
    if (rc) { ffq_shard_destroy(s); return rc; }
    *out = s;
    return FFQ_OK;
}

extern "C" int ffq_shard_halo(ffq_shard *s, int64_t *tail, int64_t *head)
{
    if (!s || !tail || !head) return fail(FFQ_E_ARG, "ffq_shard_halo: NULL argument");
    sh_halo_sizes(s->B, s->rank, s->tail_bytes, s->head_bytes, tail, head);
    return FFQ_OK;
}

// one exchange: this rank provides its own bytes out of `ext` and receives what the plan sends it into dst_ext (stream
// offset dst_start at index 0)
static int shard_serve(ffq_shard *s, const std::vector<ShPiece> &plan, uint8_t *ext, int64_t tail, uint8_t *dst_ext,
                       int64_t dst_start, hipStream_t st)
{
    const int64_t own_lo = s->lo;
    ShPtrFn provide = [=](int64_t a, int64_t) { return ext + tail + (a - own_lo); };
    ShPtrFn accept = [=](int64_t a, int64_t) { return dst_ext + (a - dst_start); };
    for (const ShPiece &p : plan) {
        if (p.src == s->rank) s->handoff_bytes += p.b - p.a;
        if (p.dst == s->rank) s->handoff_bytes += p.b - p.a;
    }
    return s->tr->exchange(plan, provide, accept, st);
}

// diagnostics: the stall asked for at this stage, once (ffq_shard_inject_stall)
static void shard_stall(ffq_shard *s, int stage, hipStream_t st)
{
    if (s->stall_stage != stage) return;
    s->stall_stage = FFQ_SHARD_STAGE_NONE;
    *s->h_stall = 0;
    hipLaunchKernelGGL(k_shard_stall, dim3(1), dim3(64), 0, st, (const int *)s->hm_stall, (unsigned long long)(s->stall_s * 1e8));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess || sh_debug()) fprintf(stderr, "[ffq shard %d/%d] stall injected at stage '%s' for up to %.1f s: %s\n", s->rank, s->world, sh_stage_name(stage), s->stall_s, hipGetErrorString(e));
}

// step 1 alone (a caller that scans by other means): fills ext[:tail] and ext[tail + own : tail + own + head]
static int shard_handoff(ffq_shard *s, uint8_t *ext, int64_t tail, bool overlap)
{
    const bool stall = s->stall_stage == FFQ_SHARD_STAGE_HANDOFF;
    if (s->world < 2 && !stall) return FFQ_OK;
    std::vector<ShPiece> plan;
    if (s->world >= 2) sh_halo_plan(s->B, s->tail_bytes, s->head_bytes, plan);
    mark_other(s->c);
    hipStream_t st = sh_xs(s, overlap);
    HIPCHK(hipEventRecord(s->ev_x[0], st));
    shard_stall(s, FFQ_SHARD_STAGE_HANDOFF, st);
    int rc = plan.empty() ? FFQ_OK : shard_serve(s, plan, ext, tail, ext, s->lo - tail, st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(s->ev_x[1], st));
    if (st != s->c->stream) HIPCHK(hipStreamWaitEvent(s->c->stream, s->ev_x[1], 0));
    s->handoff_timed = true;
    return FFQ_OK;
}

// step 4: the gather of the words the scan stream has just been given to produce -- on the GATHER stream, which waits for
// them: the collective (tens of microseconds with peers) does not sit between this step's scan and the next one's, already
// queued behind it on the scan stream.  (Serial: on the scan stream itself, in order.)
static int shard_gather(ffq_shard *s, bool waited = false)
{
    hipStream_t gs = sh_gs(s);
    if (!waited) {
        HIPCHK(hipEventRecord(s->ev_w, s->c->stream));
        if (gs != s->c->stream) HIPCHK(hipStreamWaitEvent(gs, s->ev_w, 0));
    }
    // (what a watchdog trip reads to say whose words are there: "look-ahead had" is never negative)
    if (s->world > 1 && !strcmp(s->tr->name(), "rccl")) HIPCHK(hipMemsetAsync(s->d_all, 0xFF, (size_t)s->world * SH_WORDS * 8, gs));
    HIPCHK(hipEventRecord(s->ev_g[0], gs));
    shard_stall(s, FFQ_SHARD_STAGE_GATHER, gs);
    int rc = s->tr->gather_enqueue(s->d_words, s->d_all, s->h_all, gs);
    if (rc) return rc;
    HIPCHK(hipEventRecord(s->ev_g[1], gs));
    return FFQ_OK;
}

// steps 3-4 behind a scan of view v that searched from buffer offset `offset`
static int shard_words_and_gather(ffq_shard *s, int64_t offset)
{
    // (on the gather stream too, behind ONE marker on the scan stream: the two lower bounds are a handful of dependent
    // memory round trips -- 17 us per step when they sat between this step's scan and the next one's)
    ffq_ctx *c = s->c;
    hipStream_t gs = sh_gs(s);
    mark_other(c);
    HIPCHK(hipEventRecord(s->ev_w, c->stream));
    if (gs != c->stream) HIPCHK(hipStreamWaitEvent(gs, s->ev_w, 0));
    hipLaunchKernelGGL(k_shard_words, dim3(1), dim3(64), 0, gs, (const DevRes *)c->dres, (const int64_t *)s->d_table,
                       s->table_cap, (s->flags & FFQ_F_DECODE_QUAL) ? s->qual_cap : (int64_t)-1, s->v, offset, s->head_bytes, s->d_words, s->hm_own);
    HIPCHK(hipGetLastError());
    return shard_gather(s, true);
}

// words set on the host (a rank that learns that the chain passes over its whole range): the next gather carries them
static int shard_gather_host_words(ffq_shard *s)
{
    hipStream_t st = s->c->stream;
    mark_other(s->c);
    HIPCHK(hipMemcpyAsync(s->d_words, s->h_own, 16 * 8, hipMemcpyHostToDevice, st));
    return shard_gather(s);
}

// ---- the watchdog ------------------------------------------------------------------------------------------------------
// where a step that does not come back is stuck: the hand-off's end mark, the mark behind the scan front, the gather's
static int shard_stage(ffq_shard *s)
{
    if (s->handoff_timed && hipEventQuery(s->ev_x[1]) == hipErrorNotReady) return FFQ_SHARD_STAGE_HANDOFF;
    if (hipEventQuery(s->ev_w) == hipErrorNotReady) return FFQ_SHARD_STAGE_SCAN;
    return FFQ_SHARD_STAGE_GATHER;
}

// whose words the unfinished gather has delivered so far, as far as this rank can see (the buffer was filled with 0xFF in
// front of the collective): read on a stream of its own, with a deadline of its own
static std::string shard_words_seen(ffq_shard *s)
{
    // (everything it needs was made with the shard: a hipHostFree / hipFree / hipStreamDestroy HERE would wait for the device --
    // for the very kernel that is stuck; found with real peers, tests/test_multigpu.py: the trip came back after the stall's
    // 60 s instead of the deadline's 4)
    std::string have, lack;
    const size_t nw = (size_t)s->world * SH_WORDS;
    if (!s->diag_st || !s->diag_ev || !s->h_diag) return "which ranks' words have arrived could not be read";
    if (hipMemcpyAsync(s->h_diag, s->d_all, nw * 8, hipMemcpyDeviceToHost, s->diag_st) != hipSuccess || hipEventRecord(s->diag_ev, s->diag_st) != hipSuccess ||
        sh_wait_event(s->diag_ev, 1.0, nullptr) != 0)
        return "which ranks' words have arrived could not be read";
    for (int r = 0; r < s->world; r++) {
        std::string &dst = s->h_diag[(size_t)r * SH_WORDS + 4] >= 0 ? have : lack;
        if (!dst.empty()) dst += ", ";
        dst += std::to_string(r);
    }
    return "words present from rank(s) [" + have + "], absent from [" + lack + "]";
}

// the watchdog has tripped (w: 1 the deadline, 2 an asynchronous RCCL error): where, who, FFQ_E_TIMEOUT / FFQ_E_HIP
static int shard_tripped(ffq_shard *s, double waited, int w = 1, int at = -1)
{
    ShTransport *tr = s->tr;
    const int stage = at >= 0 ? at : shard_stage(s);
    s->last_stage = stage;
    tr->poisoned = true;
    s->pending = false;
    if (w == 2)
        return fail(FFQ_E_HIP, "ffq_shard_step_wait: rank %d of %d: RCCL reports an asynchronous error at stage '%s' (%s step): %s", s->rank, s->world,
                    sh_stage_name(stage), tr->serial ? "serial" : "pipelined", tr->async_error() ? tr->async_error() : "?");
    std::string seen = (stage == FFQ_SHARD_STAGE_GATHER && s->world > 1 && !strcmp(tr->name(), "rccl")) ? "; " + shard_words_seen(s) : std::string();
    return fail(FFQ_E_TIMEOUT, "ffq_shard_step_wait: rank %d of %d: no progress within %.1f s at stage '%s' (transport %s, %s step; FFQ_SHARD_TIMEOUT_S)%s",
                s->rank, s->world, waited, sh_stage_name(stage), tr->name(), tr->serial ? "serial" : "pipelined", seen.c_str());
}

// the scan stream drained, with the watchdog's deadline
static int shard_sync_scan_stream(ffq_shard *s)
{
    ffq_ctx *c = s->c;
    const double keep = c->watchdog_s;
    c->watchdog_s = s->tr->timeout_s;
    const int rc = ctx_wait(c, nullptr, c->stream);
    c->watchdog_s = keep;
    return rc == FFQ_E_TIMEOUT ? shard_tripped(s, s->tr->timeout_s) : rc;
}

// waits for one mark of the pending step with the watchdog's deadline
static int shard_wait_mark(ffq_shard *s, hipEvent_t ev)
{
    double waited = 0;
    const int w = sh_wait_event(ev, s->tr->timeout_s, s->tr, &waited);
    if (w <= 0) return w;
    return shard_tripped(s, waited, w);
}

extern "C" int ffq_shard_step_submit(ffq_shard *s, uint8_t *d_ext, int overlap_handoff, uint32_t flags, int qual_add,
                                     int64_t *d_table, int64_t table_cap, int8_t *d_qual, int64_t qual_cap, int64_t *d_qoff)
{
    if (!s || !d_ext) return fail(FFQ_E_ARG, "ffq_shard_step_submit: NULL argument");
    if (s->pending) return fail(FFQ_E_ARG, "ffq_shard_step_submit: a step is already pending on this shard");
    if (s->tr->poisoned) return fail(FFQ_E_ARG, "ffq_shard_step_submit: this shard's last step did not come back (watchdog): only ffq_shard_abort / _destroy are left");
    ffq_ctx *c = s->c;
    HIPCHK(hipSetDevice(c->device));
    int64_t tail, head;
    sh_halo_sizes(s->B, s->rank, s->tail_bytes, s->head_bytes, &tail, &head);
    s->v = sh_view(s, tail, head);
    s->ext = d_ext; s->start = -1;
    s->flags = flags; s->qual_add = qual_add;        // (FFQ_F_NO_TIMING: no marks; the hand-off's wait in front of the scan rules the barrier-free dispatch out)
    s->d_table = d_table; s->table_cap = table_cap; s->d_qual = d_qual; s->qual_cap = qual_cap; s->d_qoff = d_qoff;
    s->handoff_bytes = 0; s->handoff_timed = false;
    s->local_fail = 0; s->last_stage = FFQ_SHARD_STAGE_NONE;
    s->from_file = s->fd >= 0 && d_ext == s->file_ext;
    // The scan's scratch FIRST: a scratch that has to grow waits for the scan stream (and hipFree for the device), and that
    // wait must not sit behind this step's hand-off -- a collective whose peer may never come.
    int rc = scan_reserve(c, s->v.n_bytes, flags, table_cap, qual_cap);
    if (rc) return rc;
    rc = s->from_file ? FFQ_OK : shard_handoff(s, d_ext, tail, overlap_handoff != 0);
    if (rc) return rc;
    if (s->stall_stage == FFQ_SHARD_STAGE_SCAN) { mark_other(c); shard_stall(s, FFQ_SHARD_STAGE_SCAN, c->stream); }
    c->watchdog_s = s->tr->timeout_s;        // (a front that must grow a scratch of its own waits for the stream: behind the hand-off by now)
    rc = ffq_scan_submit(c, d_ext, s->v.n_bytes, s->v.sentinel, 0, s->v.eof, s->v.add, s->flags, qual_add, d_table, table_cap,
                         d_qual, qual_cap, d_qoff);
    c->watchdog_s = 0;
    if (rc == FFQ_E_TIMEOUT)
        return shard_tripped(s, s->tr->timeout_s, 1, (s->handoff_timed && hipEventQuery(s->ev_x[1]) == hipErrorNotReady) ? FFQ_SHARD_STAGE_HANDOFF : FFQ_SHARD_STAGE_SCAN);
    if (rc) return rc;
    if (s->v.n_bytes == 0) {
        // an empty view (nothing is enqueued for it, the device holds no result block of this scan): its words on the host --
        // no rows, the search "ended" at the view's start, more look-ahead wanted unless the view ends the stream
        int64_t *h = s->h_own;
        sh_words_empty(s->v, s->head_bytes, h);
        h[8] = h[9] = h[10] = 0;
        rc = shard_gather_host_words(s);
    } else rc = shard_words_and_gather(s, 0);
    if (rc) return rc;
    s->pending = true;
    return FFQ_OK;
}

// a synchronous scan of the current view from stream offset `start` (< 0: the view's beginning), its words gathered
static int shard_local(ffq_shard *s, ffq_scan_result *res, int64_t start)
{
    const int64_t offset = start < 0 ? 0 : std::max(start, s->v.start) - s->v.add;
    s->start = start;
    s->c->watchdog_s = s->tr->timeout_s;
    int rc = ffq_scan_device(s->c, s->ext, s->v.n_bytes, s->v.sentinel, offset, s->v.eof, s->v.add, s->flags, s->qual_add,
                             s->d_table, s->table_cap, s->d_qual, s->qual_cap, s->d_qoff, res);
    s->c->watchdog_s = 0;
    if (rc == FFQ_E_TIMEOUT) return shard_tripped(s, s->tr->timeout_s);
    if (rc && rc != FFQ_E_TABLE_FULL) {
        // (every rank gathers at the end of a repair round: mine says "failed", and the step ends with INTERNAL everywhere)
        s->local_fail = rc; s->local_msg = ffq_last_error();
        sh_words_failed(s->v, s->h_own);
        s->h_own[8] = s->h_own[9] = s->h_own[10] = 0;
        return shard_gather_host_words(s);
    }
    return shard_words_and_gather(s, offset);
}

extern "C" int ffq_shard_step_wait(ffq_shard *s, ffq_shard_result *out)
{
    if (!s || !out) return fail(FFQ_E_ARG, "ffq_shard_step_wait: NULL argument");
    if (!s->pending) return fail(FFQ_E_ARG, "ffq_shard_step_wait: no step is pending on this shard");
    memset(out, 0, sizeof *out);
    ffq_ctx *c = s->c;
    HIPCHK(hipSetDevice(c->device));
    const int W = s->world, rank = s->rank;
    const std::vector<int64_t> &B = s->B;
    out->nranks = s->tr->nranks(s->tr->serial ? 0 : 1);
    out->serial = s->tr->serial ? 1 : 0;
    // The watchdog's first look: the mark behind the scan front.  Once it is through, the hand-off is (the scan waited
    // for it) and ffq_scan_wait has nothing left to wait for but its own later tiers -- local work.
    int rc = shard_wait_mark(s, s->ev_w);
    if (rc) return rc;
    s->pending = false;
    // (a scan that needs a later tier queues its kernels behind whatever the scan stream holds by now -- serial: this
    // step's gather, the next step's hand-off -- and waits for them: with the watchdog too)
    c->watchdog_s = s->tr->timeout_s;
    rc = ffq_scan_wait(c, &out->scan);
    c->watchdog_s = 0;
    if (rc == FFQ_E_TIMEOUT) return shard_tripped(s, s->tr->timeout_s);
    // A scan that failed HERE (not the stream's error: a kernel invariant, no memory for a later tier) must not leave the
    // peers waiting in the next collective: such a scan's words say "not ready" (its result block is marked), every rank
    // gathers once more, and this rank's words then say "failed" -- INTERNAL on every rank.
    int &local_fail = s->local_fail;
    std::string &local_msg = s->local_msg;
    local_fail = 0;
    if (rc && rc != FFQ_E_TABLE_FULL) { local_fail = rc; local_msg = ffq_last_error(); }
    int rounds = 0, regathers = 0;
    float ms = 0;
    for (;;) {
        rc = shard_wait_mark(s, s->ev_g[1]);
        if (!rc) rc = s->tr->gather_finish(s->h_all, s->ev_g[1]);
        if (rc) { if (rc == FFQ_E_TIMEOUT) { s->tr->poisoned = true; if (!s->last_stage) s->last_stage = FFQ_SHARD_STAGE_GATHER; } return rc; }
        if (hipEventElapsedTime(&ms, s->ev_g[0], s->ev_g[1]) == hipSuccess) out->allgather_ms += ms;
        const int64_t *A = s->h_all;
        auto word = [&](int r, int k) { return A[(size_t)r * SH_WORDS + k]; };
        const ShRound d = sh_decide(A, W, rank, B, s->v);          // (ffq_shard_proto.h: the protocol's one statement)
        if (sh_debug()) {
            // FFQ_SHARD_DEBUG=1: what this rank sees in every round (a run on several GPUs that nobody can attach to)
            std::string line;
            char buf[256];
            for (int r = 0; r < W; r++) {
                snprintf(buf, sizeof buf, " [%d: exit %lld first %lld n %lld want %lld had %lld err %lld/%lld search %lld]", r, (long long)word(r, 0), (long long)word(r, 1),
                         (long long)word(r, 2), (long long)word(r, 3), (long long)word(r, 4), (long long)word(r, 5), (long long)word(r, 6), (long long)word(r, 7));
                line += buf;
            }
            fprintf(stderr, "[ffq shard %d/%d] round %d regather %d kind %d view [%lld | %lld, %lld | %lld] rows %lld..%lld of %lld:%s\n", rank, W, rounds, regathers, (int)d.kind,
                    (long long)s->v.tail, (long long)s->v.lo, (long long)s->v.hi, (long long)s->v.head, (long long)s->h_own[8], (long long)s->h_own[9], (long long)s->h_own[10], line.c_str());
        }
        if (d.kind == ShRound::TABLE_FULL) {
            out->scan.n_records = d.need;
            return fail(FFQ_E_TABLE_FULL, "rank %d: offset table too small (%lld records in its view)", d.who, (long long)d.need);
        }
        if (d.kind == ShRound::QUAL_FULL) {
            out->scan.n_records = 0; out->scan.n_qual_bytes = d.need;
            return fail(FFQ_E_TABLE_FULL, "rank %d: quality buffer too small (%lld decoded bytes in its view)", d.who, (long long)d.need);
        }
        if (d.kind == ShRound::INTERNAL) {
            if (local_fail) return fail(local_fail, "%s", local_msg.c_str());            // (mine: my own error, not "rank r failed")
            return fail(FFQ_E_INTERNAL, "sharded scan: rank %d %s", d.who, d.what);
        }
        if (d.kind == ShRound::NOT_READY) {
            // some rank's scan needed a later tier (a host round trip inside its ffq_scan_wait): every rank's scan is through
            // by now -- the words once more
            if (++regathers > 4) return fail(FFQ_E_INTERNAL, "sharded scan: a rank's result does not become ready");
            if (local_fail) {
                sh_words_failed(s->v, s->h_own);
                s->h_own[8] = s->h_own[9] = s->h_own[10] = 0;
                rc = shard_gather_host_words(s);
            } else {
                int64_t off = s->start < 0 ? 0 : std::max(s->start, s->v.start) - s->v.add;
                rc = shard_words_and_gather(s, off);
            }
            if (rc) return rc;
            continue;
        }
        if (local_fail) return fail(local_fail, "%s", local_msg.c_str());
        if (d.kind == ShRound::STREAM_ERROR) { out->err_state = d.err_state; out->err_byte = d.err_byte; break; }
        if (d.kind == ShRound::SETTLED) break;
        const std::vector<int> &grow = d.grow;
        if (++rounds > sh_max_rounds(W)) return fail(FFQ_E_INTERNAL, "sharded scan does not settle (%d rounds)", rounds);
        const bool i_grow = d.i_grow, i_force = d.i_force;
        int64_t start = s->start;
        if (!grow.empty() && s->from_file) {
            // every rank's bytes are the file's: a rank that needs more look-ahead reads it itself, nobody serves anybody
            if (i_grow) {
                ShView nv = sh_view(s, s->v.tail, word(rank, 3));
                uint8_t *g = nullptr;
                const int64_t cap = nv.n_bytes + 64;
                if (hipMalloc((void **)&g, (size_t)cap) != hipSuccess) return fail(FFQ_E_NOMEM, "ffq_shard: no memory for a view of %lld bytes", (long long)cap);
                mark_other(c);
                hipError_t e = hipMemcpyAsync(g, s->ext, (size_t)s->v.n_bytes, hipMemcpyDeviceToDevice, c->stream);
                // (waits with the watchdog's deadline; views that are replaced go to the graveyard, not to hipFree: a free waits
                // for the DEVICE -- for the other lane's hand-off, for a collective whose peer may never come)
                rc = e != hipSuccess ? fail(FFQ_E_HIP, "ffq_shard: %s", hipGetErrorString(e)) : shard_sync_scan_stream(s);
                int64_t got = 0;
                const int64_t more = nv.head - s->v.head;
                if (!rc) rc = stage_fd2d(c, g + s->v.n_bytes, s->fd, s->hi + s->v.head, more, &got);
                if (!rc && got != more) rc = fail(FFQ_E_ARG, "ffq_shard: the file ends at byte %lld, its bounds say %lld", (long long)(s->hi + s->v.head + got), (long long)s->total);
                if (rc) { s->graveyard.push_back(g); return rc; }
                if (s->grown) s->graveyard.push_back(s->grown);
                s->grown = g; s->grown_cap = cap;
                s->ext = g; s->v = nv;
            }
        } else if (!grow.empty()) {
            std::vector<ShPiece> plan;
            sh_grow_plan(A, B, grow, plan);
            uint8_t *src_ext = s->ext;
            uint8_t *dst_ext = s->ext;
            int64_t dst_start = s->v.start;
            ShView nv = s->v;
            uint8_t *old_grown = nullptr;
            if (i_grow) {
                // a view with more look-ahead (the caller's buffer has room for its own halo only)
                nv = sh_view(s, s->v.tail, word(rank, 3));
                if (nv.n_bytes + 64 > s->grown_cap || s->ext == s->grown) {
                    uint8_t *g = nullptr;
                    const int64_t cap = nv.n_bytes + 64;
                    if (hipMalloc((void **)&g, (size_t)cap) != hipSuccess) return fail(FFQ_E_NOMEM, "ffq_shard: no memory for a view of %lld bytes", (long long)cap);
                    old_grown = s->grown;
                    s->grown = g; s->grown_cap = cap;
                }
                mark_other(c);
                HIPCHK(hipMemcpyAsync(s->grown, s->ext, (size_t)s->v.n_bytes, hipMemcpyDeviceToDevice, c->stream));
                dst_ext = s->grown; dst_start = nv.start;
            }
            mark_other(c);
            // (src_ext may BE the old grown view -- a rank growing a second time that a neighbour's look-ahead reaches into in
            // the same round: it is freed only once the exchange that reads it is through)
            rc = shard_serve(s, plan, src_ext, s->v.tail, dst_ext, dst_start, c->stream);
            if (old_grown) s->graveyard.push_back(old_grown);      // (freed with the shard: the exchange that reads it may still be in flight)
            if (rc) return rc;
            if (i_grow) { s->ext = s->grown; s->v = nv; }
        }
        bool rescan = i_grow || i_force;
        if (i_force) {
            if (d.passed_over) {
                // the chain passes over my whole range (or ends before it): I own nothing
                int64_t *h = s->h_own;
                sh_words_passed_over(s->v, d.prev_exit, d.prev_search, h);
                h[8] = 0; h[9] = 0; h[10] = 0;
                s->start = d.prev_search;
                rc = shard_gather_host_words(s);
                if (rc) return rc;
                continue;
            }
            start = d.prev_search;
        }
        if (rescan) rc = shard_local(s, &out->scan, start);
        else rc = shard_gather_host_words(s);      // (my words stand: the others' rounds need them again)
        if (rc) return rc;
    }
    // the hand-off's own time (events on the stream it ran on)
    if (s->handoff_timed && hipEventElapsedTime(&ms, s->ev_x[0], s->ev_x[1]) == hipSuccess) out->handoff_ms = ms;
    const int64_t *h = s->h_own;
    out->n_rows = h[10]; out->row_lo = h[8]; out->row_hi = h[9];
    out->exit_pos = h[0]; out->first_pos = h[1];
    out->n_own_records = h[9] - h[8];
    int64_t base = 0, tot = 0;
    for (int r = 0; r < W; r++) {
        const int64_t cnt = s->h_all[(size_t)r * SH_WORDS + 2];
        if (r < rank) base += cnt;
        tot += cnt;
    }
    out->record_base = base; out->total_records = tot;
    out->rounds = rounds; out->regathers = regathers;
    out->halo_source = s->from_file ? 1 : 0;
    out->handoff_bytes = s->handoff_bytes;
    out->d_ext = s->ext; out->tail = s->v.tail; out->head = s->v.head;
    return FFQ_OK;
}

// step 1 alone, for a caller that runs the scan itself: the halos of d_ext from the ranks that own them (on the scan
// stream, or -- overlap -- on the hand-off stream with the scan stream waiting for its end)
extern "C" int ffq_shard_exchange_halo(ffq_shard *s, uint8_t *d_ext, int overlap)
{
    if (!s || !d_ext) return fail(FFQ_E_ARG, "ffq_shard_exchange_halo: NULL argument");
    HIPCHK(hipSetDevice(s->c->device));
    int64_t tail, head;
    sh_halo_sizes(s->B, s->rank, s->tail_bytes, s->head_bytes, &tail, &head);
    return shard_handoff(s, d_ext, tail, overlap != 0);
}

// n bytes from d_src to d_dst through the shard's transport, this rank being both ends (ncclSend + ncclRecv to itself in
// one group on the hand-off stream): what a world of one rank can exercise of the hand-off path
extern "C" int ffq_shard_self_exchange(ffq_shard *s, const uint8_t *d_src, uint8_t *d_dst, int64_t n)
{
    if (!s || !d_src || !d_dst || n < 0) return fail(FFQ_E_ARG, "ffq_shard_self_exchange: bad argument");
    HIPCHK(hipSetDevice(s->c->device));
    std::vector<ShPiece> plan{ShPiece{s->rank, s->rank, 0, n}};
    ShPtrFn provide = [=](int64_t a, int64_t) { return const_cast<uint8_t *>(d_src) + a; };
    ShPtrFn accept = [=](int64_t a, int64_t) { return d_dst + a; };
    hipStream_t st = sh_xs(s, true);
    if (st == s->c->stream) mark_other(s->c);
    int rc = s->tr->exchange(plan, provide, accept, st);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));
    return FFQ_OK;
}

// This rank's range of a FILE: bytes [lo - tail, hi + head) of fd (bounds are file offsets) into d_ext -- pread by the
// context's helper threads into pinned slots, over the link on two copy streams (stage_fd2d).  What the reference's
// single reader does for the whole stream (/root/reference/src/fastqandfurious.py:30-36 read(), :241-245 the first fill)
// happens here once per rank, for its range; the carry of :274-279 is the look-ahead, which comes from the file as well.
extern "C" int ffq_shard_load_fd(ffq_shard *s, int fd, uint8_t *d_ext, int64_t *n_bytes)
{
    if (!s) return fail(FFQ_E_ARG, "ffq_shard_load_fd: NULL shard");
    if (s->pending) return fail(FFQ_E_ARG, "ffq_shard_load_fd: a step is pending on this shard");
    if (fd < 0) { s->fd = -1; s->file_ext = nullptr; if (n_bytes) *n_bytes = 0; return FFQ_OK; }      // back to hand-offs between ranks
    if (!d_ext) return fail(FFQ_E_ARG, "ffq_shard_load_fd: NULL buffer");
    HIPCHK(hipSetDevice(s->c->device));
    int64_t tail, head;
    sh_halo_sizes(s->B, s->rank, s->tail_bytes, s->head_bytes, &tail, &head);
    const int64_t n = tail + (s->hi - s->lo) + head;
    int64_t got = 0;
    const int rc = stage_fd2d(s->c, d_ext, fd, s->lo - tail, n, &got);
    if (rc) return rc;
    if (got != n) return fail(FFQ_E_ARG, "ffq_shard_load_fd: the file ends at byte %lld, the bounds say %lld", (long long)(s->lo - tail + got), (long long)s->total);
    s->fd = fd; s->file_ext = d_ext;
    if (n_bytes) *n_bytes = n;
    return FFQ_OK;
}

// ---- a rank's range of a file that does NOT fit its GPU: the same step over SLABS ----------------------------------------
// What the reference's loop does for any size of stream -- scan the buffer, keep buf[offset:] (the unfinished entry), read
// more (/root/reference/src/fastqandfurious.py:251-279, the carry at :274-279) -- happens INSIDE the rank: its view
// [lo - tail, hi + head) goes through ONE device buffer of slab_bytes, slab after slab; slab k + 1 begins at the byte the
// search of slab k stopped at (the iterator's `offset`: the chain's exit of slab k is the entry of slab k + 1, exactly --
// no guess between slabs), its rows go behind slab k's in the caller's table, and the bytes are dropped.  At the rank's two
// EDGES nothing changes: the entry is a guess from the run-in, the eight words are what sh_words_from makes of the rows and
// of the last slab's end, one gather, sh_decide; a rank whose guess its left neighbour's chain contradicts streams its range
// again from that neighbour's exit (rare: "tricky" input); a look-ahead that must grow is simply read on (the file is right
// there).  The scan is ~100 x faster than the load, so the slabs are loaded and scanned in turn, not overlapped.
// No decode here: the qualities of a range that does not fit would not fit either.
static int slab_pass(ffq_shard *s, int fd, int64_t slab_bytes, uint32_t flags, int64_t *d_table, int64_t table_cap,
                     int64_t start, int64_t *head_io, ffq_scan_result *last, int64_t *n_rows_out, int64_t *n_slabs, int64_t *n_loaded)
{
    ffq_ctx *c = s->c;
    const int64_t origin = s->origin, total = s->total, lo = s->lo, hi = s->hi;
    int64_t tail, head0;
    sh_halo_sizes(s->B, s->rank, s->tail_bytes, s->head_bytes, &tail, &head0);
    int64_t head = std::max(*head_io, head0);
    const int64_t vstart = lo - tail;
    // P: the stream offset the next search starts at (the iterator's sentinel sits at origin - 1)
    int64_t P = start < 0 ? vstart - (vstart == origin ? 1 : 0) : std::max(start, vstart);
    int64_t nrows = 0;
    memset(last, 0, sizeof *last);
    last->end_state = FFQ_END_REFILL; last->last_status = FFQ_POS_HEAD_BEG;
    for (int i = 0; i < 6; i++) last->last_pos[i] = -1;
    for (;;) {
        const int64_t E = hi + head;                                   // the view's end for now
        // the buffer begins at the byte the search starts at; one that begins the stream has the sentinel in front
        const int64_t bstart = std::max(origin, P);
        const int sent = bstart == origin ? 1 : 0;
        const int64_t want = std::min(slab_bytes, E - bstart);
        if (want <= 0) break;
        int64_t got = 0;
        int rc = stage_fd2d(c, s->slab, fd, bstart, want, &got);
        if (rc) return rc;
        if (got != want) return fail(FFQ_E_ARG, "ffq_shard: the file ends at byte %lld, its bounds say %lld", (long long)(bstart + got), (long long)total);
        *n_loaded += got; (*n_slabs)++;
        const int64_t bend = bstart + got;
        const int64_t offset = P - bstart + sent;
        ffq_scan_result r;
        rc = ffq_scan_device(c, s->slab, got, sent, offset, bend == total ? 1 : 0, bstart - sent, flags & ~(uint32_t)(FFQ_F_DECODE_QUAL | FFQ_F_SINGLE_PASS),
                             0, d_table + nrows * 6, std::max<int64_t>(table_cap - nrows, 0), nullptr, 0, nullptr, &r);
        if (rc == FFQ_E_TABLE_FULL) {
            // the rows the whole view will need, from the rows per byte so far
            nrows += r.n_records;
            *n_rows_out = (int64_t)((double)nrows * ((double)(E - vstart) / (double)std::max<int64_t>(bend - vstart, 1)) * 1.05) + 1024;
            *last = r;
            return FFQ_E_TABLE_FULL;
        }
        if (rc) return rc;
        nrows += r.n_records;
        *last = r;
        last->end_offset = (bstart + r.end_offset - sent);             // (as a STREAM offset: the byte the next search starts at)
        if (r.end_state != FFQ_END_REFILL) break;                      // the stream's end, or its error
        const int64_t Pn = bstart + r.end_offset - sent;
        if (bend >= E) {
            // the view is through.  Does the chain know its exit?  (the first record start at / behind hi: a row, or the entry
            // the search stopped at)  If not, the record that straddles the edge is longer than the look-ahead: read on.
            const bool inc = r.last_status != FFQ_POS_HEAD_BEG && r.last_pos[0] >= 0;
            int64_t last_p0 = -1;
            if (nrows > 0 && nrows <= table_cap) {
                HIPCHK(hipMemcpyAsync(c->h_word, d_table + (nrows - 1) * 6, 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                last_p0 = c->h_word[0];
            }
            if (hi >= total || last_p0 >= hi || (inc && r.last_pos[0] >= hi) || E >= total) break;
            ShView v = sh_make_view(lo, hi, total, origin, tail, head);
            head = sh_more_head(v, s->head_bytes);
            P = Pn;
            continue;
        }
        if (Pn <= P && got >= slab_bytes) {
            // no record ended inside a whole slab: one record is longer than the slab -- a larger one
            const int64_t cap = 2 * slab_bytes;
            uint8_t *g = nullptr;
            HIPCHK(hipStreamSynchronize(c->stream));
            if (hipMalloc((void **)&g, (size_t)cap + 64) != hipSuccess) return fail(FFQ_E_NOMEM, "ffq_shard: a record longer than the slab (%lld bytes) and no memory for a larger one", (long long)slab_bytes);
            (void)hipFree(s->slab);
            s->slab = g; s->slab_cap = cap; slab_bytes = cap;
        }
        P = Pn;
    }
    *head_io = head;
    *n_rows_out = nrows;
    return FFQ_OK;
}

extern "C" int ffq_shard_scan_fd_slabs(ffq_shard *s, int fd, int64_t slab_bytes, uint32_t flags, int64_t *d_table, int64_t table_cap,
                                       ffq_shard_result *out)
{
    if (!s || !out || fd < 0 || (table_cap > 0 && !d_table)) return fail(FFQ_E_ARG, "ffq_shard_scan_fd_slabs: bad argument");
    if (s->pending) return fail(FFQ_E_ARG, "ffq_shard_scan_fd_slabs: a step is pending on this shard");
    if (s->tr->poisoned) return fail(FFQ_E_ARG, "ffq_shard_scan_fd_slabs: this shard's last step did not come back (watchdog)");
    if (flags & FFQ_F_DECODE_QUAL) return fail(FFQ_E_ARG, "ffq_shard_scan_fd_slabs: no decode over slabs (the qualities of a range that does not fit would not fit either)");
    ffq_ctx *c = s->c;
    HIPCHK(hipSetDevice(c->device));
    memset(out, 0, sizeof *out);
    slab_bytes = std::max<int64_t>((slab_bytes + 15) & ~(int64_t)15, 1 << 16);
    if (s->slab_cap < slab_bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        (void)hipFree(s->slab); s->slab = nullptr; s->slab_cap = 0;
        if (hipMalloc((void **)&s->slab, (size_t)slab_bytes + 64) != hipSuccess) return fail(FFQ_E_NOMEM, "ffq_shard: no memory for a slab of %lld bytes", (long long)slab_bytes);
        s->slab_cap = slab_bytes;
    }
    int rc = scan_reserve(c, s->slab_cap, flags, table_cap, 0);
    if (rc) return rc;
    const int W = s->world, rank = s->rank;
    const std::vector<int64_t> &B = s->B;
    int64_t tail, head;
    sh_halo_sizes(B, rank, s->tail_bytes, s->head_bytes, &tail, &head);
    out->nranks = s->tr->nranks(s->tr->serial ? 0 : 1);
    out->serial = s->tr->serial ? 1 : 0;
    s->handoff_bytes = 0; s->handoff_timed = false; s->local_fail = 0; s->last_stage = FFQ_SHARD_STAGE_NONE;
    s->flags = flags; s->d_table = d_table; s->table_cap = table_cap; s->ext = nullptr; s->from_file = true;
    int64_t n_slabs = 0, n_loaded = 0, nrows = 0, start = -1;
    int64_t *h = s->h_own;
    int64_t i0 = 0, i1 = 0;
    // one streaming pass of my view from `start` (< 0: the view's beginning, a guess) and its eight words
    auto pass = [&](int64_t st) -> int {
        start = st;
        i0 = i1 = 0; nrows = 0;
        s->v = sh_view(s, tail, head);
        if (s->v.n_bytes == 0) { sh_words_empty(s->v, s->head_bytes, h); h[8] = h[9] = h[10] = 0; return FFQ_OK; }
        ffq_scan_result last;
        int r = slab_pass(s, fd, s->slab_cap, flags, d_table, table_cap, st, &head, &last, &nrows, &n_slabs, &n_loaded);
        out->scan = last;
        if (r == FFQ_E_TABLE_FULL) {
            const int64_t tf[SH_WORDS] = {SH_UNKNOWN, SH_UNKNOWN, 0, 0, head, SH_ERR_TABLE_FULL, nrows, 0};
            memcpy(h, tf, sizeof tf); h[8] = h[9] = h[10] = 0;
            return FFQ_OK;
        }
        if (r) { s->local_fail = r; s->local_msg = ffq_last_error(); sh_words_failed(s->v, h); h[8] = h[9] = h[10] = 0; return FFQ_OK; }
        s->v = sh_view(s, tail, head);                                 // (the look-ahead the pass ended with)
        int64_t cut[6];
        if ((r = ffq_table_cut(c, d_table, nrows, sh_lo_bound(s->v), sh_hi_bound(s->v), cut))) {     // (the peers hear of it in the gather)
            s->local_fail = r; s->local_msg = ffq_last_error(); sh_words_failed(s->v, h); h[8] = h[9] = h[10] = 0; return FFQ_OK;
        }
        ShScanFacts f;
        f.n = nrows; f.i0 = cut[0]; f.i1 = cut[1]; f.p_i0 = cut[2]; f.p_i1 = cut[3]; f.q1 = cut[5];
        f.end_state = last.end_state; f.last_status = last.last_status; f.last_pos0 = last.last_pos[0];
        f.end_offset = last.end_offset - s->v.add;                     // (sh_words_from adds v.add back: stream offsets)
        // (the pass reads its own look-ahead: a view that ends short of the stream ends in a refill, one that ends it in OK)
        if (!s->v.eof && f.end_state == FFQ_END_OK) f.end_state = FFQ_END_REFILL;
        const int64_t off0 = st < 0 ? 0 : std::max(st, s->v.start) - s->v.add;
        sh_words_from(s->v, f, off0, s->head_bytes, h);
        h[3] = 0;                                                      // (a look-ahead that had to grow was read on, above)
        i0 = cut[0]; i1 = cut[1];
        h[8] = i0; h[9] = i1; h[10] = nrows;
        return FFQ_OK;
    };
    if ((rc = pass(-1))) return rc;
    if ((rc = shard_gather_host_words(s))) return rc;
    int rounds = 0;
    float ms = 0;
    for (;;) {
        rc = shard_wait_mark(s, s->ev_g[1]);
        if (!rc) rc = s->tr->gather_finish(s->h_all, s->ev_g[1]);
        if (rc) { if (rc == FFQ_E_TIMEOUT) { s->tr->poisoned = true; if (!s->last_stage) s->last_stage = FFQ_SHARD_STAGE_GATHER; } return rc; }
        if (hipEventElapsedTime(&ms, s->ev_g[0], s->ev_g[1]) == hipSuccess) out->allgather_ms += ms;
        const int64_t *A = s->h_all;
        const ShRound d = sh_decide(A, W, rank, B, s->v);
        if (sh_debug()) {
            std::string line;
            char buf[256];
            for (int r = 0; r < W; r++) {
                snprintf(buf, sizeof buf, " [%d: exit %lld first %lld n %lld err %lld/%lld search %lld]", r, (long long)A[r * SH_WORDS], (long long)A[r * SH_WORDS + 1],
                         (long long)A[r * SH_WORDS + 2], (long long)A[r * SH_WORDS + 5], (long long)A[r * SH_WORDS + 6], (long long)A[r * SH_WORDS + 7]);
                line += buf;
            }
            fprintf(stderr, "[ffq shard %d/%d slabs] round %d kind %d slabs %lld rows %lld..%lld of %lld:%s\n", rank, W, rounds, (int)d.kind, (long long)n_slabs,
                    (long long)h[8], (long long)h[9], (long long)h[10], line.c_str());
        }
        if (d.kind == ShRound::TABLE_FULL) {
            out->scan.n_records = d.need;
            return fail(FFQ_E_TABLE_FULL, "rank %d: offset table too small (%lld records in its view)", d.who, (long long)d.need);
        }
        if (d.kind == ShRound::INTERNAL || d.kind == ShRound::QUAL_FULL || d.kind == ShRound::NOT_READY) {
            if (s->local_fail) return fail(s->local_fail, "%s", s->local_msg.c_str());
            return fail(FFQ_E_INTERNAL, "sharded scan: rank %d %s", d.who, d.what);
        }
        if (s->local_fail) return fail(s->local_fail, "%s", s->local_msg.c_str());
        if (d.kind == ShRound::STREAM_ERROR) { out->err_state = d.err_state; out->err_byte = d.err_byte; break; }
        if (d.kind == ShRound::SETTLED) break;
        if (++rounds > sh_max_rounds(W)) return fail(FFQ_E_INTERNAL, "sharded scan does not settle (%d rounds)", rounds);
        if (d.i_force) {
            if (d.passed_over) {
                sh_words_passed_over(s->v, d.prev_exit, d.prev_search, h);
                h[8] = h[9] = h[10] = 0;
                i0 = i1 = 0; nrows = 0;
                start = d.prev_search;
            } else if ((rc = pass(d.prev_search))) return rc;           // my range again, from the left neighbour's exit
        }
        if ((rc = shard_gather_host_words(s))) return rc;               // (mine stand, or are new: the others' rounds need them)
    }
    out->n_rows = h[10]; out->row_lo = h[8]; out->row_hi = h[9];
    out->exit_pos = h[0]; out->first_pos = h[1];
    out->n_own_records = h[9] - h[8];
    int64_t base = 0, tot = 0;
    for (int r = 0; r < W; r++) {
        const int64_t cnt = s->h_all[(size_t)r * SH_WORDS + 2];
        if (r < rank) base += cnt;
        tot += cnt;
    }
    out->record_base = base; out->total_records = tot;
    out->rounds = rounds;
    out->halo_source = 1;
    out->handoff_bytes = 0;
    out->n_slabs = n_slabs; out->bytes_read = n_loaded;
    out->d_ext = nullptr; out->tail = tail; out->head = head;
    return FFQ_OK;
}

extern "C" const char *ffq_shard_transport(ffq_shard *s) { return s && s->tr ? s->tr->name() : ""; }

extern "C" int ffq_shard_get_info(ffq_shard *s, ffq_shard_info *out)
{
    if (!s || !out) return fail(FFQ_E_ARG, "ffq_shard_get_info: NULL argument");
    memset(out, 0, sizeof *out);
    ShTransport *t = s->tr;
    out->rank = s->rank; out->world = s->world;
    out->nranks_handoff = t->nranks(0); out->nranks_gather = t->nranks(1);
    out->serial = t->serial ? 1 : 0;
    out->last_stage = s->last_stage;
    if (!out->last_stage) if (ShLocal *l = dynamic_cast<ShLocal *>(t)) out->last_stage = l->last_stage;
    out->poisoned = t->poisoned ? 1 : 0;
    out->timeout_s = t->timeout_s;
    out->n_bus = std::min<int>(s->world, FFQ_SHARD_MAX_INFO_RANKS);
    for (int r = 0; r < FFQ_SHARD_MAX_INFO_RANKS; r++) out->bus_id[r] = (r < out->n_bus && r < (int)t->bus.size()) ? t->bus[(size_t)r] : -1;
    return FFQ_OK;
}

extern "C" int ffq_shard_set_timeout(ffq_shard *s, double seconds)
{
    if (!s || !(seconds >= 0)) return fail(FFQ_E_ARG, "ffq_shard_set_timeout: bad argument");
    s->tr->timeout_s = seconds;
    return FFQ_OK;
}

extern "C" int ffq_shard_set_serial(ffq_shard *s, int on)
{
    if (!s) return fail(FFQ_E_ARG, "ffq_shard_set_serial: NULL shard");
    if (s->pending) return fail(FFQ_E_ARG, "ffq_shard_set_serial: a step is pending on this shard");
    if (s->tr->poisoned) return fail(FFQ_E_ARG, "ffq_shard_set_serial: this shard's last step did not come back: build a new one (FFQ_SHARD_SERIAL=1)");
    if (!on && s->tr->nranks(1) == 0) return fail(FFQ_E_ARG, "ffq_shard_set_serial: created serial: there is no second communicator to go back to");
    // (what the other mode's streams still hold must be through before the first step of this one)
    HIPCHK(hipSetDevice(s->c->device));
    if (s->comm) HIPCHK(hipStreamSynchronize(s->comm));
    if (s->gstream) HIPCHK(hipStreamSynchronize(s->gstream));
    HIPCHK(hipStreamSynchronize(s->c->stream));
    s->tr->serial = on != 0;
    return FFQ_OK;
}

extern "C" int ffq_shard_inject_stall(ffq_shard *s, int stage, double seconds)
{
    if (!s || stage < FFQ_SHARD_STAGE_NONE || stage > FFQ_SHARD_STAGE_GATHER || !(seconds >= 0) || seconds > 120)
        return fail(FFQ_E_ARG, "ffq_shard_inject_stall: bad argument (stage 0..3, at most 120 s)");
    s->stall_stage = stage; s->stall_s = seconds;
    return FFQ_OK;
}
