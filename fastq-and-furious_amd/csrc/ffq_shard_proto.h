// ffq_shard_proto.h -- the PROTOCOL of the byte-range shards, and nothing of a device: what a rank's scan says about its
// two edges (the eight words), what everybody's words mean for the next round (settled / an error of the stream / a
// look-ahead to grow / an entry to re-enter from the left neighbour's exit), who hands which bytes to whom, the barrier of
// the in-process world.  ONE statement of it: the device step (ffq_shard.h: HIP streams, RCCL) and the host step
// (ffq_shard_host.h: caller-supplied scan / exchange / gather, what the CPU tests drive over gloo) both run on these
// functions, so a change of the protocol lands once.  No HIP in here (tests/tsan_host_threads.cpp includes it for the
// barrier it runs its ranks over); under hipcc sh_words_from is also device code (k_shard_words calls it).
//
// What is sharded is the record chain of readfastq_iter (/root/reference/src/fastqandfurious.py:251-279): rank r owns the
// stream bytes [S_r, S_r+1) and every record whose '@' lies in them; what the reference does with a record that does not
// fit its buffer -- keep buf[offset:] and read more (:274-279) -- happens per range edge.
#pragma once
#include "../../include/ffq.h"

#include <stdint.h>

#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#define FFQ_HD __host__ __device__
#else
#define FFQ_HD
#endif

namespace ffq {

constexpr int64_t SH_NONE = -1;            // no record starts at / behind the bound: the view reaches the end of the stream
constexpr int64_t SH_UNKNOWN = -2;         // not known yet (more look-ahead needed, or the guessed entry led nowhere)
constexpr int64_t SH_ERR_TABLE_FULL = 100; // (beside the FFQ_END_ERR_* codes) the caller's table cannot hold the view's rows
constexpr int64_t SH_ERR_QUAL_FULL = 102;  // ... nor its quality buffer the view's decoded bytes
constexpr int64_t SH_NOT_READY = 101;      // this rank's scan needs a later tier (host round trip): gather again when it is through
constexpr int SH_WORDS = 8;                // exit, first, own count, look-ahead wanted, look-ahead had, error, error byte / rows needed, exit's search start

struct ShView {                            // a rank's [tail | own | head] buffer and its coordinates
    int64_t lo, hi, total, origin;         // my range, the stream's end, the offset of the stream's first byte
    int64_t tail, head;
    int64_t start, add, n_bytes;           // stream offset of ext[0]; what turns buffer coordinates into stream offsets
    int32_t sentinel, eof;
};

static inline ShView sh_make_view(int64_t lo, int64_t hi, int64_t total, int64_t origin, int64_t tail, int64_t head)
{
    ShView v;
    v.lo = lo; v.hi = hi; v.total = total; v.origin = origin;
    v.tail = tail; v.head = head;
    v.start = lo - tail;
    v.sentinel = v.start == origin ? 1 : 0;          // the iterator's b'\n' in front of the stream (:245)
    v.eof = (hi + head == total) ? 1 : 0;
    v.add = v.start - (v.sentinel ? 1 : 0);
    v.n_bytes = tail + (hi - lo) + head;
    return v;
}

// What one scan of a view says, as far as the words need it: n rows; i0 / i1 = the first rows whose pos0 is >= lo / >= hi
// (-inf for a range that starts the stream, +inf for one that ends it); pos0 of those rows; pos5 of the row in front of i1.
struct ShScanFacts {
    int64_t n, i0, i1, p_i0, p_i1, q1;
    int32_t end_state, last_status;
    int64_t last_pos0, end_offset;
};

FFQ_HD static inline int64_t sh_lo_bound(const ShView &v) { return v.lo == v.origin ? -(1ll << 62) : v.lo; }
FFQ_HD static inline int64_t sh_hi_bound(const ShView &v) { return v.hi == v.total ? (1ll << 62) : v.hi; }
FFQ_HD static inline int64_t sh_more_head(const ShView &v, int64_t head_bytes)
{
    int64_t want = 2 * v.head;
    if (want < head_bytes) want = head_bytes;
    if (want < 4096) want = 4096;
    const int64_t room = v.total - v.hi;
    return want < room ? want : room;
}

// The eight words of a scan that stood (no fallback pending, rows fit the table).  `offset`: buffer coordinate the scan's
// first search started at.
FFQ_HD static inline void sh_words_from(const ShView &v, const ShScanFacts &f, int64_t offset, int64_t head_bytes, int64_t w[SH_WORDS])
{
    const int64_t n = f.n;
    const int end = f.end_state;
    const int good = v.eof ? FFQ_END_OK : FFQ_END_REFILL;
    // the entry the chain stops at (incomplete / invalid): a record start like the rows'
    const bool have_inc = end != FFQ_END_OK && f.last_status != FFQ_POS_HEAD_BEG && f.last_pos0 >= 0;
    const int64_t p_inc = have_inc ? f.last_pos0 : 0;
    const int64_t unknown = (v.eof && end == FFQ_END_OK) ? SH_NONE : SH_UNKNOWN;
    // first: the first record start in my range; exit: the first at / behind my right edge
    int64_t first = (f.i0 < n) ? f.p_i0 : (have_inc && p_inc >= v.lo) ? p_inc : unknown;
    int64_t exitp = (v.hi < v.total) ? ((f.i1 < n) ? f.p_i1 : (have_inc && p_inc >= v.hi) ? p_inc : unknown) : SH_NONE;
    w[2] = f.i1 - f.i0;
    w[3] = 0; w[4] = v.head; w[5] = 0; w[6] = 0;
    // where the search that found the exit started (the iterator's `offset`, :254): the right neighbour re-enters there if
    // its own guess does not hold
    w[7] = (f.i1 < n) ? ((f.i1 > 0) ? f.q1 - 1 : offset + v.add) : f.end_offset + v.add;
    if (end == FFQ_END_ERR_FINAL_QUAL || end == FFQ_END_ERR_INCOMPLETE || end == FFQ_END_ERR_INVALID) {
        // a stream error: mine if the failing entry starts in my range (or nowhere: no entry at all)
        if (!have_inc || (v.lo <= p_inc && p_inc < v.hi) || (v.hi == v.total && p_inc >= v.lo)) {
            w[5] = end; w[6] = f.end_offset + v.add;
        } else if (p_inc < v.lo) first = exitp = SH_UNKNOWN;            // the guessed entry led nowhere
    } else if (end != good) w[5] = FFQ_E_INTERNAL;
    if (!v.eof && !w[5] && exitp == SH_UNKNOWN && !(have_inc && p_inc < v.lo) && end == FFQ_END_REFILL)
        w[3] = sh_more_head(v, head_bytes);        // the record that straddles my right edge does not end inside the look-ahead
    w[0] = exitp; w[1] = first;
}

// the words of an empty view (nothing to scan): no rows, the search "ended" at the view's start, more look-ahead wanted
// unless the view ends the stream
static inline void sh_words_empty(const ShView &v, int64_t head_bytes, int64_t w[SH_WORDS])
{
    const int64_t unknown = v.eof ? SH_NONE : SH_UNKNOWN;
    w[1] = unknown; w[0] = (v.hi < v.total) ? unknown : SH_NONE; w[2] = 0;
    w[3] = (!v.eof && w[0] == SH_UNKNOWN) ? sh_more_head(v, head_bytes) : 0;
    w[4] = v.head; w[5] = 0; w[6] = 0; w[7] = v.add;
}

// the words of a rank that learns that the chain passes over its whole range (or ends before it): it owns nothing
static inline void sh_words_passed_over(const ShView &v, int64_t prev_exit, int64_t prev_search, int64_t w[SH_WORDS])
{
    w[0] = prev_exit; w[1] = prev_exit; w[2] = 0; w[3] = 0; w[4] = v.head; w[5] = 0; w[6] = 0; w[7] = prev_search;
}

// ---- who hands which bytes to whom ------------------------------------------------------------------------------------
struct ShPiece { int src, dst; int64_t a, b; };          // stream bytes [a, b) go from rank src to rank dst

static inline void sh_range_plan(const std::vector<int64_t> &B, int dst, int64_t lo, int64_t hi, std::vector<ShPiece> &plan)
{
    for (int p = 0; p + 1 < (int)B.size(); p++) {
        const int64_t a = std::max(lo, B[p]), b = std::min(hi, B[p + 1]);
        if (a < b && p != dst) plan.push_back(ShPiece{p, dst, a, b});
    }
}

// (tail, head) of a rank's first scan: the same rule on every rank, so that each knows what the others need without asking
static inline void sh_halo_sizes(const std::vector<int64_t> &B, int rank, int64_t tail_bytes, int64_t head_bytes, int64_t *tail, int64_t *head)
{
    *tail = std::min(tail_bytes, B[rank] - B[0]);
    *head = std::min(head_bytes, B.back() - B[rank + 1]);
}

static inline void sh_halo_plan(const std::vector<int64_t> &B, int64_t tail_bytes, int64_t head_bytes, std::vector<ShPiece> &plan)
{
    for (int q = 0; q + 1 < (int)B.size(); q++) {
        int64_t t, h;
        sh_halo_sizes(B, q, tail_bytes, head_bytes, &t, &h);
        sh_range_plan(B, q, B[q] - t, B[q], plan);
        sh_range_plan(B, q, B[q + 1], B[q + 1] + h, plan);
    }
}

// ---- one look at everybody's words -------------------------------------------------------------------------------------
// exit[r] must equal first[r + 1]; rank 0's start is exact, so that proves every range by induction.  A rank whose
// look-ahead ends inside the record that straddles its edge asks for more and scans again; a rank whose guessed entry its
// left neighbour's chain contradicts scans again from that neighbour's exit; each round settles the first unsettled rank.
struct ShRound {
    enum Kind { SETTLED, STREAM_ERROR, TABLE_FULL, QUAL_FULL, INTERNAL, NOT_READY, REPAIR } kind = SETTLED;
    int who = -1;                        // TABLE_FULL / QUAL_FULL / INTERNAL: the rank
    int64_t need = 0;                    // TABLE_FULL: rows its view holds; QUAL_FULL: decoded bytes
    int32_t err_state = 0;               // STREAM_ERROR: FFQ_END_ERR_* and the byte the reference's ValueError names
    int64_t err_byte = 0;
    std::vector<int> grow, force;        // REPAIR: ranks that read more look-ahead, ranks that re-enter from the left
    bool i_grow = false, i_force = false;
    bool passed_over = false;            // i_force: the left neighbour's chain passes over my whole range
    int64_t prev_exit = 0, prev_search = 0;
    const char *what = "";
};

static inline ShRound sh_decide(const int64_t *A, int W, int rank, const std::vector<int64_t> &B, const ShView &mine)
{
    auto word = [&](int r, int k) { return A[(size_t)r * SH_WORDS + k]; };
    ShRound d;
    bool not_ready = false;
    for (int r = 0; r < W; r++) {
        if (word(r, 5) == SH_ERR_TABLE_FULL) { d.kind = ShRound::TABLE_FULL; d.who = r; d.need = word(r, 6); return d; }
        if (word(r, 5) == SH_ERR_QUAL_FULL) { d.kind = ShRound::QUAL_FULL; d.who = r; d.need = word(r, 6); return d; }
        if (word(r, 5) == FFQ_E_INTERNAL) { d.kind = ShRound::INTERNAL; d.who = r; d.what = "unexpected end state of its scan"; return d; }
        not_ready = not_ready || word(r, 5) == SH_NOT_READY;
    }
    if (not_ready) { d.kind = ShRound::NOT_READY; return d; }
    for (int r = 0; r < W; r++) if (word(r, 3) > 0) d.grow.push_back(r);
    for (int r = 1; r < W; r++)
        if (B[r] > B[0] && word(r - 1, 0) != SH_UNKNOWN && word(r, 1) != word(r - 1, 0)) d.force.push_back(r);
    if (d.grow.empty() && d.force.empty()) {
        for (int r = 0; r < W; r++) {
            if (word(r, 5)) {
                // The byte the iterator names is its `offset` when the failing search started: pos5 - 1 of the last COMPLETE
                // record in front of the failing entry (:254, :275).  A rank that owns no row in front of that entry does not
                // know it: the nearest rank to the left that owns a row (or rank 0, whose start is exact) has it as the
                // start of the search that found its exit.
                int64_t byte = word(r, 6);
                if (r > 0 && word(r, 2) == 0) {
                    int q = r - 1;
                    while (q > 0 && word(q, 2) == 0) q--;
                    byte = word(q, 7);
                }
                d.kind = ShRound::STREAM_ERROR; d.err_state = (int32_t)word(r, 5); d.err_byte = byte;
                return d;                           // (every rank reports the same error)
            }
            if (word(r, 0) == SH_UNKNOWN) { d.kind = ShRound::INTERNAL; d.who = r; d.what = "has no exit and nobody can move"; return d; }
        }
        d.kind = ShRound::SETTLED;
        return d;
    }
    d.kind = ShRound::REPAIR;
    d.i_grow = std::find(d.grow.begin(), d.grow.end(), rank) != d.grow.end();
    d.i_force = std::find(d.force.begin(), d.force.end(), rank) != d.force.end();
    if (d.i_force) {
        d.prev_exit = word(rank - 1, 0);
        d.prev_search = word(rank - 1, 7);
        d.passed_over = d.prev_exit == SH_NONE || (d.prev_exit >= mine.hi && mine.hi < mine.total);
    }
    return d;
}

// the bytes the growing ranks of a round need: [hi + had, hi + wanted) of each, from whoever owns them
static inline void sh_grow_plan(const int64_t *A, const std::vector<int64_t> &B, const std::vector<int> &grow, std::vector<ShPiece> &plan)
{
    for (int r : grow) sh_range_plan(B, r, B[r + 1] + A[(size_t)r * SH_WORDS + 4], B[r + 1] + A[(size_t)r * SH_WORDS + 3], plan);
}

static inline int sh_max_rounds(int W) { return 2 * W + 48; }

// the words of a rank whose own step failed (a scan error, no memory): sh_decide turns them into INTERNAL on EVERY rank, so
// that nobody waits in the next collective for a rank that has already gone home
static inline void sh_words_failed(const ShView &v, int64_t w[SH_WORDS])
{
    w[0] = SH_UNKNOWN; w[1] = SH_UNKNOWN; w[2] = 0; w[3] = 0; w[4] = v.head; w[5] = FFQ_E_INTERNAL; w[6] = 0; w[7] = 0;
}

// the step watchdog's default: FFQ_SHARD_TIMEOUT_S seconds without progress at one stage of a step (0: wait for ever)
static inline double sh_default_timeout()
{
    const char *e = getenv("FFQ_SHARD_TIMEOUT_S");
    if (e && *e) { const double t = atof(e); return t < 0 ? 0 : t; }
    return 30.0;
}

static inline const char *sh_stage_name(int stage)
{
    return stage == FFQ_SHARD_STAGE_HANDOFF ? "hand-off" : stage == FFQ_SHARD_STAGE_SCAN ? "scan" : stage == FFQ_SHARD_STAGE_GATHER ? "gather" : "none";
}

}  // namespace ffq

// k logical ranks as threads of ONE process: a barrier that can be broken (a rank that fails must not leave the others
// waiting), that gives up after a deadline (a rank that never arrives must not either: the watchdog of the in-process
// world -- whoever runs out of time breaks the barrier for everybody and says who was missing) and the slots their words
// meet in
struct ffq_shard_world {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t gen = 0;
    bool broken = false, timed_out = false;
    std::vector<char> here;                       // ranks that have arrived at the barrier now forming
    std::vector<int> absent;                      // after a time-out: the ranks that had not
    // 1: through; 0: broken (another rank failed, or ran out of time: timed_out says which); -1: THIS rank's wait ran out
    int wait_for(int rank, double seconds)
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return 0;
        if ((int)here.size() != world) here.assign((size_t)world, 0);
        if (rank >= 0 && rank < world) here[(size_t)rank] = 1;
        const uint64_t g = gen;
        if (++waiting == world) { waiting = 0; gen++; std::fill(here.begin(), here.end(), 0); cv.notify_all(); return 1; }
        auto pred = [&] { return gen != g || broken; };
        if (seconds > 0) {
            if (!cv.wait_for(lk, std::chrono::duration<double>(seconds), pred)) {
                absent.clear();
                for (int r = 0; r < world; r++) if (!here[(size_t)r]) absent.push_back(r);
                broken = true; timed_out = true;
                cv.notify_all();
                return -1;
            }
        } else cv.wait(lk, pred);
        return (gen != g) ? 1 : 0;               // (a barrier that completed stays completed even if it broke right after)
    }
    bool wait() { return wait_for(-1, 0) > 0; }
    void abort() { std::lock_guard<std::mutex> lk(m); broken = true; cv.notify_all(); }
    // "1, 3" -- the ranks a time-out found missing
    std::string absent_list()
    {
        std::lock_guard<std::mutex> lk(m);
        std::string s;
        for (int r : absent) { if (!s.empty()) s += ", "; s += std::to_string(r); }
        return s.empty() ? std::string("none") : s;
    }
    std::vector<const void *> providers;          // (per rank: what the transport parks for the others to read)
    std::vector<int64_t> slots;
};
