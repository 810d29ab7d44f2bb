// ffq_shard_host.h -- one step of the byte-range shards over HOST memory with caller-supplied transport (and, for tests,
// scan): the same protocol functions as the device step (ffq_shard_proto.h), driven synchronously.  Who calls it:
//   * the CPU test-suite: ranks are processes of a gloo group (world 2 / 3) or threads, the exchange / gather callbacks go
//     through torch.distributed, the scan callback is the test's own engine -- so the protocol the product runs is the one the
//     multi-process CPU tests exercise (there used to be a second, Python statement of it);
//   * a functional dry run of bench.py's N > 1 path on ONE GPU (FFQ_BENCH_DRY_MULTI: RCCL refuses two ranks per device):
//     scan == NULL means ffq_scan_host on the caller's context, the transport is gloo.
// It is NOT a CPU fallback of anything: without a scan callback it needs a context (a gfx950 device) like every other
// compute entry point.  Included at the end of ffq_hip.hip; tests/test_tsan.py runs it under ThreadSanitizer (k ranks as
// threads, a -fsanitize=thread build of the library's host code).
#pragma once
#include "ffq_shard_proto.h"

#include <cstdlib>
#include <cstring>
#include <string>

extern "C" void ffq_shard_host_free(void *p) { free(p); }

extern "C" int ffq_shard_host_step(const ffq_shard_host_ops *ops, ffq_ctx *ctx, int rank, int world, const int64_t *bounds,
                                   int64_t tail_bytes, int64_t head_bytes, uint8_t *h_ext, int64_t *h_table, int64_t table_cap,
                                   ffq_shard_result *out)
{
    using namespace ffq;
    if (!ops || !bounds || !out || !ops->allgather || (world > 1 && !ops->exchange)) return fail(FFQ_E_ARG, "ffq_shard_host_step: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(FFQ_E_ARG, "ffq_shard_host_step: rank %d of %d", rank, world);
    if (tail_bytes < 1 || head_bytes < 1) return fail(FFQ_E_ARG, "ffq_shard_host_step: tail_bytes and head_bytes must be at least 1");
    for (int r = 0; r < world; r++)
        if (bounds[r] > bounds[r + 1]) return fail(FFQ_E_ARG, "ffq_shard_host_step: bounds must not decrease");
    if (!ops->scan && !ctx) return fail(FFQ_E_ARG, "ffq_shard_host_step: neither a scan callback nor a context");
    memset(out, 0, sizeof *out);
    const std::vector<int64_t> B(bounds, bounds + world + 1);
    const int64_t lo = B[rank], hi = B[rank + 1], total = B[world], origin = B[0];
    int64_t tail, head;
    sh_halo_sizes(B, rank, tail_bytes, head_bytes, &tail, &head);
    ShView v = sh_make_view(lo, hi, total, origin, tail, head);
    if (v.n_bytes > 0 && (!h_ext || !h_table)) return fail(FFQ_E_ARG, "ffq_shard_host_step: NULL buffer");
    uint8_t *ext = h_ext, *grown = nullptr;
    int64_t handoff_bytes = 0;
    int rc = FFQ_OK;

    // one collective exchange: this rank provides its own bytes out of `src` (its range starts at src + src_tail) and receives
    // what the plan sends it into dst (stream offset dst_start at index 0)
    auto serve = [&](const std::vector<ShPiece> &plan, uint8_t *src, int64_t src_tail, uint8_t *dst, int64_t dst_start) -> int {
        std::vector<ffq_shard_piece> ps(plan.size());
        for (size_t i = 0; i < plan.size(); i++) {
            const ShPiece &p = plan[i];
            ps[i].src = p.src; ps[i].dst = p.dst; ps[i].a = p.a; ps[i].b = p.b; ps[i].ptr = nullptr;
            if (p.src == rank) { ps[i].ptr = src + src_tail + (p.a - lo); handoff_bytes += p.b - p.a; }
            if (p.dst == rank) { ps[i].ptr = dst + (p.a - dst_start); handoff_bytes += p.b - p.a; }
        }
        const int r = ops->exchange(ops->user, ps.data(), (int)ps.size());
        return r ? fail(r < 0 ? r : FFQ_E_INTERNAL, "ffq_shard_host_step: the exchange callback failed (%d)", r) : FFQ_OK;
    };
    if (world > 1) {
        std::vector<ShPiece> plan;
        sh_halo_plan(B, tail_bytes, head_bytes, plan);
        if ((rc = serve(plan, ext, tail, ext, v.start))) return rc;
    }

    int64_t w[SH_WORDS], row_lo = 0, row_hi = 0, nrows = 0, start = -1;
    // a scan of the current view from stream offset st (< 0: the view's beginning) and its words
    auto local = [&](int64_t st) -> int {
        start = st;
        row_lo = row_hi = nrows = 0;
        if (v.n_bytes == 0) { sh_words_empty(v, head_bytes, w); return FFQ_OK; }
        const int64_t offset = st < 0 ? 0 : std::max(st, v.start) - v.add;
        ffq_scan_result &res = out->scan;
        memset(&res, 0, sizeof res);
        int r;
        if (!ops->scan) r = ffq_scan_host(ctx, ext, v.n_bytes, v.sentinel, offset, v.eof, v.add, 0, 0, h_table, table_cap, nullptr, 0, nullptr, &res);
        else r = ops->scan(ops->user, ext, v.n_bytes, v.sentinel, offset, v.eof, v.add, h_table, table_cap, &res);
        if (r && r != FFQ_E_TABLE_FULL) return r < 0 ? r : fail(FFQ_E_INTERNAL, "ffq_shard_host_step: the scan callback failed (%d)", r);
        const int64_t n = res.n_records;
        if (r == FFQ_E_TABLE_FULL || n > table_cap) {
            const int64_t tf[SH_WORDS] = {SH_UNKNOWN, SH_UNKNOWN, 0, 0, v.head, SH_ERR_TABLE_FULL, n, 0};
            memcpy(w, tf, sizeof tf);
            return FFQ_OK;
        }
        auto lower = [&](int64_t value) {
            int64_t a = 0, b = n;
            while (a < b) { const int64_t m = (a + b) >> 1; if (h_table[m * 6] < value) a = m + 1; else b = m; }
            return a;
        };
        ShScanFacts f;
        f.n = n; f.i0 = lower(sh_lo_bound(v)); f.i1 = lower(sh_hi_bound(v));
        f.p_i0 = f.i0 < n ? h_table[f.i0 * 6] : -1;
        f.p_i1 = f.i1 < n ? h_table[f.i1 * 6] : -1;
        f.q1 = f.i1 > 0 ? h_table[(f.i1 - 1) * 6 + 5] : -1;
        f.end_state = res.end_state; f.last_status = res.last_status; f.last_pos0 = res.last_pos[0]; f.end_offset = res.end_offset;
        sh_words_from(v, f, offset, head_bytes, w);
        row_lo = f.i0; row_hi = f.i1; nrows = n;
        return FFQ_OK;
    };
    // A failure of THIS rank's scan (the callback's, ffq_scan_host's) must not leave the peers waiting in the next gather for a
    // rank that has gone home: its words say "failed" -- sh_decide makes that INTERNAL on every rank -- and it returns its
    // own error once everybody has seen them.
    int local_fail = 0;
    std::string local_msg;
    auto failed = [&](int code) { local_fail = code; local_msg = ffq_last_error(); sh_words_failed(v, w); row_lo = row_hi = nrows = 0; };
    if ((rc = local(-1))) failed(rc);

    std::vector<int64_t> all((size_t)world * SH_WORDS);
    int rounds = 0;
    for (;;) {
        const int r = ops->allgather(ops->user, w, all.data());
        if (r) { free(grown); return fail(r < 0 ? r : FFQ_E_INTERNAL, "ffq_shard_host_step: the gather callback failed (%d)", r); }
        const int64_t *A = all.data();
        const ShRound d = sh_decide(A, world, rank, B, v);
        if (local_fail) { free(grown); return fail(local_fail, "%s", local_msg.c_str()); }
        if (d.kind == ShRound::TABLE_FULL) {
            out->scan.n_records = d.need;
            free(grown);
            return fail(FFQ_E_TABLE_FULL, "rank %d: offset table too small (%lld records in its view)", d.who, (long long)d.need);
        }
        if (d.kind == ShRound::INTERNAL || d.kind == ShRound::NOT_READY || d.kind == ShRound::QUAL_FULL) { free(grown); return fail(FFQ_E_INTERNAL, "sharded scan: rank %d %s", d.who, d.what); }
        if (d.kind == ShRound::STREAM_ERROR) { out->err_state = d.err_state; out->err_byte = d.err_byte; break; }
        if (d.kind == ShRound::SETTLED) break;
        if (++rounds > sh_max_rounds(world)) { free(grown); return fail(FFQ_E_INTERNAL, "sharded scan does not settle (%d rounds)", rounds); }
        int64_t st = start;
        if (!d.grow.empty()) {
            std::vector<ShPiece> plan;
            sh_grow_plan(A, B, d.grow, plan);
            uint8_t *dst = ext, *old = nullptr;
            int64_t dst_start = v.start;
            ShView nv = v;
            if (d.i_grow) {
                // a view with more look-ahead (the caller's buffer has room for its own halo only)
                nv = sh_make_view(lo, hi, total, origin, v.tail, A[(size_t)rank * SH_WORDS + 3]);
                uint8_t *g = static_cast<uint8_t *>(malloc((size_t)nv.n_bytes + 64));
                if (!g) { free(grown); return fail(FFQ_E_NOMEM, "ffq_shard_host_step: no memory for a view of %lld bytes", (long long)nv.n_bytes); }
                memcpy(g, ext, (size_t)v.n_bytes);
                old = grown;                    // (still the source of what this rank provides in this round)
                grown = g; dst = g; dst_start = nv.start;
            }
            rc = serve(plan, ext, v.tail, dst, dst_start);
            free(old);
            if (rc) { free(grown); return rc; }
            if (d.i_grow) { ext = grown; v = nv; }
        }
        if (d.i_force) {
            if (d.passed_over) {
                // the chain passes over my whole range (or ends before it): I own nothing
                sh_words_passed_over(v, d.prev_exit, d.prev_search, w);
                row_lo = row_hi = nrows = 0;
                start = d.prev_search;
                continue;
            }
            st = d.prev_search;
        }
        if (d.i_grow || d.i_force) { if ((rc = local(st))) failed(rc); }
        // (else: my words stand; the others' rounds need them again)
    }
    out->n_rows = nrows; out->row_lo = row_lo; out->row_hi = row_hi;
    out->exit_pos = w[0]; out->first_pos = w[1];
    out->n_own_records = row_hi - row_lo;
    int64_t base = 0, tot = 0;
    for (int r = 0; r < world; r++) {
        const int64_t cnt = all[(size_t)r * SH_WORDS + 2];
        if (r < rank) base += cnt;
        tot += cnt;
    }
    out->record_base = base; out->total_records = tot;
    out->rounds = rounds;
    out->handoff_bytes = handoff_bytes;
    out->d_ext = ext; out->tail = v.tail; out->head = v.head;       // (ext != h_ext: a grown view; the caller frees it, ffq_shard_host_free)
    return FFQ_OK;
}
