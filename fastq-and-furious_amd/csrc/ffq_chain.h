// ffq_chain.h -- the record chain over the line index, wave-centric kernels.
//
// What is computed: the chain of readfastq_iter
//   /root/reference/src/fastqandfurious.py:251-279
// (search "\n@" from offset; scanner call; offset = pos5 - 1; repeat) with the C
// extension's scanner (/root/reference/src/_fastqandfurious.c:25-153).
//
// A "group" is OWN_T consecutive line-index tiles.  ONE WAVE per group (64-lane
// workgroups: no workgroup barriers, every sync is wave-local):
//   1. loads the line index of a window [run-in tile | OWN_T own tiles |
//      look-ahead tile] into LDS (a few hundred entries);
//   2. makes every "\n@" match of the run-in tail and the own tiles a NODE and
//      computes that node's scanner call and successor (thread per node);
//   3. follows the successor links from the window's earliest node by pointer
//      doubling with bottom-up marking: marked nodes = the chain, with ranks;
//   4. stages the chain's records of the own tiles as 16-byte group-relative
//      tuples and writes a per-group summary (entry candidate, exit candidate,
//      record count, quality bytes).
//
// Speculation.  For group 0 the chain start is exact.  For g > 0 the chain is
// started at the earliest candidate of the run-in (the last RUNIN_BYTES of the
// previous tile): a chain started at a false '@' candidate (a quality line that
// begins with '@') re-synchronises with the true chain within a few records.
// k_resolve_* then checks y[g+1] == exit[g] for every group up to the one the
// chain ends in; with group 0 exact this proves every guess by induction.  On
// any mismatch the serial walker redoes the buffer (still on the GPU).
//
//   k_chain_wave   steps 1-4                                  (latency-bound)
//   k_resolve_a/b  verification + exclusive scan of counts    (tiny)
//   k_expand       staged tuples -> int64[n][6] rows (+ quality CSR offsets)
//                  48 B written per record, coalesced through LDS
#pragma once
#include "ffq_dev.h"

namespace ffq {

constexpr int OWN_T = 2;                   // own tiles per group
constexpr int NTW = OWN_T + 2;             // + run-in tile + look-ahead tile
constexpr int RUNIN_BYTES = 8192;          // tail of the previous tile used as run-in
// LDS entry word: window-relative position (17 bits) | flags << 17 | node id << 19
constexpr uint32_t WP_MASK = 0x1FFFFu;
constexpr int WF_SHIFT = 17;
constexpr int WN_SHIFT = 19;
constexpr uint32_t WN_MASK = 0x3FFu;
constexpr uint32_t NO_NODE = 0x3FFu;
constexpr uint16_t NX_OUT = 0xFFFF, NX_NOCAND = 0xFFFE, NX_STOP = 0xFFFD;
constexpr uint16_t NM_EXT = 0xFFFF;
constexpr uint16_t UNMARKED = 0xFFFF;

constexpr int64_t Y_NOCAND = -1, X_END_TERM = -2, Y_UNRES = -3, X_END_FINAL = -4;

constexpr int RES_BLOCK = 1024;            // groups per k_resolve_a workgroup

struct StageRec { uint32_t p0, p1, p3, p4; };     // relative to the group's window origin

struct GroupTerm {
    int64_t pos[6];
    int32_t status;
    int32_t pad;
};

struct ChainBufs {
    int64_t *y;          // [ng] entry candidate ('\n' buffer coordinate) or Y_*
    int64_t *exit;       // [ng] first chain candidate past the own tiles, or Y_NOCAND / X_END_*
    uint32_t *cnt;       // [ng] records of the own tiles
    uint32_t *flags;     // [ng] bit0: irregular (does not fit this kernel's LDS budget)
    uint32_t *lines;     // [ng] newlines in the own tiles
    int64_t *qb;         // [ng] quality bytes of those records
    GroupTerm *term;     // [ng] scanner status/posbuffer where the chain stops (only if it does)
    StageRec *stage;     // [ng][nmax]
    int64_t *rloc;       // [ng] exclusive prefix of cnt inside the resolve block
    int64_t *qloc;       // [ng] same for qb
    int64_t *part;       // [nblk][4] block totals (cnt, qb, lines, -) -> exclusive prefixes
    int32_t *mins;       // [2] first terminating group, first bad group
    int32_t nmax;
    int32_t ng;
};

struct DevRes {
    int64_t n_records, n_qual_bytes, n_lines, end_offset;
    int64_t last_pos[6];
    int32_t last_status, end_state, fallback, term_group;
    int32_t has_final, pad;
};

// window accessor: flat LDS index while inside the window, global index beyond
struct WH {
    int32_t idx;     // >= 0: window entry; -1: use g; -2: before the window's first entry
    H g;
};
struct WAcc {
    typedef WH Hd;
    const LineIndex &L;
    const uint32_t *went;
    int32_t nwin;
    int32_t wt1;          // first tile after the window
    int64_t wpos0;        // buffer coordinate of window-relative position 0
    __device__ WAcc(const LineIndex &l, const uint32_t *we, int32_t nw, int32_t t1, int64_t p0)
        : L(l), went(we), nwin(nw), wt1(t1), wpos0(p0) {}
    __device__ bool next(Hd &h) const {
        if (h.idx != -1) {
            const int32_t j = (h.idx == -2) ? 0 : h.idx + 1;
            if (j < nwin) { h.idx = j; return true; }
            h.idx = -1;
            h.g = H{wt1 - 1, 0x7FFFFFF0};     // "after the last entry of tile wt1-1"
        }
        if (h.g.tile >= 0 && h.g.i != 0x7FFFFFF0 && h.g.i + 1 < (int32_t)L.cnt[h.g.tile]) {
            h.g.i++;
            return true;
        }
        int32_t t = h.g.tile + 1;
        if (t < 0) t = 0;
        while (t < L.ntiles && L.cnt[t] == 0) t++;
        if (t >= L.ntiles) return false;
        h.g.tile = t; h.g.i = 0;
        return true;
    }
    __device__ void get(const Hd &h, int64_t &P, int &fl) const {
        if (h.idx >= 0) {
            const uint32_t e = went[h.idx];
            P = wpos0 + (int64_t)(e & WP_MASK);
            fl = (int)((e >> WF_SHIFT) & 3u);
            return;
        }
        GAcc(L).get(h.g, P, fl);
    }
};

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    return (uint32_t)__shfl((int)wave_incl_scan(v), 63);
}

template <int NMAX, int EMAX>
__global__ __launch_bounds__(64) void k_chain_wave(LineIndex L, int64_t offset, int eof, ChainBufs B)
{
    __shared__ uint32_t went[EMAX];
    __shared__ uint32_t qlen[NMAX];
    __shared__ uint16_t nidx[NMAX];      // node -> window entry index of its "\n@"
    __shared__ uint16_t nm[NMAX];        // node -> window entry index of its "\n+" (NM_EXT: recompute)
    __shared__ uint16_t nxtE[NMAX];      // node -> window entry index of the successor candidate / NX_*
    __shared__ uint16_t S[NMAX];         // pointer doubling: node reached
    __shared__ uint16_t cn[NMAX];        //                   steps taken
    __shared__ uint16_t dist[NMAX];      // rank along the chain (UNMARKED: not on it)
    __shared__ int8_t nstat[NMAX];

    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const int own0 = g * OWN_T;
    const int own1 = min(own0 + OWN_T, L.ntiles);
    const bool has_runin = own0 > 0;
    const int wt0 = has_runin ? own0 - 1 : 0;
    const int wt1 = min(own1 + 1, L.ntiles);
    const int nwt = wt1 - wt0;
    const int sent = (wt0 == 0 && L.s) ? 1 : 0;
    const int64_t wpos0 = (int64_t)wt0 << TILE_SHIFT;
    const int64_t len = L.len();

    // ---- window directory + first 256 entries of every tile, one memory round trip ----
    int tc[NTW];
    uint2 ev[NTW];
#pragma unroll
    for (int k = 0; k < NTW; k++) {
        tc[k] = (k < nwt) ? (int)L.cnt[wt0 + k] : 0;
        ev[k] = make_uint2(0, 0);
        if (k < nwt) ev[k] = *reinterpret_cast<const uint2 *>(L.ent + (int64_t)(wt0 + k) * SLOT + 4 * lane);
    }
    int tb[NTW + 1];
    tb[0] = sent;
    bool irregular = false;
    uint32_t lines = 0;
#pragma unroll
    for (int k = 0; k < NTW; k++) {
        tb[k + 1] = tb[k] + tc[k];
        if (tc[k] > SLOT) irregular = true;
        if (k < nwt && wt0 + k >= own0 && wt0 + k < own1) lines += (uint32_t)tc[k];
    }
    const int nwin = tb[NTW];
    if (nwin > EMAX) irregular = true;
    if (irregular) {
        if (lane == 0) {
            B.y[g] = Y_UNRES; B.exit[g] = Y_UNRES; B.cnt[g] = 0; B.qb[g] = 0;
            B.flags[g] = 1; B.lines[g] = lines;
        }
        return;
    }
    if (sent && lane == 0) {
        const uint8_t b0 = L.n > 0 ? L.d[0] : 0;
        const uint32_t fl = (b0 == '@') ? FL_AT : (b0 == '+') ? FL_PLUS : 0;
        went[0] = 0u | (fl << WF_SHIFT) | (NO_NODE << WN_SHIFT);
    }
    int below = 0;      // entries of the run-in tile in front of its last RUNIN_BYTES
#pragma unroll
    for (int k = 0; k < NTW; k++) {
        const int c = tc[k];
        const uint32_t relb = (uint32_t)(k << TILE_SHIFT) + (uint32_t)L.s;
        for (int j0 = 0; j0 < c; j0 += 256) {
            uint2 v = ev[k];
            if (j0 > 0) v = *reinterpret_cast<const uint2 *>(L.ent + (int64_t)(wt0 + k) * SLOT + j0 + 4 * lane);
            const uint32_t x[4] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = j0 + 4 * lane + i;
                const bool ok = j < c;
                if (ok) went[tb[k] + j] = (relb + (x[i] & OFF_MASK)) | ((x[i] >> 14) << WF_SHIFT) | (NO_NODE << WN_SHIFT);
                if (k == 0 && has_runin)
                    below += __popcll(__ballot(ok && (x[i] & OFF_MASK) < (uint32_t)(TILE - RUNIN_BYTES)));
            }
        }
    }
    const int lo_idx = has_runin ? tb[0] + below : 0;     // first entry that may become a node
    const int own_lo = has_runin ? tb[1] : 0;             // entry index boundaries of the own tiles
    int own_hi = 0;
#pragma unroll
    for (int k = 0; k <= NTW; k++)
        if (k == own1 - wt0) own_hi = tb[k];
    __syncthreads();

    // ---- nodes: the "\n@" matches of [lo_idx, own_hi) at >= offset ---------------------
    int ncomp = 0;
    for (int j0 = lo_idx; j0 < own_hi; j0 += 64) {
        const int j = j0 + lane;
        bool isc = false;
        uint32_t e = 0;
        if (j < own_hi) {
            e = went[j];
            isc = ((e >> WF_SHIFT) & FL_AT) && (wpos0 + (int64_t)(e & WP_MASK) >= offset);
        }
        const unsigned long long bal = __ballot(isc);
        const int r = ncomp + __popcll(bal & ((1ull << lane) - 1ull));
        if (isc && r < NMAX - 1) {
            nidx[r] = (uint16_t)j;
            went[j] = (e & ~(WN_MASK << WN_SHIFT)) | ((uint32_t)r << WN_SHIFT);
        }
        ncomp += __popcll(bal);
    }
    if (ncomp >= NMAX) {         // node id NMAX-1 == NO_NODE is reserved
        if (lane == 0) {
            B.y[g] = Y_UNRES; B.exit[g] = Y_UNRES; B.cnt[g] = 0; B.qb[g] = 0;
            B.flags[g] = 1; B.lines[g] = lines;
        }
        return;
    }
    __syncthreads();

    // ---- one scanner call + successor search per node ------------------------------------
    const WAcc acc(L, went, nwin, wt1, wpos0);
    for (int c = lane; c < ncomp; c += 64) {
        const int k = nidx[c];
        WH hk; hk.idx = k; hk.g = H{0, 0};
        WH hm, hm1;
        Rec r;
        compute_record(acc, hk, wpos0 + (int64_t)(went[k] & WP_MASK), len, eof, r, hm, hm1);
        nstat[c] = (int8_t)(r.final_ ? ST_FINAL : r.status);
        const bool emits = (r.status == ST_COMPLETE) || r.final_;
        qlen[c] = emits ? (uint32_t)(r.p5 - r.p4) : 0u;
        nm[c] = (emits && hm.idx >= 0 && hm1.idx >= 0) ? (uint16_t)hm.idx : NM_EXT;
        uint16_t nx = NX_STOP;
        if (r.status == ST_COMPLETE) {
            WH hs; int64_t Ps;
            if (find_cand(acc, hm1, r.p5 - 1, hs, Ps)) nx = (hs.idx >= 0) ? (uint16_t)hs.idx : NX_OUT;
            else nx = NX_NOCAND;
        }
        nxtE[c] = nx;
    }
    __syncthreads();

    // ---- the chain from the earliest node: pointer doubling with bottom-up marking -------
    // before round k the marked set is every chain node at distance < 2^k from the start;
    // a marked node whose 2^k-step jump is exact marks its target at distance + 2^k.
    int rounds = 1;
    while ((1 << rounds) < ncomp) rounds++;
    constexpr int PER = (NMAX + 63) / 64;
    int e0 = 0;                       // start node of the speculative chain
    int lastn = -1, ynode = -1;
    bool unresolved = false;
    for (int attempt = 0; attempt < 4 && ncomp > 0; attempt++) {
        for (int c = lane; c < ncomp; c += 64) {
            const uint16_t nx = nxtE[c];
            uint16_t s = (uint16_t)c;
            if (nx < NX_STOP && (int)nx < own_hi) {
                const uint32_t nid = (went[nx] >> WN_SHIFT) & WN_MASK;
                if (nid != NO_NODE) s = (uint16_t)nid;
            }
            S[c] = s;
            cn[c] = (s != c) ? 1 : 0;
            dist[c] = (c == e0) ? 0 : UNMARKED;
        }
        __syncthreads();
        for (int k = 0; k < rounds; k++) {
            uint16_t s1[PER], s2[PER], c1[PER], c2[PER], dd[PER];
            bool mk[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int c = lane + u * 64;
                mk[u] = false;
                if (c < ncomp) {
                    s1[u] = S[c]; c1[u] = cn[c];
                    s2[u] = S[s1[u]]; c2[u] = cn[s1[u]];
                    dd[u] = dist[c];
                    mk[u] = (dd[u] != UNMARKED) && (c1[u] == (uint16_t)(1u << k));
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int c = lane + u * 64;
                if (c < ncomp) {
                    if (mk[u]) dist[s1[u]] = (uint16_t)(dd[u] + (1u << k));
                    S[c] = s2[u];
                    cn[c] = (uint16_t)(c1[u] + c2[u]);
                }
            }
            __syncthreads();
        }
        // last marked node and first marked node of the own tiles
        uint32_t lastkey = 0, ykey = 0xFFFFFFFFu;
        for (int c = lane; c < ncomp; c += 64) {
            const uint16_t dc = dist[c];
            if (dc == UNMARKED) continue;
            const uint32_t key = ((uint32_t)dc << 16) | (uint32_t)c;
            lastkey = max(lastkey, key);
            if ((int)nidx[c] >= own_lo) ykey = min(ykey, key);
        }
        lastkey = wave_max_u32(lastkey);
        ykey = wave_min_u32(ykey);
        lastn = (int)(lastkey & 0xFFFFu);
        ynode = (ykey == 0xFFFFFFFFu) ? -1 : (int)(ykey & 0xFFFFu);
        const bool died_in_runin = ((int)nidx[lastn] < own_lo) && (nstat[lastn] != ST_COMPLETE);
        if (!died_in_runin) break;
        // this chain stops inside the run-in (it started at a false candidate):
        // restart from the next run-in node it did not visit
        uint32_t nxt = 0xFFFFFFFFu;
        for (int c = lane; c < ncomp; c += 64)
            if (c > e0 && dist[c] == UNMARKED && (int)nidx[c] < own_lo) nxt = min(nxt, (uint32_t)c);
        nxt = wave_min_u32(nxt);
        __syncthreads();
        if (nxt == 0xFFFFFFFFu || attempt == 3) { unresolved = true; break; }
        e0 = (int)nxt;
    }

    // ---- summary (lane 0) + staging of the own tiles' records -------------------------------
    int64_t Y = Y_UNRES, EX = Y_UNRES;
    int32_t tstatus = 0;
    bool have_term = false;
    Rec tr;
    tr.p0 = tr.p1 = tr.p3 = tr.p4 = tr.p5 = -1; tr.status = 0; tr.final_ = false;
    if (lane == 0 && !unresolved) {
        if (ncomp == 0) {
            // no candidate in the run-in tail / own tiles: the chain passes over this group
            WH hb; hb.g = H{0, 0};
            hb.idx = (own_hi > 0) ? own_hi - 1 : -2;
            WH hs; int64_t Ps;
            Y = find_cand(acc, hb, offset, hs, Ps) ? Ps : Y_NOCAND;
            EX = Y;
            if (Y == Y_NOCAND) { have_term = true; tstatus = ST_HEAD_BEG; }
        } else {
            const int st = nstat[lastn];
            int64_t after = Y_NOCAND;      // the candidate the chain continues with after lastn
            if (st == ST_COMPLETE) {
                const uint16_t nx = nxtE[lastn];
                if (nx == NX_NOCAND) after = Y_NOCAND;
                else if (nx != NX_OUT) after = wpos0 + (int64_t)(went[nx] & WP_MASK);
                else {
                    const int k = nidx[lastn];
                    WH hk; hk.idx = k; hk.g = H{0, 0};
                    WH hm, hm1, hs; Rec r; int64_t Ps;
                    compute_record(acc, hk, wpos0 + (int64_t)(went[k] & WP_MASK), len, eof, r, hm, hm1);
                    after = find_cand(acc, hm1, r.p5 - 1, hs, Ps) ? Ps : Y_NOCAND;
                }
                EX = after;
                if (after == Y_NOCAND) { have_term = true; tstatus = ST_HEAD_BEG; }
            } else {
                // the chain stops at lastn: keep the scanner's posbuffer of that call
                const int k = nidx[lastn];
                WH hk; hk.idx = k; hk.g = H{0, 0};
                WH hm, hm1;
                compute_record(acc, hk, wpos0 + (int64_t)(went[k] & WP_MASK), len, eof, tr, hm, hm1);
                EX = (st == ST_FINAL) ? X_END_FINAL : X_END_TERM;
                have_term = true; tstatus = tr.status;
            }
            Y = (ynode >= 0) ? wpos0 + (int64_t)(went[nidx[ynode]] & WP_MASK) : EX;
            if (ynode < 0 && EX < 0 && EX != Y_NOCAND) Y = Y_UNRES;   // cannot happen: died in run-in
        }
    }
    // records of the own tiles, in chain order
    uint32_t cnt = 0;
    unsigned long long qsum = 0;
    bool bad_range = false;
    if (!unresolved && ynode >= 0) {
        const uint32_t d0 = dist[ynode];
        StageRec *st = B.stage + (int64_t)g * B.nmax;
        for (int c = lane; c < ncomp; c += 64) {
            const uint16_t dc = dist[c];
            if (dc == UNMARKED || (int)nidx[c] < own_lo) continue;
            const int s = nstat[c];
            if (s != ST_COMPLETE && s != ST_FINAL) continue;
            const int k = nidx[c];
            int64_t p0, p1, p3, p4;
            const uint16_t mi = nm[c];
            if (mi != NM_EXT) {
                p0 = (int64_t)(went[k] & WP_MASK) + 1;
                p1 = (int64_t)(went[k + 1] & WP_MASK);
                p3 = (int64_t)(went[mi] & WP_MASK);
                p4 = (int64_t)(went[mi + 1] & WP_MASK) + 1;
            } else {
                WH hk; hk.idx = k; hk.g = H{0, 0};
                WH hm, hm1; Rec r;
                compute_record(acc, hk, wpos0 + (int64_t)(went[k] & WP_MASK), len, eof, r, hm, hm1);
                p0 = r.p0 - wpos0; p1 = r.p1 - wpos0; p3 = r.p3 - wpos0; p4 = r.p4 - wpos0;
                if (p4 > 0xFFFFFFF0ll) bad_range = true;
            }
            StageRec o;
            o.p0 = (uint32_t)p0; o.p1 = (uint32_t)p1; o.p3 = (uint32_t)p3; o.p4 = (uint32_t)p4;
            st[dc - d0] = o;
            cnt++;
            qsum += qlen[c];
        }
    }
    cnt = wave_sum_u32(cnt);
    const uint32_t qlo = wave_sum_u32((uint32_t)(qsum & 0xFFFFFu)), qhi = wave_sum_u32((uint32_t)(qsum >> 20));
    const bool anybad = __ballot(bad_range) != 0ull;
    if (lane == 0) {
        const bool bad = unresolved || anybad;
        B.y[g] = bad ? Y_UNRES : Y;
        B.exit[g] = bad ? Y_UNRES : EX;
        B.cnt[g] = cnt;
        B.qb[g] = ((int64_t)qhi << 20) + (int64_t)qlo;
        B.flags[g] = anybad ? 1u : 0u;
        B.lines[g] = lines;
        if (have_term) {
            GroupTerm &t = B.term[g];
            t.status = tstatus;
            t.pos[0] = tr.p0; t.pos[1] = tr.p1; t.pos[2] = (tr.p1 >= 0) ? tr.p1 + 1 : -1;
            t.pos[3] = tr.p3; t.pos[4] = tr.p4; t.pos[5] = tr.p5;
        }
    }
}

// =========================================================================
// k_resolve_a: RES_BLOCK groups per workgroup.  Finds the first group the chain
// ends in and the first group whose guess is not confirmed by its predecessor's
// exit; exclusive-scans counts inside the block.
// k_resolve_b: one workgroup.  Scans the block totals, decides parallel vs
// serial, fills the result block (end state per fastqandfurious.py:256-279).
// =========================================================================
__global__ __launch_bounds__(RES_BLOCK) void k_resolve_a(ChainBufs B)
{
    __shared__ long long s_c[RES_BLOCK], s_q[RES_BLOCK];
    __shared__ int s_term, s_bad;
    __shared__ unsigned long long s_lines;
    const int tid = threadIdx.x;
    const int g = blockIdx.x * RES_BLOCK + tid;
    if (tid == 0) { s_term = 0x7FFFFFFF; s_bad = 0x7FFFFFFF; s_lines = 0; }
    __syncthreads();
    long long c = 0, q = 0;
    if (g < B.ng) {
        const int64_t y = B.y[g], ex = B.exit[g];
        c = B.cnt[g]; q = B.qb[g];
        if (ex == Y_NOCAND || ex == X_END_TERM || ex == X_END_FINAL) atomicMin(&s_term, g);
        bool bad = (B.flags[g] & 1u) || (y == Y_UNRES);
        if (g > 0) {
            const int64_t pe = B.exit[g - 1];
            if (pe >= 0 && y != pe) bad = true;
        }
        if (bad) atomicMin(&s_bad, g);
        atomicAdd(&s_lines, (unsigned long long)B.lines[g]);
    }
    s_c[tid] = c; s_q[tid] = q;
    __syncthreads();
    for (int d = 1; d < RES_BLOCK; d <<= 1) {
        long long v = 0, vq = 0;
        if (tid >= d) { v = s_c[tid - d]; vq = s_q[tid - d]; }
        __syncthreads();
        s_c[tid] += v; s_q[tid] += vq;
        __syncthreads();
    }
    if (g < B.ng) { B.rloc[g] = s_c[tid] - c; B.qloc[g] = s_q[tid] - q; }
    if (tid == RES_BLOCK - 1) {
        B.part[blockIdx.x * 4 + 0] = s_c[tid];
        B.part[blockIdx.x * 4 + 1] = s_q[tid];
    }
    if (tid == 0) {
        B.part[blockIdx.x * 4 + 2] = (long long)s_lines;
        if (s_term != 0x7FFFFFFF) atomicMin(&B.mins[0], s_term);
        if (s_bad != 0x7FFFFFFF) atomicMin(&B.mins[1], s_bad);
    }
}

__global__ __launch_bounds__(1024) void k_resolve_b(ChainBufs B, int nblk, int eof, int64_t offset,
                                                    int64_t add, DevRes *res)
{
    __shared__ long long s_c[1024], s_q[1024], s_l[1024];
    const int tid = threadIdx.x;
    // exclusive scan of the block totals (nblk <= 1024 per pass, carried across passes)
    long long carry_c = 0, carry_q = 0, carry_l = 0;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + tid;
        const long long c = (b < nblk) ? B.part[b * 4 + 0] : 0, q = (b < nblk) ? B.part[b * 4 + 1] : 0,
                        l = (b < nblk) ? B.part[b * 4 + 2] : 0;
        s_c[tid] = c; s_q[tid] = q; s_l[tid] = l;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            long long v = 0, vq = 0, vl = 0;
            if (tid >= d) { v = s_c[tid - d]; vq = s_q[tid - d]; vl = s_l[tid - d]; }
            __syncthreads();
            s_c[tid] += v; s_q[tid] += vq; s_l[tid] += vl;
            __syncthreads();
        }
        if (b < nblk) { B.part[b * 4 + 0] = carry_c + s_c[tid] - c; B.part[b * 4 + 1] = carry_q + s_q[tid] - q; }
        const long long tc = s_c[1023], tq = s_q[1023], tl = s_l[1023];
        __syncthreads();
        carry_c += tc; carry_q += tq; carry_l += tl;
    }
    if (tid == 0) {
        const int tterm = B.mins[0], tbad = B.mins[1];
        const bool fallback = (tterm >= B.ng) || (tbad <= tterm);     // mins start at 0x7F7F7F7F
        res->n_lines = carry_l;
        res->fallback = fallback ? 1 : 0;
        res->term_group = fallback ? -1 : tterm;
        res->end_offset = offset;
        res->has_final = 0;
        res->n_records = 0;
        res->n_qual_bytes = 0;
        if (!fallback) {
            res->n_records = B.part[(tterm / RES_BLOCK) * 4 + 0] + B.rloc[tterm] + B.cnt[tterm];
            res->n_qual_bytes = B.part[(tterm / RES_BLOCK) * 4 + 1] + B.qloc[tterm] + B.qb[tterm];
            const GroupTerm &tg = B.term[tterm];
            const int64_t ex = B.exit[tterm];
            const int st = tg.status;
            res->last_status = st;
            for (int i = 0; i < 6; i++) res->last_pos[i] = (tg.pos[i] >= 0) ? tg.pos[i] + add : -1;
            int end;
            if (ex == X_END_FINAL) { end = 0; res->has_final = 1; }
            else if (ex == Y_NOCAND) end = eof ? 0 : 1;
            else if (eof) end = (st == ST_QUAL_END) ? 2 : (st == ST_INVALID) ? 4 : 3;
            else end = (st == ST_INVALID) ? 4 : 1;
            res->end_state = end;
        }
    }
}

// =========================================================================
// k_expand: one wave per group.  Turns the staged group-relative tuples into the
// int64[n][6] rows (pos0..pos5 + add), rows written as whole 1 KiB lines through
// an LDS transpose; optional CSR offsets of the decoded qualities.
// Algorithmic traffic: 16 B read + 48 B (+ 8 B) written per record.
// =========================================================================
__global__ __launch_bounds__(64) void k_expand(ChainBufs B, const DevRes *__restrict__ res, int64_t add,
                                               int64_t *__restrict__ table, int64_t table_cap,
                                               int64_t *__restrict__ qoff)
{
    __shared__ __attribute__((aligned(16))) int64_t s_rows[64 * 6];
    const int g = blockIdx.x, lane = threadIdx.x;
    if (res->fallback || g > res->term_group) return;
    const uint32_t cnt = B.cnt[g];
    if (cnt == 0) return;
    const int64_t r0 = B.part[(g / RES_BLOCK) * 4 + 0] + B.rloc[g];
    int64_t q0 = B.part[(g / RES_BLOCK) * 4 + 1] + B.qloc[g];
    const int own0 = g * OWN_T;
    const int64_t base = ((int64_t)((own0 > 0) ? own0 - 1 : 0) << TILE_SHIFT) + add;
    const StageRec *st = B.stage + (int64_t)g * B.nmax;
    for (uint32_t d0 = 0; d0 < cnt; d0 += 64) {
        const uint32_t d = d0 + lane;
        const bool ok = d < cnt;
        StageRec r = {0, 0, 0, 0};
        if (ok) r = st[d];
        const int64_t p0 = base + r.p0, p1 = base + r.p1, p3 = base + r.p3, p4 = base + r.p4;
        const int64_t p5 = p4 + p3 - p1 - 1;
        if (qoff) {
            const uint32_t ql = ok ? (uint32_t)(p5 - p4) : 0u;
            // quality lengths are < 2^31; chunk sums of 64 fit 64 bits via two 32-bit scans
            const uint32_t lo = wave_incl_scan(ql & 0xFFFFu), hi = wave_incl_scan(ql >> 16);
            const int64_t incl = ((int64_t)hi << 16) + (int64_t)lo;
            if (ok && r0 + d < table_cap) qoff[r0 + d] = q0 + incl - ql;
            q0 += ((int64_t)__shfl((int)hi, 63) << 16) + (int64_t)(uint32_t)__shfl((int)lo, 63);
        }
        int64_t *mine = s_rows + lane * 6;
        mine[0] = p0; mine[1] = p1; mine[2] = p1 + 1; mine[3] = p3; mine[4] = p4; mine[5] = p5;
        __syncthreads();
        const uint32_t nrow = min(64u, cnt - d0);
        const int64_t rowbase = r0 + d0;
        // 48 B rows, 16-byte pieces: piece q of the chunk belongs to row q / 3
        const longlong2 *src = reinterpret_cast<const longlong2 *>(s_rows);
        longlong2 *dst = reinterpret_cast<longlong2 *>(table + rowbase * 6);
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const uint32_t q = lane + u * 64;
            if (q < nrow * 3 && rowbase + q / 3 < table_cap) dst[q] = src[q];
        }
        __syncthreads();
    }
}

}  // namespace ffq
