// ffq_dense.h -- the record chain through DENSE regions of the line index.
//
// A tile with more than SLOT newlines (lines under 16 bytes on average: reads of a dozen bases with short
// headers -- the shape of the reference's own test template, /root/reference/tests.py:8-35 --, blocks of blank
// lines) keeps its entries in the overflow pool; a group of tiles with such a tile in its window does not fit
// the LDS budget of k_chain_wave (ffq_chain.h) and is WALKED here: one wave per group, from the entry the
// verification proves (a guess in the first pass, the predecessor's exit in the repair passes), to the first
// chain candidate past the group's own tiles.
//
// What is computed is the chain of /root/reference/src/fastqandfurious.py:251-279 with the scanner of
// /root/reference/src/_fastqandfurious.c:25-153, as everywhere.  The chain is sequential -- record i + 1 is
// searched from pos5(i) - 1 -- but through a dense region its records are SHORT, so the wave does not follow it
// call by call (that was k_group_walk until round 3: the walkers' wave-wide searches, ~2 us per record, 4-6 ms
// for one 64 KiB group of 30-byte records).  It takes the index a WINDOW of DW_W entries at a time:
//   1. the window's entries -> LDS (position relative to the window's first tile | flags | node id);
//   2. every "\n@" match of the window is a NODE; a thread per node makes that node's scanner call from the
//      DW_B entries that follow it (header end, the "\n+" match, the '+' line's end, the first "\n@" at or
//      behind pos5 - 1: its successor) -- the same rules in the same order as the C scanner; whatever does not
//      fit that (a record of more lines, the end of the buffer near, an INVALID '+' line) is marked generic;
//   3. the chain is followed from the window's first node RUN BY RUN: consecutive nodes whose successor is the
//      very next node are taken 64 at a time with one ballot, and their records staged by all lanes at once; a
//      node that ends a run (a false candidate skipped, a generic node) is one step;
//   4. a node too close to the window's end starts the next window; a generic node is one call of the
//      walkers' wave-wide scanner (wv_record_t / wv_find_t, ffq_dev.h) and the next window starts at the
//      candidate that call continues with.
// Records are staged as in k_chain_wave (16-byte group-relative tuples) in a chunk of DCHUNK records that the
// group takes from a second stage the first time it is walked (ChainBufs::sbase / dstage); k_expand turns them
// into rows.  The summary (entry candidate, exit candidate, count, quality bytes, terminal posbuffer) is what
// k_chain_wave writes, and k_resolve_* verifies it the same way.
#pragma once
#include "ffq_chain.h"

namespace ffq {

constexpr int DW_W = 1024;                 // index entries per window
constexpr int DW_B = 14;                   // entries a node's scanner call may look at, its own included
constexpr int DW_TILES = 32;               // tiles a window reaches over at most (positions stay below 2^20)
constexpr uint32_t DP_MASK = 0xFFFFFu;     // LDS word: window position (20 bits) | flags << 20 | node id << 22
constexpr int DF_SHIFT = 20, DN_SHIFT = 22;
constexpr uint32_t DK_OK = 0, DK_MORE = 1, DK_GEN = 2;      // node word: successor node (11 bits) | kind << 11 | mi << 13
static_assert(DW_TILES * TILE + 1 <= (int)DP_MASK, "window positions must fit DP_MASK");
static_assert(DW_W <= 1024, "node ids are 10 bits");

struct DwLds {
    __attribute__((aligned(16))) uint32_t went[DW_W + 16];
    uint32_t info[DW_W];
    uint16_t nidx[DW_W];
};

// index of the first entry of a tile (c entries at src, sorted by offset) whose offset is >= off; c if none.
// By the whole wave: 64 probes per round trip (a tile of 16384 entries: three).
__device__ __forceinline__ uint32_t dw_lower_bound(const uint16_t *__restrict__ src, uint32_t c, uint32_t off, int lane)
{
    uint32_t lo = 0, hi = c;               // the answer lies in [lo, hi]
    while (hi - lo > 64u) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t j = lo + step * (uint32_t)(lane + 1) - 1u;          // last entry of piece `lane`
        const bool ge = (j >= hi) || (((uint32_t)src[j] & OFF_MASK) >= off);
        const unsigned long long m = __ballot(ge);                          // (monotone; lane 63 is always set)
        const uint32_t f = (uint32_t)(__ffsll((long long)m) - 1);
        const uint32_t nlo = lo + step * f, nhi = min(lo + step * (f + 1u) - 1u, hi);
        lo = nlo; hi = nhi;
    }
    const uint32_t j = lo + (uint32_t)lane;
    const unsigned long long m = __ballot(j < hi && (((uint32_t)src[j] & OFF_MASK) >= off));
    return m ? lo + (uint32_t)(__ffsll((long long)m) - 1) : hi;
}

// where tile t keeps its entries (the slot, or the pool for a dense tile); nullptr: a dense tile whose entries
// the pool does not hold (the index kernel has flagged ERR_POOL: this scan is run again with a larger pool)
__device__ __forceinline__ const uint16_t *dw_tile_entries(const LineIndex &L, int t, uint32_t c)
{
    if (c <= (uint32_t)SLOT) return L.ent + (int64_t)t * SLOT;
    const unsigned long long at = L.ovf[t] & OVF_MASK;
    return (at + c <= L.pool_cap) ? L.pool + at : nullptr;
}

// (the walk of ONE group by one wave; k_dense_walk below calls it per wave, or for every group of the first pass's list)
__device__ __forceinline__ void dense_walk_group(const LineIndex &L, const ChainBufs &B, int64_t offset, int eof, int speculate,
                                                 int walk_all, Ctl *ctl, const int g, DwLds &sm)
{
    const int lane = threadIdx.x & 63;
    if (g >= B.ng || !(B.flags[g] & 1u)) return;
    const int64_t fpos = speculate ? FORCE_NONE : B.force[g];
    if (g > 0 && fpos == FORCE_NONE && !speculate) return;
    const int own0 = g * OWN_T, own1 = min(own0 + OWN_T, L.ntiles);
    if (!walk_all) {
        // usual configuration: only a group with a DENSE tile in its window is walked here; one
        // that merely exceeds this configuration's LDS budget (lines of ~32 bytes) is better off
        // with the dense configuration of k_chain_wave
        const int wt0 = own0 > 0 ? own0 - 1 : 0, wt1 = min(own1 + 1, L.ntiles);
        const bool dense_here = wt0 + lane < wt1 && L.cnt[wt0 + lane] > (uint32_t)SLOT;
        if (__ballot(dense_here) == 0ull) return;
    }
    uint32_t *went = sm.went, *info = sm.info;
    uint16_t *nidx = sm.nidx;
    const int64_t own_beg = ((int64_t)own0 << TILE_SHIFT) + L.s;        // coordinate of the own tiles' first byte
    const int64_t own_end = ((int64_t)own1 << TILE_SHIFT) + L.s;        // first coordinate past the own tiles
    const int64_t wpos0 = (int64_t)(own0 > 0 ? own0 - 1 : 0) << TILE_SHIFT;
    const int64_t len = L.len();
    // the group's chunk of the walked groups' stage
    int sb = B.sbase[g] - 1;
    if (sb < 0) {
        uint32_t got = 0;
        if (lane == 0) got = atomicAdd(B.dhead, 1u);
        sb = (int)__shfl((int)got, 0);
        if (sb >= B.dchunks) { if (lane == 0) atomicOr(&ctl->err, ERR_DSTAGE); return; }
        if (lane == 0) B.sbase[g] = sb + 1;
    }
    StageRec *stg = B.dstage + (int64_t)sb * DCHUNK;

    // entry: exact for group 0 (the scan's search offset) and in a repair pass (the predecessor's
    // exit); a GUESS in the first pass -- the first candidate of the run-in, as k_chain_wave does:
    // a chain started at a false candidate has the run-in to fall in with the true one, and the
    // verification (y[g] == exit[g-1]) decides.  Records of the run-in are walked, not staged.
    const bool guess = g > 0 && fpos == FORCE_NONE;
    const int64_t X = (g == 0) ? offset : guess ? max(own_beg - RUNIN_BYTES, offset) : fpos;
    // first index entry at coordinate >= X (coordinate = tile << TILE_SHIFT | offset, + L.s; the sentinel is coordinate 0)
    H start;
    if (L.s && X <= 0) start = H{-1, 0};
    else {
        const int64_t xr = X - L.s;
        const int64_t xt = xr >> TILE_SHIFT;
        if (xt >= (int64_t)L.ntiles) start = H{L.ntiles, 0};
        else {
            const uint32_t c = L.cnt[(int)xt];
            const uint16_t *src = dw_tile_entries(L, (int)xt, c);
            if (!src) return;
            start = H{(int)xt, (int32_t)dw_lower_bound(src, c, (uint32_t)(xr & OFF_MASK), lane)};
        }
    }

    Rec r;
    r.p0 = r.p1 = r.p3 = r.p4 = r.p5 = -1; r.status = ST_HEAD_BEG; r.final_ = false;
    uint32_t n = 0;
    unsigned long long qsum = 0;           // (per lane; summed at the end)
    int64_t Y = Y_UNRES, EX = Y_UNRES;
    bool have_term = false, first_window = true;

    for (;;) {                              // one window per turn
        // ---- 1. the window: up to DW_W entries from `start` on, over at most DW_TILES tiles -------------
        wave_sync();
        const int sent = start.tile < 0 ? 1 : 0;
        const int t0 = max(start.tile, 0);
        const uint32_t i0 = start.tile < 0 ? 0u : (uint32_t)start.i;
        const int64_t wbase = (int64_t)t0 << TILE_SHIFT;               // window position 0 (the sentinel's coordinate when t0 == 0)
        const bool tl = lane < DW_TILES && t0 + lane < L.ntiles;
        const uint32_t cl = tl ? L.cnt[t0 + lane] : 0u;
        const uint32_t av = cl - ((lane == 0) ? min(i0, cl) : 0u);
        const uint32_t incl = wave_incl_scan(av);
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        const uint32_t ex = incl - av + (uint32_t)sent;               // window index of this tile's first entry taken
        const uint32_t take = (ex < (uint32_t)DW_W) ? min(av, (uint32_t)DW_W - ex) : 0u;
        const int wn = (int)min((uint32_t)DW_W, (uint32_t)sent + total);
        const bool idx_end = (t0 + DW_TILES >= L.ntiles) && ((uint32_t)sent + total <= (uint32_t)DW_W);   // the index ends with the window
        if (sent && lane == 0) {
            const uint8_t b0 = L.n > 0 ? L.d[0] : 0;
            went[0] = ((b0 == '@') ? (uint32_t)FL_AT : (b0 == '+') ? (uint32_t)FL_PLUS : 0u) << DF_SHIFT;
        }
        {
            unsigned long long tm = __ballot(take > 0u);
            while (tm) {
                const int q = __ffsll((long long)tm) - 1;
                tm &= tm - 1ull;
                const uint32_t cq = (uint32_t)__shfl((int)cl, q), tk = (uint32_t)__shfl((int)take, q),
                               wq = (uint32_t)__shfl((int)ex, q);
                const uint32_t aq = (q == 0) ? i0 : 0u;
                const uint16_t *src = dw_tile_entries(L, t0 + q, cq);
                if (!src) return;
                const uint32_t relb = ((uint32_t)q << TILE_SHIFT) + (uint32_t)L.s;
                for (uint32_t j0 = 0; j0 < tk; j0 += 512u) {
                    const uint32_t j = j0 + 8u * (uint32_t)lane;
                    if (j >= tk) continue;
                    uint32_t x[8];
                    if (aq + j + 8u <= cq) {
                        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(2)));
                        const u32x4u v = *reinterpret_cast<const u32x4u *>(src + aq + j);
                        x[0] = v.x & 0xFFFFu; x[1] = v.x >> 16; x[2] = v.y & 0xFFFFu; x[3] = v.y >> 16;
                        x[4] = v.z & 0xFFFFu; x[5] = v.z >> 16; x[6] = v.w & 0xFFFFu; x[7] = v.w >> 16;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; e++) x[e] = (aq + j + (uint32_t)e < cq) ? (uint32_t)src[aq + j + e] : 0u;
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (j + (uint32_t)e < tk) went[wq + j + e] = (relb + (x[e] & OFF_MASK)) | ((x[e] >> 14) << DF_SHIFT);
                }
            }
        }
        wave_sync();
        // window entry index -> handle in the global index
        auto handle_of = [&](int e) -> H {
            if (sent && e == 0) return H{-1, 0};
            const unsigned long long m = __ballot(take > 0u && ex <= (uint32_t)e);
            const int q = 63 - __clzll((long long)m);
            return H{t0 + q, (int32_t)((uint32_t)e - (uint32_t)__shfl((int)ex, q) + ((q == 0) ? i0 : 0u))};
        };

        // ---- 2. nodes: the "\n@" matches of the window, numbered in entry order ---------------------------
        // n_ob / n_oe: nodes in front of the own tiles (a guess's run-in: walked, not staged) / in front of their end
        const int64_t ob_rel = guess ? own_beg - wbase : (int64_t)0, oe_rel = own_end - wbase;
        const uint32_t ob32 = (uint32_t)min(max(ob_rel, (int64_t)0), (int64_t)0x7FFFFFF0);
        const uint32_t oe32 = (uint32_t)min(max(oe_rel, (int64_t)0), (int64_t)0x7FFFFFF0);
        int nn = 0, n_ob = 0, n_oe = 0;
        for (int p = 0; p * 64 < wn; p++) {
            const int e = p * 64 + lane;
            const uint32_t w = (e < wn) ? went[e] : 0u;
            const bool at = ((w >> DF_SHIFT) & (uint32_t)FL_AT) != 0u;
            const unsigned long long m = __ballot(at);
            if (at) {
                const int id = nn + bits_below_lane(m);
                nidx[id] = (uint16_t)e;
                went[e] = w | ((uint32_t)id << DN_SHIFT);
            }
            nn += __popcll(m);
            n_ob += __popcll(__ballot(at && (w & DP_MASK) < ob32));
            n_oe += __popcll(__ballot(at && (w & DP_MASK) < oe32));
        }
        wave_sync();
        if (first_window && g > 0 && !guess) {
            // a repair pass enters the group AT the predecessor's exit: that must be the window's first candidate
            if (nn == 0 || wbase + (int64_t)(went[nidx[0]] & DP_MASK) != fpos) return;    // not a candidate: leave it to the later tiers
        }
        if (nn == 0) {
            // no candidate in the window (blank lines, lines of a long record): the next one by the wave-wide search,
            // which steps over dense tiles without a "\n@" on their ovf[] word alone
            H hs; int64_t Ps; int fls;
            const H from = (wn > 0) ? handle_of(wn - 1) : H{t0 + DW_TILES - 1, 0x7FFFFFF0};      // (behind the window's tiles)
            if (idx_end || !wv_find_t<true>(L, from, FL_AT, X, hs, Ps, fls)) {
                EX = Y_NOCAND; have_term = true; r.p0 = r.p1 = r.p3 = r.p4 = r.p5 = -1; r.status = ST_HEAD_BEG; r.final_ = false;
                break;
            }
            if (Ps >= own_end) { EX = Ps; break; }
            start = hs; first_window = false;
            continue;
        }

        // ---- 3. one scanner call per node, from the DW_B entries that follow it -----------------------------
        const int64_t lr64 = len - wbase;
        const uint32_t lenrel = (uint32_t)min(max(lr64, (int64_t)0), (int64_t)0x7FFFFFF0);
        for (int c0 = 0; c0 < nn; c0 += 64) {
            const int c = c0 + lane;
            if (c >= nn) continue;
            const int k = nidx[c];
            uint32_t inf = (idx_end ? DK_GEN : DK_MORE) << 11;
            if (k + DW_B <= wn) {
                inf = DK_GEN << 11;
                uint32_t w[DW_B];
#pragma unroll
                for (int i = 0; i < DW_B; i++) w[i] = went[k + i];
                const uint32_t P0 = w[0] & DP_MASK, P1 = w[1] & DP_MASK;
                if ((w[DW_B - 1] & DP_MASK) + 4u < lenrel) {          // every buffer-end rule of the scanner is out of reach
                    uint32_t pm = 0;                                  // "\n+" at >= seq_beg + 1 (:87-88)
#pragma unroll
                    for (int i = 2; i <= 7; i++)
                        if (((w[i] >> DF_SHIFT) & (uint32_t)FL_PLUS) && (w[i] & DP_MASK) >= P1 + 2u) pm |= 1u << i;
                    if (pm) {
                        const int mi = __ffs((int)pm) - 1;
                        const uint32_t P3 = went[k + mi] & DP_MASK, Pq = went[k + mi + 1] & DP_MASK;
                        const bool invalid = (Pq - P3 - 1u > 1u) && (Pq - P3 != P1 - P0);      // :109-117
                        const uint32_t qe = Pq + P3 - P1;                                       // :129 (pos4 + pos3 - pos2)
                        uint32_t am = 0;                              // the next call's "\n@" at >= pos5 - 1 (:62, fastqandfurious.py:254)
#pragma unroll
                        for (int i = 4; i < DW_B; i++)
                            if (((w[i] >> DF_SHIFT) & (uint32_t)FL_AT) && (w[i] & DP_MASK) + 1u >= qe && i >= mi + 2) am |= 1u << i;
                        if (!invalid && am) {
                            const int sj = __ffs((int)am) - 1;
                            inf = (went[k + sj] >> DN_SHIFT) | (DK_OK << 11) | ((uint32_t)mi << 13);
                        }
                    }
                }
            }
            info[c] = inf;
        }
        wave_sync();

        // ---- 4. the chain through the window, run by run ---------------------------------------------------
        const int64_t delta = wbase - wpos0;       // window position -> group-relative position (>= 0 for what is staged)
        int cur = 0;
        bool progressed = false, done = false, bail = false, next_window = false;
        while (!done && !next_window) {
            if (cur >= n_oe) { EX = wbase + (int64_t)(went[nidx[cur]] & DP_MASK); done = true; break; }
            const int b = cur & 63, base = cur - b;
            const int c = base + lane;
            const uint32_t inf = (c < nn) ? info[c] : (DK_GEN << 11);
            const bool stop = !((inf >> 11 & 3u) == DK_OK && (inf & 0x7FFu) == (uint32_t)(c + 1)) || c >= n_oe;
            const unsigned long long nsm = __ballot(stop) & (~0ull << b);
            const int r_rel = nsm ? __ffsll((long long)nsm) - 1 : 64;
            const int rn = base + r_rel;                                           // the node the run ends in front of
            const uint32_t inf_r = (uint32_t)__shfl((int)inf, r_rel & 63);
            const bool inc_r = r_rel < 64 && rn < n_oe && (inf_r >> 11 & 3u) == DK_OK;     // ... is a member with a jump behind it
            const bool mem = lane >= b && (lane < r_rel || (inc_r && lane == r_rel));
            const bool stg_ = mem && c >= n_ob;
            const unsigned long long smk = __ballot(stg_);
            const uint32_t cr = (uint32_t)__popcll(smk);
            if (cr) {
                if (n + cr > (uint32_t)DCHUNK) { bail = true; break; }
                uint32_t myp0 = 0;
                if (stg_) {
                    const int k = nidx[c];
                    const int mi = (int)((inf >> 13) & 15u);
                    StageRec o;
                    myp0 = went[k] & DP_MASK;
                    o.p0 = (uint32_t)(delta + (int64_t)myp0 + 1);
                    o.p1 = (uint32_t)(delta + (int64_t)(went[k + 1] & DP_MASK));
                    o.p3 = (uint32_t)(delta + (int64_t)(went[k + mi] & DP_MASK));
                    o.p4 = (uint32_t)(delta + (int64_t)(went[k + mi + 1] & DP_MASK) + 1);
                    stg[n + (uint32_t)bits_below_lane(smk)] = o;
                    qsum += (unsigned long long)(o.p3 - o.p1 - 1u);
                }
                if (Y == Y_UNRES) Y = wbase + (int64_t)(uint32_t)__shfl((int)myp0, __ffsll((long long)smk) - 1);
                n += cr;
            }
            if (__ballot(mem)) progressed = true;
            if (r_rel == 64) { cur = base + 64; continue; }             // (the run goes on in the next 64 nodes)
            if (rn >= n_oe) { cur = rn; continue; }                      // (the exit: taken at the top)
            const uint32_t kind = inf_r >> 11 & 3u;
            if (kind == DK_OK) { cur = (int)(inf_r & 0x7FFu); continue; }
            if (kind == DK_MORE && progressed) { start = handle_of(nidx[rn]); next_window = true; break; }
            // a generic node: its call and the candidate the chain continues with by the walkers' wave-wide searches
            {
                const int kr = nidx[rn];
                const int64_t Pk = wbase + (int64_t)(went[kr] & DP_MASK);
                const bool own = rn >= n_ob;
                H hm1;
                wv_record_t<true>(L, handle_of(kr), Pk, len, eof, r, hm1);
                if (!own) {
                    if (r.status != ST_COMPLETE) { bail = true; break; }   // the guessed chain ends in the run-in: no guess
                } else {
                    if (Y == Y_UNRES) Y = Pk;
                    if (r.status == ST_COMPLETE || r.final_) {
                        if (n >= (uint32_t)DCHUNK || r.p4 - wpos0 > 0xFFFFFFF0ll) { bail = true; break; }
                        if (lane == 0) {
                            stg[n] = StageRec{(uint32_t)(r.p0 - wpos0), (uint32_t)(r.p1 - wpos0), (uint32_t)(r.p3 - wpos0),
                                              (uint32_t)(r.p4 - wpos0)};
                            qsum += (unsigned long long)(r.p5 - r.p4);
                        }
                        n++;
                    }
                }
                if (r.final_) { EX = X_END_FINAL; have_term = true; done = true; break; }
                if (r.status != ST_COMPLETE) { EX = X_END_TERM; have_term = true; done = true; break; }
                H hs; int64_t Ps; int fls;
                if (!wv_find_t<true>(L, hm1, FL_AT, r.p5 - 1, hs, Ps, fls)) {
                    EX = Y_NOCAND; have_term = true; done = true;
                    r.p0 = r.p1 = r.p3 = r.p4 = r.p5 = -1; r.status = ST_HEAD_BEG; r.final_ = false;
                    break;
                }
                if (Ps >= own_end) { EX = Ps; done = true; break; }
                start = hs; next_window = true;
            }
        }
        if (bail) return;                  // (the group stays flagged: the later tiers take it)
        if (done) break;
        first_window = false;
    }
    if (Y == Y_UNRES) Y = EX;                                           // no member in the own tiles: the chain passes over
    const uint32_t qlo = wave_sum_u32((uint32_t)(qsum & 0xFFFFFu)), qmd = wave_sum_u32((uint32_t)((qsum >> 20) & 0xFFFFFu)),
                   qhi = wave_sum_u32((uint32_t)(qsum >> 40));
    if (lane != 0) return;
    B.y[g] = Y; B.exit[g] = EX; B.cnt[g] = n;
    B.qb[g] = (int64_t)qlo + ((int64_t)qmd << 20) + ((int64_t)qhi << 40);
    B.flags[g] = 0;
    if (have_term) {
        GroupTerm &t = B.term[g];
        t.status = r.status;
        t.pos[0] = r.p0; t.pos[1] = r.p1; t.pos[2] = (r.p1 >= 0) ? r.p1 + 1 : -1;
        t.pos[3] = r.p3; t.pos[4] = r.p4; t.pos[5] = r.p5;
    }
}

// use_list: the first pass -- the groups k_chain_wave flagged are on B.ilist, a small grid takes them one after another (a
// launch over ALL groups, nearly all of which return at once, cost 26 us per 10 GiB); else (repair passes) every group's
// flags and forced entry are looked at.
__global__ __launch_bounds__(256) void k_dense_walk(LineIndex L, ChainBufs B, int64_t offset, int eof, int speculate,
                                                    int walk_all, Ctl *ctl, int use_list)
{
    __shared__ DwLds lds_all[4];
    // (four groups per workgroup, one wave each, no barrier)
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (use_list) {
        const uint32_t nl = *B.icnt;
        for (uint32_t i = blockIdx.x * 4 + wid; i < nl; i += gridDim.x * 4)
            dense_walk_group(L, B, offset, eof, speculate, walk_all, ctl, (int)B.ilist[i], lds_all[wid]);
        return;
    }
    dense_walk_group(L, B, offset, eof, speculate, walk_all, ctl, blockIdx.x * 4 + wid, lds_all[wid]);
}

}  // namespace ffq
