// ffq_kernels.h -- the gfx950 kernels of the FASTQ buffer-scan path.
//
//   k_scan_lines     bytes -> line index            (HBM-bound, the dominant kernel)
//   k_chain<false>   line index -> per-group chain summaries (speculative)
//   k_resolve        verifies the speculation, prefix-sums record counts
//   k_chain<true>    emits the int64[n][6] offset table (+ quality CSR offsets)
//   k_chain_serial   single-lane walker: exact on any input, used when the
//                    speculation cannot be verified
//   k_decode_quals   Phred decode of every record's quality span
//   k_finalize       end offset / result block
//
// What is computed is the record chain of
//   /root/reference/src/fastqandfurious.py:251-279 (readfastq_iter)
// with the scanner of
//   /root/reference/src/_fastqandfurious.c:25-153  (entrypos, C extension).
#pragma once
#include "ffq_dev.h"

namespace ffq {

// =========================================================================
// k_scan_lines: one 256-thread workgroup per 16 KiB tile.  Wave w owns the
// contiguous 4 KiB [w*4096, (w+1)*4096) of the tile and reads it as four
// coalesced 1 KiB rows (16 B per lane).  Per row: SWAR newline mask,
// wave-prefix-sum (DPP) of the per-lane counts, entries (offset | flags)
// written in position order into an LDS list, which the workgroup then
// copies to the tile's slot with 16-byte stores.
// Algorithmic HBM traffic: TILE bytes read + 2 bytes per newline written.
// =========================================================================
__device__ __forceinline__ uint4 load_tail16(const uint8_t *d, int64_t n, int64_t at)
{
    uint32_t w[4] = {0, 0, 0, 0};
    for (int b = 0; b < 16; b++) {
        const int64_t p = at + b;
        if (p < n) w[b >> 2] |= (uint32_t)d[p] << ((b & 3) * 8);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void k_scan_lines(const uint8_t *__restrict__ d, int64_t n,
                                                    uint16_t *__restrict__ ent,
                                                    uint32_t *__restrict__ cnt,
                                                    unsigned long long *__restrict__ ovf,
                                                    uint16_t *__restrict__ pool,
                                                    unsigned long long pool_cap, Ctl *ctl)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_list[SLOT];
    __shared__ uint32_t s_wtot[4];
    __shared__ unsigned long long s_ovf;

    const int tile = blockIdx.x;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int64_t base = (int64_t)tile << TILE_SHIFT;
    const bool full = base + TILE <= n;

    uint4 v[4];
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        o[i] = (uint32_t)(w * 4096 + i * 1024 + l * 16);
        if (full) v[i] = *reinterpret_cast<const uint4 *>(d + base + o[i]);
        else if (base + o[i] + 16 <= n) v[i] = *reinterpret_cast<const uint4 *>(d + base + o[i]);
        else v[i] = load_tail16(d, n, base + o[i]);
    }
    // the byte that follows this wave's 4 KiB span (wave-uniform address)
    const int64_t nxa = base + (int64_t)(w + 1) * 4096;
    const uint32_t nxw = (nxa < n) ? (uint32_t)d[nxa] : 0u;

    uint32_t m[4], c[4], nf[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        m[i] = nl_mask16(v[i]);
        c[i] = __popc(m[i]);
    }
    // first byte of the NEXT 16-byte piece in position order
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t fb = v[i].x & 0xFFu;
        const uint32_t dn = (uint32_t)__shfl_down((int)fb, 1);
        const uint32_t wrap = (i < 3) ? (uint32_t)__shfl((int)(v[(i + 1) & 3].x & 0xFFu), 0) : nxw;
        nf[i] = (l == 63) ? wrap : dn;
    }
    // wave prefix sums of the four row counts, two 16-bit fields per register
    const uint32_t s01 = wave_incl_scan(c[0] | (c[1] << 16));
    const uint32_t s23 = wave_incl_scan(c[2] | (c[3] << 16));
    const uint32_t t01 = (uint32_t)__shfl((int)s01, 63), t23 = (uint32_t)__shfl((int)s23, 63);
    uint32_t ex[4], rowtot[4];
    ex[0] = (s01 & 0xFFFFu) - c[0];  rowtot[0] = t01 & 0xFFFFu;
    ex[1] = (s01 >> 16) - c[1];      rowtot[1] = t01 >> 16;
    ex[2] = (s23 & 0xFFFFu) - c[2];  rowtot[2] = t23 & 0xFFFFu;
    ex[3] = (s23 >> 16) - c[3];      rowtot[3] = t23 >> 16;
    const uint32_t wtot = rowtot[0] + rowtot[1] + rowtot[2] + rowtot[3];
    if (l == 0) s_wtot[w] = wtot;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t t = s_wtot[q];
        if (q < w) wbase += t;
        total += t;
    }
    const bool dense = total > (uint32_t)SLOT;
    if (dense) {   // rare: avg line shorter than 16 bytes over the whole tile
        if (tid == 0) {
            const unsigned long long at = atomicAdd(&ctl->pool_head, (unsigned long long)total);
            s_ovf = at;
            if (at + total > pool_cap) atomicOr(&ctl->err, ERR_POOL);
        }
        __syncthreads();
    }
    const unsigned long long pbase = dense ? s_ovf : 0ull;
    const bool pool_ok = dense && (pbase + total <= pool_cap);

    uint32_t rb = wbase;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t mm = m[i];
        uint32_t idx = rb + ex[i];
        while (mm) {
            const uint32_t p = (uint32_t)__ffs((int)mm) - 1u;
            mm &= mm - 1u;
            const uint32_t nb = (p < 15u) ? get_byte(v[i], p + 1u) : nf[i];
            const uint32_t fl = (nb == '@') ? (uint32_t)FL_AT : (nb == '+') ? (uint32_t)FL_PLUS : 0u;
            const uint16_t e = (uint16_t)((o[i] + p) | (fl << 14));
            if (!dense) s_list[idx] = e;
            else if (pool_ok) pool[pbase + idx] = e;
            idx++;
        }
        rb += rowtot[i];
    }
    if (tid == 0) {
        cnt[tile] = total;
        ovf[tile] = pbase;
    }
    if (!dense) {
        __syncthreads();
        const uint32_t nvec = (total * 2u + 15u) >> 4;      // 16-byte pieces
        uint4 *dst = reinterpret_cast<uint4 *>(ent + (int64_t)tile * SLOT);
        const uint4 *src = reinterpret_cast<const uint4 *>(s_list);
        for (uint32_t q = tid; q < nvec; q += 256) dst[q] = src[q];
    }
}

// =========================================================================
// Chain kernels.
// A "group" is OWN consecutive tiles.  The workgroup of group g loads the
// line index of a window = RUNIN tiles before + OWN tiles + AHEAD tiles after
// into LDS, computes one scanner call per "\n@" candidate of the run-in and
// own regions (thread per candidate), and follows the successor links
// (offset = pos5-1 -> next "\n@", fastqandfurious.py:254) by pointer doubling.
//
// Speculation: the chain's entry into the own region (Y) is guessed as the
// exit of the chain that starts at the earliest run-in candidate (a chain
// started at a false '@' candidate, e.g. a quality line beginning with '@',
// re-synchronises with the true chain within a few records).
// k_resolve checks yguess[g+1] == exit[g] for every group up to the chain's
// end, which proves all guesses by induction from group 0 (whose Y is exact).
// Any mismatch -> the serial walker redoes the buffer (still on the GPU).
// =========================================================================
constexpr int RUNIN = 1, OWN = 4, AHEAD = 1;
constexpr int NT = RUNIN + OWN + AHEAD;
constexpr int E_MAX = NT * SLOT + 16;      // window entries (+ sentinel)
constexpr int C_MAX = 1024;                // candidates in run-in + own
constexpr uint32_t W_POS = 0x3FFFFu;       // 18 bits: window-relative position
constexpr int W_NODE_SHIFT = 18;           // 10 bits: node id of a candidate entry
constexpr uint32_t W_NODE_MASK = 0x3FFu;
constexpr uint32_t NO_NODE = 0x3FFu;       // candidate without a node id
constexpr uint16_t NX_OUT = 0xFFFF, NX_NOCAND = 0xFFFE, NX_STOP = 0xFFFD;
constexpr uint16_t NM_EXT = 0xFFFF;
constexpr uint16_t UNMARKED = 0xFFFF;

constexpr int64_t Y_NOCAND = -1, X_END_TERM = -2, Y_UNRES = -3, X_END_FINAL = -4;

struct GroupSum {
    int64_t yguess;       // >=0: buffer coordinate of the '\n' of the entry candidate
    int64_t exit;         // >=0: first chain candidate past the own region; X_*/Y_NOCAND
    int64_t qbytes;
    uint32_t count;
    uint32_t lines;
    int32_t flags;        // bit0: irregular (window does not fit the LDS budget)
    int32_t term_status;
    int64_t term_pos[6];
};

struct DevRes {
    int64_t n_records, n_qual_bytes, n_lines, end_offset;
    int64_t last_pos[6];
    int32_t last_status, end_state, fallback, term_group;
    int32_t has_final, pad;
};

// window accessor: flat LDS index while inside the window, global beyond
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
            if (wt1 <= 0) h.g = H{-1, 0x7FFFFFF0};
        }
        // global continuation: entries of tiles >= h.g.tile+1, or further in h.g.tile
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
            P = wpos0 + (int64_t)(e & W_POS);
            fl = (int)(e >> 30);
            return;
        }
        GAcc(L).get(h.g, P, fl);
    }
};

template <bool EMIT>
__global__ __launch_bounds__(256) void k_chain(LineIndex L, int64_t offset, int eof, int64_t add,
                                               GroupSum *__restrict__ sums,
                                               const int64_t *__restrict__ ystart,
                                               const int64_t *__restrict__ rbase,
                                               const int64_t *__restrict__ qbase,
                                               const DevRes *__restrict__ dres,
                                               int64_t *__restrict__ table, int64_t table_cap,
                                               int64_t *__restrict__ qoff, Ctl *ctl)
{
    __shared__ uint32_t went[E_MAX];
    __shared__ uint16_t cidx[C_MAX];     // node -> window entry index
    __shared__ uint16_t nm[C_MAX];       // node -> window index of its "\n+" entry (NM_EXT: recompute)
    __shared__ uint16_t nxtE[C_MAX];     // node -> window entry index of the successor candidate
    __shared__ uint16_t S[C_MAX];        // pointer doubling: node reached
    __shared__ uint16_t cn[C_MAX];       //                   steps taken
    __shared__ uint32_t Q[C_MAX];        //                   quality bytes of the nodes stepped over
    __shared__ uint32_t qlen[C_MAX];
    __shared__ int8_t nstat[C_MAX];
    __shared__ uint16_t dist[EMIT ? C_MAX : 1];   // EMIT: rank of a chain member (UNMARKED otherwise)
    __shared__ uint32_t qpre[EMIT ? C_MAX : 1];   // EMIT: quality bytes before the member
    __shared__ int32_t s_tcnt[NT + 1];
    __shared__ int32_t s_tbase[NT + 2];
    __shared__ int32_t s_wc[4];
    __shared__ int32_t s_misc[8];
    __shared__ long long s_y;

    const int g = blockIdx.x;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;

    if (EMIT) {
        if (dres->fallback) return;
        if (ystart[g] < 0) return;        // chain does not reach this group
    }

    const int own0 = g * OWN;
    const int own1 = min(own0 + OWN, L.ntiles);
    const int wt0 = max(own0 - RUNIN, 0);
    const int wt1 = min(own1 + AHEAD, L.ntiles);
    const int nwt = wt1 - wt0;
    const int sent = (wt0 == 0 && L.s) ? 1 : 0;
    const int64_t wpos0 = (int64_t)wt0 << TILE_SHIFT;     // buffer coord of rel 0 (rel includes +s)
    const int64_t len = L.len();

    // ---- window directory ------------------------------------------------
    if (tid < nwt) s_tcnt[tid] = (int32_t)L.cnt[wt0 + tid];
    __syncthreads();
    if (tid == 0) {
        int32_t b = sent, irregular = 0, lines = 0;
        for (int t = 0; t < nwt; t++) {
            s_tbase[t] = b;
            if (s_tcnt[t] > SLOT) irregular = 1;
            b += s_tcnt[t];
            if (wt0 + t >= own0 && wt0 + t < own1) lines += s_tcnt[t];
        }
        s_tbase[nwt] = b;
        if (b > E_MAX) irregular = 1;
        s_misc[0] = irregular;
        s_misc[1] = lines;
    }
    __syncthreads();
    const int nwin = s_tbase[nwt];
    if (s_misc[0]) {
        if (!EMIT) {
            if (tid == 0) {
                GroupSum &o = sums[g];
                o.yguess = Y_UNRES; o.exit = Y_UNRES; o.qbytes = 0; o.count = 0;
                o.lines = (uint32_t)s_misc[1]; o.flags = 1; o.term_status = 0;
            }
        } else if (tid == 0) atomicOr(&ctl->err, ERR_INTERNAL);
        return;
    }
    // entry index boundaries of the own region
    const int own_lo = (g == 0) ? 0 : s_tbase[own0 - wt0];
    const int own_hi = s_tbase[own1 - wt0];
    const int64_t own_hi_pos = ((int64_t)own1 << TILE_SHIFT) + L.s;   // buffer coord of first byte after own

    // ---- load the window's entries ---------------------------------------
    if (sent && tid == 0) {
        const uint8_t b0 = L.n > 0 ? L.d[0] : 0;
        const uint32_t fl = (b0 == '@') ? FL_AT : (b0 == '+') ? FL_PLUS : 0;
        went[0] = 0u | (NO_NODE << W_NODE_SHIFT) | (fl << 30);
    }
    for (int t = 0; t < nwt; t++) {
        const int c = s_tcnt[t];
        const uint16_t *src = L.ent + (int64_t)(wt0 + t) * SLOT;
        const uint32_t relb = (uint32_t)(t << TILE_SHIFT) + (uint32_t)L.s;
        for (int i = tid; i < c; i += 256) {
            const uint32_t e = src[i];
            went[s_tbase[t] + i] = (relb + (e & OFF_MASK)) | (NO_NODE << W_NODE_SHIFT) | ((e >> 14) << 30);
        }
    }
    __syncthreads();

    // ---- compact the candidates of run-in + own into node ids -------------
    // each wave takes a contiguous quarter of [0, own_hi), two passes
    const int per = (own_hi + 3) >> 2;
    const int q0 = min(w * per, own_hi), q1 = min(q0 + per, own_hi);
    int wc = 0;
    for (int j = q0 + l; j - l < q1; j += 64) {
        bool isc = false;
        if (j < q1) {
            const uint32_t e = went[j];
            isc = ((e >> 30) & FL_AT) && (wpos0 + (int64_t)(e & W_POS) >= offset);
        }
        wc += __popcll(__ballot(isc));
    }
    if (l == 0) s_wc[w] = wc;
    __syncthreads();
    int nbase = 0, ncomp = 0;
    for (int q = 0; q < 4; q++) { if (q < w) nbase += s_wc[q]; ncomp += s_wc[q]; }
    if (ncomp >= C_MAX) {      // node id C_MAX-1 == NO_NODE is reserved
        if (!EMIT) {
            if (tid == 0) {
                GroupSum &o = sums[g];
                o.yguess = Y_UNRES; o.exit = Y_UNRES; o.qbytes = 0; o.count = 0;
                o.lines = (uint32_t)s_misc[1]; o.flags = 1; o.term_status = 0;
            }
        } else if (tid == 0) atomicOr(&ctl->err, ERR_INTERNAL);
        return;
    }
    for (int j = q0 + l; j - l < q1; j += 64) {
        bool isc = false;
        uint32_t e = 0;
        if (j < q1) {
            e = went[j];
            isc = ((e >> 30) & FL_AT) && (wpos0 + (int64_t)(e & W_POS) >= offset);
        }
        const unsigned long long bal = __ballot(isc);
        if (isc) {
            const int r = nbase + __popcll(bal & ((1ull << l) - 1ull));
            cidx[r] = (uint16_t)j;
            went[j] = (e & ~(W_NODE_MASK << W_NODE_SHIFT)) | ((uint32_t)r << W_NODE_SHIFT);
        }
        nbase += __popcll(bal);
    }
    __syncthreads();

    // ---- one scanner call per node ----------------------------------------
    const WAcc acc(L, went, nwin, wt1, wpos0);
    for (int c = tid; c < ncomp; c += 256) {
        const int k = cidx[c];
        WH hk; hk.idx = k; hk.g = H{0, 0};
        WH hm, hm1;
        Rec r;
        const int64_t Pk = wpos0 + (int64_t)(went[k] & W_POS);
        compute_record(acc, hk, Pk, len, eof, r, hm, hm1);
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

    // ---- pointer doubling inside each region --------------------------------
    for (int c = tid; c < ncomp; c += 256) {
        const int reg = (cidx[c] < own_lo) ? 0 : 1;
        const uint16_t nx = nxtE[c];
        uint16_t s = (uint16_t)c;
        if (nx < NX_STOP && nx < own_hi) {
            const int nreg = ((int)nx < own_lo) ? 0 : 1;
            const uint32_t nid = (went[nx] >> W_NODE_SHIFT) & W_NODE_MASK;
            if (nreg == reg && nid != NO_NODE) s = (uint16_t)nid;
        }
        S[c] = s;
        cn[c] = (s != c) ? 1 : 0;
        Q[c] = (s != c) ? qlen[c] : 0u;
        if (EMIT) { dist[c] = UNMARKED; qpre[c] = 0; }
    }
    __syncthreads();

    // ---- pointer doubling (summary pass): S = last node of the chain inside its region
    int rounds = 1;
    while ((1 << rounds) < ncomp) rounds++;
    if (!EMIT) {
        for (int k = 0; k < rounds; k++) {
            uint16_t s1[C_MAX / 256], s2[C_MAX / 256], c1[C_MAX / 256], c2[C_MAX / 256];
            uint32_t g1[C_MAX / 256], g2[C_MAX / 256];
#pragma unroll
            for (int u = 0; u < C_MAX / 256; u++) {
                const int c = tid + u * 256;
                if (c < ncomp) {
                    s1[u] = S[c]; c1[u] = cn[c]; g1[u] = Q[c];
                    s2[u] = S[s1[u]]; c2[u] = cn[s1[u]]; g2[u] = Q[s1[u]];
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < C_MAX / 256; u++) {
                const int c = tid + u * 256;
                if (c < ncomp) {
                    S[c] = s2[u];
                    cn[c] = (uint16_t)(c1[u] + c2[u]);
                    Q[c] = g1[u] + g2[u];
                }
            }
            __syncthreads();
        }
    }

    // ---- the chain's entry candidate Y --------------------------------------
    // EMIT: given (verified by k_resolve).  Group 0: exact, the first "\n@" at >= offset.
    // Otherwise guessed: the exit of the chain that starts at the EARLIEST run-in
    // candidate -- it had the whole run-in region to re-synchronise with the true
    // chain.  With no usable run-in chain: the first candidate at/after the own region.
    if (tid == 0) {
        long long y = Y_UNRES;
        if (EMIT) y = ystart[g];
        else if (g == 0) {
            WH hb; hb.idx = -2; hb.g = H{0, 0};
            WH hs; int64_t Ps;
            y = find_cand(acc, hb, offset, hs, Ps) ? Ps : Y_NOCAND;
        } else {
            bool got = false;
            for (int c = 0; c < ncomp && c < 8 && cidx[c] < own_lo && !got; c++) {
                const int last = S[c];
                if (nstat[last] != ST_COMPLETE) continue;      // this chain dies inside the run-in
                const uint16_t nx = nxtE[last];
                if (nx == NX_NOCAND) y = Y_NOCAND;
                else if (nx != NX_OUT) y = wpos0 + (long long)(went[nx] & W_POS);
                else {   // successor beyond the window: recompute through the global index
                    const int k = cidx[last];
                    WH hk; hk.idx = k; hk.g = H{0, 0};
                    WH hm, hm1, hs; Rec r; int64_t Ps;
                    compute_record(acc, hk, wpos0 + (int64_t)(went[k] & W_POS), len, eof, r, hm, hm1);
                    y = find_cand(acc, hm1, r.p5 - 1, hs, Ps) ? Ps : Y_NOCAND;
                }
                got = true;
            }
            if (!got) {
                WH hb; hb.g = H{0, 0};
                hb.idx = (own_lo > 0) ? own_lo - 1 : -2;
                WH hs; int64_t Ps;
                const int64_t own_lo_pos = ((int64_t)own0 << TILE_SHIFT) + L.s;
                y = find_cand(acc, hb, max(offset, own_lo_pos), hs, Ps) ? Ps : Y_NOCAND;
            }
        }
        s_y = y;
    }
    __syncthreads();
    const int64_t Y = s_y;

    // locate Y's node (thread 0), decide what this group does
    if (tid == 0) {
        int32_t ynode = -1, kind;      // kind: 0 walk from ynode, 1 skip, 2 chain over, 3 unresolved
        if (Y == Y_UNRES) kind = 3;
        else if (Y < 0) kind = 2;
        else if (Y >= own_hi_pos) kind = 1;
        else {
            const int64_t rel = Y - wpos0;
            int lo = 0, hi = ncomp;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((int64_t)(went[cidx[mid]] & W_POS) < rel) lo = mid + 1; else hi = mid;
            }
            if (lo < ncomp && (int64_t)(went[cidx[lo]] & W_POS) == rel && cidx[lo] >= own_lo) {
                ynode = lo; kind = 0;
            } else kind = 3;
        }
        s_misc[2] = ynode;
        s_misc[3] = kind;
        if (EMIT && kind == 0) { dist[ynode] = 0; qpre[ynode] = 0; }
    }
    __syncthreads();
    const int ynode = s_misc[2], kind = s_misc[3];
    if (EMIT && kind != 0) {
        if (kind == 3 && tid == 0) atomicOr(&ctl->err, ERR_INTERNAL);
        return;
    }

    // ---- emit pass: doubling rounds with bottom-up marking from Y ------------
    // before round k the marked set is every chain node at distance < 2^k from Y;
    // a marked node whose 2^k-step jump is exact marks its target (distance + 2^k).
    if (EMIT) {
        for (int k = 0; k < rounds; k++) {
            uint16_t s1[C_MAX / 256], s2[C_MAX / 256], c1[C_MAX / 256], c2[C_MAX / 256];
            uint32_t g1[C_MAX / 256], g2[C_MAX / 256];
            bool mk[C_MAX / 256];
            uint16_t dd[C_MAX / 256];
            uint32_t qq[C_MAX / 256];
#pragma unroll
            for (int u = 0; u < C_MAX / 256; u++) {
                const int c = tid + u * 256;
                mk[u] = false;
                if (c < ncomp) {
                    s1[u] = S[c]; c1[u] = cn[c]; g1[u] = Q[c];
                    s2[u] = S[s1[u]]; c2[u] = cn[s1[u]]; g2[u] = Q[s1[u]];
                    dd[u] = dist[c]; qq[u] = qpre[c];
                    mk[u] = (dd[u] != UNMARKED) && (c1[u] == (uint16_t)(1u << k));
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < C_MAX / 256; u++) {
                const int c = tid + u * 256;
                if (c < ncomp) {
                    if (mk[u]) {
                        dist[s1[u]] = (uint16_t)(dd[u] + (1u << k));
                        qpre[s1[u]] = qq[u] + g1[u];
                    }
                    S[c] = s2[u];
                    cn[c] = (uint16_t)(c1[u] + c2[u]);
                    Q[c] = g1[u] + g2[u];
                }
            }
            __syncthreads();
        }
    }

    if (!EMIT) {
        // ---- summary -------------------------------------------------------
        if (tid == 0) {
            GroupSum &o = sums[g];
            o.yguess = Y; o.flags = 0; o.lines = (uint32_t)s_misc[1];
            o.count = 0; o.qbytes = 0; o.term_status = 0;
            for (int i = 0; i < 6; i++) o.term_pos[i] = -1;
            if (kind == 3) { o.yguess = Y_UNRES; o.exit = Y_UNRES; }
            else if (kind == 2) { o.exit = Y_NOCAND; o.term_status = ST_HEAD_BEG; }
            else if (kind == 1) { o.exit = Y; }
            else {
                const int last = S[ynode];
                const int st = nstat[last];
                const bool emits = (st == ST_COMPLETE) || (st == ST_FINAL);
                o.count = (uint32_t)cn[ynode] + (emits ? 1u : 0u);
                o.qbytes = (int64_t)Q[ynode] + (emits ? (int64_t)qlen[last] : 0);
                if (st == ST_COMPLETE) {
                    const uint16_t nx = nxtE[last];
                    if (nx == NX_NOCAND) { o.exit = Y_NOCAND; o.term_status = ST_HEAD_BEG; }
                    else if (nx != NX_OUT) o.exit = wpos0 + (int64_t)(went[nx] & W_POS);
                    else {
                        const int k = cidx[last];
                        WH hk; hk.idx = k; hk.g = H{0, 0};
                        WH hm, hm1, hs; Rec r; int64_t Ps;
                        compute_record(acc, hk, wpos0 + (int64_t)(went[k] & W_POS), len, eof, r, hm, hm1);
                        if (find_cand(acc, hm1, r.p5 - 1, hs, Ps)) o.exit = Ps;
                        else { o.exit = Y_NOCAND; o.term_status = ST_HEAD_BEG; }
                    }
                } else {
                    // the chain stops at `last`: report the scanner's posbuffer for that call
                    const int k = cidx[last];
                    WH hk; hk.idx = k; hk.g = H{0, 0};
                    WH hm, hm1; Rec r;
                    compute_record(acc, hk, wpos0 + (int64_t)(went[k] & W_POS), len, eof, r, hm, hm1);
                    o.exit = (st == ST_FINAL) ? X_END_FINAL : X_END_TERM;
                    o.term_status = r.status;
                    o.term_pos[0] = r.p0; o.term_pos[1] = r.p1;
                    o.term_pos[2] = (r.p1 >= 0) ? r.p1 + 1 : -1;
                    o.term_pos[3] = r.p3; o.term_pos[4] = r.p4; o.term_pos[5] = r.p5;
                }
            }
        }
        return;
    } else {
        // ---- emission: every marked node is a record of the chain ------------
        const int64_t r0 = rbase[g], qb0 = qbase[g];
        for (int c = tid; c < ncomp; c += 256) {
            const uint16_t dc = dist[c];
            if (dc == UNMARKED) continue;
            const int st = nstat[c];
            if (st != ST_COMPLETE && st != ST_FINAL) continue;
            const int64_t row = r0 + dc;
            if (qoff) {
                if (row < table_cap) qoff[row] = qb0 + qpre[c];
            }
            if (row >= table_cap) continue;
            int64_t p0, p1, p3, p4, p5;
            const uint16_t mi = nm[c];
            const int k = cidx[c];
            if (mi != NM_EXT) {
                p0 = wpos0 + (int64_t)(went[k] & W_POS) + 1;
                p1 = wpos0 + (int64_t)(went[k + 1] & W_POS);
                p3 = wpos0 + (int64_t)(went[mi] & W_POS);
                p4 = wpos0 + (int64_t)(went[mi + 1] & W_POS) + 1;
                p5 = p4 + p3 - p1 - 1;
            } else {
                WH hk; hk.idx = k; hk.g = H{0, 0};
                WH hm, hm1; Rec r;
                compute_record(acc, hk, wpos0 + (int64_t)(went[k] & W_POS), len, eof, r, hm, hm1);
                p0 = r.p0; p1 = r.p1; p3 = r.p3; p4 = r.p4; p5 = r.p5;
            }
            int64_t *o = table + row * 6;
            // rows are 48 bytes and 16-byte aligned: three 16-byte stores
            longlong2 *o2 = reinterpret_cast<longlong2 *>(o);
            o2[0] = make_longlong2(p0 + add, p1 + add);
            o2[1] = make_longlong2(p1 + 1 + add, p3 + add);
            o2[2] = make_longlong2(p4 + add, p5 + add);
        }
    }
}

// =========================================================================
// k_resolve: one workgroup.  Verifies the per-group guesses, finds the group
// the chain ends in, exclusive-scans record counts / quality bytes, and fills
// the result block (end state per fastqandfurious.py:256-279).
// =========================================================================
__global__ __launch_bounds__(1024) void k_resolve(const GroupSum *__restrict__ sums, int ngroups,
                                                  int eof, int64_t offset, int64_t add,
                                                  int64_t *__restrict__ ystart,
                                                  int64_t *__restrict__ rbase,
                                                  int64_t *__restrict__ qbase, DevRes *res)
{
    __shared__ int s_term, s_bad;
    __shared__ long long s_part[1024], s_partq[1024];
    __shared__ unsigned long long s_lines;
    const int tid = threadIdx.x;
    if (tid == 0) { s_term = 0x7FFFFFFF; s_bad = 0x7FFFFFFF; s_lines = 0; }
    __syncthreads();
    int lt = 0x7FFFFFFF, lb = 0x7FFFFFFF;
    unsigned long long ll = 0;
    for (int t = tid; t < ngroups; t += 1024) {
        const GroupSum &sg = sums[t];
        ll += sg.lines;
        const int64_t ex = sg.exit;
        if (ex == Y_NOCAND || ex == X_END_TERM || ex == X_END_FINAL) lt = min(lt, t);
        bool bad = (sg.flags & 1) || (sg.yguess == Y_UNRES);
        if (t > 0) {
            const int64_t pe = sums[t - 1].exit;
            if (pe >= 0 && sg.yguess != pe) bad = true;
        }
        if (bad) lb = min(lb, t);
    }
    atomicMin(&s_term, lt);
    atomicMin(&s_bad, lb);
    atomicAdd(&s_lines, ll);
    __syncthreads();
    const int tterm = s_term, tbad = s_bad;
    const bool fallback = (tterm == 0x7FFFFFFF) || (tbad <= tterm);

    // exclusive scan of counts over groups 0..tterm
    const int per = (ngroups + 1023) / 1024;
    const int a0 = min(tid * per, ngroups), a1 = min(a0 + per, ngroups);
    long long ps = 0, pq = 0;
    for (int t = a0; t < a1; t++)
        if (!fallback && t <= tterm) { ps += sums[t].count; pq += sums[t].qbytes; }
    s_part[tid] = ps; s_partq[tid] = pq;
    __syncthreads();
    // Hillis-Steele over 1024 partials
    for (int d = 1; d < 1024; d <<= 1) {
        long long v = 0, vq = 0;
        if (tid >= d) { v = s_part[tid - d]; vq = s_partq[tid - d]; }
        __syncthreads();
        s_part[tid] += v; s_partq[tid] += vq;
        __syncthreads();
    }
    long long run = s_part[tid] - ps, runq = s_partq[tid] - pq;
    for (int t = a0; t < a1; t++) {
        const bool on = !fallback && t <= tterm;
        ystart[t] = on ? sums[t].yguess : (int64_t)-1;
        rbase[t] = run; qbase[t] = runq;
        if (on) { run += sums[t].count; runq += sums[t].qbytes; }
    }
    if (tid == 1023) {
        res->n_records = s_part[1023];
        res->n_qual_bytes = s_partq[1023];
    }
    if (tid == 0) {
        res->n_lines = (int64_t)s_lines;
        res->fallback = fallback ? 1 : 0;
        res->term_group = fallback ? -1 : tterm;
        res->end_offset = offset;
        res->has_final = 0;
        if (!fallback) {
            const GroupSum &tg = sums[tterm];
            const int st = tg.term_status;
            res->last_status = st;
            for (int i = 0; i < 6; i++) res->last_pos[i] = (tg.term_pos[i] >= 0) ? tg.term_pos[i] + add : -1;
            int end;
            if (tg.exit == X_END_FINAL) { end = 0; res->has_final = 1; }
            else if (tg.exit == Y_NOCAND) end = eof ? 0 : 1;
            else if (eof) end = (st == ST_QUAL_END) ? 2 : (st == ST_INVALID) ? 4 : 3;
            else end = (st == ST_INVALID) ? 4 : 1;
            res->end_state = end;
        }
    }
}

// =========================================================================
// k_chain_serial: the whole chain by one lane over the global index.  Exact on
// any input (dense tiles, records longer than the window, chains that do not
// re-synchronise); ~microseconds per record.
// =========================================================================
__global__ void k_chain_serial(LineIndex L, int64_t offset, int eof, int64_t add,
                               int64_t *__restrict__ table, int64_t table_cap,
                               int64_t *__restrict__ qoff, DevRes *res)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const GAcc a(L);
    const int64_t len = L.len();
    int64_t n = 0, qb = 0, off = offset;
    H k, hm, hm1;
    int64_t Pk;
    Rec r;
    int status = ST_HEAD_BEG, end;
    r.p0 = r.p1 = r.p3 = r.p4 = r.p5 = -1; r.final_ = false;
    bool have = find_cand(a, a.before(), offset, k, Pk);
    for (;;) {
        if (!have) { status = ST_HEAD_BEG; r.p0 = r.p1 = r.p3 = r.p4 = r.p5 = -1; r.final_ = false; break; }
        compute_record(a, k, Pk, len, eof, r, hm, hm1);
        status = r.status;
        if (status != ST_COMPLETE && !r.final_) break;
        if (n < table_cap) {
            int64_t *o = table + n * 6;
            o[0] = r.p0 + add; o[1] = r.p1 + add; o[2] = r.p1 + 1 + add;
            o[3] = r.p3 + add; o[4] = r.p4 + add; o[5] = r.p5 + add;
            if (qoff) qoff[n] = qb;
        }
        n++;
        qb += r.p5 - r.p4;
        if (r.final_) break;
        off = r.p5 - 1;
        have = find_cand(a, hm1, r.p5 - 1, k, Pk);
    }
    if (r.final_) end = 0;
    else if (status == ST_HEAD_BEG) end = eof ? 0 : 1;
    else if (eof) end = (status == ST_QUAL_END) ? 2 : (status == ST_INVALID) ? 4 : 3;
    else end = (status == ST_INVALID) ? 4 : 1;
    res->n_records = n;
    res->n_qual_bytes = qb;
    res->end_offset = off;
    res->last_status = status;
    res->end_state = end;
    res->has_final = r.final_ ? 1 : 0;
    const int64_t p[6] = {r.p0, r.p1, r.p1 >= 0 ? r.p1 + 1 : -1, r.p3, r.p4, r.p5};
    for (int i = 0; i < 6; i++) res->last_pos[i] = p[i] >= 0 ? p[i] + add : -1;
    int64_t nl = 0;
    for (int t = 0; t < L.ntiles; t++) nl += L.cnt[t];
    res->n_lines = nl;
}

// k_finalize: the iterator's `offset` at exit = pos5 - 1 of the last COMPLETE
// record (fastqandfurious.py:254), read back from the table; qoff[n].
__global__ void k_finalize(DevRes *res, const int64_t *__restrict__ table, int64_t table_cap,
                           int64_t add, int64_t offset, int64_t *__restrict__ qoff)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (res->fallback) return;
    const int64_t ncomplete = res->n_records - (res->has_final ? 1 : 0);
    if (ncomplete > 0 && ncomplete <= table_cap) res->end_offset = table[(ncomplete - 1) * 6 + 5] - add - 1;
    else res->end_offset = offset;
    if (qoff && res->n_records <= table_cap) qoff[res->n_records] = res->n_qual_bytes;
}

__global__ void k_finalize_serial(DevRes *res, int64_t table_cap, int64_t *__restrict__ qoff)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (qoff && res->n_records <= table_cap) qoff[res->n_records] = res->n_qual_bytes;
}

// =========================================================================
// k_decode_quals: out[qoff[i] + b] = buf[pos4_i + b] + qadd, one wave per record
// (arrayadd_b over each record's quality slice, _fastqandfurious.c:161-185).
// =========================================================================
__global__ __launch_bounds__(256) void k_decode_quals(const uint8_t *__restrict__ d, int s,
                                                      const int64_t *__restrict__ table,
                                                      const int64_t *__restrict__ qoff,
                                                      const DevRes *__restrict__ res,
                                                      int64_t table_cap, int64_t add, int qadd,
                                                      int8_t *__restrict__ out, int64_t out_cap)
{
    const int64_t n = min(res->n_records, table_cap);
    const int l = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const uint8_t v = (uint8_t)qadd;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t p4 = table[i * 6 + 4] - add, p5 = table[i * 6 + 5] - add;
        const int64_t qo = qoff[i];
        const uint8_t *src = d + (p4 - s);
        const int64_t m = p5 - p4;
        for (int64_t b = l; b < m; b += 64)
            if (qo + b < out_cap) out[qo + b] = (int8_t)(uint8_t)(src[b] + v);
    }
}

// lower bound over one column of the (record-ordered) offset table
__global__ void k_table_lower_bound(const int64_t *__restrict__ table, int64_t n, int col,
                                    int64_t value, int64_t *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (table[mid * 6 + col] < value) lo = mid + 1; else hi = mid;
    }
    *out = lo;
}

// =========================================================================
// arrayadd_b / arrayadd_q (_fastqandfurious.c:161-217), in place, wrapping
// =========================================================================
__global__ __launch_bounds__(256) void k_arrayadd_b(uint8_t *__restrict__ a, int64_t n, uint32_t v)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    const uintptr_t mis = (16 - (reinterpret_cast<uintptr_t>(a) & 15)) & 15;
    const int64_t head = min((int64_t)mis, n);
    const int64_t nvec = (n - head) >> 4;
    uint4 *av = reinterpret_cast<uint4 *>(a + head);
    const uint32_t vv = v * 0x01010101u;
    for (int64_t i = gid; i < nvec; i += gsz) {
        uint4 x = av[i];
        // per-byte add without carries across bytes
        auto addb = [vv](uint32_t y) {
            return ((y & 0x7F7F7F7Fu) + (vv & 0x7F7F7F7Fu)) ^ ((y ^ vv) & 0x80808080u);
        };
        x.x = addb(x.x); x.y = addb(x.y); x.z = addb(x.z); x.w = addb(x.w);
        av[i] = x;
    }
    const int64_t tail0 = head + (nvec << 4);
    for (int64_t i = gid; i < head; i += gsz) a[i] = (uint8_t)(a[i] + v);
    for (int64_t i = tail0 + gid; i < n; i += gsz) a[i] = (uint8_t)(a[i] + v);
}

__global__ __launch_bounds__(256) void k_arrayadd_q(unsigned long long *__restrict__ a, int64_t n,
                                                    unsigned long long v)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    const bool al = (reinterpret_cast<uintptr_t>(a) & 15) == 0;
    if (al) {
        const int64_t nvec = n >> 1;
        ulonglong2 *av = reinterpret_cast<ulonglong2 *>(a);
        for (int64_t i = gid; i < nvec; i += gsz) {
            ulonglong2 x = av[i];
            x.x += v; x.y += v;
            av[i] = x;
        }
        if ((n & 1) && gid == 0) a[n - 1] += v;
    } else {
        for (int64_t i = gid; i < n; i += gsz) a[i] += v;
    }
}

// =========================================================================
// synthetic inputs
// =========================================================================
__device__ __forceinline__ void put_header(uint8_t *o, int64_t i)
{
    o[0] = '@'; o[1] = 'S'; o[2] = 'Y'; o[3] = 'N'; o[4] = '.';
    int64_t x = i;
    for (int k = 9; k >= 0; k--) { o[5 + k] = (uint8_t)('0' + (x % 10)); x /= 10; }
    o[15] = '/'; o[16] = '1'; o[17] = '\n';
}

__global__ __launch_bounds__(256) void k_synth_single(uint8_t *__restrict__ out, int64_t first,
                                                      int64_t count, uint64_t seed)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const int64_t i = first + r;
    uint8_t *o = out + r * 322;
    put_header(o, i);
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    for (int j = 0; j < 150; j++) {
        const uint64_t h = splitmix64(seed ^ (((uint64_t)i << 9) | (uint64_t)j));
        o[18 + j] = (uint8_t)acgt[h & 3];
        o[171 + j] = (uint8_t)(33 + (h >> 8) % 41);
    }
    o[168] = '\n'; o[169] = '+'; o[170] = '\n';
    o[321] = '\n';
}

__host__ __device__ inline int64_t synth_wrapped_size(int64_t i, uint64_t seed)
{
    const uint64_t h = splitmix64(seed ^ (uint64_t)i);
    const int64_t Lr = 50 + (int64_t)(h % 251);
    const int64_t nl = (Lr + 79) / 80;
    const int64_t rep = ((h >> 32) % 4 == 0) ? 16 : 0;
    return 18 + Lr + nl + 1 + rep + 1 + Lr + nl;
}

__global__ __launch_bounds__(256) void k_synth_wrapped(uint8_t *__restrict__ out,
                                                       const int64_t *__restrict__ start,
                                                       int64_t first, int64_t count, uint64_t seed)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const int64_t i = first + r;
    const uint64_t hh = splitmix64(seed ^ (uint64_t)i);
    const int Lr = 50 + (int)(hh % 251);
    const bool rep = ((hh >> 32) % 4 == 0);
    uint8_t *o = out + start[r];
    put_header(o, i);
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    int w = 18;
    for (int j = 0; j < Lr; j++) {
        const uint64_t h = splitmix64(seed ^ (((uint64_t)i << 9) | (uint64_t)j));
        o[w++] = (uint8_t)acgt[h & 3];
        if ((j % 80) == 79 || j == Lr - 1) o[w++] = '\n';
    }
    o[w++] = '+';
    if (rep) { for (int k = 1; k < 17; k++) o[w++] = o[k]; }
    o[w++] = '\n';
    for (int j = 0; j < Lr; j++) {
        const uint64_t h = splitmix64(seed ^ (((uint64_t)i << 9) | (uint64_t)j));
        o[w++] = (uint8_t)(33 + (h >> 8) % 41);
        if ((j % 80) == 79 || j == Lr - 1) o[w++] = '\n';
    }
}

// =========================================================================
// self-test kernel: wave scan and newline mask against scalar code
// =========================================================================
__global__ void k_selftest(const uint8_t *__restrict__ bytes, uint32_t *__restrict__ bad)
{
    const int l = threadIdx.x & 63;
    const uint4 v = *reinterpret_cast<const uint4 *>(bytes + threadIdx.x * 16);
    const uint32_t m = nl_mask16(v);
    uint32_t ref = 0;
    for (int b = 0; b < 16; b++)
        if (bytes[threadIdx.x * 16 + b] == '\n') ref |= 1u << b;
    if (m != ref) atomicAdd(bad, 1u);
    for (uint32_t q = 0; q < 16; q++)
        if (get_byte(v, q) != bytes[threadIdx.x * 16 + q]) atomicAdd(bad, 1u);
    const uint32_t c = __popc(m) | ((uint32_t)(l + 1) << 16);
    const uint32_t s = wave_incl_scan(c);
    uint32_t e0 = 0, e1 = 0;
    for (int j = 0; j <= l; j++) {
        e0 += __popc((uint32_t)__shfl((int)m, j));
        e1 += j + 1;
    }
    if ((s & 0xFFFF) != e0 || (s >> 16) != e1) atomicAdd(bad, 1u);
}

}  // namespace ffq
