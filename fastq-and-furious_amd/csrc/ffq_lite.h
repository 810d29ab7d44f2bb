// ffq_lite.h -- the record chain of an ORDINARY group of tiles, with a fraction of k_chain_wave's instructions.
//
// k_chain_wave (ffq_chain.h) is bound by instruction issue: 2390 VALU instructions per wave and group on S-wrapped,
// half of them spent on being general (window words that carry a node id merged per entry, passes for tiles of any
// density, search offsets, the sentinel, buffer ends, forced entries, restarts, pointer doubling, records of any
// length).  Nearly every group needs none of that.  k_chain_lite takes a group ONLY IF it is ordinary -- not the first
// one (sentinel, search offset), its whole window [run-in tile | OWN_T own tiles | look-ahead tile] inside the buffer
// with room to spare (so that no buffer-end rule of the scanner can apply), every tile of it one pass of six entries
// per lane -- computes exactly what k_chain_wave computes for it (same nodes, same scanner rules in the same order:
// /root/reference/src/_fastqandfurious.c:25-153, same summary), and DECLINES (flag bit 3) the moment anything else
// turns up on the chain: a node whose call does not resolve within the LT_B entries behind it (a record of many lines,
// an INVALID '+' line, a successor far away), a chain that stops, too many nodes.  Declined groups are run by
// k_chain_wave right behind (only_deferred == 3); repair passes are k_chain_wave's as before.
//   window:  the entries as they are stored, 2 bytes each, six per lane and tile, one pass each
//   nodes:   the "\n@" matches of the run-in tail and the own tiles, numbered in entry order
//   calls:   a thread per node, from the LT_B entries behind its own (distances modulo the tile size)
//   chain:   run by run from the window's first node (a run = nodes whose successor is the very next node: one
//            ballot per 64), records of the own tiles staged by all lanes of a run at once
#pragma once
#include "ffq_chain.h"

namespace ffq {

constexpr int LT_E = 384;                  // entries of a tile the kernel takes: one pass of six per lane
constexpr int LT_LA = 64;                  // entries of the look-ahead tile it looks at
constexpr int LT_WIN = (OWN_T + 1) * LT_E + LT_LA;
constexpr int LT_NODES = 320;
constexpr int LT_B = 14;                   // words a node's call looks at, its own included
constexpr uint32_t LK_OK = 0, LK_GEN = 1, LK_SLOW = 2;  // node word: successor node (10 bits, NO_NODE: past the own tiles) | kind << 10 | mi << 12 | sj << 16
constexpr uint32_t FLAG_LITE_DECLINED = 8u;

__device__ __forceinline__ void lite_decline(const ChainBufs &B, int g, int why = 0)
{
    if (PROBES && B.prof) atomicAdd(&B.prof[8 + why], 1ull);      // (instrumented build: why groups are declined)
    B.flags[g] = FLAG_LITE_DECLINED;
    B.dlist[atomicAdd(B.dcnt, 1u)] = (uint32_t)g;
}

// LDS per wave: the window's entries as they are stored (16 bits each: offset in the tile | flags << 14), tile after tile
// with no gap, + the node list (entry index | tile of the window << 11 | "the tile ends within LT_B entries" << 14).
// Positions inside ONE scanner call are taken relative to the call's "\n@" entry, modulo the tile size: the LT_B entries a
// call looks at span less than a tile (checked per node), so (off_j - off_0) & (TILE - 1) is the exact distance whether or
// not a tile boundary lies in between -- no per-entry word has to be built, the window is 2 bytes per entry, and eight
// waves per SIMD fit where the word window allowed four.
// the first group whose window does not lie at least four bytes in front of the buffer's end (host side: those groups, and
// group 0, are launched as k_chain_wave's from the start)
static inline int lite_first_end_group(int64_t n_bytes, int64_t ntiles, int ngroups)
{
    int gl = ngroups;
    while (gl > 1) {
        const int g = gl - 1;
        const bool fits = (int64_t)g * OWN_T + OWN_T + 1 <= ntiles && (((int64_t)g * OWN_T + OWN_T + 1) << TILE_SHIFT) + 4 < n_bytes;
        if (fits) break;
        gl--;
    }
    return gl;
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void k_chain_lite(LineIndex L, int64_t offset, ChainBufs B, int ng, int gl, int ablate)
{
    __shared__ __attribute__((aligned(4))) uint16_t raw_all[WPB][LT_WIN + 16];
    __shared__ uint16_t nidx_all[WPB][LT_NODES + 4];
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int g = blockIdx.x * WPB + wid;
    if (g >= ng) return;
    uint16_t *raw = raw_all[wid];
    uint16_t *nidx = nidx_all[wid];
    const int own0 = g * OWN_T, own1 = own0 + OWN_T, wt0 = own0 - 1;
    const int64_t wpos0 = (int64_t)wt0 << TILE_SHIFT;
    // not the first group (sentinel, search offset), not the last ones (every position of the window lies at least four
    // bytes in front of the buffer's end: none of the scanner's buffer-end rules can apply), the search offset in front
    // of the run-in tail (every "\n@" there is a candidate)
    // (the first group and those from gl on -- lite_first_end_group -- are k_chain_wave's: on the list of declined groups
    // with the rest; round 4 gave them two launches of their own in front of this kernel, the first of which -- an ordinary
    // launch, one workgroup, 29 us -- ran with the GPU otherwise idle)
    if (g == 0 || g >= gl) { if (lane == 0) lite_decline(B, g, 0); return; }
    bool ok = own1 + 1 <= L.ntiles && (((int64_t)(wt0 + OWN_T + 2) << TILE_SHIFT) + L.s + 4 < L.len()) &&
              offset <= wpos0 + L.s + (TILE - RUNIN_BYTES);
    if (!ok) { if (lane == 0) lite_decline(B, g, 0); return; }
    // ---- the window's entries, one memory round trip --------------------------------------------------------------
    const uint32_t cl = (lane < OWN_T + 2) ? L.cnt[wt0 + lane] : 0u;
    uint32_t ev[OWN_T + 1][3];
#pragma unroll
    for (int k = 0; k <= OWN_T; k++) {
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
        const u32x3 t = *reinterpret_cast<const u32x3 *>(L.ent + (int64_t)(wt0 + k) * SLOT + 6 * lane);
        ev[k][0] = t.x; ev[k][1] = t.y; ev[k][2] = t.z;
    }
    const uint32_t la_raw = (uint32_t)L.ent[(int64_t)(wt0 + OWN_T + 1) * SLOT + lane];
    int tc[OWN_T + 2], tb[OWN_T + 2];
    uint32_t lines = 0;
    tb[0] = 0;
#pragma unroll
    for (int k = 0; k <= OWN_T + 1; k++) {
        tc[k] = (int)__builtin_amdgcn_readlane((int)cl, k);
        if (k <= OWN_T) { tb[k + 1] = tb[k] + tc[k]; if (tc[k] > LT_E) ok = false; }
        if (k >= 1 && tc[k] < LT_B) ok = false;                      // (a call's entries cross one tile boundary at most)
        if (k >= 1 && k <= OWN_T) lines += (uint32_t)tc[k];
    }
    bool dense = false;
#pragma unroll
    for (int k = 0; k <= OWN_T + 1; k++) dense = dense || tc[k] > SLOT;
    if (dense) {
        // a DENSE tile in the window: k_chain_wave would only find that the group does not fit it -- say so at once (flag
        // bit 0, on the walker's list: ffq_dense.h), as that kernel does
        if (lane == 0) {
            B.y[g] = Y_UNRES; B.exit[g] = Y_UNRES; B.cnt[g] = 0; B.qb[g] = 0; B.flags[g] = 1; B.lines[g] = lines;
            B.ilist[atomicAdd(B.icnt, 1u)] = (uint32_t)g;
        }
        return;
    }
    if (!ok) { if (lane == 0) lite_decline(B, g, 1); return; }
    const int lac = min(tc[OWN_T + 1], LT_LA);
    const int own_hi = tb[OWN_T + 1], nwin = own_hi + lac;
    // ---- entries -> LDS as they are; nodes (the "\n@" matches of the run-in tail and the own tiles) numbered ------------
    int ncomp = 0, n_runin = 0;
#pragma unroll
    for (int k = 0; k <= OWN_T; k++) {
        const int c = tc[k];
        const int nv = min(max(c - 6 * lane, 0), 6);
        uint32_t at = 0;
#pragma unroll
        for (int h = 0; h < 3; h++) at |= (((ev[k][h] >> 14) & 1u) | ((ev[k][h] >> 29) & 2u)) << (2 * h);
        at &= (1u << nv) - 1u;
        if (k == 0) {
            // the run-in tile: only its tail makes candidates (offset >= TILE - RUNIN_BYTES: bit 13 of the offset)
            static_assert(RUNIN_BYTES * 2 == TILE, "the tail test reads one bit of the offset");
            uint32_t tailm = 0;
#pragma unroll
            for (int h = 0; h < 3; h++) tailm |= (((ev[k][h] >> 13) & 1u) | ((ev[k][h] >> 28) & 2u)) << (2 * h);
            at &= tailm;
        }
        const uint32_t nc = __popc(at);
        const uint32_t incl = wave_incl_scan(nc);
        uint32_t id = (uint32_t)ncomp + incl - nc;
        ncomp += __builtin_amdgcn_readlane((int)incl, 63);            // (a scalar: what is decided on it below is decided once per wave)
        if (6 * lane < c) {
            // (what lies past the tile's count is overwritten by the next tile's entries, written later, or lies past nwin)
            uint16_t *dst = raw + tb[k] + 6 * lane;
#pragma unroll
            for (int h = 0; h < 3; h++) { dst[2 * h] = (uint16_t)ev[k][h]; dst[2 * h + 1] = (uint16_t)(ev[k][h] >> 16); }
            uint32_t mrem = at;
            while (mrem) {
                const int i = __ffs((int)mrem) - 1;
                mrem &= mrem - 1u;
                const int at_i = 6 * lane + i;
                nidx[min(id, (uint32_t)LT_NODES)] = (uint16_t)((tb[k] + at_i) | (k << 11) | ((c - at_i < LT_B) ? 0x4000 : 0));
                id++;
            }
        }
        if (k == 0) n_runin = ncomp;
    }
    if (lane < lac) raw[own_hi + lane] = (uint16_t)la_raw;
    if (ncomp == 0 || ncomp > LT_NODES) { if (lane == 0) lite_decline(B, g, 2); return; }     // (no candidate: the chain passes over -- a search of its own)
    if (lane < 3) nidx[ncomp + lane] = 0x7FF;                 // behind the last node: no entry index any successor could have
    wave_sync();
    if (PROBES && ablate == 1) { if (lane == 0) B.lines[g] = lines + (uint32_t)ncomp; return; }      // (instruction counts per phase: tools/pmc_insts.sh)
    // ---- one scanner call per node ------------------------------------------------------------------------------------
    // A record of this kernel's kind: header line, mi - 1 sequence lines, the '+' line, as many quality lines -- entry k + mi
    // is the "\n+" match, k + mi + 1 the '+' line's end, and the next call's "\n@" (the first one at >= pos5 - 1 behind the
    // '+' line's end, :62 / fastqandfurious.py:254) is entry k + 2 mi: that is what is TESTED (the entry in front of it must
    // lie in front of pos5 - 1: positions grow with the index, so no earlier entry can be the match); a node whose call
    // looks different (another wrapping of the qualities, an INVALID '+' line, more than six sequence lines) is not taken
    constexpr int NB = LT_NODES / 64;
    constexpr uint32_t TM = (uint32_t)TILE - 1u;
    uint32_t infr[NB], kreg[NB];           // node u * 64 + lane: its word (successor | kind | mi) and its list entry
    unsigned long long NS[NB];             // (a run = nodes whose successor is the very next node; NS = the nodes that end one)
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const int c = u * 64 + lane;
        uint32_t inf = LK_GEN << 10, kw = 0;
        if (u * 64 < ncomp && c < ncomp) {                     // (the first test is wave-uniform)
            kw = nidx[c];
            const int k = (int)(kw & 0x7FFu);
            if (k + LT_B < nwin) {
                uint32_t r[8];
#pragma unroll
                for (int i = 0; i < 8; i++) r[i] = raw[k + i];
                const uint32_t r13 = raw[k + LT_B - 1];
                // distances from the call's own entry, modulo the tile: exact if the LT_B entries span less than a tile --
                // no tile boundary among them, or the last one's offset already below the first one's
                const bool span_ok = !(kw & 0x4000u) || (r13 & TM) < (r[0] & TM);
                const uint32_t P1 = (r[1] - r[0]) & TM;
                // "\n+" at >= seq_beg + 1 (:87-88): entry 2 must lie two bytes behind the header's end; entries 3 .. 7 do
                // anyway (positions grow by at least one per entry) -- for them the flag alone
                static_assert(FL_PLUS == 2, "the '+' flag is bit 15 of an entry");
                uint32_t pm = (((r[2] >> 15) & 1u) && ((r[2] - r[0]) & TM) >= P1 + 2u) ? 4u : 0u;
#pragma unroll
                for (int i = 3; i <= 7; i++) pm |= ((r[i] >> 15) & 1u) << i;
                if (pm && span_ok) {
                    const int mi = __ffs((int)pm) - 1;        // 2 .. 7: 2 mi <= LT_B
                    const uint32_t P3 = (raw[k + mi] - r[0]) & TM, Pq = (raw[k + mi + 1] - r[0]) & TM;
                    const uint32_t rs = raw[k + 2 * mi], rp = raw[k + 2 * mi - 1];
                    const bool invalid = (Pq - P3 - 1u > 1u) && (Pq - P3 != P1);           // :109-117 (head_end - pos0 + 1 = P1)
                    const uint32_t qe = Pq + P3 - P1;                                       // :129
                    const bool succ_ok = ((rs >> 14) & (uint32_t)FL_AT) && ((rs - r[0]) & TM) + 1u >= qe &&
                                         (mi == 2 || ((rp - r[0]) & TM) + 1u < qe);
                    const int sj = 2 * mi;
                    // not the usual shape (qualities wrapped otherwise than the read, a line too many or too few -- and nearly
                    // every FALSE candidate, which no chain visits): the rule itself, entry by entry, is applied by the whole
                    // wave IF the chain gets there (LK_SLOW, below) -- as a loop in here every wave paid for it in every batch
                    if (!invalid && !succ_ok) inf = (LK_SLOW << 10) | ((uint32_t)mi << 12);
                    if (!invalid && succ_ok) {
                        // the successor as a node: one of the next three (every "\n@" of the own tiles is one), or an entry
                        // of the look-ahead tile
                        const uint32_t t = (uint32_t)(k + sj);
                        uint32_t nx = 0xFFFFu;
                        if ((nidx[c + 3] & 0x7FFu) == t) nx = (uint32_t)(c + 3);
                        if ((nidx[c + 2] & 0x7FFu) == t) nx = (uint32_t)(c + 2);
                        if ((nidx[c + 1] & 0x7FFu) == t) nx = (uint32_t)(c + 1);
                        if (t >= (uint32_t)own_hi) nx = NO_NODE;
                        if (nx != 0xFFFFu) inf = nx | (LK_OK << 10) | ((uint32_t)mi << 12) | ((uint32_t)sj << 16);
                    }
                }
            }
        }
        infr[u] = inf; kreg[u] = kw;
        NS[u] = __ballot(!(((inf >> 10) & 3u) == LK_OK && (inf & WN_MASK) == (uint32_t)(c + 1)));
    }
    if (PROBES && ablate == 2) { uint32_t x = 0; for (int u = 0; u < NB; u++) x += infr[u] + kreg[u]; if (x == 0x12345u) B.lines[g] = x; return; }
    // ---- chain membership: run by run, on the scalar side (the chain only moves forward: batch after batch) -------------
    // (the walk only marks where a run of members starts and where it ends; runs are disjoint and in order, so the
    // members of a batch are (ends << 1) - starts, modulo 2^64 for a run that ends with the batch)
    // (cur only grows -- a successor lies behind its node, a restart behind the run -- so the walk ends by itself; a walk
    // that is over, well or badly, parks cur behind the last batch: one test per turn)
    constexpr int CUR_DONE = NB * 64;
    unsigned long long MB[NB];
    int cur = 0, lastn = -1;
    uint32_t last_inf = 0, last_k = 0;
    bool bad = false;
#pragma unroll
    for (int u = 0; u < NB; u++) {
        unsigned long long st = 0ull, en = 0ull;
        while ((cur >> 6) == u) {
            const int b = cur & 63;
            st |= 1ull << b;
            const unsigned long long ns = NS[u] >> b;
            if (!ns) { en |= 1ull << 63; cur = (u + 1) * 64; if (cur >= ncomp) { bad = true; cur = CUR_DONE; } break; }
            const int r = b + __ffsll((long long)ns) - 1;
            en |= 1ull << r;
            uint32_t ir = (uint32_t)__builtin_amdgcn_readlane((int)infr[u], r);
            if (PROBES && B.prof && lane == 0) atomicAdd(&B.prof[14], 1ull);
            if (((ir >> 10) & 3u) == LK_SLOW) {
                if (PROBES && B.prof && lane == 0) atomicAdd(&B.prof[13], 1ull);
                // the successor rule entry by entry, by the whole wave: the first "\n@" behind the '+' line's end at >= pos5 - 1
                const uint32_t kw_ = (uint32_t)__builtin_amdgcn_readlane((int)kreg[u], r);
                const int k = (int)(kw_ & 0x7FFu), mi = (int)((ir >> 12) & 15u), c = u * 64 + r;
                const uint32_t r0 = raw[k];
                const uint32_t P1 = ((uint32_t)raw[k + 1] - r0) & TM, P3 = ((uint32_t)raw[k + mi] - r0) & TM,
                               Pq = ((uint32_t)raw[k + mi + 1] - r0) & TM;
                const uint32_t qe = Pq + P3 - P1;
                const int j = mi + 2 + lane;
                const uint32_t rj = (j < LT_B) ? (uint32_t)raw[k + j] : 0u;
                const unsigned long long hit = __ballot(j < LT_B && ((rj >> 14) & (uint32_t)FL_AT) && ((rj - r0) & TM) + 1u >= qe);
                ir = LK_GEN << 10;
                if (hit) {
                    const int sj = mi + 2 + (__ffsll((long long)hit) - 1);
                    const uint32_t t = (uint32_t)(k + sj);
                    uint32_t nx = 0xFFFFu;
                    if (t >= (uint32_t)own_hi) nx = NO_NODE;
                    else {
                        const unsigned long long nm = __ballot(lane < 8 && c + 1 + lane < ncomp && (uint32_t)(nidx[c + 1 + lane] & 0x7FFu) == t);
                        if (nm) nx = (uint32_t)(c + 1 + (__ffsll((long long)nm) - 1));
                    }
                    if (nx != 0xFFFFu) ir = nx | (LK_OK << 10) | ((uint32_t)mi << 12) | ((uint32_t)sj << 16);
                }
            }
            if (((ir >> 10) & 3u) != LK_OK) {
                // a node this kernel does not take lies on the chain.  In the run-in that is a chain started at a false
                // candidate (a quality line that begins with '@': one group in twenty starts so) which led nowhere: start
                // again at the next candidate of the run-in -- the entry stays a guess, the verification decides (what was
                // walked so far lies in front of the own tiles: nothing of it is staged)
                if (u * 64 + r + 1 < n_runin) cur = u * 64 + r + 1;
                else { bad = true; cur = CUR_DONE; }
            }
            else if ((ir & WN_MASK) == NO_NODE) {               // the chain leaves the own tiles
                lastn = u * 64 + r; last_inf = ir; last_k = (uint32_t)__builtin_amdgcn_readlane((int)kreg[u], r);
                cur = CUR_DONE;
            } else cur = (int)(ir & WN_MASK);
        }
        MB[u] = (en << 1) - st;
    }
    if (PROBES && ablate == 3) { if (lane == 0) B.lines[g] = lines + (uint32_t)lastn + (uint32_t)MB[0]; return; }
    if (bad || lastn < 0) { if (lane == 0) lite_decline(B, g, 3); return; }
    // ---- records of the own tiles, staged in chain order -----------------------------------------------------------------
    uint32_t ntot = 0;
    unsigned long long OWN[NB];
#pragma unroll
    for (int u = 0; u < NB; u++) {
        // members at or behind node n_runin
        const int lo = n_runin - u * 64;
        OWN[u] = MB[u] & (lo <= 0 ? ~0ull : lo >= 64 ? 0ull : (~0ull << lo));
        ntot += (uint32_t)__popcll(OWN[u]);
    }
    if (ntot > (uint32_t)B.nmax) { if (lane == 0) lite_decline(B, g, 4); return; }
    StageRec *stg = B.stage + (int64_t)g * B.nmax;
    uint32_t nbase = 0, qsum = 0;
    int64_t Y = Y_UNRES;
    // window position of a node's own entry: its tile of the window, its offset there (+ the sentinel's shift)
    auto node_pos = [&](uint32_t kw_) -> uint32_t {
        return (((kw_ >> 11) & 7u) << TILE_SHIFT) + (uint32_t)L.s + ((uint32_t)raw[kw_ & 0x7FFu] & TM);
    };
#pragma unroll
    for (int u = 0; u < NB; u++) {
        if (!OWN[u]) continue;                                  // (wave-uniform)
        if (bit_of_lane(OWN[u], lane)) {
            const int k = (int)(kreg[u] & 0x7FFu);
            const int mi = (int)((infr[u] >> 12) & 15u);
            const uint32_t r0 = raw[k];
            // 8 bytes per record (StageRec8): pos0 counted from the own tiles' first byte -- the node lies in one of them,
            // window tile 1 .. OWN_T --, pos1 / pos3 / pos4 from pos0 (a call spans less than a tile)
            const uint32_t d0 = ((((kreg[u] >> 11) & 7u) - 1u) << TILE_SHIFT) + (r0 & TM);
            const uint32_t d1 = ((raw[k + 1] - r0) & TM) - 1u, d3 = ((raw[k + mi] - r0) & TM) - 1u, d4 = (raw[k + mi + 1] - r0) & TM;
            reinterpret_cast<uint2 *>(stg)[nbase + (uint32_t)bits_below_lane(OWN[u])] = make_uint2(d0 | (d1 << 16), d3 | (d4 << 16));
            qsum += d3 - d1 - 1u;
        }
        if (Y == Y_UNRES)
            Y = wpos0 + (int64_t)node_pos((uint32_t)__builtin_amdgcn_readlane((int)kreg[u], __ffsll((long long)OWN[u]) - 1));
        nbase += (uint32_t)__popcll(OWN[u]);
    }
    // the candidate the chain goes on with: entry k + sj of its last node
    const int kl = (int)(last_k & 0x7FFu);
    const int64_t EX = wpos0 + (int64_t)(node_pos(last_k) + (((uint32_t)raw[kl + (int)((last_inf >> 16) & 15u)] - (uint32_t)raw[kl]) & TM));
    if (Y == Y_UNRES) Y = EX;
    const uint32_t qtot = wave_sum_u32(qsum);                 // (a group's qualities are < 2^17 bytes)
    if (lane == 0) {
        B.y[g] = Y; B.exit[g] = EX; B.cnt[g] = ntot; B.qb[g] = (int64_t)qtot; B.lines[g] = lines;
        B.flags[g] = FLAG_STAGE8;
    }
}

}  // namespace ffq
