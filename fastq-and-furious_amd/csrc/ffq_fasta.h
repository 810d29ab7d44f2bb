// ffq_fasta.h -- FASTA records on the device (widening row).
//
// The reference's scanner (/root/reference/src/fastqandfurious.py:103-143, entrypos_fasta) finds
// "\n>" from `offset`, the end of that header line, and the next "\n>" searched from the byte
// AFTER the header's newline.  Repeated with offset := pos[3] it yields every entry of a buffer.
// Over the line index (built with '>' in place of '@' as the AT character) that chain is local:
// a "\n>" entry starts a record unless the entry right before it is itself a record start (the
// search for the end of a sequence skips the header's own newline, :133) -- so inside a run of
// consecutive "\n>" entries every other one is a start, counted from the run's first entry at
// or after `offset` (its parity comes from a wave-wide look at the entries in front, 64 per step).
// No speculation, no chain walk:
//   k_fa_count   one wave per tile: record starts per tile
//   k_scan_i64   exclusive scan of those counts (ffq_kernels.h)
//   k_fa_rows    one wave per tile: pos0, pos1, pos2 of every start at its rank, and pos3 (= pos0 of the
//                next start, minus one) of every start but the tile's last
//   k_fa_fix     pos3 of each tile's last start, from the row behind it (one thread per tile); status and
//                posbuffer of the last start, which the buffer's end cuts short (it is never COMPLETE)
#pragma once
#include "ffq_chain.h"

namespace ffq {

struct FaHdr {
    long long n_starts;
    long long last_p0, last_p1;      // buffer coordinates of the last start's '>' and header end (-1: none)
    int32_t last_has_next;           // the entry after the last start's "\n>" exists
    int32_t pad;
};

// flags + position of entry j of tile t (c = its count); tile -1 is the sentinel
__device__ __forceinline__ uint32_t fa_entry(const LineIndex &L, int t, int j, uint32_t c)
{
    return (c <= (uint32_t)SLOT) ? (uint32_t)L.ent[(int64_t)t * SLOT + j] : L.pooled(t, (uint32_t)j);
}

// is entry j of tile t (count c) an eligible "\n>": the flag, at buffer coordinate >= offset
__device__ __forceinline__ bool fa_eligible(const LineIndex &L, int64_t offset, int t, uint32_t j, uint32_t c)
{
    if (j >= c) return false;
    const uint32_t e = fa_entry(L, t, (int)j, c);
    const int64_t P = ((int64_t)t << TILE_SHIFT) + (e & OFF_MASK) + L.s;
    return ((e >> 14) & FL_AT) && P >= offset;
}

// Parity of the run of eligible "\n>" entries that ends right in front of tile t's first entry
// (wave-uniform; the wave looks at 64 entries per step, so a run of n entries costs n / 64 steps
// per tile -- a lane-by-lane walk back from every entry was quadratic in the run's length).
__device__ int fa_run_parity_before(const LineIndex &L, int64_t offset, int t)
{
    const int lane = threadIdx.x & 63;
    int par = 0;
    for (int tt = t - 1;; tt--) {
        while (tt >= 0 && L.cnt[tt] == 0) tt--;
        if (tt < 0) {
            // before tile 0 there is only the sentinel (coordinate 0)
            if (L.s && L.n > 0 && L.d[0] == '>' && 0 >= offset) par ^= 1;
            return par;
        }
        const uint32_t c = L.cnt[tt];
        for (int64_t jend = c; jend > 0; jend -= 64) {
            const int nvalid = (int)min((int64_t)64, jend);
            // lane i looks at entry jend - 1 - i: lane 0 is the entry nearest to tile t
            const bool at = lane < nvalid && fa_eligible(L, offset, tt, (uint32_t)(jend - 1 - lane), c);
            const unsigned long long m = __ballot(at);
            const int k = (~m == 0ull) ? 64 : (__ffsll((long long)~m) - 1);      // consecutive ones from lane 0
            par ^= min(k, nvalid) & 1;
            if (k < nvalid) return par;
        }
    }
}

// Round 5: the two kernels below were chains of dependent memory round trips per wave (the tile's count, the count of the
// tile in front, its entries, then the tile's own entries 64 at a time: seven of them for the usual tile, 80 us for the
// row kernel of a GiB whatever it held).  Now two: the counts of tile t and t - 1, then -- together -- the last 64 entries
// of tile t - 1 (where the parity of the run in front is usually decided at once) and the tile's first 384 entries.
constexpr int FA_PRE = 6;        // chunks of 64 entries loaded up front: 384 entries = 16 KiB of lines of 43 bytes or more (60-column FASTA: 268)
struct FaPre {
    uint32_t c, cp;            // entries of tile t, of tile t - 1
    uint32_t e[FA_PRE];        // entries 64 q + lane of tile t (q < FA_PRE), as stored; valid where the tile is not pooled
    uint32_t ep;               // entry cp - 1 - lane of tile t - 1
    bool own, prev;            // e[] / ep were loaded (tiles within their slots)
};

__device__ __forceinline__ FaPre fa_preload(const LineIndex &L, int t)
{
    const int lane = threadIdx.x & 63;
    FaPre p;
    p.c = L.cnt[t];
    p.cp = t > 0 ? L.cnt[t - 1] : 0u;
    asm volatile("" ::"v"(p.c), "v"(p.cp));
    p.own = p.c <= (uint32_t)SLOT;
    p.prev = t > 0 && p.cp > 0u && p.cp <= (uint32_t)SLOT;
    const uint16_t *src = L.ent + (int64_t)t * SLOT;
#pragma unroll
    for (int q = 0; q < FA_PRE; q++) p.e[q] = p.own ? (uint32_t)src[min(64 * q + lane, SLOT - 1)] : 0u;
    p.ep = p.prev ? (uint32_t)L.ent[(int64_t)(t - 1) * SLOT + max((int)p.cp - 1 - lane, 0)] : 0u;
    asm volatile("" ::"v"(p.e[0]), "v"(p.e[1]), "v"(p.e[2]), "v"(p.e[3]), "v"(p.e[4]), "v"(p.e[5]), "v"(p.ep));
    static_assert(FA_PRE == 6, "the pin above names every element");
    return p;
}

// fa_run_parity_before from the preloaded entries where they decide it (a run that ends inside tile t - 1's last 64
// entries: always, but for runs of dozens of empty headers), the walk otherwise
__device__ __forceinline__ int fa_run_parity_pre(const LineIndex &L, int64_t offset, int t, const FaPre &p)
{
    if (p.prev) {
        const int lane = threadIdx.x & 63;
        const int nvalid = (int)min(64u, p.cp);
        const int64_t P = ((int64_t)(t - 1) << TILE_SHIFT) + (p.ep & OFF_MASK) + L.s;
        const bool at = lane < nvalid && ((p.ep >> 14) & FL_AT) && P >= offset;
        const unsigned long long m = __ballot(at);
        const int k = (~m == 0ull) ? 64 : (__ffsll((long long)~m) - 1);
        if (k < nvalid) return k & 1;
    }
    return fa_run_parity_before(L, offset, t);
}

// Starts among entries [j0, j0 + 64) of tile t: a "\n>" entry starts a record iff the run of
// eligible entries right in front of it has even length.  carry = parity of the run that ends in
// front of j0 (updated for the next chunk).  Returns this lane's answer.
__device__ __forceinline__ bool fa_chunk_starts_of(bool at, uint32_t j0, uint32_t c, int &carry);
__device__ __forceinline__ bool fa_chunk_starts(const LineIndex &L, int64_t offset, int t, uint32_t j0, uint32_t c,
                                                int &carry)
{
    return fa_chunk_starts_of(fa_eligible(L, offset, t, j0 + (uint32_t)(threadIdx.x & 63), c), j0, c, carry);
}
// (at: this lane's entry j0 + lane is an eligible "\n>")
__device__ __forceinline__ bool fa_chunk_starts_of(bool at, uint32_t j0, uint32_t c, int &carry)
{
    const int lane = threadIdx.x & 63;
    const int nvalid = (int)min((uint32_t)64, c - j0);
    const unsigned long long m = __ballot(at);
    // eligible entries right below this lane's bit
    int k = 0;
    if (lane > 0) {
        const unsigned long long y = m << (64 - lane);          // bit lane-1 on top
        k = (~y == 0ull) ? 64 : __clzll((long long)~y);
        k = min(k, lane);
    }
    const bool st = at && (((k + (k == lane ? carry : 0)) & 1) == 0);
    // the run at the end of this chunk
    const unsigned long long z = m << (64 - nvalid);
    int k2 = (~z == 0ull) ? 64 : __clzll((long long)~z);
    k2 = min(k2, nvalid);
    carry = (k2 == nvalid) ? (carry ^ (nvalid & 1)) : (k2 & 1);
    return st;
}

__global__ __launch_bounds__(256) void k_fa_count(LineIndex L, int64_t offset, unsigned int *__restrict__ cnt_start)
{
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wid;
    if (t >= L.ntiles) return;
    const FaPre pre = fa_preload(L, t);
    const uint32_t c = pre.c;
    uint32_t n = 0;
    int carry = c ? fa_run_parity_pre(L, offset, t, pre) : 0;
    const int64_t tbase = ((int64_t)t << TILE_SHIFT) + L.s;
#pragma unroll
    for (int q = 0; q < FA_PRE; q++) {
        if (pre.own && (uint32_t)(64 * q) < c) {
            const uint32_t j = (uint32_t)(64 * q + lane);
            const bool at = j < c && ((pre.e[q] >> 14) & FL_AT) && tbase + (int64_t)(pre.e[q] & OFF_MASK) >= offset;
            n += (uint32_t)__popcll(__ballot(fa_chunk_starts_of(at, (uint32_t)(64 * q), c, carry)));
        }
    }
    for (uint32_t j0 = pre.own ? (uint32_t)(64 * FA_PRE) : 0u; j0 < c; j0 += 64) {
        const bool st = fa_chunk_starts(L, offset, t, j0, c, carry);
        n += (uint32_t)__popcll(__ballot(st));
    }
    (void)lane;
    // the sentinel (a virtual "\n" at coordinate 0) belongs to tile 0's count
    if (t == 0 && L.s && L.n > 0 && L.d[0] == '>' && offset <= 0) n += 1;
    if (lane == 0) cnt_start[t] = n;
}

__global__ __launch_bounds__(256) void k_fa_rows(LineIndex L, int64_t offset, int64_t add,
                                                 const long long *__restrict__ base,
                                                 const long long *__restrict__ total, int64_t *__restrict__ table,
                                                 int64_t table_cap, FaHdr *hdr, const unsigned int *__restrict__ cnt_start)
{
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wid;
    if (t >= L.ntiles) return;
    const long long ntot = *total;
    long long rank = base[t];
    const unsigned int mine = cnt_start[t];
    if (t == 0 && lane == 0) hdr->n_starts = ntot;
    if (mine == 0u) return;               // (no start in this tile -- most tiles of a file of long entries: nothing to write)
    const FaPre pre = fa_preload(L, t);
    const uint32_t c = pre.c;
    __shared__ __attribute__((aligned(16))) int64_t s_rows_all[4][64 * 6];       // a chunk's rows, compact
    int64_t *s_rows = s_rows_all[wid];
    long long pend = -1;                  // rank of the start whose pos3 is still open (wave-uniform)
    const int64_t len = L.len();
    // the sentinel start (rank 0 of tile 0)
    const bool sent_start = (t == 0 && L.s && L.n > 0 && L.d[0] == '>' && offset <= 0);
    if (sent_start) {
        if (lane == 0) {
            // header end = the first newline of the data = entry 0 of the first non-empty tile
            int tn = 0;
            while (tn < L.ntiles && L.cnt[tn] == 0) tn++;
            const bool has = tn < L.ntiles;
            const int64_t p1 = has ? ((int64_t)tn << TILE_SHIFT) + (fa_entry(L, tn, 0, L.cnt[tn]) & OFF_MASK) + L.s : -1;
            if (rank < table_cap) {
                int64_t *o = table + rank * 6;
                o[0] = 1 + add; o[1] = p1 + add; o[2] = p1 + 1 + add; o[4] = -1; o[5] = -1;
            }
            if (rank == ntot - 1) { hdr->last_p0 = 1; hdr->last_p1 = p1; hdr->last_has_next = has ? 1 : 0; }
        }
        pend = rank;
        rank += 1;
    }
    int carry = c ? fa_run_parity_pre(L, offset, t, pre) : 0;
    const int64_t tbase = ((int64_t)t << TILE_SHIFT) + L.s;
    for (uint32_t j0 = 0; j0 < c; j0 += 64) {
        const uint32_t j = j0 + lane;
        // this lane's entry, once: eligibility, position, and (through the neighbour lane) the header's end
        uint32_t e;
        if (pre.own && j0 < (uint32_t)(64 * FA_PRE)) {
            const uint32_t q = j0 >> 6;
            e = q == 0 ? pre.e[0] : q == 1 ? pre.e[1] : q == 2 ? pre.e[2] : q == 3 ? pre.e[3] : q == 4 ? pre.e[4] : pre.e[5];
            if (j >= c) e = 0u;
        }
        else e = (j < c) ? fa_entry(L, t, (int)j, c) : 0u;
        const int64_t Pe = tbase + (e & OFF_MASK);
        const bool st = fa_chunk_starts_of(j < c && ((e >> 14) & FL_AT) && Pe >= offset, j0, c, carry);
        const int64_t P = st ? Pe : 0;
        const unsigned long long m = __ballot(st);
        // pos3 of a start = the newline in front of the NEXT start's '>': the next start of this chunk (its P from that
        // lane); the chunk's last start waits for the first start of a later chunk (pend = its rank, wave-uniform)
        const unsigned long long above = (lane == 63) ? 0ull : (m & ~((2ull << lane) - 1ull));
        const int nl = above ? __ffsll((long long)above) - 1 : lane;
        const int64_t Pn = ((int64_t)__shfl((int)(P >> 32), nl) << 32) | (uint32_t)__shfl((int)(uint32_t)P, nl);
        const uint32_t e_next = (uint32_t)__shfl((int)e, min(lane + 1, 63));       // entry j + 1, if it is in this chunk
        if (m != 0ull && pend >= 0) {
            const int fl = __ffsll((long long)m) - 1;
            if (lane == fl && pend < table_cap) table[pend * 6 + 3] = P + add;
            pend = -1;
        }
        const int below = __popcll(m & ((1ull << lane) - 1ull));
        if (st) {
            const long long r = rank + below;
            // header end: the next entry (the following tiles when this is the tile's last)
            int64_t p1;
            bool has = true;
            if (lane < 63 && j + 1 < c) p1 = tbase + (e_next & OFF_MASK);
            else {
                int tn = t, jn = (int)j + 1;
                if (jn >= (int)c) {
                    tn = t + 1; jn = 0;
                    while (tn < L.ntiles && L.cnt[tn] == 0) tn++;
                    has = tn < L.ntiles;
                }
                p1 = has ? ((int64_t)tn << TILE_SHIFT) + (fa_entry(L, tn, jn, L.cnt[tn]) & OFF_MASK) + L.s : -1;
            }
            // the row, compact in LDS (row `below` of this chunk)
            int64_t *o = s_rows + below * 6;
            o[0] = P + 1 + add; o[1] = p1 + add; o[2] = p1 + 1 + add; o[3] = Pn + add; o[4] = -1; o[5] = -1;
            if (r == ntot - 1) { hdr->last_p0 = P + 1; hdr->last_p1 = p1; hdr->last_has_next = has ? 1 : 0; }
        }
        wave_sync();
        // rows of a chunk are consecutive in the table: 16-byte pieces, consecutive lanes -> consecutive pieces.  The piece
        // that holds pos3 of the chunk's LAST start (still open) is not written: its pos2 goes out alone.
        const int nst = __popcll(m);
        for (int q = lane; q < 3 * nst; q += 64) {
            const int row = q / 3, part = q - 3 * row;
            if (rank + row >= table_cap) continue;
            const longlong2 vv = *reinterpret_cast<const longlong2 *>(s_rows + row * 6 + 2 * part);
            int64_t *dst = table + (rank + row) * 6 + 2 * part;
            if (row == nst - 1 && part == 1) dst[0] = vv.x;
            else *reinterpret_cast<longlong2 *>(dst) = vv;
        }
        wave_sync();
        if (m != 0ull) pend = rank + nst - 1;          // the chunk's last start
        rank += nst;
    }
    // the tile's last start: its successor lies in a later tile (k_fa_fix), or there is none
    if (pend >= 0 && lane == 0 && pend < table_cap) table[pend * 6 + 3] = -1;
    (void)len;
}

// pos3 of every COMPLETE entry, and the result block
__global__ __launch_bounds__(256) void k_fa_fix(LineIndex L, int64_t offset, int64_t add, const FaHdr *__restrict__ hdr,
                                                const unsigned int *__restrict__ cnt_start, const long long *__restrict__ base,
                                                int64_t *__restrict__ table, int64_t table_cap, DevRes *res, Pub pb)
{
    const long long ntot = hdr->n_starts;
    const long long ncomplete = ntot > 0 ? ntot - 1 : 0;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // a tile
    if (i < L.ntiles && cnt_start[i] > 0) {
        const long long r = base[i] + (long long)cnt_start[i] - 1;             // the tile's last start
        if (r < ncomplete && r + 1 < table_cap) table[r * 6 + 3] = table[(r + 1) * 6 + 0] - 1;
    }
    if (i != 0) return;
    // block 0 thread 0: the last call of the chain (never COMPLETE: no "\n>" follows it)
    const int64_t len = L.len();
    res->fallback = 0;
    res->n_records = ncomplete;
    res->n_qual_bytes = 0;
    res->has_final = 0;
    res->term_group = -1;
    res->end_state = 0;
    for (int q = 0; q < 6; q++) res->last_pos[q] = -1;
    if (ntot == 0) {
        res->last_status = ST_HEAD_BEG;
        res->end_offset = offset;
    } else {
        const int64_t p0 = hdr->last_p0, p1 = hdr->last_p1;
        res->end_offset = p0 - 1;                       // the "\n>" the last call matched
        res->last_pos[0] = p0 + add;
        if (!hdr->last_has_next) res->last_status = ST_HEAD_END;
        else {
            res->last_pos[1] = p1 + add;
            if (p1 + 1 >= len) res->last_status = ST_SEQ_BEG;
            else {
                res->last_pos[2] = p1 + 1 + add;
                // buf[-1] == '\n' ? len - 1 : len  (:137-140); coordinate len-1 is data byte len-1-s
                const uint8_t lastb = L.d[L.n - 1];
                res->last_pos[3] = ((lastb == '\n') ? len - 1 : len) + add;
                res->last_status = ST_SEQ_END;
            }
        }
    }
    // the last COMPLETE row's pos3 when the table could not hold the row after it
    if (ncomplete > 0 && ncomplete - 1 < table_cap && ncomplete >= table_cap)
        table[(ncomplete - 1) * 6 + 3] = hdr->last_p0 - 1 + add;
    publish(pb, res);
}

}  // namespace ffq
