// ffq_ranked.h -- the record chain by list ranking: the tier for input whose records are long
// compared with a group of tiles (wrapped reads of kilobases and more), where a guessed chain
// entry inside a quality block never falls in with the true chain and the group kernels of
// ffq_chain.h have nothing to verify against.
//
// The chain of readfastq_iter (/root/reference/src/fastqandfurious.py:251-279) is a linked
// list: every "\n@" match of the buffer (a CANDIDATE) has exactly one successor -- the scanner
// call from it (/root/reference/src/_fastqandfurious.c:25-153) gives pos5, and the next search
// finds the first candidate at >= pos5 - 1 -- and the records are the list that starts at the
// first candidate at >= offset.  No speculation at all:
//
//   k_rk_count   candidates per tile                          (one wave per tile)
//   k_scan_i64v  exclusive scan -> candidate ordinals
//   k_rk_list    candidate c -> its line-index entry
//   k_rk_succ    ONE WAVE PER CANDIDATE: the scanner call and the successor search, with the wave-wide
//                searches of the serial walker (a few memory round trips whatever the record's
//                length); every candidate in parallel
//   k_rk_root    the list head; newlines of the buffer
//   k_rk_round   pointer doubling with rank marking, ceil(log2(candidates)) launches of one
//                thread per candidate: after round k every list member of rank < 2^(k+1) knows
//                its rank
//   k_rk_emit    members write their row at table[rank]; the member the list ends at writes the
//                result block
//
// Cost: ~64 bytes of scratch and a handful of index look-ups per CANDIDATE, not per byte: the
// longer the records, the cheaper (20 kb reads: 11 candidates per 40 KB).  Exact on any input the
// line index describes; the one-wave serial walker stays behind it as the last resort.
#pragma once

namespace ffq {

constexpr uint32_t RK_NONE = 0xFFFFFFFFu;

struct RankRec {
    int64_t p1, p3, p4;        // buffer coordinates (pos0 = candidate + 1, pos5 = p4 + p3 - p1 - 1)
    int32_t status;
    int32_t final_;
};

struct RankBufs {
    long long *tbase;          // [ntiles] candidates per tile -> exclusive prefix
    H *cand;                   // [nc] line-index entry of candidate c
    RankRec *rec;              // [nc]
    uint32_t *succ;            // [nc] successor candidate or RK_NONE
    uint32_t *S[2], *C[2];     // pointer doubling: node reached, steps taken
    uint32_t *D;               // rank or RK_NONE
    uint32_t *root;            // [1] the list head or RK_NONE
    int64_t nc;
};

// is the virtual sentinel a candidate (the stream starts with '@')
__device__ __forceinline__ int rk_sentinel_cand(const LineIndex &L)
{
    return (L.s && L.n > 0 && L.d[0] == '@') ? 1 : 0;
}

__device__ __forceinline__ uint32_t rk_entry(const LineIndex &L, int t, uint32_t c, uint32_t j)
{
    return (c <= (uint32_t)SLOT) ? (uint32_t)L.ent[(int64_t)t * SLOT + j] : L.pooled(t, j);
}

// the FL_AT bits of entries j .. j + 3 of a tile that fits its slot (j a multiple of 4: one 8-byte load), as bits 0 .. 3
__device__ __forceinline__ uint32_t rk_at4(const LineIndex &L, int t, uint32_t j)
{
    static_assert(FL_AT == 1, "the '@' flag is bit 14 of an entry");
    const uint2 w = *reinterpret_cast<const uint2 *>(L.ent + (int64_t)t * SLOT + j);
    return ((w.x >> 14) & 1u) | ((w.x >> 29) & 2u) | (((w.y >> 14) & 1u) << 2) | (((w.y >> 30) & 1u) << 3);
}

// AT entries among entries [0, upto) of tile t (wave-uniform)
__device__ __forceinline__ uint32_t rk_count_at(const LineIndex &L, int t, uint32_t upto)
{
    const int lane = threadIdx.x & 63;
    const uint32_t c = L.cnt[t];
    uint32_t n = 0;
    if (c <= (uint32_t)SLOT) {
        // four entries per lane: a tile of 80-column lines (~200 entries) is ONE memory round trip instead of four
        for (uint32_t j0 = 0; j0 < upto; j0 += 256) {
            const uint32_t j = j0 + 4u * (uint32_t)lane;
            uint32_t m = (j < upto) ? rk_at4(L, t, j) : 0u;
            if (j + 4u > upto) m &= (1u << (upto > j ? upto - j : 0u)) - 1u;
            n += (uint32_t)__popc(m);
        }
        return wave_sum_u32(n);
    }
    for (uint32_t j0 = 0; j0 < upto; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool at = j < upto && ((rk_entry(L, t, c, j) >> 14) & FL_AT);
        n += (uint32_t)__popcll(__ballot(at));
    }
    return n;
}

__global__ __launch_bounds__(256) void k_rk_count(LineIndex L, long long *__restrict__ tbase)
{
    const int t = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (t >= L.ntiles) return;
    uint32_t n = rk_count_at(L, t, L.cnt[t]);
    if (t == 0) n += (uint32_t)rk_sentinel_cand(L);
    if ((threadIdx.x & 63) == 0) tbase[t] = (long long)n;
}

__global__ __launch_bounds__(256) void k_rk_list(LineIndex L, RankBufs R)
{
    const int t = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= L.ntiles) return;
    long long base = R.tbase[t];
    if (t == 0 && rk_sentinel_cand(L)) {
        if (lane == 0) R.cand[0] = H{-1, 0};
        base = 1;
    }
    const uint32_t c = L.cnt[t];
    if (c <= (uint32_t)SLOT) {
        for (uint32_t j0 = 0; j0 < c; j0 += 256) {
            const uint32_t j = j0 + 4u * (uint32_t)lane;
            uint32_t m = (j < c) ? rk_at4(L, t, j) : 0u;
            if (j + 4u > c) m &= (1u << (c > j ? c - j : 0u)) - 1u;
            const uint32_t k = (uint32_t)__popc(m);
            const uint32_t incl = wave_incl_scan(k);
            long long o = base + (long long)(incl - k);
            while (m) {
                const int q = __ffs((int)m) - 1;
                m &= m - 1u;
                if (o < R.nc) R.cand[o] = H{t, (int32_t)(j + (uint32_t)q)};
                o++;
            }
            base += (long long)(uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        return;
    }
    for (uint32_t j0 = 0; j0 < c; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool at = j < c && ((rk_entry(L, t, c, j) >> 14) & FL_AT);
        const unsigned long long m = __ballot(at);
        if (at) {
            const long long o = base + __popcll(m & ((1ull << lane) - 1ull));
            if (o < R.nc) R.cand[o] = H{t, (int32_t)j};
        }
        base += __popcll(m);
    }
}

// the first "\n@" entry at buffer coordinate >= minP behind entry `from` (what wv_find(L, from, FL_AT, minP, ...) returns),
// with four entries per lane in the tile the position falls into when that tile fits its slot: the successor search of a
// candidate is one memory round trip there instead of up to four
__device__ __forceinline__ bool rk_find_at(const LineIndex &L, H from, int64_t minP, H &out, int64_t &Pout, int &flout)
{
    const int lane = threadIdx.x & 63;
    const int64_t tmin = (minP - L.s) >> TILE_SHIFT;
    if (from.tile >= 0 && tmin >= (int64_t)from.tile && tmin < (int64_t)L.ready) {
        const int t = (int)tmin;
        const uint32_t c = L.cnt[t];
        if (c <= (uint32_t)SLOT && c > 0) {
            const uint32_t i0 = (t == from.tile) ? (uint32_t)from.i + 1u : 0u;
            for (uint32_t j0 = i0 & ~3u; j0 < c; j0 += 256) {
                const uint32_t j = j0 + 4u * (uint32_t)lane;
                int hit = -1;
                uint32_t e_hit = 0;
                if (j < c) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(L.ent + (int64_t)t * SLOT + j);
                    const uint32_t e[4] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
#pragma unroll
                    for (int q = 3; q >= 0; q--) {
                        const int64_t P = ((int64_t)t << TILE_SHIFT) + (e[q] & OFF_MASK) + L.s;
                        if (j + (uint32_t)q < c && j + (uint32_t)q >= i0 && ((e[q] >> 14) & FL_AT) && P >= minP) { hit = q; e_hit = e[q]; }
                    }
                }
                const unsigned long long m = __ballot(hit >= 0);
                if (m) {
                    const int w = __ffsll((long long)m) - 1;
                    const int q = __shfl(hit, w);
                    const uint32_t e = (uint32_t)__shfl((int)e_hit, w);
                    out = H{t, (int32_t)(j0 + 4u * (uint32_t)w + (uint32_t)q)};
                    Pout = ((int64_t)t << TILE_SHIFT) + (e & OFF_MASK) + L.s;
                    flout = (int)(e >> 14);
                    return true;
                }
            }
            // nothing in that tile: the general search goes on behind its last entry
            return wv_find(L, H{t, (int32_t)c - 1}, FL_AT, minP, out, Pout, flout);
        }
    }
    return wv_find(L, from, FL_AT, minP, out, Pout, flout);
}

// ordinal of the candidate at line-index entry h
__device__ __forceinline__ long long rk_ordinal(const LineIndex &L, const RankBufs &R, H h)
{
    if (h.tile < 0) return 0;
    const long long before = (long long)rk_count_at(L, h.tile, (uint32_t)h.i);
    return (h.tile == 0 ? (long long)rk_sentinel_cand(L) : R.tbase[h.tile]) + before;
}

// The usual candidate, in ONE look at the index: the call from a "\n@" whose record -- and the "\n@" the chain goes on with --
// lie within the 256 entries behind it in the SAME tile (reads of a few kilobases wrapped at 80 columns: 40-80 lines), far
// from the buffer's end (none of the scanner's buffer-end rules can apply).  Four entries per lane from one 8-byte load; the
// three searches of the call (/root/reference/src/_fastqandfurious.c:70-71, :87-88, :102-103), the successor search
// (fastqandfurious.py:254) and the successor's ordinal (this candidate's + the "\n@" entries in between) are ballots over
// those registers -- where the general path below makes a dozen dependent look-ups.  false: not such a candidate (the
// general path decides everything).
__device__ __forceinline__ bool rk_succ_window(const LineIndex &L, const RankBufs &R, int64_t c, H k)
{
    const int lane = threadIdx.x & 63;
    if (k.tile < 0) return false;
    const int t = k.tile;
    const uint32_t ct = L.cnt[t];
    const int64_t tbase = ((int64_t)t << TILE_SHIFT) + L.s, len = L.len();
    if (ct > (uint32_t)SLOT || tbase + TILE + 4 >= len) return false;
    const uint32_t i = (uint32_t)k.i, j0 = i & ~3u, j = j0 + 4u * (uint32_t)lane;
    uint32_t e[4] = {0u, 0u, 0u, 0u};
    if (j < ct) {
        const uint2 w = *reinterpret_cast<const uint2 *>(L.ent + (int64_t)t * SLOT + j);
        e[0] = w.x & 0xFFFFu; e[1] = w.x >> 16; e[2] = w.y & 0xFFFFu; e[3] = w.y >> 16;
    }
    const uint32_t hi = min(ct, j0 + 256u);                    // entries [j0, hi) are in the registers
    auto entry_at = [&](uint32_t idx) -> uint32_t {            // idx wave-uniform, in [j0, hi)
        const uint32_t d = idx - j0;
        const uint32_t q = d & 3u;
        const uint32_t v = q == 0 ? e[0] : q == 1 ? e[1] : q == 2 ? e[2] : e[3];
        return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(d >> 2));
    };
    // first entry of index > after (and < hi) with the flag, at position >= minP (tile-relative); -1: none in the window
    auto first_flag = [&](uint32_t after, uint32_t flag_bit, int64_t minP) -> int {
        uint32_t m4 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t idx = j + (uint32_t)q;
            const bool ok = idx > after && idx < hi && ((e[q] >> flag_bit) & 1u) && tbase + (int64_t)(e[q] & OFF_MASK) >= minP;
            m4 |= (ok ? 1u : 0u) << q;
        }
        const unsigned long long m = __ballot(m4 != 0u);
        if (!m) return -1;
        const int w = __ffsll((long long)m) - 1;
        const uint32_t mw = (uint32_t)__builtin_amdgcn_readlane((int)m4, w);
        return (int)(j0 + 4u * (uint32_t)w) + (__ffs((int)mw) - 1);
    };
    if (i + 1u >= hi) return false;
    const int64_t Pk = tbase + (int64_t)(entry_at(i) & OFF_MASK);
    const int64_t he = tbase + (int64_t)(entry_at(i + 1u) & OFF_MASK);                    // :70-71 (the next newline)
    const int mi = first_flag(i + 1u, 15u, he + 2);                                       // :87-88 "\n+" at >= seq_beg + 1
    if (mi < 0 || (uint32_t)mi + 1u >= hi) return false;
    const int64_t se = tbase + (int64_t)(entry_at((uint32_t)mi) & OFF_MASK);
    const int64_t qhe = tbase + (int64_t)(entry_at((uint32_t)mi + 1u) & OFF_MASK);        // :102-103
    const int64_t p0 = Pk + 1;
    if ((qhe - se - 1 > 1) && (qhe - se != he - p0 + 1)) {                               // :109-117
        if (lane == 0) {
            R.rec[c] = RankRec{he, se, -1, ST_INVALID, 0};
            R.succ[c] = RK_NONE; R.S[0][c] = (uint32_t)c; R.C[0][c] = 0u; R.D[c] = RK_NONE;
        }
        return true;
    }
    const int64_t p4 = qhe + 1, qe = p4 + se - he - 1;                                    // :129
    if (qe + 2 >= len) return false;                                                     // (:130-133: the general path's)
    const int ni = first_flag((uint32_t)mi + 1u, 14u, qe - 1);                            // fastqandfurious.py:254: the next "\n@" at >= pos5 - 1
    if (ni < 0) return false;
    // its ordinal: this candidate's + 1 + the "\n@" entries strictly between the two
    uint32_t between = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t idx = j + (uint32_t)q;
        between += (idx > i && idx < (uint32_t)ni && ((e[q] >> 14) & 1u)) ? 1u : 0u;
    }
    const long long o = (long long)c + 1 + (long long)wave_sum_u32(between);
    if (lane == 0) {
        const uint32_t nx = (o < R.nc) ? (uint32_t)o : RK_NONE;
        R.rec[c] = RankRec{he, se, p4, ST_COMPLETE, 0};
        R.succ[c] = nx;
        R.S[0][c] = (nx != RK_NONE) ? nx : (uint32_t)c;
        R.C[0][c] = (nx != RK_NONE) ? 1u : 0u;
        R.D[c] = RK_NONE;
    }
    return true;
}

__global__ __launch_bounds__(256) void k_rk_succ(LineIndex L, RankBufs R, int eof)
{
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= R.nc) return;
    const H k = R.cand[c];
    if (rk_succ_window(L, R, c, k)) return;
    int64_t Pk; int fl;
    GAcc(L).get(k, Pk, fl);
    Rec r; H hm1;
    wv_record(L, k, Pk, L.len(), eof, r, hm1);
    uint32_t nx = RK_NONE;
    if (r.status == ST_COMPLETE) {
        H kn; int64_t Pn; int fln;
        if (rk_find_at(L, hm1, r.p5 - 1, kn, Pn, fln)) {
            const long long o = rk_ordinal(L, R, kn);
            nx = (o < R.nc) ? (uint32_t)o : RK_NONE;
        }
    }
    if (lane == 0) {
        R.rec[c] = RankRec{r.p1, r.p3, r.p4, r.status, r.final_ ? 1 : 0};
        R.succ[c] = nx;
        R.S[0][c] = (nx != RK_NONE) ? nx : (uint32_t)c;
        R.C[0][c] = (nx != RK_NONE) ? 1u : 0u;
        R.D[c] = RK_NONE;
    }
}

// the list head: the first candidate at buffer coordinate >= offset (its rank is 0); the newlines
// of the buffer for the result block.  One workgroup.
__global__ __launch_bounds__(1024) void k_rk_root(LineIndex L, RankBufs R, int64_t offset, DevRes *res)
{
    __shared__ unsigned long long s_nl[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    unsigned long long nlp = 0;
    {
        // (sixteen counts per thread and step, four 16-byte loads in flight: one count per step took 62 us for 4 GiB)
        const int64_t nt = L.ntiles, n16 = nt & ~(int64_t)15;
        for (int64_t t0 = (int64_t)tid * 16; t0 < n16; t0 += 1024 * 16) {
            const uint4 *p = reinterpret_cast<const uint4 *>(L.cnt + t0);
            const uint4 a = p[0], b = p[1], c4 = p[2], d = p[3];
            nlp += (unsigned long long)a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c4.x + c4.y + c4.z + c4.w + d.x + d.y + d.z + d.w;
        }
        for (int64_t t = n16 + tid; t < nt; t += 1024) nlp += L.cnt[t];
    }
    const uint32_t lo = wave_sum_u32((uint32_t)(nlp & 0xFFFFFu)), hi = wave_sum_u32((uint32_t)(nlp >> 20));
    if (lane == 0) s_nl[wid] = ((unsigned long long)hi << 20) + lo;
    __syncthreads();
    if (wid != 0) return;
    H k; int64_t Pk; int flk;
    const bool have = wv_find(L, H{-2, 0}, FL_AT, offset, k, Pk, flk);
    long long o = -1;
    if (have) o = rk_ordinal(L, R, k);
    if (lane == 0) {
        unsigned long long nl = 0;
        for (int q = 0; q < 16; q++) nl += s_nl[q];
        res->n_lines = (int64_t)nl;
        const uint32_t root = (have && o < R.nc) ? (uint32_t)o : RK_NONE;
        R.root[0] = root;
        if (root != RK_NONE) R.D[root] = 0u;
    }
}

// Round k of the doubling.  Invariant before: S[c] = the 2^k-th successor of c (or the list's last
// node from c if there are fewer), C[c] = steps actually taken, D[c] = rank for the members of
// rank < 2^k.  A member whose jump is a full 2^k steps marks the node it reaches.  (A node marked
// early within the round -- its marker ran first -- marks on with a correct rank: every writer
// of a word writes the same value.)
__global__ __launch_bounds__(256) void k_rk_round(int64_t nc, const uint32_t *__restrict__ Sin,
                                                  const uint32_t *__restrict__ Cin, uint32_t *__restrict__ Sout,
                                                  uint32_t *__restrict__ Cout, uint32_t *D, int k)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    const uint32_t s1 = Sin[c], c1 = Cin[c];
    const uint32_t s2 = Sin[s1], c2 = Cin[s1];
    const uint32_t dd = D[c];
    if (dd != RK_NONE && c1 == (1u << k)) D[s1] = dd + (1u << k);
    Sout[c] = s2;
    Cout[c] = c1 + c2;
}

// rows of the members (COMPLETE records and the final one) at table[rank]; the member the list
// ends at fills the result block (k_finalize adds the iterator's exit offset from the table)
__global__ __launch_bounds__(256) void k_rk_emit(LineIndex L, RankBufs R, int eof, int64_t offset, int64_t add,
                                                 int64_t *__restrict__ table, int64_t table_cap, DevRes *res)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && R.root[0] == RK_NONE) {
        // no "\n@" at or after offset: the chain ends at once with MISSING_SEQHEADER_BEGIN
        res->fallback = 0; res->n_records = 0; res->n_qual_bytes = 0; res->end_offset = offset;
        res->last_status = ST_HEAD_BEG; res->end_state = eof ? 0 : 1; res->has_final = 0; res->term_group = -1;
        for (int i = 0; i < 6; i++) res->last_pos[i] = -1;
    }
    if (c >= R.nc) return;
    const uint32_t rank = R.D[c];
    if (rank == RK_NONE) return;
    const RankRec r = R.rec[c];
    int64_t Pk; int fl;
    GAcc(L).get(R.cand[c], Pk, fl);
    const int64_t p0 = Pk + 1;
    const bool row = r.status == ST_COMPLETE || r.final_;
    const int64_t p5 = p0 >= 0 && row ? r.p4 + r.p3 - r.p1 - 1 : -1;
    if (row && (int64_t)rank < table_cap) {
        int64_t *o = table + (int64_t)rank * 6;
        o[0] = p0 + add; o[1] = r.p1 + add; o[2] = r.p1 + 1 + add; o[3] = r.p3 + add; o[4] = r.p4 + add; o[5] = p5 + add;
    }
    const bool last = r.status != ST_COMPLETE || R.succ[c] == RK_NONE;
    if (!last) return;
    res->fallback = 0;
    res->term_group = -1;
    res->n_qual_bytes = 0;
    res->end_offset = offset;              // (k_finalize: pos5 - 1 of the last COMPLETE record, if there is one)
    if (r.status == ST_COMPLETE) {
        // the list ends behind a COMPLETE record: the next call finds no "\n@" at all
        res->n_records = (int64_t)rank + 1;
        res->has_final = 0;
        res->last_status = ST_HEAD_BEG;
        res->end_state = eof ? 0 : 1;
        for (int i = 0; i < 6; i++) res->last_pos[i] = -1;
        return;
    }
    res->n_records = (int64_t)rank + (r.final_ ? 1 : 0);
    res->has_final = r.final_;
    res->last_status = r.status;
    int end;
    if (r.final_) end = 0;
    else if (eof) end = (r.status == ST_QUAL_END) ? 2 : (r.status == ST_INVALID) ? 4 : 3;
    else end = (r.status == ST_INVALID) ? 4 : 1;
    res->end_state = end;
    const int64_t p[6] = {p0, r.p1, r.p1 >= 0 ? r.p1 + 1 : -1, r.p3, r.p4, r.final_ ? r.p4 + r.p3 - r.p1 - 1 : -1};
    for (int i = 0; i < 6; i++) res->last_pos[i] = p[i] >= 0 ? p[i] + add : -1;
}

}  // namespace ffq
