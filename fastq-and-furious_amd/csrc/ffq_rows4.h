// ffq_rows4.h -- fast path of the record chain for plain four-line FASTQ.
//
// On input where every record is exactly four lines (header, sequence, '+' line,
// quality) the chain of readfastq_iter (/root/reference/src/fastqandfurious.py:251-279)
// has a closed form over NEWLINE ORDINALS: if the chain's first "\n@" match is newline
// number j0, record k is made of newlines j0+4k .. j0+4k+4.  The
// scanner call of /root/reference/src/_fastqandfurious.c:25-153 on record k then reads
//   e0 = nl[j0+4k]    the "\n@" match            pos0 = P(e0)+1
//   e1 = nl[j0+4k+1]  header end  (:70-71)        pos1 = P(e1), pos2 = pos1+1
//   e2 = nl[j0+4k+2]  "\n+" match (:87-88)        pos3 = P(e2)
//   e3 = nl[j0+4k+3]  '+' line end (:102-103)     pos4 = P(e3)+1
//   e4 = nl[j0+4k+4]  next "\n@" match (:62 of the next call), must lie at >= pos5-1
// with pos5 = pos4 + pos3 - pos2 (:129).  k_rows4 evaluates every record in parallel
// under that assumption AND checks, per record, exactly the conditions under which
// the reference would have made the same choices (flags of e2 / e4, non-empty
// sequence, '+' line length rule, buffer-end rules).  A record that does not satisfy
// them is "irregular".  k_finalize4 accepts the result only if no irregular record
// lies in front of the record the chain ends at; otherwise the general kernels
// (ffq_chain.h) redo the work from the same line index.  So the output is either the
// reference's chain, bit for bit, or discarded.
//
//   k_sum64       newlines per superblock of 64 tiles (from the tile counts)
//   k_sbscan      exclusive scan of those, first candidate j0
//   k_rows4       one wave per tile: rows of the records whose "\n@" lies in the tile,
//                 written as whole lines through an LDS transpose (48 B / record)
//   k_finalize4   validity, record count, end state
#pragma once
#include "ffq_chain.h"

namespace ffq {

constexpr int SB_TILES = 64;                  // tiles per superblock (one ordinal base each)
constexpr int R4_LIST = SLOT + 8;             // tile entries + look-ahead entries in LDS

struct Fast4Hdr {
    long long j0;                  // ordinal of the chain's first "\n@" match; -1: there is none
    int32_t attempt;               // 0: fast path not applicable (decided before k_rows4)
    int32_t dense_seen;            // k_rows4<., false> met a dense tile (more than SLOT newlines): the DENSE instantiation can take it
    unsigned long long irr_min;    // smallest irregular record index (~0: none)
    unsigned long long term_min;   // smallest (k << 24 | tile & 0xFFFFFF) where the chain ends (~0: none)
    long long n_lines;
};

struct TermInfo4 {
    int64_t pos[6];
    int32_t status;
    int32_t final_;
};

// per tile, for the decoded-quality offsets: its records and their quality bytes
struct TileQ {
    long long kfirst;              // record index of the tile's first record
    int32_t nrec;                  // records whose "\n@" lies in the tile (0: none)
    uint32_t qsum;                 // their quality bytes
};

// global ordinal of the first entry of tile t (the sentinel, if any, is ordinal 0)
__device__ __forceinline__ long long tile_ordinal_base(const LineIndex &L, const long long *sbbase, int t, int lane)
{
    const int sb = t / SB_TILES, t0 = sb * SB_TILES;
    const uint32_t c = (t0 + lane < t) ? L.cnt[t0 + lane] : 0u;       // SB_TILES == 64 lanes
    return sbbase[sb] + (long long)wave_sum_u32(c) + L.s;
}

// sums of 64 consecutive u32 values at a stride of `stride` words: one wave per sum
// (newlines per superblock from cnt[]; quality bytes per superblock from tileq[].qsum)
__global__ __launch_bounds__(256) void k_sum64(const uint32_t *__restrict__ src, int stride, int64_t nsrc,
                                               unsigned int *__restrict__ dst, int ndst)
{
    const int b = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    if (b >= ndst) return;
    const int64_t i = (int64_t)b * 64 + lane;
    const uint32_t v = (i < nsrc) ? src[i * stride] : 0u;
    const uint32_t t = wave_sum_u32(v);
    if (lane == 0) dst[b] = t;
}

// presum: per-superblock newline counts computed by k_sum64 (large buffers: many workgroups sum,
// this one only scans), or nullptr: this workgroup sums the tile counts itself (one launch less)
__global__ __launch_bounds__(1024) void k_sbscan(LineIndex L, int nsb, long long *__restrict__ sbbase, int64_t offset,
                                                 Fast4Hdr *hdr, const unsigned int *__restrict__ presum)
{
    __shared__ long long s_w[16];
    __shared__ uint32_t s_sum[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // warm this CU's cache with what the serial candidate search below will read one dependent
    // load at a time: the first bytes, and the first entries of the tile that holds `offset`
    if (wid == 15) {
        const int tq = (int)min((int64_t)max(L.ntiles - 1, 0), max((int64_t)0, (offset - L.s) >> TILE_SHIFT));
        uint32_t sink = (L.n > 0) ? (uint32_t)L.d[0] : 0u;
        if (L.ntiles > 0) sink += L.cnt[tq] + L.ent[(int64_t)tq * SLOT + lane * 4];
        if (tq + 1 < L.ntiles) sink += L.cnt[tq + 1] + L.ent[(int64_t)(tq + 1) * SLOT + (lane & 15) * 4];
        asm volatile("" ::"v"(sink));
    }
    long long carry = 0;
    if (presum) {
        // all of this thread's values first (one round trip), then scans that only touch LDS
        for (int c0 = 0; c0 < nsb; c0 += 16 * 1024) {
            uint32_t pv[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int b = c0 + k * 1024 + tid;
                pv[k] = (b < nsb) ? presum[b] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int b = c0 + k * 1024 + tid;
                if (c0 + k * 1024 >= nsb) continue;          // workgroup-uniform
                const uint32_t v = pv[k];
                const uint32_t incl = wave_incl_scan(v);
                if (lane == 63) s_w[wid] = (long long)incl;
                __syncthreads();
                long long wpre = 0, tot = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const long long t = s_w[q];
                    if (q < wid) wpre += t;
                    tot += t;
                }
                if (b < nsb) sbbase[b] = carry + wpre + (long long)(incl - v);
                __syncthreads();
                carry += tot;
            }
        }
    } else
    for (int b0 = 0; b0 < nsb; b0 += 1024) {
        const int b = b0 + tid;
        // newlines per superblock.  A wave owns 64 superblocks = 16 KiB of tile counts and reads
        // them as 16 fully coalesced 1 KiB loads (4 superblocks each, 16 lanes per superblock);
        // a thread-per-superblock read of the same bytes is bound by one CU's cache-line rate.
        uint32_t part[16];
        const int64_t tw0 = ((int64_t)(b0 + wid * 64) << 6) + lane * 4;    // this lane's first tile, piece 0
        if (tw0 + (15 << 8) + 4 <= L.ntiles - (63 - lane) * 4) {
            // the whole 16 KiB of counts of this wave lies inside the index (wave-uniform test:
            // the last lane's last piece): 16 loads, no branch between them, one wait
            uint4 x[16];
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = *reinterpret_cast<const uint4 *>(L.cnt + tw0 + ((int64_t)j << 8));
#pragma unroll
            for (int j = 0; j < 16; j++) part[j] = x[j].x + x[j].y + x[j].z + x[j].w;
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int64_t t = tw0 + ((int64_t)j << 8);
                uint32_t a = 0;
                for (int q = 0; q < 4; q++)
                    if (t + q < L.ntiles) a += L.cnt[t + q];
                part[j] = a;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            // sum over each row of 16 lanes: the row's last lane ends up with the superblock's sum
            uint32_t r = part[j];
            r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x111, 0xf, 0xf, false);  // row_shr:1
            r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x112, 0xf, 0xf, false);  // row_shr:2
            r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xf, false);  // row_shr:4
            r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xf, false);  // row_shr:8
            if ((lane & 15) == 15) s_sum[wid][4 * j + (lane >> 4)] = r;
        }
        __syncthreads();
        const uint32_t v = (b < nsb) ? s_sum[wid][lane] : 0u;      // <= 64 tiles * 16384 newlines
        // wave scan (sums of 64 superblocks fit 32 bits), then the 16 wave totals through LDS
        const uint32_t incl = wave_incl_scan(v);
        if (lane == 63) s_w[wid] = (long long)incl;
        __syncthreads();
        long long wpre = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const long long t = s_w[q];
            if (q < wid) wpre += t;
            tot += t;
        }
        if (b < nsb) sbbase[b] = carry + wpre + (long long)(incl - v);
        __syncthreads();
        carry += tot;
    }
    __syncthreads();
    if (tid == 0) {
        hdr->n_lines = carry;
        hdr->irr_min = ~0ull;
        hdr->term_min = ~0ull;
        hdr->attempt = 1;
        hdr->dense_seen = 0;
        hdr->j0 = -1;
        // the chain's first "\n@" match at buffer coordinate >= offset, as an ordinal
        const GAcc a(L);
        H h = a.before();
        long long ord = -1;           // ordinal of h
        long long found = -1;
        int64_t P; int fl;
        // entries in front of the tile that holds `offset` cannot match: skip whole tiles
        int tskip = (int)min((int64_t)L.ntiles, max((int64_t)0, (offset - L.s) >> TILE_SHIFT));
        if (tskip > 0) {
            long long o = sbbase[tskip / SB_TILES] + L.s;
            for (int t = (tskip / SB_TILES) * SB_TILES; t < tskip; t++) o += L.cnt[t];
            // continue right before the first entry of tile tskip
            h = H{tskip - 1, 0x7FFFFFF0};
            ord = o - 1;
        }
        bool gave_up = false;
        for (int steps = 0;; steps++) {
            if (steps > 4096) { gave_up = true; break; }
            bool ok;
            if (h.tile >= 0 && h.i == 0x7FFFFFF0) {       // "after tile h.tile"
                int t = h.tile + 1;
                while (t < L.ntiles && L.cnt[t] == 0) t++;
                ok = t < L.ntiles;
                if (ok) { h.tile = t; h.i = 0; }
            } else ok = a.next(h);
            if (!ok) break;
            ord++;
            a.get(h, P, fl);
            if ((fl & FL_AT) && P >= offset) { found = ord; break; }
        }
        if (gave_up) hdr->attempt = 0;
        else {
            hdr->j0 = found;
            if (found >= 0) {
                // a wrapped / non-four-line first record: do not even try
                H h2 = h;
                if (a.next(h2) && a.next(h2)) {
                    a.get(h2, P, fl);
                    if (!(fl & FL_PLUS)) hdr->attempt = 0;
                }
            }
        }
    }
}

// One wave per tile, four tiles per workgroup (no workgroup barrier).  (Round 3: a wave taking the PAIR of tiles 2w, 2w + 1 with
// both tiles' loads issued up front -- shared ordinal base, tile 2w's look-ahead = tile 2w + 1's own entries -- measured
// slower: 39.1 against 37.6 us per GiB, 460 against 397 us per 10 GiB in the fused variant;
// profiles/r03_probes/rows4_two_tiles_per_wave_ab.txt.)
// FUSED: the variant behind the single-pass index + decode kernel (ffq_fused.h); a template parameter so that the
// usual instantiation carries none of it (as a run-time test it cost the kernel 3 VGPRs and 4 us per GiB).
// DENSE: the instantiation that also takes DENSE tiles (more than SLOT newlines in 16 KiB: reads of a dozen bases with short
// headers, the shape of /root/reference/tests.py:8-35), their entries read from the overflow pool a chunk of SLOT list elements
// at a time -- the closed form over newline ordinals does not care how close the newlines are.  The usual instantiation refuses
// such a tile (Fast4Hdr::dense_seen) and is the same code as before; the host runs this one for a context that has met one.
constexpr uint8_t FZ_ALL = 0xFE;      // fz_phase of a tile that wrote all of its bytes (k_scan_ident, ffq_fused.h)
template <int FUSED, bool DENSE>
__global__ __launch_bounds__(256, DENSE ? 4 : 8) void k_rows4(LineIndex L, const long long *__restrict__ sbbase, int eof,
                                               int64_t add, Fast4Hdr *hdr, TermInfo4 *__restrict__ tinfo,
                                               int64_t *__restrict__ table, int64_t table_cap,
                                               uint32_t *__restrict__ qrel, TileQ *__restrict__ tileq,
                                               int64_t *__restrict__ p4s, int64_t p4_cap,
                                               int64_t fz_stride, const uint8_t *__restrict__ fz_phase,
                                               int64_t *__restrict__ fz_qoff)
{
    // fz_*: the decoded qualities were written by the index pass itself (k_scan_seg, ffq_fused.h) on the
    // assumption that every fourth line is a record's quality, whole -- each line into the output segment
    // (fz_stride bytes per tile) of the tile that holds the newline in front of it.  This kernel then only
    // has to say where each record's bytes START (fz_qoff): that segment + the lengths of the segment's
    // earlier lines -- and to verify the assumption, per tile (the lines taken for quality lines, fz_phase,
    // against the newline ordinals) and per record (the quality line is exactly as long as the sequence
    // line; the chain starts at the buffer's first newline).
    __shared__ __attribute__((aligned(16))) uint16_t s_ent_all[4][R4_LIST];   // the tile's own entries as stored (offset | flags << 14),
                                                                              // then (usually) the first five of the next tile
    __shared__ uint32_t s_la_all[4][8];             // look-ahead entries: position - tile base, flags << 30
    __shared__ __attribute__((aligned(16))) int32_t s_rows_all[4][64 * 6];    // row fields relative to the tile
    // (the wave index through readfirstlane: the compiler then keeps everything derived from the
    // tile number -- bases, limits, addresses -- in scalar registers)
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wid;
    if (t >= L.ntiles) return;
    // the header words are only TESTED after the tile's own loads have been issued: a branch on
    // them up here would put a second memory round trip in front of every wave
    const int attempt = hdr->attempt;
    const long long j0 = hdr->j0;
    const unsigned long long irr0 = hdr->irr_min;     // (as they stand when this wave starts: they only ever fall)
    const unsigned long long term0 = hdr->term_min;
    uint16_t *s_ent = s_ent_all[wid];
    uint32_t *s_la = s_la_all[wid];
    int32_t *s_rows = s_rows_all[wid];
    const int64_t len = L.len();

    // everything this tile usually needs in ONE memory round trip: its count, its first 256
    // entries (4 per lane), the counts for its ordinal base, and -- speculatively -- the
    // first 8 entries of the next tile as look-ahead
    // All of these loads are unconditional (clamped addresses, values masked afterwards): any
    // branch between them makes the compiler wait for the earlier ones first, and the prologue
    // was four dependent memory round trips instead of one.
    const int tn = min(t + 1, L.ntiles - 1);                        // next tile, clamped
    const int sb = t / SB_TILES, t0 = sb * SB_TILES;
    const uint16_t *src = L.ent + (int64_t)t * SLOT;
    const uint32_t c_raw = L.cnt[t];
    const uint2 v0 = *reinterpret_cast<const uint2 *>(src + 4 * lane);
    const uint32_t c1_raw = L.cnt[tn];
    const uint2 vla_raw = *reinterpret_cast<const uint2 *>(L.ent + (int64_t)tn * SLOT + 4 * (lane & 1));
    const uint32_t cb_raw = L.cnt[min(t0 + lane, L.ntiles - 1)];     // SB_TILES == 64 lanes
    const long long sbb = sbbase[sb];
    constexpr bool fused = FUSED != 0, in_place = FUSED == 2;
    uint32_t fzph = 0;
    if (fused) fzph = fz_phase[t];
    // pin the loads here: without a use in front of the early exits below the compiler sinks
    // each load behind the branch that precedes its first use
    asm volatile("" ::"v"(c_raw), "v"(v0.x), "v"(v0.y), "v"(c1_raw), "v"(vla_raw.x), "v"(vla_raw.y), "v"(cb_raw),
                 "v"(sbb), "s"(attempt), "s"(j0), "v"(fzph));
    const int c = (int)c_raw;
    const bool have_next = t + 1 < L.ntiles;
    const int c1 = have_next ? (int)c1_raw : 0;
    const uint2 vla = vla_raw;                                       // used by lanes 0, 1 when have_next
    // global ordinal of entry 0 of this tile (the sentinel, if any, is ordinal 0)
    const long long ob = sbb + (long long)wave_sum_u32((t0 + lane < t) ? cb_raw : 0u) + L.s;
    if (!attempt) return;
    // An irregular record in front of everything this tile could hold: whatever the tile finds, the attempt is refused (or the
    // chain ended before that record and none of this tile's rows is wanted): k_finalize4 accepts only irr_min > the chain's
    // last record.  On hostile text -- every record irregular, each through the careful path below -- the refused attempt
    // took 4.9 ms per GiB (tools/stress_huge.py); the waves that start after the first such record is known now leave here.
    // (the same behind the chain's end, term_min = record << 24 | tile: rows behind it are nobody's)
    if (ob - j0 > 0 && ((unsigned long long)((ob - j0) >> 2) > irr0 || (unsigned long long)((ob - j0) >> 2) > (term0 >> 24))) return;
    if (c > SLOT && (!DENSE || fused)) {     // dense tile: not this instantiation's (the DENSE one, or the general path)
        // (every dense tile says the same: look before storing -- 65536 atomics / stores onto one address took 2.9 ms)
        if (lane == 0 && (hdr->irr_min != 0ull || !hdr->dense_seen)) { atomicMin(&hdr->irr_min, 0ull); hdr->dense_seen = 1; }
        return;
    }
    if (fused) {
        // the lines the index pass took for quality lines: those that follow entries = phase (mod 4); by the
        // ordinals a quality line follows entry e iff ordinal(e) = j0 + 3 (mod 4), ordinal(e) = ob + e.  (A
        // tile without a newline starts no line: nothing was decoded for it.)  The chain must start at the
        // buffer's very first newline (nothing in front of it that could pass for a quality line).
        const uint32_t want = (uint32_t)(((j0 + 3 - ob) % 4 + 4) % 4);
        // (in place: a tile that wrote all of its bytes assumed nothing)
        const bool wrong = in_place ? (fzph != (uint32_t)FZ_ALL && fzph != want) : (c > 0 && fzph != want);
        // (look before storing: a refused buffer says so from every tile)
        if ((wrong || j0 != 0) && lane == 0 && hdr->irr_min != 0ull) atomicMin(&hdr->irr_min, 0ull);
    }
    // the sentinel is entry -1 of tile 0 (ordinal 0): give tile 0 a list that starts with it
    const int pre = (t == 0 && L.s) ? 1 : 0;
    if (c + pre == 0 && t != 0) return;   // no newline in the tile (long reads): no record starts here
    const int64_t tbase = ((int64_t)t << TILE_SHIFT) + L.s;            // buffer coordinate of tile offset 0
    const int64_t tb_add = tbase + add;                                // (wave-uniform: scalar registers)
    const int32_t lim = (int32_t)min(len - tbase, (int64_t)0x7FFFFFF0);   // len relative to the tile
    if (j0 < 0) {
        // no "\n@" at all: the chain ends at once with MISSING_SEQHEADER_BEGIN
        if (t == 0 && lane == 0) {
            TermInfo4 &ti = tinfo[0];
            for (int i = 0; i < 6; i++) ti.pos[i] = -1;
            ti.status = ST_HEAD_BEG; ti.final_ = 0;
            atomicMin(&hdr->term_min, 0ull);
        }
        return;
    }
    // A tile's list = [sentinel] + its entries; it is taken in CHUNKS of SLOT elements -- one chunk for every tile but a
    // dense one (DENSE instantiation only: the loop below is a single turn otherwise, decided at compile time).
    const int ntot = pre + c;                                          // list elements this tile owns
    const int nch = DENSE ? (ntot + SLOT - 1) / SLOT : 1;
    const uint16_t *dsrc = nullptr;                                    // a dense tile's entries in the pool
    if (DENSE && c > SLOT) {
        const unsigned long long at = L.ovf[t] & OVF_MASK;
        if (at + (unsigned long long)c > L.pool_cap) {                 // (the pool ran out: ERR_POOL, this scan is run again)
            if (lane == 0) atomicMin(&hdr->irr_min, 0ull);
            return;
        }
        dsrc = L.pool + at;
    }
    long long kfirst = -1;                           // per tile: first record, records, their quality bytes
    int nrec_tile = 0;
    bool tile_term_done = false;
    uint32_t qrun = 0;
    for (int ch = 0; ch < nch; ch++) {
    const int e_lo = DENSE ? ch * SLOT : 0;                            // first list element of the chunk
    const int nown = DENSE ? min(SLOT, ntot - e_lo) : ntot;            // list elements this chunk owns
    const bool sent0 = pre && e_lo == 0;                               // chunk element 0 is the sentinel
    const long long obl = ob - pre + e_lo;                             // ordinal of chunk element 0
    if (DENSE && ch > 0) wave_sync();                                  // (the lists are rewritten)
    if (DENSE && dsrc) {
        // chunk elements x = 0 .. nown - 1 are physical entries e_lo + x - pre of the tile: eight per lane and step
        for (int x0 = 8 * lane; x0 < nown; x0 += 512) {
            const int ph = e_lo + x0 - pre;
            if (ph >= 0 && ph + 8 <= c) {
                typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(2)));
                const u32x4u v = *reinterpret_cast<const u32x4u *>(dsrc + ph);
                *reinterpret_cast<uint4 *>(s_ent + x0) = make_uint4(v.x, v.y, v.z, v.w);
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (ph + q >= 0 && ph + q < c) s_ent[x0 + q] = dsrc[ph + q];
            }
        }
    }
    if (sent0) {
        if (lane == 0) {
            const uint8_t b0 = L.n > 0 ? L.d[0] : 0;
            const uint32_t fl = (b0 == '@') ? FL_AT : (b0 == '+') ? FL_PLUS : 0;
            s_ent[0] = (uint16_t)(fl << 14);             // the sentinel: coordinate 0, special-cased below
        }
        if (!(DENSE && dsrc)) {
            const uint32_t x[4] = {v0.x & 0xFFFFu, v0.x >> 16, v0.y & 0xFFFFu, v0.y >> 16};
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (4 * lane + i < c) s_ent[1 + 4 * lane + i] = (uint16_t)x[i];
        }
    } else if (!(DENSE && dsrc)) {
        // four entries per lane as loaded, one 8-byte LDS store; what lies past the tile's count is
        // never read as an entry of the tile
        *reinterpret_cast<uint2 *>(s_ent + 4 * lane) = v0;
    }
    if (!(DENSE && dsrc))
        for (int i = 256 + lane; i < c; i += 64) {              // more than 256 lines in the tile
            s_ent[pre + i] = src[i];
        }
    // look-ahead: the list elements that follow the chunk (a record needs 4 more newlines)
    int nl = nown;
    bool idx_end = false;           // the look-ahead ran into the end of the index
    // la_simple: the look-ahead is five consecutive entries of ONE tile -- the first five of tile t + 1 (la_add = TILE) or, behind
    // a chunk that does not end its dense tile, the tile's own next five (la_add = 0); they are then ALSO list
    // elements nown .. nown + 4 of s_ent as stored (position = la_add + offset), which is what the
    // usual-record test below reads
    const bool last_ch = !DENSE || ch == nch - 1;
    bool la_simple;
    int32_t la_add = TILE;
    if (last_ch) {
        la_simple = have_next && c1 >= 5 && (c1 <= SLOT || DENSE);
        if (la_simple && (!DENSE || c1 <= SLOT)) {
            if (lane < 2) {
                const uint32_t x[4] = {vla.x & 0xFFFFu, vla.x >> 16, vla.y & 0xFFFFu, vla.y >> 16};
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (4 * lane + i < 5) {
                        s_la[4 * lane + i] = ((1u << TILE_SHIFT) + (x[i] & OFF_MASK)) | ((x[i] >> 14) << 30);
                        s_ent[nl + 4 * lane + i] = (uint16_t)x[i];
                    }
            }
            nl += 5;
        } else if (la_simple) {
            // (DENSE: the next tile is dense, its first five entries are in the pool)
            const unsigned long long at1 = L.ovf[t + 1] & OVF_MASK;
            if (at1 + 5ull <= L.pool_cap) {
                if (lane < 5) {
                    const uint32_t e = L.pool[at1 + lane];
                    s_la[lane] = ((1u << TILE_SHIFT) + (e & OFF_MASK)) | ((e >> 14) << 30);
                    s_ent[nl + lane] = (uint16_t)e;
                }
                nl += 5;
            } else la_simple = false;       // (ERR_POOL: the records at the chunk's end come out irregular, the scan is run again)
        }
    } else {
        la_simple = ntot - (e_lo + nown) >= 5;
        la_add = 0;
        if (la_simple) {
            if (lane < 5) {
                const uint32_t e = dsrc[e_lo + nown - pre + lane];
                s_la[lane] = (e & OFF_MASK) | ((e >> 14) << 30);
                s_ent[nl + lane] = (uint16_t)e;
            }
            nl += 5;
        }
    }
    if (!la_simple) {
        // element by element: what is left of this tile behind the chunk (DENSE), then the following tiles
        int tt = last_ch ? t + 1 : t, pp = last_ch ? 0 : e_lo + nown - pre, got = 0;
        while (got < 5) {
            if (tt >= L.ntiles) { idx_end = true; break; }
            const int cc = (tt == t) ? c : (int)L.cnt[tt];
            if (cc > SLOT && !DENSE) break;
            if (pp >= cc) { tt++; pp = 0; if (tt - t > 60000) break; continue; }       // positions must stay below 2^30
            const uint16_t *es = L.ent + (int64_t)tt * SLOT;
            if (cc > SLOT) {
                const unsigned long long at1 = L.ovf[tt] & OVF_MASK;
                if (at1 + (unsigned long long)cc > L.pool_cap) break;
                es = L.pool + at1;
            }
            const int take = min(cc - pp, 5 - got);
            if (lane < take) {
                const uint32_t e = es[pp + lane];
                s_la[got + lane] = (((uint32_t)(tt - t) << TILE_SHIFT) + (e & OFF_MASK)) | ((e >> 14) << 30);
            }
            got += take;
            pp += take;
        }
        nl += got;
    }
    wave_sync();

    // records whose "\n@" is list element i: ordinal obl + i = j0 + 4k
    long long kfirst_ch = -1;
    int nrec_ch = 0;
    {
        // first owned element with ordinal >= j0 and (ordinal - j0) % 4 == 0
        long long i0 = (obl >= j0) ? ((4 - ((obl - j0) & 3)) & 3) : (j0 - obl);
        if (i0 < nown) {
            kfirst_ch = (obl + i0 - j0) >> 2;
            nrec_ch = (int)((nown - i0 + 3) >> 2);
            if (kfirst < 0) kfirst = kfirst_ch;
        }
        // positions of list elements: (s_pos & 0x0FFFFFFF) + tbase, the sentinel is coordinate 0
        for (int r0 = 0; r0 < nrec_ch; r0 += 64) {
            const int r = r0 + lane;
            const bool act = r < nrec_ch;
            const int i = (int)i0 + 4 * r;
            int cls = 0;          // 0 regular COMPLETE, 1 irregular, 2 chain ends here (status below), 3 complete and last
            bool fin = false;
            // ---- the usual record: all five list elements at hand and every rule of the scanner
            //      met, tested branch-free on tile-relative 32-bit positions.  Whatever fails the
            //      test (the ends of the buffer, irregular text, a look-ahead that is not simply
            //      the next tile's) takes the rule-by-rule path below, in 64 bits.
            int32_t f0 = 0, f1 = 0, f3 = 0, f4 = 0, f5 = 0;      // row fields, relative to tbase
            bool usual = false;
            {
                const int ic = min(i, R4_LIST - 5);
                const uint32_t e0 = s_ent[ic], e1 = s_ent[ic + 1], e2 = s_ent[ic + 2], e3 = s_ent[ic + 3],
                               e4 = s_ent[ic + 4];
                const int32_t x0 = (sent0 && ic == 0) ? -1 : (int32_t)(e0 & OFF_MASK) + ((ic >= nown) ? la_add : 0);
                const int32_t x1 = (int32_t)(e1 & OFF_MASK) + ((ic + 1 >= nown) ? la_add : 0);
                const int32_t x2 = (int32_t)(e2 & OFF_MASK) + ((ic + 2 >= nown) ? la_add : 0);
                const int32_t x3 = (int32_t)(e3 & OFF_MASK) + ((ic + 3 >= nown) ? la_add : 0);
                const int32_t x4 = (int32_t)(e4 & OFF_MASK) + ((ic + 4 >= nown) ? la_add : 0);
                const int32_t qe = x3 + x2 - x1;                 // pos4 + (pos3 - pos2) = x3 + 1 + x2 - x1 - 1
                usual = act & (i + 4 < nl) & (la_simple | (i + 4 < nown))
                      & (x1 <= lim - 2)                                              // header line ends inside
                      & (((e2 >> 14) & FL_PLUS) != 0) & (x2 >= x1 + 2)               // '+' line right after ONE sequence line
                      & (x2 + 2 < lim) & (x3 <= lim - 2)
                      & !((x3 - x2 - 1 > 1) & (x3 - x2 != x1 - x0))                  // '+' line length rule
                      & (qe + 2 < lim)
                      & (((e4 >> 14) & FL_AT) != 0) & (x4 >= qe - 1);                // the next call finds e4
                f0 = x0 + 1; f1 = x1; f3 = x2; f4 = x3 + 1; f5 = qe;
                // the single pass decoded the whole quality LINE: it must end where pos5 says
                if (fused && usual && x4 != qe) atomicMin(&hdr->irr_min, (unsigned long long)(kfirst_ch + r));
            }
            if (act && !usual) {
                int64_t p0 = 0, p1 = 0, p3 = 0, p4 = 0, p5 = 0;
                int status = ST_COMPLETE;
                const long long k = kfirst_ch + r;
                const int have = nl - i;           // list elements from e0 on (e0 included)
                // list element i+q: an own entry (the sentinel is element 0 of tile 0) or look-ahead
                auto POS = [&](int q) -> int64_t {
                    const int e = i + q;
                    if (e >= nown) return tbase + (int64_t)(s_la[e - nown] & 0x3FFFFFFFu);
                    if (sent0 && e == 0) return (int64_t)0;
                    return tbase + (int64_t)(s_ent[e] & OFF_MASK);
                };
                auto FLG = [&](int q) -> uint32_t {
                    const int e = i + q;
                    return (e >= nown) ? (s_la[e - nown] >> 30) : ((uint32_t)s_ent[e] >> 14);
                };
                const int64_t P0 = POS(0);
                p0 = P0 + 1; p1 = p3 = p4 = p5 = -1;
                // fewer elements than needed: a real end only if the index itself ends there
                if (have < 2) { cls = idx_end ? 2 : 1; status = ST_HEAD_END; }
                else {
                    const int64_t P1 = POS(1);
                    if (P1 > len - 2) { cls = 2; status = ST_HEAD_END; }
                    else {
                        p1 = P1;
                        if (have < 3) { cls = idx_end ? 2 : 1; status = ST_SEQ_END; }
                        else {
                            const int64_t P2 = POS(2);
                            if (!(FLG(2) & FL_PLUS) || P2 < P1 + 2) cls = 1;      // wrapped / empty read: not this path
                            else {
                                p3 = P2;
                                if (P2 + 2 >= len) { cls = 2; status = ST_QUALHEAD_END; }
                                else if (have < 4) { cls = idx_end ? 2 : 1; status = ST_QUALHEAD_END; }
                                else {
                                    const int64_t P3 = POS(3);
                                    if (P3 > len - 2) { cls = 2; status = ST_QUALHEAD_END; }
                                    else if ((P3 - P2 - 1 > 1) && (P3 - P2 != P1 - P0)) { cls = 2; status = ST_INVALID; }
                                    else {
                                        p4 = P3 + 1;
                                        const int64_t qe = p4 + P2 - P1 - 1;
                                        if (qe + 2 >= len) {
                                            cls = 2; status = ST_QUAL_END;
                                            if (eof && qe < len) { p5 = qe; fin = true; }
                                        } else {
                                            p5 = qe;
                                            // the next call must find e4: "\n@" at >= qe - 1
                                            if (have < 5) cls = idx_end ? 3 : 1;
                                            else if (!(FLG(4) & FL_AT) || POS(4) < qe - 1) cls = 1;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                if (fused && (cls == 0 || cls == 3 || (cls == 2 && fin)) && !(have >= 5 && POS(4) == p5)) cls = 1;   // (see above)
                {
                    // the smallest k of the wave's irregular records (lanes are in k order), and only if it can lower the
                    // minimum: a buffer of nothing but irregular records (hostile text: 5 M of them per GiB) put every one of
                    // them through an atomic on this one address -- 5 ms of a refused attempt
                    const unsigned long long im = __ballot(cls == 1);
                    if (cls == 1 && lane == __ffsll((long long)im) - 1 &&
                        (unsigned long long)k < __hip_atomic_load(&hdr->irr_min, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        atomicMin(&hdr->irr_min, (unsigned long long)k);
                }
                if (cls == 2 || cls == 3) {
                    // where the chain ends: cls 2 -> at record k (status of its call); cls 3 -> after
                    // record k: the next call finds no "\n@" at all
                    // (look first, as above: on hostile text every other record "ends the chain")
                    const unsigned long long kk = (unsigned long long)(cls == 3 ? k + 1 : k);
                    const unsigned long long mine = (kk << 24) | (unsigned long long)(t & 0xFFFFFF);
                    if (mine < __hip_atomic_load(&hdr->term_min, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&hdr->term_min, mine);
                }
                // (a row that is written has all its fields; they lie within 2^31 of the tile)
                f0 = (int32_t)(p0 - tbase); f1 = (int32_t)(p1 - tbase); f3 = (int32_t)(p3 - tbase);
                f4 = (int32_t)(p4 - tbase); f5 = (int32_t)(p5 - tbase);
                // per tile: the terminal information of the smallest k (lanes are in k order; only
                // lanes on this path can be terminal, the ballot is among them)
                const unsigned long long tm = __ballot(cls == 2 || cls == 3);
                if (tm != 0ull && !tile_term_done && lane == __ffsll((long long)tm) - 1) {
                    TermInfo4 ti;
                    if (cls == 3) {
                        for (int q = 0; q < 6; q++) ti.pos[q] = -1;
                        ti.status = ST_HEAD_BEG; ti.final_ = 0;
                    } else {
                        ti.pos[0] = p0; ti.pos[1] = p1; ti.pos[2] = (p1 >= 0) ? p1 + 1 : -1;
                        ti.pos[3] = p3; ti.pos[4] = p4; ti.pos[5] = p5;
                        ti.status = status; ti.final_ = fin ? 1 : 0;
                    }
                    tinfo[t] = ti;
                }
            }
            if (__ballot(act && (cls == 2 || cls == 3)) != 0ull) tile_term_done = true;      // (wave-uniform)
            // rows: COMPLETE records (cls 0, 3) and the final record
            const bool emit = act && (cls == 0 || cls == 3 || (cls == 2 && fin));
            if (in_place) {
                // the record's bytes lie where they lie in the input: pos4 in the coordinates of d
                if (emit && kfirst_ch + r < p4_cap) fz_qoff[kfirst_ch + r] = ((long long)t << TILE_SHIFT) + f4;
            } else if (fused) {
                // start of the record's bytes in the decoded stream = quality bytes in front of pos4
                const uint32_t ql = emit ? (uint32_t)(f5 - f4) : 0u;
                const uint32_t incl = wave_incl_scan(ql);
                if (emit && kfirst_ch + r < p4_cap) {
                    // the newline in front of pos4 (the '+' line's end) lies at tile-relative f4 - 1: in this tile, or
                    // in a later one -- whose earlier bytes then all belong to this record: its segment's first line
                    const int32_t x3 = f4 - 1;
                    long long q = ((long long)t + (x3 >> TILE_SHIFT)) * fz_stride;
                    if (x3 < TILE) {
                        // this tile's segment: first the quality line of an earlier record whose '+' line ends in
                        // this tile (the entry in front of the tile's first record start), then the tile's own
                        // records in front of this one
                        int32_t head = 0;
                        const int e0 = (int)i0 - pre;            // physical entry of the tile's first record start
                        if (kfirst_ch > 0 && e0 >= 1)
                            head = (int32_t)(s_ent[e0 + pre] & OFF_MASK) - (int32_t)(s_ent[e0 - 1 + pre] & OFF_MASK) - 1;
                        q += head + (long long)(qrun + incl - ql);
                    }
                    fz_qoff[kfirst_ch + r] = q;
                }
                qrun += (uint32_t)__shfl((int)incl, 63);
            } else if (qrel) {
                // tile-relative offsets of the decoded qualities (32 bits, scratch); k_qfix4 adds the
                // tile's base and writes the caller's 64-bit offsets
                const uint32_t ql = emit ? (uint32_t)(f5 - f4) : 0u;
                const uint32_t incl = wave_incl_scan(ql);
                if (act && kfirst_ch + r < p4_cap) qrel[kfirst_ch + r] = qrun + incl - ql;
                // pos4 once more, compact: the decode reads 8 bytes per record instead of the row's line
                if (emit && kfirst_ch + r < p4_cap) p4s[kfirst_ch + r] = tb_add + f4;
                qrun += (uint32_t)__shfl((int)incl, 63);
            }
            int2 *mine = reinterpret_cast<int2 *>(s_rows + lane * 6);
            mine[0] = make_int2(f0, f1); mine[1] = make_int2(f1 + 1, f3); mine[2] = make_int2(f4, f5);
            wave_sync();
            // rows of one chunk are consecutive in the table: 16-byte pieces, consecutive lanes ->
            // consecutive pieces; a row is written iff its record emits
            const unsigned long long em = __ballot(emit);
            const int64_t rowbase = kfirst_ch + r0;
            const int2 *srcr = reinterpret_cast<const int2 *>(s_rows);
            longlong2 *dst = reinterpret_cast<longlong2 *>(table + rowbase * 6);
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int q = lane + u * 64;
                const int row = q / 3;
                if (((em >> row) & 1ull) && rowbase + row < table_cap) {
                    // written once, read by someone else later: non-temporal
                    typedef long long i64x2 __attribute__((ext_vector_type(2)));
                    const int2 vv = srcr[q];
                    i64x2 t; t.x = tb_add + vv.x; t.y = tb_add + vv.y;
                    __builtin_nontemporal_store(t, reinterpret_cast<i64x2 *>(dst + q));
                }
            }
            wave_sync();
        }
    }
    nrec_tile += nrec_ch;
    }   // chunks
    if (qrel && lane == 0 && nrec_tile > 0) tileq[t] = TileQ{kfirst, nrec_tile, qrun};
}

// exclusive scan of the per-superblock quality bytes (one workgroup)
__global__ __launch_bounds__(1024) void k_qscan4(const unsigned int *__restrict__ sbq, int nsb,
                                                 long long *__restrict__ sbqbase)
{
    __shared__ long long s_v[1024];
    const int tid = threadIdx.x;
    long long carry = 0;
    for (int b0 = 0; b0 < nsb; b0 += 1024) {
        const int b = b0 + tid;
        const long long v = (b < nsb) ? (long long)sbq[b] : 0;
        s_v[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            long long x = 0;
            if (tid >= d) x = s_v[tid - d];
            __syncthreads();
            s_v[tid] += x;
            __syncthreads();
        }
        if (b < nsb) sbqbase[b] = carry + s_v[tid] - v;
        const long long tot = s_v[1023];
        __syncthreads();
        carry += tot;
    }
}

// One wave per PAIR of tiles: tile-relative quality offsets -> stream offsets, directory of the
// stream.  A tile is two dependent memory round trips (its TileQ, then its offsets) and little
// else: with two tiles per wave, loaded together, half as many of those chains are in flight for
// the same bytes (182 -> ~120 us per 10 GiB).
__global__ __launch_bounds__(256) void k_qfix4(int ntiles, const Fast4Hdr *__restrict__ hdr,
                                               const TileQ *__restrict__ tileq,
                                               const long long *__restrict__ sbqbase,
                                               const uint32_t *__restrict__ qrel, int64_t *__restrict__ qoff,
                                               int64_t table_cap, int64_t *__restrict__ qdir, int64_t qdir_cap)
{
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int t = (blockIdx.x * 4 + wid) * 2;            // tiles t and t + 1 (same superblock: SB_TILES is even)
    if (t >= ntiles) return;
    // the loads go out together (clamped addresses, masked values): a branch between them -- the
    // test of the header word included -- is a memory round trip each
    const int attempt = hdr->attempt;
    const int t0 = (t / SB_TILES) * SB_TILES;
    const TileQ me0 = tileq[t];
    TileQ me1 = tileq[min(t + 1, ntiles - 1)];
    const uint32_t bq = tileq[min(t0 + lane, ntiles - 1)].qsum;                  // SB_TILES == 64 lanes
    const long long sbb = sbqbase[t / SB_TILES];
    asm volatile("" ::"v"(me0.kfirst), "v"(me0.nrec), "v"(me0.qsum), "v"(me1.kfirst), "v"(me1.nrec), "v"(me1.qsum),
                 "v"(bq), "v"(sbb), "s"(attempt));
    if (!attempt) return;
    if (t + 1 >= ntiles) me1.nrec = 0;
    const int64_t base0 = sbb + (int64_t)wave_sum_u32((t0 + lane < t) ? bq : 0u);
    const int64_t base1 = base0 + (int64_t)(me0.nrec > 0 ? me0.qsum : 0u);
    // first 64 records of both tiles: both loads before either is used
    const int64_t i0 = me0.kfirst + lane, i1 = me1.kfirst + lane;
    const bool a0 = lane < me0.nrec && i0 < table_cap, a1 = lane < me1.nrec && i1 < table_cap;
    const int64_t l0 = a0 ? (int64_t)qrel[i0] : 0, l1 = a1 ? (int64_t)qrel[i1] : 0;
    asm volatile("" ::"v"(l0), "v"(l1));
    auto fix = [&](const TileQ &me, int64_t base, int r0, int64_t loc, bool act) {
        const int r = r0 + lane;
        const int64_t idx = me.kfirst + r;
        // the record's bytes end where the next record's begin
        int64_t nxt = ((int64_t)__shfl((int)(loc >> 32), lane + 1) << 32) | (uint32_t)__shfl((int)(uint32_t)loc, lane + 1);
        if (r + 1 >= me.nrec) nxt = me.qsum;
        else if (lane == 63) nxt = (idx + 1 < table_cap) ? (int64_t)qrel[idx + 1] : (int64_t)me.qsum;
        if (act) {
            qoff[idx] = base + loc;
            qdir_mark(qdir, qdir_cap, base + loc, nxt - loc, idx);
        }
    };
    auto rest = [&](const TileQ &me, int64_t base) {
        for (int r0 = 64; r0 < me.nrec; r0 += 64) {
            const int64_t idx = me.kfirst + r0 + lane;
            const bool act = r0 + lane < me.nrec && idx < table_cap;
            fix(me, base, r0, act ? (int64_t)qrel[idx] : 0, act);
        }
    };
    if (me0.nrec > 0) fix(me0, base0, 0, l0, a0);
    if (me1.nrec > 0) fix(me1, base1, 0, l1, a1);
    if (me0.nrec > 64) rest(me0, base0);
    if (me1.nrec > 64) rest(me1, base1);
}

// total of the decoded stream and its closing offset (after k_finalize4 and k_qfix4)
__global__ void k_qtotal4(DevRes *res, const int64_t *__restrict__ table, int64_t table_cap,
                          int64_t *__restrict__ qoff, Pub pb)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (res->fallback) { publish(pb, res); return; }
    const int64_t n = res->n_records;
    int64_t tot = 0;
    if (n > 0 && n <= table_cap) tot = qoff[n - 1] + (table[(n - 1) * 6 + 5] - table[(n - 1) * 6 + 4]);
    res->n_qual_bytes = tot;
    if (n <= table_cap) qoff[n] = tot;
    publish(pb, res);
}

// validity + result block (end state per fastqandfurious.py:256-279)
__global__ void k_finalize4(LineIndex L, Fast4Hdr *hdr, const TermInfo4 *__restrict__ tinfo, int eof,
                            int64_t offset, int64_t add, const int64_t *__restrict__ table, int64_t table_cap,
                            DevRes *res, Pub pb, const uint32_t *__restrict__ fz_bad = nullptr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    res->n_lines = hdr->n_lines;
    res->n_qual_bytes = 0;
    res->term_group = -1;
    res->has_final = 0;
    const unsigned long long tm = hdr->term_min, im = hdr->irr_min;
    const unsigned long long tk = tm >> 24;
    res->fused_bad = fz_bad ? (int32_t)*fz_bad : 0;
    res->fast4_dense = hdr->dense_seen;
    if (!hdr->attempt || tm == ~0ull || (im != ~0ull && im <= tk) || (fz_bad && *fz_bad)) { res->fallback = 1; res->fast4_hint = 0; publish(pb, res); return; }
    res->fallback = 0;
    res->fast4_hint = 1;
    const int tt = (int)(tm & 0xFFFFFF);
    const TermInfo4 ti = tinfo[tt];
    const int status = ti.status;
    res->last_status = status;
    for (int i = 0; i < 6; i++) res->last_pos[i] = (ti.pos[i] >= 0) ? ti.pos[i] + add : -1;
    const long long ncomplete = (long long)tk;
    res->n_records = ncomplete + (ti.final_ ? 1 : 0);
    res->has_final = ti.final_ ? 1 : 0;
    int end;
    if (ti.final_) end = 0;
    else if (status == ST_HEAD_BEG) end = eof ? 0 : 1;
    else if (eof) end = (status == ST_QUAL_END) ? 2 : (status == ST_INVALID) ? 4 : 3;
    else end = (status == ST_INVALID) ? 4 : 1;
    res->end_state = end;
    if (ncomplete > 0 && ncomplete <= table_cap) res->end_offset = table[(ncomplete - 1) * 6 + 5] - add - 1;
    else res->end_offset = offset;
    publish(pb, res);
}

}  // namespace ffq
