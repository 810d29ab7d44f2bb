// ffq_dev.h -- device-side data structures and helpers shared by the kernels
// of libffq_hip.so (gfx950 only; wave64 is hard-coded).
//
// Line index.  The scan kernel (k_scan_lines) turns the byte stream into a
// two-level index of newline positions: the buffer is cut into TILE-byte
// tiles, tile t owns a SLOT-entry slot of u16 entries, one per '\n' in the
// tile, in position order:
//     entry = offset_in_tile (14 bits) | AT << 14 | PLUS << 15
// AT / PLUS say that the byte AFTER the newline exists and is '@' / '+': they
// are exactly the matches of the reference's memmem("\n@") / memmem("\n+")
// (/root/reference/src/_fastqandfurious.c:62,87).  cnt[t] is the number of
// newlines in the tile; a tile with more than SLOT of them stores its entries
// in an overflow pool at ovf[t].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffq {

// Diagnostics (ablation switches of the kernels, look-back / pipeline / read probes) exist only in the
// instrumented build of this library, libffq_probe.so (-DFFQ_PROBES; tools/ only).  In the product build
// every `PROBES && ...` test is constant-false and the code behind it is not compiled.
#ifdef FFQ_PROBES
constexpr bool PROBES = true;
#else
constexpr bool PROBES = false;
#endif

constexpr int TILE_SHIFT = 14;
constexpr int TILE = 1 << TILE_SHIFT;          // bytes per line-index tile
constexpr int SLOT = 1024;                     // u16 entries per tile slot
constexpr uint32_t OFF_MASK = TILE - 1;
constexpr int FL_AT = 1, FL_PLUS = 2;

// scanner status codes (_fastqandfurious.c:7-15)
constexpr int ST_INVALID = -1, ST_HEAD_BEG = 0, ST_HEAD_END = 1, ST_SEQ_BEG = 2,
              ST_SEQ_END = 3, ST_QUAL_BEG = 4, ST_QUAL_END = 5, ST_COMPLETE = 6,
              ST_QUALHEAD_END = 7;
constexpr int ST_FINAL = 21;   // internal: status 5 at eof, accepted by the final-record rule

// control-block error bits
constexpr uint32_t ERR_POOL = 1u;       // overflow pool exhausted -> host grows it and re-runs
constexpr uint32_t ERR_INTERNAL = 2u;   // an invariant of the chain kernels failed
constexpr uint32_t ERR_DSTAGE = 4u;     // the walked groups' stage ran out of chunks -> host grows it and re-runs the chain

// ovf[t] of a dense tile: where its entries start in the pool, and in the two top bits whether ANY of them carries
// FL_AT / FL_PLUS -- a search for a flagged entry steps over a dense tile without one (a block of blank lines)
// on that word alone
constexpr unsigned long long OVF_MASK = (1ull << 62) - 1ull;

// The overflow pool is cut into POOL_NB equal regions, tile t allocates in region t % POOL_NB: one bump counter
// for all dense tiles was 65536 agent-scope atomics per GiB of dense input on ONE address, served one after the other.
constexpr int POOL_NB = 64;

struct Ctl {
    uint32_t err;
    uint32_t pool_any;             // a dense tile allocated in this scan: the publisher gathers and zeroes pool_heads
    unsigned long long pool_head;  // host mirror only: entries the pool must hold for this scan (POOL_NB x the fullest region)
    unsigned long long pool_heads[POOL_NB];
};

struct LineIndex {
    const uint8_t *d;          // the bytes (read only for the sentinel's flags)
    int64_t n;                 // number of bytes
    int32_t s;                 // 1: a virtual '\n' sits at buffer coordinate 0
    int32_t ntiles;
    int32_t ready;             // tiles [0, ready) of the index are valid (chunked pipelining)
    int32_t pad_;
    const uint16_t *ent;       // [ntiles][SLOT]
    const uint32_t *cnt;       // [ntiles]
    const unsigned long long *ovf;   // [ntiles] dense tiles: pool offset | (OR of the tile's entry flags) << 62
    const uint16_t *pool;
    unsigned long long pool_cap;     // entries the pool holds
    __device__ __forceinline__ int64_t len() const { return n + s; }
    // entry j of dense tile t.  A scan whose dense tiles outgrow the pool is run again with a
    // larger one, but the kernels queued behind the index kernel of the failed attempt still
    // run: what such a tile did not get to store reads as 0 instead of past the allocation.
    __device__ __forceinline__ uint32_t pooled(int t, uint32_t j) const {
        const unsigned long long at = (ovf[t] & OVF_MASK) + j;
        return at < pool_cap ? (uint32_t)pool[at] : 0u;
    }
};

// Handle of one line-index entry.  tile == -2: before everything,
// tile == -1: the sentinel.
struct H {
    int32_t tile;
    int32_t i;
};

// ---- accessor over the global index (slow path, serial walker) -----------
struct GAcc {
    typedef H Hd;
    const LineIndex &L;
    __device__ GAcc(const LineIndex &l) : L(l) {}
    __device__ __forceinline__ Hd before() const { return H{-2, 0}; }
    __device__ bool next(Hd &h) const {
        if (h.tile == -2 && L.s) { h.tile = -1; h.i = 0; return true; }
        if (h.tile >= 0 && h.i + 1 < (int32_t)L.cnt[h.tile]) { h.i++; return true; }
        int32_t t = h.tile < 0 ? 0 : h.tile + 1;
        while (t < L.ready && L.cnt[t] == 0) t++;
        if (t >= L.ready) return false;
        h.tile = t; h.i = 0;
        return true;
    }
    __device__ void get(const Hd &h, int64_t &P, int &fl) const {
        if (h.tile < 0) {
            P = 0;
            const uint8_t b = L.n > 0 ? L.d[0] : 0;
            fl = (b == '@') ? FL_AT : (b == '+') ? FL_PLUS : 0;
            return;
        }
        const uint32_t c = L.cnt[h.tile];
        const uint32_t e = (c <= (uint32_t)SLOT) ? (uint32_t)L.ent[(int64_t)h.tile * SLOT + h.i]
                                                 : L.pooled(h.tile, (uint32_t)h.i);
        P = ((int64_t)h.tile << TILE_SHIFT) + (e & OFF_MASK) + L.s;
        fl = (int)(e >> 14);
    }
};

// One scanner call over the line index (wv_record below): the posbuffer fields, the status, and
// whether the final-record rule of the iterator (fastqandfurious.py:259-266) accepted it at eof.
//   /root/reference/src/_fastqandfurious.c:25-153  (C extension entrypos)
struct Rec {
    int64_t p0, p1, p3, p4, p5;
    int32_t status;
    bool final_;
};

// ---- wave-wide search over the global index ---------------------------------
// The chain is sequential, but each of its searches is not: the wave looks at 64 index entries (or
// 64 tile counts) per step and jumps straight to the tile a position bound falls into.
// first entry after `from` whose flags meet `mask` (0: any entry) at buffer coordinate >= minP;
// wave-uniform arguments and result
// WIDE: with the 512-entry steps over unflagged stretches of dense tiles (below).  k_group_walk -- the
// kernel dense regions go to -- and the group kernel's search behind a group without candidates take it;
// the others do not: inlined into the group kernel's generic node path it cost that kernel a wave of
// occupancy (118 -> 133 VGPRs), and the list-ranking and serial tiers ran 5-7 % slower per record with it.
template <bool WIDE>
__device__ __forceinline__ bool wv_find_t(const LineIndex &L, H from, int mask, int64_t minP, H &out, int64_t &Pout, int &flout)
{
    const int lane = threadIdx.x & 63;
    int t, i;
    if (from.tile == -2) {
        if (L.s) {
            const uint8_t b = L.n > 0 ? L.d[0] : 0;
            const int fl = (b == '@') ? FL_AT : (b == '+') ? FL_PLUS : 0;
            if ((mask == 0 || (fl & mask)) && 0 >= minP) { out = H{-1, 0}; Pout = 0; flout = fl; return true; }
        }
        t = 0; i = 0;
    } else if (from.tile == -1) { t = 0; i = 0; }
    else { t = from.tile; i = from.i + 1; }
    // entries of the tiles in front of the one minP falls into lie in front of minP
    const int64_t tmin = (minP - L.s) >> TILE_SHIFT;
    if (tmin > (int64_t)t) { t = (int)min(tmin, (int64_t)L.ready); i = 0; }
    while (t < L.ready) {
        const uint32_t c = L.cnt[t];
        // (dense tile searched for a flag, after a step without a hit: see below)
        const bool wide = WIDE && mask != 0 && c > (uint32_t)SLOT;
        const unsigned long long ov = wide ? L.ovf[t] : 0ull;
        const unsigned long long at0 = ov & OVF_MASK;
        const uint32_t fm = ((uint32_t)mask << 14) * 0x00010001u;           // the flag bits of both halves of a dword
        // a dense tile none of whose entries carries the flag (ovf[t]'s top bits): not looked at at all
        const uint32_t cend = (wide && !((int)(ov >> 62) & mask)) ? 0u : c;
        for (uint32_t j0 = (uint32_t)i; j0 < cend; j0 += 64) {
            const uint32_t j = j0 + lane;
            bool ok = false;
            int64_t P = 0;
            uint32_t e = 0;
            if (j < c) {
                e = (c <= (uint32_t)SLOT) ? (uint32_t)L.ent[(int64_t)t * SLOT + j] : L.pooled(t, j);
                P = ((int64_t)t << TILE_SHIFT) + (e & OFF_MASK) + L.s;
                ok = (mask == 0 || ((int)(e >> 14) & mask)) && P >= minP;
            }
            const unsigned long long m = __ballot(ok);
            if (m) {
                const int w = __ffsll((long long)m) - 1;
                out = H{t, (int32_t)(j0 + w)};
                Pout = ((int64_t)__shfl((int)(P >> 32), w) << 32) | (uint32_t)__shfl((int)(uint32_t)P, w);
                flout = __shfl((int)(e >> 14), w);
                return true;
            }
            if (wide) {
                // 64 entries of a dense tile without the flag: from here on 512 pooled entries per step
                // (eight per lane, one 16-byte load), stepping over the stretches where no entry carries
                // it -- a block of blank lines is nothing else; the step that holds a flagged entry goes
                // back to the exact one above.  (2 MB of blank lines at the end of a buffer: 14 -> 2 ms;
                // short records, found within the first step, do not come here; tools/cliffs.py)
                while (j0 + 64u + 512u <= c && at0 + j0 + 64u + 512u <= L.pool_cap) {
                    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(2)));
                    const u32x4u v = *reinterpret_cast<const u32x4u *>(L.pool + at0 + j0 + 64u + 8u * (uint32_t)lane);
                    if (__ballot(((v.x | v.y | v.z | v.w) & fm) != 0u)) break;
                    j0 += 512u;
                }
            }
        }
        // next non-empty tile, 64 counts at a time (WIDE, searching for a flag: next tile that can hold one -- 2 MB of
        // blank lines are 128 dense tiles without any flag: two steps instead of 4096 of 512 entries each)
        t++; i = 0;
        while (t < L.ready) {
            const uint32_t cl = (t + lane < L.ready) ? L.cnt[t + lane] : 1u;
            bool may = cl != 0u;
            if (WIDE && mask != 0 && cl > (uint32_t)SLOT && t + lane < L.ready) may = ((int)(L.ovf[t + lane] >> 62) & mask) != 0;
            const unsigned long long m = __ballot(may);
            if (m) { t += __ffsll((long long)m) - 1; break; }
            t += 64;
        }
    }
    return false;
}

__device__ bool wv_find(const LineIndex &L, H from, int mask, int64_t minP, H &out, int64_t &Pout, int &flout)
{
    return wv_find_t<false>(L, from, mask, minP, out, Pout, flout);
}

// the scanner call (/root/reference/src/_fastqandfurious.c:25-153) with the wave's searches: same rules, same order
template <bool WIDE>
__device__ __forceinline__ void wv_record_t(const LineIndex &L, H k, int64_t Pk, int64_t len, int eof, Rec &r, H &hm1)
{
    const int64_t NONE = -(1ll << 62);
    r.p0 = Pk + 1; r.p1 = r.p3 = r.p4 = r.p5 = -1; r.final_ = false;
    hm1 = k;
    int64_t P; int fl;
    H j = k;
    if (!wv_find_t<WIDE>(L, j, 0, NONE, j, P, fl)) { r.status = ST_HEAD_END; return; }          // :70-71
    if (P > len - 2) { r.status = ST_HEAD_END; return; }
    r.p1 = P;
    const int64_t p2 = P + 1;
    if (!wv_find_t<WIDE>(L, j, FL_PLUS, p2 + 1, j, P, fl)) { r.status = ST_SEQ_END; return; }   // :87-88
    r.p3 = P;
    if (P + 2 >= len) { r.status = ST_QUALHEAD_END; return; }
    if (!wv_find_t<WIDE>(L, j, 0, NONE, j, P, fl)) { r.status = ST_QUALHEAD_END; return; }      // :102-103
    if (P > len - 2) { r.status = ST_QUALHEAD_END; return; }
    hm1 = j;
    const int64_t qhe = P, se = r.p3, he = r.p1;
    if ((qhe - se - 1 > 1) && (qhe - se != he - r.p0 + 1)) { r.status = ST_INVALID; return; }   // :109-117
    r.p4 = qhe + 1;
    const int64_t qe = r.p4 + se - he - 1;                                               // :129
    if (qe + 2 >= len) {                                                                 // :130-133
        r.status = ST_QUAL_END;
        if (eof && qe < len) { r.p5 = qe; r.final_ = true; }                             // fastqandfurious.py:259-266
        return;
    }
    r.p5 = qe;
    r.status = ST_COMPLETE;
}

__device__ void wv_record(const LineIndex &L, H k, int64_t Pk, int64_t len, int eof, Rec &r, H &hm1)
{
    wv_record_t<false>(L, k, Pk, len, eof, r, hm1);
}


// ---- wave64 helpers -------------------------------------------------------
// inclusive prefix sum over the 64 lanes: six v_add_u32 with the DPP shift ON the add (row
// shifts with bound_ctrl, row broadcasts) -- the compiler's lowering of update_dpp is
// v_mov 0 + v_mov_dpp + v_add per step, three times the VALU time.  The s_nop are the two
// wait states a DPP read needs after the VALU write of its source.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x)
{
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(x));
    return x;
}

// wave-local ordering of LDS traffic (several independent waves share a workgroup)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 16-bit mask of the bytes of v equal to '\n' (bit p = byte p in memory order)
__device__ __forceinline__ uint32_t nl_mask16(const uint4 v)
{
    const uint32_t K = 0x0A0A0A0Au, L7 = 0x7F7F7F7Fu, H1 = 0x80808080u;
    uint32_t a = v.x ^ K, b = v.y ^ K, c = v.z ^ K, e = v.w ^ K;
    // bit 7 of each byte = 1 iff the byte is non-zero (exact, no cross-byte carry)
    a = (((a & L7) + L7) | a) & H1;
    b = (((b & L7) + L7) | b) & H1;
    c = (((c & L7) + L7) | c) & H1;
    e = (((e & L7) + L7) | e) & H1;
    // gather the four flag bits of each dword with one v_dot4_u32_u8 each
    uint32_t lo = __builtin_amdgcn_udot4(a, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(b, 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(c, 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(e, 0x80402010u, hi, false);
    return ((((hi << 8) | lo) >> 7) ^ 0xFFFFu) & 0xFFFFu;
}

// byte q (0..15) of v, memory order
__device__ __forceinline__ uint32_t get_byte(const uint4 v, uint32_t q)
{
    const uint32_t w = (q < 8) ? ((q < 4) ? v.x : v.y) : ((q < 12) ? v.z : v.w);
    return (w >> ((q & 3) * 8)) & 0xFFu;
}

// per-byte wrap-around add of two packed u8x4 (arrayadd_b, _fastqandfurious.c:161-185)
__device__ __forceinline__ uint32_t addb4(uint32_t y, uint32_t vv)
{
    return ((y & 0x7F7F7F7Fu) + (vv & 0x7F7F7F7Fu)) ^ ((y ^ vv) & 0x80808080u);
}

// Directory of the decoded-quality stream: qdir[b] = the record whose decoded bytes cover
// stream offset b << DQ_SHIFT.  Record r with bytes [q, q + len) owns every such boundary
// inside its range, so each entry below the stream's end has exactly one writer.
constexpr int DQ_SHIFT = 16;
__device__ __forceinline__ void qdir_mark(int64_t *__restrict__ qdir, int64_t qdir_cap, int64_t q, int64_t len,
                                          int64_t r)
{
    for (int64_t b = (q + (1 << DQ_SHIFT) - 1) >> DQ_SHIFT; (b << DQ_SHIFT) < q + len && b < qdir_cap; b++)
        qdir[b] = r;
}

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace ffq
