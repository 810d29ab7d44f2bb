// ffq_kernels.h -- the gfx950 kernels of the FASTQ buffer-scan path.
//
//   k_scan_lines      bytes -> line index            (HBM-bound, the dominant kernel)
//   k_chain_serial    single-lane walker: exact on any input, the last tier
//   k_finalize*       end offset / result block, published to host-mapped memory
//   k_decode_stream   Phred decode of every record's quality span (output-aligned stream)
//   k_sel_*, k_scan_i64, k_table_cut, k_table_lower_bound   table utilities
//   k_arrayadd_b/q    the reference's array utilities
//   k_synth_*, k_read_probe, k_selftest                     generators, diagnostics
// The record chain itself: ffq_rows4.h (four-line fast path), ffq_chain.h (general path),
// ffq_fasta.h (FASTA entries).
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
// wave-prefix-sum (fused DPP adds) of the per-lane counts, newline OFFSETS
// written in position order into an LDS list.  The bytes themselves are
// parked in LDS too, so that the AT / PLUS flags ("the byte after the
// newline is '@' / '+'") are looked up once per entry by the threads that
// copy the list to the tile's slot (16-byte stores), instead of inside the
// per-lane compaction loop: 279 instead of 462 VALU instructions per wave and
// half the registers (8 resident workgroups per CU).
// Algorithmic HBM traffic: TILE bytes read + 2 bytes per newline written.
// =========================================================================
__device__ __forceinline__ uint4 load_tail16(const uint8_t *d, int64_t n, int64_t at)
{
    if (at + 16 <= n) return *reinterpret_cast<const uint4 *>(d + at);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int b = 0; b < 16; b++) {
        const int64_t p = at + b;
        if (p < n) w[b >> 2] |= (uint32_t)d[p] << ((b & 3) * 8);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// flags of one entry: the byte after the newline at tile offset `off` (s_data holds the tile,
// nxt the first byte of the next tile, 0 past the end of the buffer)
__device__ __forceinline__ uint32_t entry_flags(const uint8_t *s_data, uint32_t off, uint32_t nxt, uint32_t at_char)
{
    const uint32_t nb = (off + 1u < (uint32_t)TILE) ? (uint32_t)s_data[off + 1u] : nxt;
    return (nb == at_char) ? (uint32_t)FL_AT : (nb == '+') ? (uint32_t)FL_PLUS : 0u;
}

// The body for one tile.  FULL: the tile lies completely inside the buffer (no bounds checks in
// the loads).  One tile per workgroup and as many resident waves as possible.  Measured on
// MI355X: 2 / 4 / 8 consecutive tiles per workgroup with the next tile's loads issued ahead of
// the scans, the barrier and the store tail run at 190 / 213 / 224 us per GiB against 190 us
// (fewer, longer workgroups fill the last round of the grid worse than the prefetch gains).  Round 3, measured
// again with the kernel at 163 us: persistent workgroups (the chip's 2048, grid-stride, the next tile's loads issued
// into a second register set the moment the current tile's bytes arrive, 64 VGPRs, no spills) 185-190 us, with or
// without the entry stores or the extra barrier and for any number of workgroups; a lane-major variant (every lane
// reads its own 64 consecutive bytes back from the parked tile: one mask, one count, one prefix sum, two compaction
// loops instead of four) 160-173 us, wrapped input +6 % -- profiles/r03_probes/*_index_kernel_ab.txt.
struct ScanLds {
    __attribute__((aligned(16))) uint8_t data[TILE];
    __attribute__((aligned(16))) uint16_t list[SLOT];
    uint32_t wtot[4];
    unsigned long long ovf;
    uint32_t dfl;                  // dense tile: OR of its entries' flags
};

// The last phase of a tile: the count, then each wave stores its own entries (ranks [wbase, wbase + wtot) of the
// tile, offsets in sm.list or -- dense tile -- in the pool), the AT / PLUS flags looked up on the way.
__device__ __forceinline__ void scan_tile_store(ScanLds &sm, const int tile, const uint32_t wbase, const uint32_t wtot,
                                                const uint32_t total, const bool dense, const unsigned long long pbase,
                                                const bool pool_ok, const uint32_t dense_any, const uint32_t nxt,
                                                uint16_t *__restrict__ ent, uint32_t *__restrict__ cnt,
                                                unsigned long long *__restrict__ ovf, uint16_t *__restrict__ pool,
                                                int ablate, uint32_t at_char)
{
    uint8_t *const s_data = sm.data;
    uint16_t *const s_list = sm.list;
    const int tid = threadIdx.x, l = tid & 63;
    if (tid == 0 && !(PROBES && ablate == 7)) {
        // no atomics here: an agent-scope atomic of 64 tiles on one address costs more than the
        // whole scan (measured: +45 us per GiB); the per-superblock sums are a kernel of their own
        cnt[tile] = total;
    }
    if (PROBES && (ablate == 6 || ablate == 7)) return;
    // Each wave stores its own entries, flags looked up on the way, and is done: no second
    // workgroup barrier, no wave waits for another one's store (a workgroup-wide copy of the
    // finished list cost 20 us per GiB in barrier + tail latency).
    if (!dense) {
        // (the usual tile on its own: the list comes out of LDS with plain ds reads -- one loop for
        // both cases reads through a flat pointer)
        // (the slot's address as a scalar + a 32-bit lane offset, spelled out)
        const uint16_t *gdst = ent + (int64_t)__builtin_amdgcn_readfirstlane(tile) * SLOT;
        for (uint32_t j = (uint32_t)l; j < wtot; j += 64) {
            const uint32_t off = (uint32_t)s_list[wbase + j];
            const uint32_t e = off | (entry_flags(s_data, off, nxt, at_char) << 14);
            // written once, read by the row / chain kernels from HBM later: non-temporal (-4...10 us per GiB)
            if (PROBES && ablate == 9) const_cast<uint16_t *>(gdst)[wbase + j] = (uint16_t)e;
            else asm volatile("global_store_short %0, %1, %2 nt" : : "v"((wbase + j) * 2u), "v"(e), "s"(gdst) : "memory");
        }
    } else {
        // (a dense tile's entries went to the pool with their flags, straight from the compaction loop: dense_any)
        // ovf[tile] (read only for tiles with cnt > SLOT): the pool offset, and whether any entry of the tile is
        // flagged at all (sm.dfl was zeroed in front of the barrier of the pool allocation)
        const uint32_t wany = (__ballot(dense_any & (uint32_t)FL_AT) ? (uint32_t)FL_AT : 0u) |
                              (__ballot(dense_any & (uint32_t)FL_PLUS) ? (uint32_t)FL_PLUS : 0u);
        if (l == 0 && wany) atomicOr(&sm.dfl, wany);
        __syncthreads();
        if (tid == 0) ovf[tile] = pbase | ((unsigned long long)sm.dfl << 62);
    }
}

// What follows the loads of a tile: m[] / c[] are the newline masks and counts of this lane's four
// 16-byte pieces at tile offsets o[], already parked in sm.data; nxt is the first byte behind the tile.
__device__ __forceinline__ void scan_tile_rest(ScanLds &sm, const int tile, const uint32_t (&m)[4], const uint32_t (&c)[4],
                                               const uint32_t (&o)[4], const uint32_t nxt,
                                               uint16_t *__restrict__ ent, uint32_t *__restrict__ cnt,
                                               unsigned long long *__restrict__ ovf, uint16_t *__restrict__ pool,
                                               unsigned long long pool_cap, Ctl *ctl, int ablate, uint32_t at_char)
{
    uint16_t *const s_list = sm.list;
    uint32_t *const s_wtot = sm.wtot;
    unsigned long long &s_ovf = sm.ovf;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    if (PROBES && ablate == 3) { if ((c[0] + c[1] + c[2] + c[3]) == 77u) cnt[tile] = 1; return; }
    // wave prefix sums of the four row counts, two 16-bit fields per register
    const uint32_t s01 = wave_incl_scan(c[0] | (c[1] << 16));
    const uint32_t s23 = wave_incl_scan(c[2] | (c[3] << 16));
    const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)s01, 63), t23 = (uint32_t)__builtin_amdgcn_readlane((int)s23, 63);   // (scalars: the row totals)
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
    if (PROBES && ablate == 5) { if (total + wbase == 0x7777u) cnt[tile] = 1; return; }
#ifdef FFQ_PROBES
    if (ablate == 8 && w == 0) {
        // PROBE (ffq_read_probe mode 7): what a decoupled look-back over the tiles' newline counts
        // costs on this part -- descriptors flag << 62 | value in ovf[] (zeroed before the launch),
        // relaxed agent-scope loads / stores, no read-modify-write, no fence
        unsigned long long *desc = ovf;
        const unsigned long long VM = (1ull << 62) - 1ull;
        if (tile == 0) {
            if (l == 0) __hip_atomic_store(desc, (2ull << 62) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (l == 0) __hip_atomic_store(desc + tile, (1ull << 62) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long excl = 0;
            int64_t pos = tile - 1;
            for (;;) {
                const int64_t idx = pos - l;
                const unsigned long long v = idx >= 0 ? __hip_atomic_load(desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
                const int fl = (int)(v >> 62);
                const unsigned long long inv = __ballot(fl == 0), inc = __ballot(fl == 2);
                const int first_inc = inc ? __ffsll((long long)inc) - 1 : 64;
                const int first_inv = inv ? __ffsll((long long)inv) - 1 : 64;
                if (first_inv < first_inc) { __builtin_amdgcn_s_sleep(1); continue; }      // a descriptor in between is not there yet
                const unsigned long long val = (l <= first_inc) ? (v & VM) : 0ull;
                excl += (unsigned long long)(uint32_t)__shfl((int)wave_incl_scan((uint32_t)(val & 0xFFFFFu)), 63) +
                        ((unsigned long long)(uint32_t)__shfl((int)wave_incl_scan((uint32_t)(val >> 20)), 63) << 20);
                if (first_inc < 64) break;
                pos -= 64;
            }
            if (l == 0) __hip_atomic_store(desc + tile, (2ull << 62) | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (ablate >= 100) {
        // PROBE (ffq_read_probe modes 100 + K + 16 * barrier + 32 * variant; tools/lookback_probe.py): the
        // same look-back with a window of 64 * K descriptors per round trip (lane l reads the descriptors
        // at distance l + 64 k, K loads in flight), optionally with the other waves waiting behind a barrier.
        // variant 1: the two stores only; 2: stores + ONE window load, nothing waited for; 3: only every 4th
        // tile takes part (descriptor tile >> 2: the traffic of 64 KiB super-tiles); 4: longer sleeps between
        // polls; 6: variant 2 with plain cached loads; 7: variant 2 with sc0 loads (coherent in this XCD's L2
        // only).  The inclusive descriptor carries the rounds (bits 40..47) and retries (48..61) of its look-back.
        const int K = (ablate - 100) & 15, variant = ((ablate - 100) >> 5) & 7;
        const bool part = variant != 3 || (tile & 3) == 3;
        if (w == 0 && part) {
            unsigned long long *desc = ovf;
            const int64_t me = variant == 3 ? (tile >> 2) : tile;
            const unsigned long long VM = (1ull << 40) - 1ull;
            const bool waits = variant == 0 || variant == 3 || variant == 4;
            if (me == 0) {
                if (l == 0) __hip_atomic_store(desc, (2ull << 62) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (l == 0) __hip_atomic_store(desc + me, (1ull << 62) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long excl = 0, rounds = 0, retries = 0;
                int64_t pos = me - 1;
                if (variant != 1)
                for (;;) {
                    unsigned long long vv[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int64_t idx = pos - l - 64 * k;
                        if (variant == 6) vv[k] = (k < K && idx >= 0) ? desc[idx] : 0ull;
                        else if (variant == 7) {
                            unsigned long long x = 0;
                            if (k < K && idx >= 0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(desc + idx) : "memory");
                            vv[k] = x;
                        }
                        else vv[k] = (k < K) ? (idx >= 0 ? __hip_atomic_load(desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62)) : 0ull;
                    }
                    rounds++;
                    unsigned long long add = 0;
                    bool retry = false, done = false;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (k < K && !retry && !done) {
                            const int fl = (int)(vv[k] >> 62);
                            const unsigned long long inv = __ballot(fl == 0), inc = __ballot(fl == 2);
                            const int first_inc = inc ? __ffsll((long long)inc) - 1 : 64;
                            const int first_inv = inv ? __ffsll((long long)inv) - 1 : 64;
                            if (first_inv < first_inc && waits) retry = true;
                            else {
                                const unsigned long long val = (l <= first_inc) ? (vv[k] & VM) : 0ull;
                                add += (unsigned long long)(uint32_t)__shfl((int)wave_incl_scan((uint32_t)(val & 0xFFFFFu)), 63) +
                                       ((unsigned long long)(uint32_t)__shfl((int)wave_incl_scan((uint32_t)(val >> 20)), 63) << 20);
                                if (first_inc < 64) done = true;
                            }
                        }
                    }
                    if (retry) { retries++; if (variant == 4) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(1); continue; }
                    excl += add;
                    if (done || !waits) break;
                    pos -= 64 * K;
                }
                if (l == 0) __hip_atomic_store(desc + me, (2ull << 62) | ((excl + total) & VM) | (min(rounds, 255ull) << 40) | (min(retries, 16383ull) << 48),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if ((ablate - 100) & 16) __syncthreads();
    }
#endif
    const bool dense = total > (uint32_t)SLOT;
    if (dense) {   // rare: avg line shorter than 16 bytes over the whole tile
        if (tid == 0) {
            // region tile % POOL_NB of the pool, a bump counter per region (ffq_dev.h, Ctl)
            const unsigned long long region = pool_cap / POOL_NB;
            const int b = tile & (POOL_NB - 1);
            const unsigned long long at = atomicAdd(&ctl->pool_heads[b], (unsigned long long)total);
            const bool fits = at + total <= region;
            // (what does not fit gets an offset past the pool: nothing of it is stored or read)
            s_ovf = fits ? (unsigned long long)b * region + at : pool_cap;
            sm.dfl = 0u;
            if (at == 0ull) ctl->pool_any = 1u;               // (the first allocation of a region says so: one store per region, not one per tile on ONE address)
            if (!fits) atomicOr(&ctl->err, ERR_POOL);
        }
        __syncthreads();
    }
    const unsigned long long pbase = dense ? s_ovf : 0ull;
    const bool pool_ok = dense && (pbase + total <= pool_cap);

    // newline offsets in position order (no flags yet); this wave's entries are ranks
    // [wbase, wbase + wtot) of the tile
    uint32_t rb = wbase;
    uint32_t dense_any = 0;         // dense tile: OR of the flags of this lane's entries
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t mm = m[i];
        uint32_t idx = rb + ex[i];
        while (mm) {
            const uint32_t p = (uint32_t)__ffs((int)mm) - 1u;
            mm &= mm - 1u;
            if (!dense) s_list[idx] = (uint16_t)(o[i] + p);
            else if (pool_ok) {
                // (every wave's bytes are parked: the barrier behind the wave totals.  The entry goes out whole, one
                // store -- until round 3 the flags were added by a second pass that read every pooled entry back)
                const uint32_t fl = entry_flags(sm.data, o[i] + p, nxt, at_char);
                dense_any |= fl;
                pool[pbase + idx] = (uint16_t)((o[i] + p) | (fl << 14));
            }
            idx++;
        }
        rb += rowtot[i];
    }
    scan_tile_store(sm, tile, wbase, wtot, total, dense, pbase, pool_ok, dense_any, nxt, ent, cnt, ovf, pool, ablate, at_char);
}

// WIDE (round 6): the same pass also writes EVERY byte decoded -- out[p] = d[p] + qadd (int8 arithmetic) at the offset the
// byte has in the input -- so that the Phred decode of ANY record layout (wrapped records: which lines are quality lines is
// known only behind the chain) needs no second read of the input: record i's bytes are out[pos4 : pos5] in the
// coordinates of d (embedded newlines of a wrapped quality come out as '\n' + qadd, as the reference's slice + arrayadd_b
// give them, /root/reference/src/_fastqandfurious.c:129, doc/user-guide.rst:126-141).  The price is the bytes nobody asked
// for (headers, bases): input + as much written instead of input read twice + the qualities written.  The output must hold
// ntiles * TILE bytes (the ragged last tile writes whole 16-byte pieces).
template <bool FULL, bool WIDE = false>
__device__ __forceinline__ void scan_tile(ScanLds &sm, const int tile, const uint8_t *__restrict__ d, int64_t n,
                                          uint16_t *__restrict__ ent, uint32_t *__restrict__ cnt,
                                          unsigned long long *__restrict__ ovf, uint16_t *__restrict__ pool,
                                          unsigned long long pool_cap, Ctl *ctl, int ablate, uint32_t at_char,
                                          int8_t *__restrict__ wout = nullptr, uint32_t wadd = 0)
{
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int64_t base = (int64_t)tile << TILE_SHIFT;

    uint4 v[4];
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (uint32_t)(w * 4096 + i * 1024 + l * 16);
    if (FULL) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // non-temporal: the input streams through once; keeping it out of the L2's way lets the
            // index lines this kernel writes leave for HBM in bulk instead of trickling out between
            // the reads (-7 us per GiB)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(d + base + o[i]));
            v[i] = make_uint4(t.x, t.y, t.z, t.w);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = load_tail16(d, n, base + o[i]);
    }
    // first byte of the next tile (workgroup-uniform): the flags of a newline at offset TILE-1
    const uint32_t nxt = (base + TILE < n) ? (uint32_t)d[base + TILE] : 0u;

    uint32_t m[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        *reinterpret_cast<uint4 *>(sm.data + o[i]) = v[i];
        m[i] = nl_mask16(v[i]);
        c[i] = __popc(m[i]);
    }
    if (WIDE) {
        const uint32_t vv = (wadd & 0xFFu) * 0x01010101u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 t; t.x = addb4(v[i].x, vv); t.y = addb4(v[i].y, vv); t.z = addb4(v[i].z, vv); t.w = addb4(v[i].w, vv);
            __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(wout + base + o[i]));
        }
    }
    scan_tile_rest(sm, tile, m, c, o, nxt, ent, cnt, ovf, pool, pool_cap, ctl, ablate, at_char);
}

// The launch: workgroup b takes whole tile tile0 + b.  Only the last tile of a buffer can be
// ragged; it is workgroup 0's second tile (ragged_tile >= 0), with bounds-checked loads: a
// variant of the whole kernel with a ragged test in front of the loads ran 3-8 us per GiB slower
// on every tile, and workgroup 0 is long done when the last round of the grid starts.
// WHOLE = false: no whole tile at all (a buffer shorter than a tile), one workgroup.
template <bool WHOLE, int MINW, bool WIDE = false>
__global__ __launch_bounds__(256, MINW) void k_scan_lines(const uint8_t *__restrict__ d, int64_t n,
                                                    uint16_t *__restrict__ ent,
                                                    uint32_t *__restrict__ cnt,
                                                    unsigned long long *__restrict__ ovf,
                                                    uint16_t *__restrict__ pool,
                                                    unsigned long long pool_cap, Ctl *ctl, int tile0,
                                                    int ablate, LineIndex Lval, LineIndex *__restrict__ d_L,
                                                    uint32_t at_char, int ragged_tile, int8_t *__restrict__ wout, uint32_t wadd)
{
    __shared__ ScanLds sm;
    if (WHOLE) scan_tile<true, WIDE>(sm, tile0 + (int)blockIdx.x, d, n, ent, cnt, ovf, pool, pool_cap, ctl, ablate, at_char, wout, wadd);
    if (blockIdx.x == 0) {
        if (ragged_tile >= 0) {
            if (WHOLE) __syncthreads();
            scan_tile<false, WIDE>(sm, ragged_tile, d, n, ent, cnt, ovf, pool, pool_cap, ctl, ablate, at_char, wout, wadd);
        }
        // the device copy of the index descriptor (out-of-line device functions take it by pointer)
        if (threadIdx.x == 0 && d_L) *d_L = Lval;
    }
}

}  // namespace ffq

#include "ffq_rows4.h"
#include "ffq_dense.h"
#include "ffq_lite.h"

namespace ffq {

// =========================================================================
// k_chain_serial: the whole chain by ONE WAVE over the global index, record after record.
// Exact on any input (dense tiles, records longer than a window, chains that do not
// re-synchronise).  The chain is sequential, but each of its searches is not: the wave looks
// at 64 index entries (or 64 tile counts) per step and jumps straight to the tile a position
// bound falls into, so a record costs a handful of memory round trips whatever its length
// (a single lane walking entry by entry paid one per line: 0.3 GB/s on long wrapped records).
// =========================================================================
// (wv_find, the wave-wide search over the global index, lives in ffq_dev.h: the group kernel uses it too)
// (wv_record, the scanner call with the wave's searches, lives in ffq_dev.h beside wv_find)
__global__ __launch_bounds__(64) void k_chain_serial(LineIndex L, int64_t offset, int eof, int64_t add,
                                                     int64_t *__restrict__ table, int64_t table_cap,
                                                     int64_t *__restrict__ qoff, int64_t *__restrict__ qdir,
                                                     int64_t qdir_cap, int64_t *__restrict__ p4s, int64_t p4_cap,
                                                     DevRes *res)
{
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t len = L.len();
    int64_t n = 0, qb = 0, off = offset;
    H k, hm1;
    int64_t Pk;
    int flk;
    Rec r;
    int status = ST_HEAD_BEG, end;
    r.p0 = r.p1 = r.p3 = r.p4 = r.p5 = -1; r.final_ = false;
    bool have = wv_find(L, H{-2, 0}, FL_AT, offset, k, Pk, flk);
    for (;;) {
        if (!have) { status = ST_HEAD_BEG; r.p0 = r.p1 = r.p3 = r.p4 = r.p5 = -1; r.final_ = false; break; }
        wv_record(L, k, Pk, len, eof, r, hm1);
        status = r.status;
        if (status != ST_COMPLETE && !r.final_) break;
        if (n < table_cap && lane == 0) {
            int64_t *o = table + n * 6;
            o[0] = r.p0 + add; o[1] = r.p1 + add; o[2] = r.p1 + 1 + add;
            o[3] = r.p3 + add; o[4] = r.p4 + add; o[5] = r.p5 + add;
            if (qoff) {
                qoff[n] = qb; qdir_mark(qdir, qdir_cap, qb, r.p5 - r.p4, n);
                if (n < p4_cap) p4s[n] = r.p4 + add;
            }
        }
        n++;
        qb += r.p5 - r.p4;
        if (r.final_) break;
        off = r.p5 - 1;
        have = wv_find(L, hm1, FL_AT, r.p5 - 1, k, Pk, flk);
    }
    // newlines of the buffer (all lanes)
    unsigned long long nlp = 0;
    for (int t = lane; t < L.ntiles; t += 64) nlp += L.cnt[t];
    const int64_t nl = (int64_t)wave_sum_u32((uint32_t)(nlp & 0xFFFFFu)) + ((int64_t)wave_sum_u32((uint32_t)(nlp >> 20)) << 20);
    if (lane != 0) return;
    if (r.final_) end = 0;
    else if (status == ST_HEAD_BEG) end = eof ? 0 : 1;
    else if (eof) end = (status == ST_QUAL_END) ? 2 : (status == ST_INVALID) ? 4 : 3;
    else end = (status == ST_INVALID) ? 4 : 1;
    res->fallback = 0;               // the walker is exact: whatever a parallel attempt left here is void
    res->n_records = n;
    res->n_qual_bytes = qb;
    res->end_offset = off;
    res->last_status = status;
    res->end_state = end;
    res->has_final = r.final_ ? 1 : 0;
    const int64_t p[6] = {r.p0, r.p1, r.p1 >= 0 ? r.p1 + 1 : -1, r.p3, r.p4, r.p5};
    for (int i = 0; i < 6; i++) res->last_pos[i] = p[i] >= 0 ? p[i] + add : -1;
    res->n_lines = nl;
}

// (the walk through dense regions -- k_group_walk until round 3 -- is k_dense_walk, ffq_dense.h)

// k_finalize: the iterator's `offset` at exit = pos5 - 1 of the last COMPLETE
// record (fastqandfurious.py:254), read back from the table; qoff[n].
// in_place_s >= 0 (the sentinel flag of a scan whose index pass decoded every byte in place): the qualities "end" where
// the last record's do in the buffer -- n_qual_bytes and qoff[n] as k_qtotal4 leaves them for the in-place single pass.
__global__ void k_finalize(DevRes *res, const int64_t *__restrict__ table, int64_t table_cap,
                           int64_t add, int64_t offset, int64_t *__restrict__ qoff, Pub pb, int in_place_s)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (res->fallback) { publish(pb, res); return; }
    const int64_t ncomplete = res->n_records - (res->has_final ? 1 : 0);
    if (ncomplete > 0 && ncomplete <= table_cap) res->end_offset = table[(ncomplete - 1) * 6 + 5] - add - 1;
    else res->end_offset = offset;
    if (qoff && in_place_s >= 0) {
        const int64_t n = res->n_records;
        res->n_qual_bytes = (n > 0 && n <= table_cap) ? table[(n - 1) * 6 + 5] - add - in_place_s : 0;
    }
    if (qoff && res->n_records <= table_cap) qoff[res->n_records] = res->n_qual_bytes;
    publish(pb, res);
}

// qoff[i] = the offset pos4 of row i has in the buffer, qoff[n] = n_qual_bytes = where the last record's qualities end there:
// the offsets of a scan whose index pass decoded every byte in place, for the tiers that do not write them on the way
// (list ranking, the one-wave walker).  Publishes.
__global__ __launch_bounds__(256) void k_qoff_in_place(DevRes *res, const int64_t *__restrict__ table, int64_t table_cap, int64_t add, int s,
                                                       int64_t *__restrict__ qoff, Pub pb)
{
    const int64_t n = res->fallback ? 0 : min(res->n_records, table_cap);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        qoff[i] = table[i * 6 + 4] - add - s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (!res->fallback) {
            const int64_t nr = res->n_records;
            const int64_t tot = (nr > 0 && nr <= table_cap) ? table[(nr - 1) * 6 + 5] - add - s : 0;
            res->n_qual_bytes = tot;
            if (nr <= table_cap) qoff[nr] = tot;
        }
        publish(pb, res);
    }
}

__global__ void k_finalize_serial(DevRes *res, int64_t table_cap, int64_t *__restrict__ qoff, Pub pb)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (qoff && res->n_records <= table_cap) qoff[res->n_records] = res->n_qual_bytes;
    publish(pb, res);
}

// publisher of a front that ends without a result kernel (the list-ranking tier and the one-wave walker are driven from
// the host's wait): the result block on the DEVICE says so -- fallback = 1, "a later tier follows" -- for whoever reads it
// behind this front without a host round trip (k_shard_words: the words of such a rank are "not ready", gathered again
// once its wait is through; before this mark they were the PREVIOUS scan's)
__global__ void k_publish(DevRes *res, Pub pb, int later_tier_follows)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (later_tier_follows) res->fallback = 1;
    publish(pb, res);
}

// =========================================================================
// k_decode_stream: Phred decode over a finished table + CSR offsets (all paths).
//   array('b').frombytes(buf[pos4:pos5]); arrayadd_b(q, -33)
//   (/root/reference/doc/user-guide.rst:130-141, src/demo/benchmark.py:159-168)
// The OUTPUT stream is cut into blocks of DQ_BLK bytes, one per workgroup, and those into
// 16-byte chunks aligned on the destination address.  qdir[b] (written by whoever produced
// qoff, see qdir_mark) names the record under the block's first byte; the workgroup keeps
// (offset, source) of the records under its block in LDS and gathers every chunk from its
// record(s) with unaligned 16-byte loads, DQ_PER chunks per thread in flight.  Stores are
// whole aligned 16-byte pieces, 1 KiB per wave instruction.
// Algorithmic traffic per record: quality bytes read + written, 16 B of (qoff, pos4) read
// (pos4 from the compact copy the row kernels write beside the table: the 48-byte rows would
// come in whole lines, 1.5 GiB instead of 0.25 per 10 GiB of input).
// =========================================================================
constexpr int DQ_PER = 4;                         // chunks per thread and batch
constexpr int DQ_BLK = 1 << DQ_SHIFT;             // output bytes per workgroup
constexpr int DQ_REC = 1024;                      // records cached in LDS per window

// dword w of the 16-byte mask with bytes [0, nb) set
__device__ __forceinline__ uint32_t lt_mask(int nb, int w)
{
    const int t = nb - 4 * w;
    return t <= 0 ? 0u : (t >= 4 ? 0xFFFFFFFFu : ((1u << (8 * t)) - 1u));
}

// 16 bytes at buffer offset a; bytes outside [0, nbytes) read as zero
__device__ __noinline__ uint4 load16_edge(const uint8_t *__restrict__ d, int64_t nbytes, int64_t a)
{
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (a + k >= 0 && a + k < nbytes) w[k >> 2] |= (uint32_t)d[a + k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint4 load16_any(const uint8_t *__restrict__ d, int64_t nbytes, int64_t a)
{
    if (a >= 0 && a + 16 <= nbytes) {
        // unaligned, non-temporal: the input passes through once more and is not needed again
        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
        const auto t = __builtin_nontemporal_load(reinterpret_cast<const u32x4u *>(d + a));
        return make_uint4(t.x, t.y, t.z, t.w);
    }
    return load16_edge(d, nbytes, a);
}

// bytes [kb, ke) of a chunk whose first byte goes to o[0]
__device__ __noinline__ void store16_part(int8_t *__restrict__ o, uint4 v, int kb, int ke)
{
    const uint32_t y[4] = {v.x, v.y, v.z, v.w};
    for (; kb < ke; kb++) {
        uint32_t wv = y[0];
#pragma unroll
        for (int w = 1; w < 4; w++)
            if ((kb >> 2) == w) wv = y[w];
        o[kb] = (int8_t)(uint8_t)(wv >> (8 * (kb & 3)));
    }
}

// records shorter than 16 bytes: chunk bytes [kb, kend) gathered byte by byte from the
// cached records m, m+1, ...
__device__ __noinline__ uint4 gather_tail(const uint8_t *__restrict__ d, const int32_t *s_q, const int64_t *s_adj,
                                          uint4 v, int m, int kb, int kend, int clo, int vhi)
{
    uint32_t y[4] = {v.x, v.y, v.z, v.w};
    while (kb < kend) {
        const int he = min(s_q[m + 1], vhi) - clo;
        const int64_t sa = s_adj[m] + clo;
        for (; kb < he; kb++) {
            const uint32_t sh = 8u * (kb & 3), val = (uint32_t)d[sa + kb] << sh, msk = ~(0xFFu << sh);
#pragma unroll
            for (int w = 0; w < 4; w++)
                if ((kb >> 2) == w) y[w] = (y[w] & msk) | val;
        }
        m++;
    }
    return make_uint4(y[0], y[1], y[2], y[3]);
}

__global__ __launch_bounds__(256) void k_decode_stream(const uint8_t *__restrict__ d, int64_t nbytes, int s,
                                                       const int64_t *__restrict__ p4s,
                                                       const int64_t *__restrict__ qoff,
                                                       const int64_t *__restrict__ qdir,
                                                       const DevRes *__restrict__ res,
                                                       int64_t table_cap, int64_t add, int qadd,
                                                       int8_t *__restrict__ out, int64_t out_cap, int ablate)
{
    __shared__ int32_t s_q[DQ_REC];               // stream offset of cached record i, relative to ob
    __shared__ int64_t s_adj[DQ_REC];             // buffer offset of the byte decoded to stream offset ob
    const int tid = threadIdx.x;
    const int64_t ob = (int64_t)blockIdx.x << DQ_SHIFT;
    int64_t rbase = qdir[blockIdx.x];             // garbage past the end of the stream: not used then
    const int64_t n = res->n_records;
    const int64_t qtotal = res->n_qual_bytes;     // == qoff[n]
    if (res->fallback || n > table_cap || n <= 0) return;        // a table that overflowed decodes nothing
    const int64_t total = min(qtotal, out_cap);
    if (ob >= total) return;
    const int oe = (int)min((int64_t)DQ_BLK, total - ob);        // block-relative from here on
    const int shiftA = (int)(reinterpret_cast<uintptr_t>(out + ob) & 15);
    int8_t *__restrict__ outb = out + ob;
    const int mean = (int)min(max(qtotal / n, (int64_t)1), (int64_t)1 << 20);
    const uint32_t vv = (uint32_t)(uint8_t)qadd * 0x01010101u;

    // chunk k covers [16 k - shiftA, +16) cut to [0, oe); a destination that is not 16-byte
    // aligned has one more (partial) chunk at the end
    const int nchunk = (oe + shiftA + 15) >> 4;
    int done = 0;                                  // chunks [0, done) are written
    int want = 0;
    for (;;) {
        // ---- window: (offset, source) of records rbase .. rbase + nrec in LDS.  Sized from the
        //      mean quality length; a window that covers no whole chunk is redone at full size
        const int rem = oe - max(16 * done - shiftA, 0);
        want = (want < 0) ? DQ_REC - 1 : min(DQ_REC - 1, rem / mean + rem / (8 * mean) + 8);
        const int nrec = (int)min((int64_t)want, n - rbase);     // >= 1
        for (int i = tid; i <= nrec; i += 256) {
            // both loads before either is used (the source position of index nrec is not needed:
            // a clamped address keeps the load unconditional)
            const int64_t qraw = qoff[rbase + i];
            const int64_t p4 = p4s[rbase + min(i, nrec - 1)];            // (pos4 of the rows, compact)
            asm volatile("" ::"v"(qraw), "v"(p4));
            const int64_t q = qraw - ob;
            s_q[i] = (int32_t)min(max(q, (int64_t)-0x7FFFFFFF), (int64_t)0x7FFFFFFF);
            if (i < nrec) s_adj[i] = p4 - add - s - q;
        }
        __syncthreads();
        const int cend = s_q[nrec];                // every byte below cend has its record cached
        const int klim = (rbase + nrec == n || cend >= oe) ? nchunk : min(nchunk, (cend + shiftA) >> 4);
        const float inv_mean = (float)nrec / (float)(cend - s_q[0]);

        for (int k0 = done; k0 < klim; k0 += 256 * DQ_PER) {
            // phase 1: the record under the first byte of each chunk; phase 2: 16 bytes from that
            // record and 16 from the next, positioned chunk-relative, all loads in flight
            // together; phase 3: byte-select, rare tails (records under 16 bytes), store
            int ci[DQ_PER], h0[DQ_PER], h1[DQ_PER];
            uint4 xa[DQ_PER], xb[DQ_PER];
#pragma unroll
            for (int j = 0; j < DQ_PER; j++) {
                const int k = k0 + j * 256 + tid;
                const int clo = 16 * k - shiftA;
                const int vlo = max(clo, 0), vhi = min(clo + 16, oe);
                ci[j] = -1;
                if (k >= klim || vlo >= vhi) continue;
                // largest cached index a with s_q[a] <= vlo (it is below nrec); equal-length
                // records make the interpolated guess exact
                int a = 0, b = nrec - 1;
                const int g = min(max((int)((float)(vlo - s_q[0]) * inv_mean), 0), nrec - 1);
                if (s_q[g] <= vlo) { a = g; if (s_q[g + 1] > vlo) b = g; } else b = g - 1;
                while (b > a) {
                    const int m = (a + b + 1) >> 1;
                    if (s_q[m] <= vlo) a = m; else b = m - 1;
                }
                ci[j] = a;
                h0[j] = min(s_q[a + 1], vhi) - clo;                 // chunk bytes [.., h0) come from record a
                h1[j] = (h0[j] < vhi - clo) ? min(s_q[a + 2], vhi) - clo : h0[j];
            }
#pragma unroll
            for (int j = 0; j < DQ_PER; j++) {
                xa[j] = xb[j] = make_uint4(0, 0, 0, 0);
                if (ci[j] < 0 || (PROBES && (ablate & 4))) continue;
                const int clo = 16 * (k0 + j * 256 + tid) - shiftA;
                xa[j] = load16_any(d, nbytes, s_adj[ci[j]] + clo);
                if (h1[j] > h0[j]) xb[j] = load16_any(d, nbytes, s_adj[ci[j] + 1] + clo);
            }
#pragma unroll
            for (int j = 0; j < DQ_PER; j++) {
                if (ci[j] < 0) continue;
                const int clo = 16 * (k0 + j * 256 + tid) - shiftA;
                const int vlo = max(clo, 0), vhi = min(clo + 16, oe);
                uint32_t y[4];
                {
                    const uint32_t A[4] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w};
                    const uint32_t B[4] = {xb[j].x, xb[j].y, xb[j].z, xb[j].w};
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const uint32_t m = lt_mask(h0[j], w);
                        y[w] = (A[w] & m) | (B[w] & ~m);
                    }
                }
                if (h1[j] < vhi - clo) {
                    const uint4 t = gather_tail(d, s_q, s_adj, make_uint4(y[0], y[1], y[2], y[3]), ci[j] + 2, h1[j],
                                                vhi - clo, clo, vhi);
                    y[0] = t.x; y[1] = t.y; y[2] = t.z; y[3] = t.w;
                }
#pragma unroll
                for (int w = 0; w < 4; w++) y[w] = addb4(y[w], vv);
                if (PROBES && (ablate & 2)) { if (y[0] == 0x12345678u && y[1] == 77u) outb[0] = 1; }
                else if (vhi - vlo == 16) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 t; t.x = y[0]; t.y = y[1]; t.z = y[2]; t.w = y[3];
                    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(outb + clo));      // write-once stream
                } else {
                    store16_part(outb + clo, make_uint4(y[0], y[1], y[2], y[3]), vlo - clo, vhi - clo);
                }
            }
        }
        if (klim >= nchunk) break;
        // ---- next window starts at the record under the first byte of chunk klim
        int nb = -1;
        if (klim > done) {
            const int vlo = 16 * klim - shiftA;    // > 0 and < cend
            int a = 0, b = nrec - 1;
            while (b > a) {
                const int m = (a + b + 1) >> 1;
                if (s_q[m] <= vlo) a = m; else b = m - 1;
            }
            nb = a;
        }
        __syncthreads();                           // the cache is rewritten next
        if (nb < 0) { want = -1; continue; }       // no whole chunk covered: full-size window, same base
        rbase += nb;
        done = klim;
        want = 0;
    }
}

// =========================================================================
// Row selection by sequence length (push-down of the length filter of
// /root/reference/doc/user-guide.rst:153-180, evaluated on the offset table):
// stable compaction of the rows with min_len <= pos3 - pos2 <= max_len.
//   k_sel_count    one row per thread: kept rows per workgroup
//   k_scan_i64     exclusive scan of those counts (one workgroup)
//   k_sel_scatter  rank inside the workgroup by ballots, 48-byte row copy
// =========================================================================
__device__ __forceinline__ bool sel_keep(const int64_t *__restrict__ table, int64_t i, int64_t n, int64_t lo,
                                         int64_t hi)
{
    if (i >= n) return false;
    const longlong2 p = *reinterpret_cast<const longlong2 *>(table + i * 6 + 2);      // pos2, pos3
    const int64_t ln = p.y - p.x;
    return ln >= lo && ln <= hi;
}

__global__ __launch_bounds__(256) void k_sel_count(const int64_t *__restrict__ table, int64_t n, int64_t lo,
                                                   int64_t hi, unsigned int *__restrict__ bcnt)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = __syncthreads_count(sel_keep(table, i, n, lo, hi) ? 1 : 0);
    if (threadIdx.x == 0) bcnt[blockIdx.x] = (unsigned int)c;
}

__global__ __launch_bounds__(1024) void k_scan_i64(const unsigned int *__restrict__ v, int64_t nv,
                                                   long long *__restrict__ base, long long *__restrict__ total)
{
    // Exclusive scan of u32 counts into i64, one workgroup.  A thread owns 32 CONSECUTIVE values (eight 16-byte
    // loads in flight, a sum in registers), the 1024 thread sums are scanned once per 32768 values, and the thread
    // writes its 32 results -- one barrier round per 32768 values (a round per 1024 values took 65 us for the
    // 65536 tiles of a GiB, 1 us each).
    __shared__ long long s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    long long carry = 0;
    const bool vec = (reinterpret_cast<uintptr_t>(v) & 15) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0;
    for (int64_t c0 = 0; c0 < nv; c0 += 32768) {
        const int64_t b0 = c0 + (int64_t)tid * 32;
        uint32_t x[32];
        if (vec && b0 + 32 <= nv) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint4 t = *reinterpret_cast<const uint4 *>(v + b0 + 4 * k);
                x[4 * k] = t.x; x[4 * k + 1] = t.y; x[4 * k + 2] = t.z; x[4 * k + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 32; k++) x[k] = (b0 + k < nv) ? v[b0 + k] : 0u;
        }
        // (values are counts per tile / block: 32 of them and a wave's 64 sums fit 32 bits ... the running total
        // of a thread is kept in 64 bits all the same)
        unsigned long long mine = 0;
#pragma unroll
        for (int k = 0; k < 32; k++) mine += x[k];
        // scan of the thread sums: within the wave in two 32-bit halves, the 16 wave totals through LDS
        const uint32_t lo = wave_incl_scan((uint32_t)(mine & 0xFFFFFFu)), hi = wave_incl_scan((uint32_t)(mine >> 24));
        const unsigned long long incl = ((unsigned long long)hi << 24) + lo;
        if (lane == 63) s_w[wid] = (long long)incl;
        __syncthreads();
        long long wpre = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const long long t = s_w[q];
            if (q < wid) wpre += t;
            tot += t;
        }
        long long run = carry + wpre + (long long)(incl - mine);
        if (vec && b0 + 32 <= nv) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                longlong2 o;
                o.x = run; run += x[2 * k];
                o.y = run; run += x[2 * k + 1];
                *reinterpret_cast<longlong2 *>(base + b0 + 2 * k) = o;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 32; k++) {
                if (b0 + k < nv) base[b0 + k] = run;
                run += x[k];
            }
        }
        __syncthreads();
        carry += tot;
    }
    if (tid == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_sel_scatter(const int64_t *__restrict__ table, int64_t n, int64_t lo,
                                                     int64_t hi, const long long *__restrict__ bbase,
                                                     int64_t *__restrict__ out, int64_t *__restrict__ idx_out)
{
    // idx_out (optional): the ordinal in `table` of every kept row -- what a host that must hand back one item per
    // ORIGINAL row (None for a dropped one, /root/reference/doc/user-guide.rst:166-170) puts the kept ones back by
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    const bool keep = sel_keep(table, i, n, lo, hi);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_w[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t wpre = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q < wid) wpre += s_w[q];
    if (!keep) return;
    const int64_t r = bbase[blockIdx.x] + wpre + __popcll(m & ((1ull << lane) - 1ull));
    const longlong2 *src = reinterpret_cast<const longlong2 *>(table + i * 6);
    longlong2 *dst = reinterpret_cast<longlong2 *>(out + r * 6);
    const longlong2 a = src[0], b = src[1], c = src[2];
    dst[0] = a; dst[1] = b; dst[2] = c;
    if (idx_out) idx_out[r] = i;
}

// =========================================================================
// Column selection (push-down of an entryfunc that builds only ONE component of an entry,
// /root/reference/doc/user-guide.rst:153-180: `buf[posarray[2]:posarray[3]]`; the header as
// entryfunc cuts it, /root/reference/src/fastqandfurious.py:161-171: `buf[pos[0] + 1:pos[1]]`):
// the packed stream buf[pos[ca] + shift : pos[cb]] (+ value) of every row and its CSR offsets.
//   k_col_sum      bytes of the component per block of 256 rows
//   k_scan_i64v    exclusive scan of those (one workgroup); total -> a result block
//   k_col_offsets  offsets[i], start[i], directory of the output stream (qdir_mark)
//   k_decode_stream (above) then copies: it is the Phred decode with another pair of columns
// =========================================================================
__device__ __forceinline__ int64_t col_len(const int64_t *__restrict__ table, int64_t i, int64_t n, int ca, int shift,
                                           int cb, int64_t &start)
{
    if (i >= n) { start = 0; return 0; }
    start = table[i * 6 + ca] + shift;
    const int64_t len = table[i * 6 + cb] - start;
    return len > 0 ? min(len, (int64_t)0x7FFFFFF0) : 0;
}

// exclusive prefix of `len` inside a 256-thread workgroup (lengths below 2^31) and the block's sum
__device__ __forceinline__ int64_t block_excl_scan_len(int64_t len, int64_t &block_sum)
{
    __shared__ long long s_w[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t lo = wave_incl_scan((uint32_t)len & 0xFFFFu), hi = wave_incl_scan((uint32_t)(len >> 16));
    const long long incl = ((long long)hi << 16) + (long long)lo;
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    long long wpre = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const long long t = s_w[q];
        if (q < wid) wpre += t;
        tot += t;
    }
    block_sum = tot;
    return wpre + incl - len;
}

__global__ __launch_bounds__(256) void k_col_sum(const int64_t *__restrict__ table, int64_t n, int ca, int shift, int cb,
                                                 long long *__restrict__ bsum)
{
    int64_t start, tot;
    const int64_t len = col_len(table, (int64_t)blockIdx.x * 256 + threadIdx.x, n, ca, shift, cb, start);
    (void)block_excl_scan_len(len, tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// in-place exclusive scan of int64 values by one workgroup; the total goes to res (n_records = n,
// n_qual_bytes = total: what k_decode_stream reads)
__global__ __launch_bounds__(1024) void k_scan_i64v(long long *__restrict__ v, int64_t nv, int64_t n_rows, DevRes *res)
{
    // A thread owns 32 CONSECUTIVE values (a sum in registers), the 1024 thread sums are scanned once per 32768 values (six
    // shuffles inside the wave, the sixteen wave totals through LDS): one barrier round per 32768 values.  (The
    // Hillis-Steele version this replaces took twenty barriers per 1024 values: 483 us for the 262144 tiles of 4 GiB --
    // a fifth of the list-ranking tier's time --, 200 us for the row blocks of a 10 GiB table.)
    __shared__ long long s_w[16];
    constexpr int PER = 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    long long carry = 0;
    for (int64_t c0 = 0; c0 < nv; c0 += 1024 * PER) {
        const int64_t b0 = c0 + (int64_t)tid * PER;
        long long x[PER];
        if (b0 + PER <= nv) {
#pragma unroll
            for (int k = 0; k < PER / 2; k++) {
                const longlong2 t = *reinterpret_cast<const longlong2 *>(v + b0 + 2 * k);      // (v: hipMalloc'ed, b0 a multiple of 32)
                x[2 * k] = t.x; x[2 * k + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; k++) x[k] = (b0 + k < nv) ? v[b0 + k] : 0;
        }
        long long mine = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) mine += x[k];
        long long incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        if (lane == 63) s_w[wid] = incl;
        __syncthreads();
        long long wpre = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const long long t = s_w[q];
            if (q < wid) wpre += t;
            tot += t;
        }
        long long run = carry + wpre + incl - mine;
        if (b0 + PER <= nv) {
#pragma unroll
            for (int k = 0; k < PER / 2; k++) {
                longlong2 o;
                o.x = run; run += x[2 * k];
                o.y = run; run += x[2 * k + 1];
                *reinterpret_cast<longlong2 *>(v + b0 + 2 * k) = o;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; k++) { if (b0 + k < nv) v[b0 + k] = run; run += x[k]; }
        }
        __syncthreads();
        carry += tot;
    }
    if (tid == 0) { res->n_records = n_rows; res->n_qual_bytes = carry; res->fallback = 0; }
}

// The same scan in two levels for long arrays (the per-tile counts of a 4 GiB buffer: 262 144 values took the one
// workgroup above 132 us -- eight rounds, every lane reading its own 256 bytes): k_scan_blksum, one workgroup per 2048
// values, their sums; k_scan_i64v over the sums (it also writes the total); k_scan_blkapply, the scan inside each block
// on top of its base.  Three launches of a few microseconds.
constexpr int SCAN_BLK = 2048;

__device__ __forceinline__ long long scan_blk_load8(const long long *__restrict__ v, int64_t nv, int64_t b0, long long (&x)[8])
{
    if (b0 + 8 <= nv) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const longlong2 t = *reinterpret_cast<const longlong2 *>(v + b0 + 2 * k);      // (v: hipMalloc'ed, b0 a multiple of 8)
            x[2 * k] = t.x; x[2 * k + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = (b0 + k < nv) ? v[b0 + k] : 0;
    }
    long long s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += x[k];
    return s;
}

__global__ __launch_bounds__(256) void k_scan_blksum(const long long *__restrict__ v, int64_t nv, long long *__restrict__ bs)
{
    __shared__ long long s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    long long x[8];
    long long s = scan_blk_load8(v, nv, (int64_t)blockIdx.x * SCAN_BLK + (int64_t)tid * 8, x);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) s_w[wid] = s;
    __syncthreads();
    if (tid == 0) bs[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// ... and of u32 counts into i64 offsets (k_scan_i64's contract: base[i] = values in front of i, *total = all of them)
__device__ __forceinline__ long long scan_blk_load8u(const unsigned int *__restrict__ v, int64_t nv, int64_t b0, uint32_t (&x)[8])
{
    if (b0 + 8 <= nv) {
        const uint4 a = *reinterpret_cast<const uint4 *>(v + b0), b = *reinterpret_cast<const uint4 *>(v + b0 + 4);   // (hipMalloc'ed, b0 a multiple of 8)
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = (b0 + k < nv) ? v[b0 + k] : 0u;
    }
    long long s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += (long long)x[k];
    return s;
}

__global__ __launch_bounds__(256) void k_scan_blksum_u32(const unsigned int *__restrict__ v, int64_t nv, long long *__restrict__ bs)
{
    __shared__ long long s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t x[8];
    long long s = scan_blk_load8u(v, nv, (int64_t)blockIdx.x * SCAN_BLK + (int64_t)tid * 8, x);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) s_w[wid] = s;
    __syncthreads();
    if (tid == 0) bs[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void k_scan_blkapply_u32(const unsigned int *__restrict__ v, int64_t nv, const long long *__restrict__ bs,
                                                           long long *__restrict__ base, long long *__restrict__ total)
{
    __shared__ long long s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * SCAN_BLK + (int64_t)tid * 8;
    const long long bb = bs[blockIdx.x];
    uint32_t x[8];
    const long long mine = scan_blk_load8u(v, nv, b0, x);
    long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    long long wpre = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q < wid) wpre += s_w[q];
    long long run = bb + wpre + incl - mine;
#pragma unroll
    for (int k = 0; k < 8; k++) { if (b0 + k < nv) base[b0 + k] = run; run += (long long)x[k]; }
    if (blockIdx.x == gridDim.x - 1 && tid == 255) *total = run;        // (values past nv count as zero)
}

__global__ __launch_bounds__(256) void k_scan_blkapply(long long *__restrict__ v, int64_t nv, const long long *__restrict__ bs)
{
    __shared__ long long s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * SCAN_BLK + (int64_t)tid * 8;
    const long long base = bs[blockIdx.x];
    long long x[8];
    const long long mine = scan_blk_load8(v, nv, b0, x);
    long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    long long wpre = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (q < wid) wpre += s_w[q];
    long long run = base + wpre + incl - mine;
    if (b0 + 8 <= nv) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            longlong2 o;
            o.x = run; run += x[2 * k];
            o.y = run; run += x[2 * k + 1];
            *reinterpret_cast<longlong2 *>(v + b0 + 2 * k) = o;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) { if (b0 + k < nv) v[b0 + k] = run; run += x[k]; }
    }
}

__global__ __launch_bounds__(256) void k_col_offsets(const int64_t *__restrict__ table, int64_t n, int ca, int shift, int cb,
                                                     const long long *__restrict__ bbase, const DevRes *__restrict__ res,
                                                     int64_t *__restrict__ coff, int64_t *__restrict__ starts,
                                                     int64_t *__restrict__ qdir, int64_t qdir_cap)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t start, tot;
    const int64_t len = col_len(table, i, n, ca, shift, cb, start);
    const int64_t off = bbase[blockIdx.x] + block_excl_scan_len(len, tot);
    if (i < n) {
        coff[i] = off;
        starts[i] = start;
        qdir_mark(qdir, qdir_cap, off, len, i);
    }
    if (i == 0) coff[n] = res->n_qual_bytes;
}

// The rows of a shard inside the table of its [tail | own | head] scan, one wave, one launch:
// i0 = first row with pos0 >= lo, i1 = first row with pos0 >= hi (64-ary searches), pos0 of
// both rows (-1 past the end) and pos5 of the rows in front of them (-1: there is none) -- the
// chain found row i from offset pos5[i - 1] - 1 (fastqandfurious.py:254).
// out = {i0, i1, pos0[i0], pos0[i1], pos5[i0 - 1], pos5[i1 - 1]}.
__global__ __launch_bounds__(64) void k_table_cut(const int64_t *__restrict__ table, int64_t n, int64_t lo,
                                                  int64_t hi, int64_t *__restrict__ out)
{
    const int lane = threadIdx.x;
    int64_t res[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int64_t value = q ? hi : lo;
        int64_t a = 0, b = n;                   // answer in [a, b]
        while (b - a > 0) {
            const int64_t stride = (b - a + 63) / 64;
            const int64_t i = a + (int64_t)lane * stride;
            const bool below = (i < b) && (table[i * 6] < value);       // monotone in i
            const int k = __popcll(__ballot(below));                    // rows probed that are below
            if (k == 0) { b = a; break; }
            const int64_t last = a + (int64_t)(k - 1) * stride;         // largest probed row below value
            a = last + 1;
            b = min(b, last + stride);
        }
        res[q] = a;
    }
    if (lane == 0) {
        out[0] = res[0];
        out[1] = res[1];
        out[2] = (res[0] < n) ? table[res[0] * 6] : -1;
        out[3] = (res[1] < n) ? table[res[1] * 6] : -1;
        out[4] = (res[0] > 0) ? table[(res[0] - 1) * 6 + 5] : -1;
        out[5] = (res[1] > 0) ? table[(res[1] - 1) * 6 + 5] : -1;
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

#ifdef FFQ_PROBES
// =========================================================================
// k_pipe_probe (ffq_read_probe modes 200 + lag; tools/pipe_probe.py): what a single-pass design would have to
// be built on -- PERSISTENT workgroups that keep streaming tiles (workgroup b takes tiles b, b + G, b + 2G, ...;
// the next tile's loads are issued before this one is counted) while the prefix over all earlier tiles is
// resolved `lag` iterations later from a two-level tree of descriptors: every workgroup publishes its
// tile's count, the last workgroup of each group of 32 sums its group's counts one iteration later, and
// `lag` iterations later every workgroup reads the 32 group sums and the counts of its own group in ONE
// round trip and carries the running base itself.  Nothing is decoded: the question is what the prefix
// costs when no tile waits for it with its loads still to come.  LDS is allocated as the real thing would
// (four workgroups per CU).  lag 0: no prefix at all (the streaming loop alone).
// =========================================================================
constexpr int PP_GROUP = 32;
struct PipeArgs {
    const uint8_t *d;
    int64_t ntiles;
    unsigned long long *descA;      // [niter * G] flag << 62 | newlines of the tile
    unsigned long long *descG;      // [niter * G / 32] flag << 62 | newlines of the group
    long long *prefix;              // [niter * G] out: newlines in front of the tile
    uint32_t *err;                  // set when a poll gave up
    int lag;
};

__device__ __forceinline__ unsigned long long pp_poll(const unsigned long long *p, bool need, uint32_t *err)
{
    unsigned long long v = need ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1ull << 62);
    for (int spins = 0; __ballot((v >> 62) == 0ull) != 0ull; spins++) {
        if (spins > (1 << 18)) { if ((threadIdx.x & 63) == 0) atomicOr(err, 1u); break; }      // never hang the GPU
        __builtin_amdgcn_s_sleep(2);
        if ((v >> 62) == 0ull) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v & ((1ull << 62) - 1ull);
}

// KIB: LDS per workgroup (36: four workgroups per CU, 72: two); the tile is parked in one of KIB / 16 slots
template <int KIB>
__global__ __launch_bounds__(256) void k_pipe_probe(PipeArgs a, uint32_t *__restrict__ sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_pad[KIB * 1024];      // the residency of the real thing
    __shared__ uint32_t s_w[2][4];
    const int G = (int)gridDim.x, b = (int)blockIdx.x, tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int ngroups = G / PP_GROUP, g = b / PP_GROUP, bi = b % PP_GROUP;
    const int64_t niter = (a.ntiles + G - 1) / G;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 cur[4], nxt[4];
    auto issue = [&](u32x4 (&v)[4], int64_t t) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[i] = u32x4{0, 0, 0, 0};
            if (t < a.ntiles)
                v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.d + (t << TILE_SHIFT) + w * 4096 + i * 1024 + l * 16));
        }
    };
    issue(cur, b);
    long long base = 0;
    uint32_t acc = 0;
    if (tid == 0) s_pad[b & 1023] = 1;
    const int lag = a.lag;
    for (int64_t it = 0; it < niter + lag; it++) {
        const int64_t T = it * G + b;
        // the descriptor loads FIRST, the next tile's loads behind them: loads return in order, and a
        // wait for the descriptors must not be a wait for 16 KiB of tile (with the order reversed the
        // loop ran at half speed: one tile in flight per workgroup instead of two)
        bool lead = a.lag && w == 0 && bi == PP_GROUP - 1 && it >= 1 && it - 1 < niter;
        bool res = a.lag && w == 0 && it >= lag && it - lag < niter;
        int64_t j = it - lag;
        const bool isg = l < 32;
        const unsigned long long *pl = a.descA + (it - 1) * G + g * PP_GROUP + (l & 31);
        const unsigned long long *pr = isg ? a.descG + j * ngroups + min(l, ngroups - 1) : a.descA + j * G + g * PP_GROUP + (l - 32);
        const bool needr = isg ? (l < ngroups) : (l - 32 < bi);
        unsigned long long vl = 1ull << 62, vr = 1ull << 62;
        if (lead && l < PP_GROUP) vl = __hip_atomic_load(pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (res && needr) vr = __hip_atomic_load(pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (it + 1 < niter) issue(nxt, T + G);
        if (it < niter) {
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                c += (uint32_t)__popc(nl_mask16(make_uint4(cur[i].x, cur[i].y, cur[i].z, cur[i].w)));
                *reinterpret_cast<u32x4 *>(s_pad + (it % (KIB / 16)) * TILE + w * 4096 + i * 1024 + l * 16) = cur[i];      // parked, as the real thing would
            }
            const uint32_t ws = (uint32_t)__shfl((int)wave_incl_scan(c), 63);
            if (l == 0) s_w[it & 1][w] = ws;
            __syncthreads();
            const uint32_t total = s_w[it & 1][0] + s_w[it & 1][1] + s_w[it & 1][2] + s_w[it & 1][3];
            acc += total;
            if (a.lag && tid == 0)
                __hip_atomic_store(a.descA + T, (1ull << 62) | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int64_t lit = it;
        if (lead) {
            // the last workgroup of a group: that group's sum of the iteration before
            pl = a.descA + (lit - 1) * G + g * PP_GROUP + (l & 31);
            if (__ballot((vl >> 62) == 0ull)) vl = (1ull << 62) | pp_poll(pl, l < PP_GROUP, a.err);
            const uint32_t sum = (uint32_t)__shfl((int)wave_incl_scan(l < PP_GROUP ? (uint32_t)vl : 0u), 63);
            if (l == 0)
                __hip_atomic_store(a.descG + (it - 1) * ngroups + g, (1ull << 62) | (unsigned long long)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (res) {
            // everyone: the prefix of the tile taken `lag` iterations ago
            if (__ballot((vr >> 62) == 0ull)) vr = (1ull << 62) | pp_poll(pr, needr, a.err);
            const uint32_t val = needr ? (uint32_t)vr : 0u;
            const uint32_t all_g = (uint32_t)__shfl((int)wave_incl_scan(isg ? val : 0u), 63);
            const uint32_t before = (uint32_t)__shfl((int)wave_incl_scan((isg && l < g) || !isg ? val : 0u), 63);
            if (l == 0 && j * G + b < a.ntiles) a.prefix[j * G + b] = base + (long long)before;
            base += (long long)all_g;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) cur[i] = nxt[i];
    }
    if (acc == 0x12345678u && s_pad[tid] == 77) sink[0] = acc;
}

// =========================================================================
// k_read_probe: pure streaming read in the launch geometry of k_scan_lines (one 16 KiB
// tile per 256-thread workgroup, four 16-byte loads per lane) or as a grid-stride loop.
// The measured ceiling the scan kernel is compared with (tools/read_probe.py).
// =========================================================================
template <int MODE>
__global__ __launch_bounds__(256) void k_read_probe(const uint8_t *__restrict__ d, int64_t ntiles,
                                                    uint32_t *__restrict__ sink)
{
    const int tid = threadIdx.x;
    uint32_t acc = 0;
    if (MODE == 0) {
        const int64_t base = (int64_t)blockIdx.x << TILE_SHIFT;
        const int w = tid >> 6, l = tid & 63;
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<const uint4 *>(d + base + w * 4096 + i * 1024 + l * 16);
#pragma unroll
        for (int i = 0; i < 4; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    } else if (MODE == 2) {
        // MODE 0 with the scan kernel's non-temporal loads
        const int64_t base = (int64_t)blockIdx.x << TILE_SHIFT;
        const int w = tid >> 6, l = tid & 63;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(d + base + w * 4096 + i * 1024 + l * 16));
#pragma unroll
        for (int i = 0; i < 4; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    } else {
        const int64_t nvec = ntiles << (TILE_SHIFT - 4);
        const int64_t stride = (int64_t)gridDim.x * 256;
        const uint4 *p = reinterpret_cast<const uint4 *>(d);
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < nvec; i += stride * 4) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = (i + k * stride < nvec) ? p[i + k * stride] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;      // never true in practice: keeps the loads alive
}

#endif  // FFQ_PROBES

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

#include "ffq_fused.h"
#include "ffq_ranked.h"
