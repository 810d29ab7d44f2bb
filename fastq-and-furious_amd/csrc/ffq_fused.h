// ffq_fused.h -- ONE pass over the input for plain four-line FASTQ with the Phred decode:
// line index AND decoded quality stream from the same read of every byte.
//
// What is computed is unchanged:
//   rows      the record chain of /root/reference/src/fastqandfurious.py:251-279 with the scanner of
//             /root/reference/src/_fastqandfurious.c:25-153 (k_rows4, from the index, as before)
//   qualities array('b').frombytes(buf[pos4:pos5]); arrayadd_b(q, qual_add) of every record, packed
//             (/root/reference/doc/user-guide.rst:126-141; _fastqandfurious.c:161-185)
// Until round 3 the decode was a second pass (k_decode_stream re-read the whole input to pick the
// quality bytes out of it: 1.5 x its algorithmic traffic).  Here the kernel that builds the line index
// also writes the decoded bytes, so the input is read once.
//
// Why that is possible without knowing the rows first.  On four-line input whose quality line is as
// long as its sequence line (k_rows4 checks both, per record), buf[pos4:pos5] is exactly the CONTENT
// OF EVERY FOURTH LINE.  Which lines those are can be told from the tile's own bytes: with line types
// H S P Q cycling, "starts with '@'" holds for every H (and some Q), "starts with '+'" for every P (and
// some Q), neither for any S -- so two consecutive lines flagged [none][+] are S P and nothing else
// (Q is followed by H, which starts with '@'; H and P themselves are flagged).  The packed stream is in file order, so a quality byte's place in it is the number of quality
// bytes in front of it: a tile needs ONE number from the rest of the buffer -- the quality bytes in
// front of the tile -- and only to know where to write.
//
// How that number arrives without stalling the stream (one dependent load behind a
// tile's own costs +52 %).  PERSISTENT workgroups: workgroup b takes tiles b, b + G, b + 2G, ...; the
// next tile's loads are in flight while this one is worked on; a tile's decoded bytes stay in REGISTERS
// (two 16-byte pieces per lane) for LAG iterations; its prefix is resolved LAG iterations later from a
// two-level tree of descriptors (every workgroup publishes its tile's count; the last workgroup of each
// group of 32 sums its group one iteration later; LAG iterations later every workgroup reads the group
// sums and its own group's counts in one round trip and carries the running base itself: no chain of
// dependent waits from iteration to iteration, no polling in the common case).  Measured skeleton:
// tools/pipe_probe.py (round 2).
//
// Everything here is speculation that k_rows4 / k_finalize4 verify (phase of every tile against the
// newline ordinals, quality line length == sequence line length per record, chain starts at the
// buffer's first newline): if anything does not hold the result is discarded and the two-pass kernels
// redo the scan, bit-exact as before.
#pragma once
// (included at the end of ffq_kernels.h: scan_tile's helpers, lt_mask, addb4, wave_sync are in scope)

namespace ffq {

constexpr int FZ_GROUP = 32;             // workgroups per descriptor group
constexpr int FZ_LAG = 3;                // iterations between a tile's count and its prefix
constexpr int FZ_MAXQ = 8192;            // decoded bytes a tile can hold back: 2 x 16 B per lane
constexpr uint8_t FZ_NOPHASE = 0xFF;
// bits of *bad
constexpr uint32_t FZ_BAD_SHAPE = 1u;    // a tile the speculation cannot take (no H S P pattern, dense, too many quality bytes)
constexpr uint32_t FZ_BAD_POLL = 2u;     // a descriptor never arrived (workgroups not co-resident?)
constexpr uint32_t FZ_BAD_INDEX = 4u;    // ... and the line index of that tile was not written (dense tile): rebuild it

struct FuseArgs {
    const uint8_t *d;
    int64_t n;
    int32_t s;                     // virtual sentinel in front (only shifts nothing here: tile offsets are data offsets)
    int32_t ntiles;
    uint16_t *ent;                 // line index, as k_scan_lines writes it
    uint32_t *cnt;
    unsigned long long *descA;     // [niter * G]       flag << 62 | quality bytes of the tile
    unsigned long long *descG;     // [niter * G / 32]  flag << 62 | quality bytes of the group
    long long *qbase;              // [ntiles] out: quality bytes in front of the tile
    uint8_t *qphase;               // [ntiles] out: which lines of the tile were taken for quality lines
    uint32_t *bad;                 // out: FZ_BAD_*
    int8_t *out;                   // decoded stream
    int64_t out_cap;
    int32_t qadd;
    uint32_t at_char;
    LineIndex Lval;                // the index descriptor, copied to *d_L for the kernels that take it by pointer
    LineIndex *d_L;
    int32_t ablate;                // (instrumented build only: bit 0 no gather, 1 no quality table, 2 no write-out, 3 no prefix, 4 no index stores)
    unsigned long long *prof;      // (instrumented build only: cycles per phase, summed over workgroups and iterations)
};

#ifdef FFQ_PROBES
#define FZ_T(k) do { if (a.prof && tid == 0) { const long long t__ = clock64(); tacc[k] += t__ - tlast; tlast = t__; } } while (0)
#else
#define FZ_T(k) do { } while (0)
#endif

__device__ __forceinline__ unsigned long long fz_poll(const unsigned long long *p, bool need, uint32_t *bad)
{
    unsigned long long v = need ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1ull << 62);
    unsigned long long t0 = 0;
    for (int spins = 0; __ballot((v >> 62) == 0ull) != 0ull; spins++) {
        // never hang the GPU: give up after 50 ms of wall clock, and at once when somebody else already has given
        // up (the scan is void then; the two-pass kernels redo it).  The workgroups wait for one another, so all
        // of them must be resident together: the grid is sized for that on an otherwise idle device, and work
        // of another queue that occupies compute units when this kernel starts can keep some of them out for
        // good (measured: torch fill kernels running beside it -> the descriptors of the late workgroups never
        // arrive).  A persistent-grid kernel cannot rule that out; it can only notice and step aside.
        if ((spins & 63) == 63) {
            const unsigned long long now = wall_clock64();               // constant 100 MHz
            if (t0 == 0) t0 = now;
            if (now - t0 > 5000000ull || (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FZ_BAD_POLL)) {
                if ((threadIdx.x & 63) == 0) atomicOr(bad, FZ_BAD_POLL);
                break;
            }
        }
        __builtin_amdgcn_s_sleep(2);
        if ((v >> 62) == 0ull) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v & ((1ull << 62) - 1ull);
}

// 16 bytes at an arbitrary LDS address (the hardware runs LDS in unaligned-access mode)
__device__ __forceinline__ uint4 lds_load16(const uint8_t *p)
{
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    const u32x4u t = *reinterpret_cast<const u32x4u *>(p);
    return make_uint4(t.x, t.y, t.z, t.w);
}

// chunk bytes [kb, kend) gathered byte by byte from quality lines m, m + 1, ... (lines shorter than 16 bytes)
__device__ __noinline__ uint4 fz_gather_tail(const uint8_t *s_data, const uint16_t *s_qs, const uint16_t *s_src, uint4 v, int m,
                                             int kb, int kend, int lo)
{
    uint32_t y[4] = {v.x, v.y, v.z, v.w};
    while (kb < kend) {
        const int he = min((int)s_qs[m + 1] - lo, kend);
        const int sa = (int)s_src[m] - ((int)s_qs[m] - lo);       // address of chunk byte 0 if it came from line m
        for (; kb < he; kb++) {
            const uint32_t sh = 8u * (kb & 3), val = (uint32_t)s_data[sa + kb] << sh, msk = ~(0xFFu << sh);
#pragma unroll
            for (int w = 0; w < 4; w++)
                if ((kb >> 2) == w) y[w] = (y[w] & msk) | val;
        }
        m++;
    }
    return make_uint4(y[0], y[1], y[2], y[3]);
}

__device__ __noinline__ void fz_store_part(int8_t *__restrict__ o, uint4 v, int nb)
{
    const uint32_t y[4] = {v.x, v.y, v.z, v.w};
    for (int kb = 0; kb < nb; kb++) {
        uint32_t wv = y[0];
#pragma unroll
        for (int w = 1; w < 4; w++)
            if ((kb >> 2) == w) wv = y[w];
        o[kb] = (int8_t)(uint8_t)(wv >> (8 * (kb & 3)));
    }
}

__global__ __launch_bounds__(256, 4) void k_scan_fused(FuseArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_data[TILE + 32];
    __shared__ __attribute__((aligned(16))) uint16_t s_list[SLOT + 8];
    __shared__ uint16_t s_qs[4][SLOT / 4 + 4];      // per wave (each builds the tile's table itself: no barrier for it)
    __shared__ uint16_t s_src[4][SLOT / 4 + 4];
    __shared__ uint32_t s_wtot[2][4];
    __shared__ long long s_qb[2];
    const int G = (int)gridDim.x, b = (int)blockIdx.x, tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int ngroups = G / FZ_GROUP, g = b / FZ_GROUP, bi = b % FZ_GROUP;
    const int64_t ntiles = a.ntiles;
    const int64_t niter = (ntiles + G - 1) / G;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 cur[4], nxt[4];
    uint32_t cur_nb = 0, nxt_nb = 0;              // first byte of the tile behind (flags of a newline at offset TILE - 1)
    uint32_t o4[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o4[i] = (uint32_t)(w * 4096 + i * 1024 + l * 16);
    auto issue = [&](u32x4 (&v)[4], uint32_t &nb, int64_t t) {
        nb = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = u32x4{0, 0, 0, 0};
        if (t >= ntiles) return;
        const int64_t base = t << TILE_SHIFT;
        if (base + TILE <= a.n) {
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.d + base + o4[i]));
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint4 t4 = load_tail16(a.d, a.n, base + o4[i]);
                v[i] = u32x4{t4.x, t4.y, t4.z, t4.w};
            }
        }
        if (base + TILE < a.n) nb = (uint32_t)a.d[base + TILE];
    };
    issue(cur, cur_nb, b);
    if (b == 0 && tid == 0 && a.d_L) *a.d_L = a.Lval;
    long long base = 0;                              // (wave 0) quality bytes in front of iteration it - LAG
    uint4 pend[FZ_LAG][2];                           // decoded pieces of the tiles taken 1 .. LAG iterations ago
    int pend_cnt[FZ_LAG];
#pragma unroll
    for (int k = 0; k < FZ_LAG; k++) { pend[k][0] = pend[k][1] = make_uint4(0, 0, 0, 0); pend_cnt[k] = 0; }
    const uint32_t vv = (uint32_t)(uint8_t)a.qadd * 0x01010101u;
#ifdef FFQ_PROBES
    long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif

    for (int64_t it = 0; it < niter + FZ_LAG; it++) {
        const int64_t T = it * G + b;
        const bool have = it < niter && T < ntiles;             // (workgroup-uniform)
        // ---- descriptor loads FIRST, the next tile's loads behind them (loads return in order) ----
        const bool lead = w == 0 && bi == FZ_GROUP - 1 && it >= 1 && it - 1 < niter;
        const bool res = w == 0 && it >= FZ_LAG && it - FZ_LAG < niter;
        const int64_t j = it - FZ_LAG;
        const bool isg = l < 32;
        const unsigned long long *pl = a.descA + (it - 1) * G + g * FZ_GROUP + (l & 31);
        const unsigned long long *pr = isg ? a.descG + j * ngroups + min(l, ngroups - 1) : a.descA + j * G + g * FZ_GROUP + (l - 32);
        const bool needr = isg ? (l < ngroups) : (l - 32 < bi);
        unsigned long long vl = 1ull << 62, vr = 1ull << 62;
        if (lead && l < FZ_GROUP) vl = __hip_atomic_load(pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (res && needr) vr = __hip_atomic_load(pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (it + 1 < niter) issue(nxt, nxt_nb, T + G);
        FZ_T(0);

        uint4 newp[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        int qcount = 0;
        if (have) {
            // ---- the line index of this tile, as scan_tile (ffq_kernels.h) builds it ----------------
            const int64_t tbase = T << TILE_SHIFT;
            const int nvalid = (int)min((int64_t)TILE, a.n - tbase);
            uint32_t m[4], c[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint4 v4 = make_uint4(cur[i].x, cur[i].y, cur[i].z, cur[i].w);
                *reinterpret_cast<uint4 *>(s_data + o4[i]) = v4;
                m[i] = nl_mask16(v4);
                c[i] = __popc(m[i]);
            }
            const uint32_t s01 = wave_incl_scan(c[0] | (c[1] << 16));
            const uint32_t s23 = wave_incl_scan(c[2] | (c[3] << 16));
            const uint32_t t01 = (uint32_t)__shfl((int)s01, 63), t23 = (uint32_t)__shfl((int)s23, 63);
            uint32_t ex[4], rowtot[4];
            ex[0] = (s01 & 0xFFFFu) - c[0];  rowtot[0] = t01 & 0xFFFFu;
            ex[1] = (s01 >> 16) - c[1];      rowtot[1] = t01 >> 16;
            ex[2] = (s23 & 0xFFFFu) - c[2];  rowtot[2] = t23 & 0xFFFFu;
            ex[3] = (s23 >> 16) - c[3];      rowtot[3] = t23 >> 16;
            const uint32_t wtot = rowtot[0] + rowtot[1] + rowtot[2] + rowtot[3];
            if (l == 0) s_wtot[it & 1][w] = wtot;
            FZ_T(1);
            __syncthreads();
            FZ_T(2);
            uint32_t wbase = 0, total = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t t = s_wtot[it & 1][q];
                if (q < w) wbase += t;
                total += t;
            }
            const bool dense = total > (uint32_t)SLOT;
            uint32_t rb = wbase;
            if (!dense) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    uint32_t mm = m[i];
                    uint32_t idx = rb + ex[i];
                    while (mm) {
                        const uint32_t p = (uint32_t)__ffs((int)mm) - 1u;
                        mm &= mm - 1u;
                        s_list[idx] = (uint16_t)(o4[i] + p);
                        idx++;
                    }
                    rb += rowtot[i];
                }
                // this wave's entries: flags looked up, stored to the index, and left in the list
                uint16_t *gdst = a.ent + T * SLOT;
                for (uint32_t jj = (uint32_t)l; jj < wtot; jj += 64) {
                    const uint32_t off = (uint32_t)s_list[wbase + jj];
                    const uint16_t e = (uint16_t)(off | (entry_flags(s_data, off, cur_nb, a.at_char) << 14));
                    s_list[wbase + jj] = e;
                    if (!(PROBES && (a.ablate & 16))) __builtin_nontemporal_store(e, gdst + wbase + jj);
                }
            }
            if (tid == 0) {
                a.cnt[T] = total;
                if (dense) atomicOr(a.bad, FZ_BAD_SHAPE | FZ_BAD_INDEX);
            }
            FZ_T(3);
            __syncthreads();                                   // the list, flags included, is complete
            FZ_T(4);
            // ---- which lines are quality lines: [@][none][+] = H S P, from the first 64 entries ------
            const int tot = dense ? 0 : (int)total;
            uint8_t phase = FZ_NOPHASE;
            {
                // (two lines are enough: [none][+] can only be S P -- H starts with '@', P with '+', and Q is followed by H)
                const uint32_t f1 = (uint32_t)s_list[min(l, max(tot - 1, 0))] >> 14;
                const uint32_t f2 = (uint32_t)s_list[min(l + 1, max(tot - 1, 0))] >> 14;
                const unsigned long long hit = __ballot(l + 1 < tot && f1 == 0u && f2 == (uint32_t)FL_PLUS);
                if (hit) phase = (uint8_t)((__ffsll((long long)hit) - 1 + 2) & 3);     // line after entry i is S: Q lines follow entries = i + 2 (mod 4)
            }
            int NQ = 0;
            const int eq0 = (phase == 3) ? -1 : (int)phase;     // first entry index >= -1 that a quality line follows
            if (phase != FZ_NOPHASE) NQ = (tot - 1 >= eq0) ? (tot - 1 - eq0) / 4 + 1 : 0;
            if (PROBES && (a.ablate & 2)) NQ = 0;
            bool shape_bad = phase == FZ_NOPHASE || NQ > SLOT / 4;
            if (NQ > SLOT / 4) NQ = SLOT / 4;
            // ---- table of the tile's quality lines (start in the tile, start in the packed bytes): every
            //      wave builds all of it for itself, 64 lines per step -----------------------------------
            uint16_t *qs = s_qs[w], *src = s_src[w];
            int run = 0;
            for (int m0 = 0; m0 < NQ; m0 += 64) {
                const int mq = m0 + l, e = eq0 + 4 * mq;
                int st = 0, en = 0;
                if (mq < NQ) {
                    st = e < 0 ? 0 : (int)(s_list[e] & OFF_MASK) + 1;
                    en = (e + 1 < tot) ? (int)(s_list[e + 1] & OFF_MASK) : nvalid;
                    if (e < 0 && T == 0) en = st;               // bytes in front of the buffer's first newline belong to no record
                }
                const int len = max(en - st, 0);
                const uint32_t incl = wave_incl_scan((uint32_t)len);
                if (mq < NQ) { qs[mq] = (uint16_t)min(run + (int)incl - len, 0xFFFF); src[mq] = (uint16_t)st; }
                run += (int)__shfl((int)incl, 63);
            }
            qcount = run;
            if (l == 0) qs[NQ] = (uint16_t)min(run, 0xFFFF);
            if (qcount > FZ_MAXQ) { shape_bad = true; qcount = 0; }
            if (shape_bad) { qcount = 0; NQ = 0; }
            if (tid == 0) {
                a.qphase[T] = shape_bad ? FZ_NOPHASE : phase;
                if (shape_bad) atomicOr(a.bad, FZ_BAD_SHAPE);
            }
            wave_sync();
            FZ_T(5);
            // ---- gather: lane takes packed pieces tid and tid + 256 -----------------------------------
            if (qcount > 0 && !(PROBES && (a.ablate & 1))) {
                const float inv = (float)NQ / (float)qcount;
#pragma unroll
                for (int jp = 0; jp < 2; jp++) {
                    const int lo = 16 * (tid + 256 * jp);
                    if (lo >= qcount) continue;
                    int mm = min(max((int)((float)lo * inv), 0), NQ - 1);
                    while ((int)qs[mm] > lo) mm--;
                    while ((int)qs[mm + 1] <= lo) mm++;
                    const int kend = min(16, qcount - lo);
                    const int h0 = min((int)qs[mm + 1] - lo, kend);
                    const uint4 A = lds_load16(s_data + (int)src[mm] + (lo - (int)qs[mm]));
                    uint32_t y[4] = {A.x, A.y, A.z, A.w};
                    if (h0 < kend) {
                        int m2 = mm + 1;
                        while ((int)qs[m2 + 1] == (int)qs[m2]) m2++;            // (empty quality lines)
                        const int h1 = min((int)qs[m2 + 1] - lo, kend);
                        const uint4 B = lds_load16(s_data + (int)src[m2] - h0);
                        const uint32_t Bw[4] = {B.x, B.y, B.z, B.w};
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t mk = lt_mask(h0, q);
                            y[q] = (y[q] & mk) | (Bw[q] & ~mk);
                        }
                        if (h1 < kend) {
                            const uint4 t4 = fz_gather_tail(s_data, qs, src, make_uint4(y[0], y[1], y[2], y[3]), m2 + 1, h1, kend, lo);
                            y[0] = t4.x; y[1] = t4.y; y[2] = t4.z; y[3] = t4.w;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) y[q] = addb4(y[q], vv);
                    newp[jp] = make_uint4(y[0], y[1], y[2], y[3]);
                }
            }
        }
        FZ_T(6);
        if (it < niter && tid == 0)
            __hip_atomic_store(a.descA + T, (1ull << 62) | (unsigned long long)qcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (PROBES && (a.ablate & 8)) {
            if (res && l == 0) s_qb[it & 1] = (long long)(j * G + b) * 7000;
        } else {
        if (lead) {
            // the last workgroup of a group: that group's sum of the iteration before
            if (__ballot((vl >> 62) == 0ull)) vl = (1ull << 62) | fz_poll(pl, l < FZ_GROUP, a.bad);
            const uint32_t sum = (uint32_t)__shfl((int)wave_incl_scan(l < FZ_GROUP ? (uint32_t)vl : 0u), 63);
            if (l == 0)
                __hip_atomic_store(a.descG + (it - 1) * ngroups + g, (1ull << 62) | (unsigned long long)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (res) {
            // everyone: the prefix of the tile taken LAG iterations ago
            if (__ballot((vr >> 62) == 0ull)) vr = (1ull << 62) | fz_poll(pr, needr, a.bad);
            const uint32_t val = needr ? (uint32_t)vr : 0u;
            const uint32_t all_g = (uint32_t)__shfl((int)wave_incl_scan(isg ? val : 0u), 63);
            const uint32_t before = (uint32_t)__shfl((int)wave_incl_scan((isg && l < g) || !isg ? val : 0u), 63);
            if (l == 0) {
                s_qb[it & 1] = base + (long long)before;
                if (j * G + b < ntiles) a.qbase[j * G + b] = base + (long long)before;
            }
            base += (long long)all_g;
        }
        }
        FZ_T(7);
        __syncthreads();                       // the prefix is there; the tile's LDS may be written again
        FZ_T(8);
        // ---- the decoded bytes of the tile taken LAG iterations ago go where they belong ------------
        if (it >= FZ_LAG && pend_cnt[FZ_LAG - 1] > 0 && !(PROBES && (a.ablate & 4))) {
            const long long qb = s_qb[it & 1];
            const int cntl = pend_cnt[FZ_LAG - 1];
#pragma unroll
            for (int jp = 0; jp < 2; jp++) {
                const int lo = 16 * (tid + 256 * jp);
                if (lo >= cntl) continue;
                const int nb = (int)min((long long)min(16, cntl - lo), a.out_cap - (qb + lo));      // (a stream that is too small keeps what fits)
                if (nb <= 0) continue;
                int8_t *dst = a.out + qb + lo;
                const uint4 p = pend[FZ_LAG - 1][jp];
                if (nb == 16) {
                    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
                    u32x4u t; t.x = p.x; t.y = p.y; t.z = p.z; t.w = p.w;
                    __builtin_nontemporal_store(t, reinterpret_cast<u32x4u *>(dst));
                } else fz_store_part(dst, p, nb);
            }
        }
#pragma unroll
        for (int k = FZ_LAG - 1; k > 0; k--) { pend[k][0] = pend[k - 1][0]; pend[k][1] = pend[k - 1][1]; pend_cnt[k] = pend_cnt[k - 1]; }
        pend[0][0] = newp[0]; pend[0][1] = newp[1]; pend_cnt[0] = qcount;
        FZ_T(9);
#pragma unroll
        for (int i = 0; i < 4; i++) cur[i] = nxt[i];
        cur_nb = nxt_nb;
#ifdef FFQ_PROBES
        if (a.prof) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        FZ_T(10);
    }
#ifdef FFQ_PROBES
    if (a.prof && tid == 0) {
        for (int k = 0; k < 11; k++) atomicAdd(a.prof + k, (unsigned long long)tacc[k]);
        atomicAdd(a.prof + 11, (unsigned long long)(niter + FZ_LAG));
    }
#endif
}

}  // namespace ffq
