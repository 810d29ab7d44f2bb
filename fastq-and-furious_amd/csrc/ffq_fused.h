// ffq_fused.h -- ONE pass over the input for plain four-line FASTQ with the Phred decode:
// line index AND decoded qualities from the same read of every byte.
//
// What is computed is unchanged:
//   rows      the record chain of /root/reference/src/fastqandfurious.py:251-279 with the scanner of
//             /root/reference/src/_fastqandfurious.c:25-153 (k_rows4, from the index, as before)
//   qualities array('b').frombytes(buf[pos4:pos5]); arrayadd_b(q, qual_add) of every record
//             (/root/reference/doc/user-guide.rst:126-141; _fastqandfurious.c:161-185)
// In the two-pass path the decode is a kernel of its own that reads the whole input again to pick the
// quality bytes out of it (1.5 x its algorithmic traffic).  Here the kernel that builds the line index
// also writes the decoded bytes.
//
// Why that is possible without knowing the rows first.  On four-line input whose quality line is as
// long as its sequence line (k_rows4 checks both, per record), buf[pos4:pos5] is exactly the CONTENT
// OF EVERY FOURTH LINE.  Which lines those are can be told from the tile's own bytes: with line types
// H S P Q cycling, "starts with '@'" holds for every H (and some Q), "starts with '+'" for every P (and
// some Q), neither for any S -- so two consecutive lines flagged [none][+] are S P and nothing else
// (Q is followed by H, which starts with '@'; H and P themselves are flagged).
//
// WHERE the bytes go is what decides the design.  Round 3 first built the packed stream (a tile needs the
// count of quality bytes in front of it: persistent workgroups, pending bytes in registers, a lagged
// two-level prefix): correct, 1.0 x the algorithmic traffic -- and 7.6 ms per 10 GiB against 5.07 for the
// two passes, bound by instruction issue at the four waves per SIMD such a kernel gets, and void whenever
// another queue kept part of its grid from becoming resident (DESIGN.md section 8;
// profiles/r03_probes/fused_ablate.txt; git history: k_scan_fused).  This version asks for nothing from the
// rest of the buffer: the output is SEGMENTED -- every 16 KiB tile of the input owns SG_STRIDE bytes of
// the output and writes the quality lines that START behind one of its newlines there, packed within the
// segment, in file order.  One workgroup per tile exactly like k_scan_lines, eight per CU, no waiting, no
// co-residency.  A record's bytes are contiguous (the line that runs over the tile's end is finished from
// the first SG_EXT bytes of the next tile, loaded with the tile); its start, qoff[i], is exact (k_rows4:
// the segment of the tile that holds the newline in front of pos4 + the lengths of the segment's earlier
// lines); its length is pos5 - pos4 of its row.  Between segments there are gaps: the stream is NOT
// packed and qoff[i + 1] - qoff[i] is not a length.  Callers ask for this layout (FFQ_F_SINGLE_PASS).
//
// Everything here is speculation that k_rows4 / k_finalize4 verify (phase of every tile against the
// newline ordinals, quality line length == sequence line length per record, chain starts at the
// buffer's first newline): if anything does not hold -- or a line is longer than a segment can take --
// the result is discarded and the two-pass kernels redo the scan (their output is packed, which is a
// special case of the same contract).
#pragma once
// (included at the end of ffq_kernels.h: scan_tile's helpers, lt_mask, addb4 are in scope)

namespace ffq {

constexpr int SG_EXT = 512;                      // bytes of the next tile a workgroup sees (the line that runs over its end)
constexpr int SG_STRIDE = TILE / 2 + SG_EXT;     // output bytes a tile owns (a multiple of 16: segments start aligned)
constexpr uint8_t FZ_NOPHASE = 0xFF;
constexpr int FZ_TAB_AT = 576;                   // byte offset of the mask table inside the (dead) entry list; the piece -> line bytes lie in front
static_assert(SG_STRIDE / 16 <= FZ_TAB_AT && FZ_TAB_AT % 16 == 0 && FZ_TAB_AT + 17 * 16 <= SLOT * 2, "tables must fit the entry list");
static_assert(SLOT / 4 <= 256, "a quality line's number must fit a byte");
// bits of *bad
constexpr uint32_t FZ_BAD_SHAPE = 1u;    // a tile the speculation cannot take (no S P pattern, dense, a line longer than SG_EXT behind
                                         // the tile, more quality bytes than a segment holds, the caller's buffer too small)
constexpr uint32_t FZ_BAD_INDEX = 4u;    // ... and the line index of that tile was not written (dense tile): rebuild it

struct SegArgs {
    const uint8_t *d;
    int64_t n;
    int32_t s;
    int32_t ntiles;
    uint16_t *ent;                 // line index, as k_scan_lines writes it
    uint32_t *cnt;
    uint8_t *qphase;               // [ntiles] out: which lines of the tile were taken for quality lines (FZ_NOPHASE: none)
    uint32_t *bad;                 // out: FZ_BAD_*
    int8_t *out;                   // decoded qualities, segmented: tile t owns [t * SG_STRIDE, (t + 1) * SG_STRIDE)
    int64_t out_cap;
    int32_t qadd;
    uint32_t at_char;
    LineIndex Lval;                // the index descriptor, copied to *d_L for the kernels that take it by pointer
    LineIndex *d_L;
};

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

__global__ __launch_bounds__(256, 8) void k_scan_seg(SegArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_data[TILE + SG_EXT + 32];
    __shared__ __attribute__((aligned(16))) uint16_t s_list[SLOT];
    __shared__ uint16_t s_qs[SLOT / 4 + 4];       // packed start of quality line m inside the segment; [NQ] = their total
    __shared__ uint16_t s_src[SLOT / 4 + 4];      // offset of its first byte in s_data
    __shared__ uint32_t s_wtot[4], s_wq[4], s_ext;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int64_t T = blockIdx.x;
    const int64_t tbase = T << TILE_SHIFT;
    if (T == 0 && tid == 0 && a.d_L) *a.d_L = a.Lval;
    uint32_t o4[4];
    uint4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o4[i] = (uint32_t)(w * 4096 + i * 1024 + l * 16);
    if (tbase + TILE <= a.n) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.d + tbase + o4[i]));
            v[i] = make_uint4(t.x, t.y, t.z, t.w);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = load_tail16(a.d, a.n, tbase + o4[i]);
    }
    // the first SG_EXT bytes of the next tile (zero past the end of the buffer): 32 lanes of wave 0
    uint4 xe = make_uint4(0, 0, 0, 0);
    if (tid < SG_EXT / 16) {
        const int64_t at = tbase + TILE + 16 * tid;
        if (at + 16 <= a.n) xe = *reinterpret_cast<const uint4 *>(a.d + at);
        else xe = load_tail16(a.d, a.n, at);
    }
    const int nvalid = (int)min((int64_t)(TILE + SG_EXT), a.n - tbase);     // bytes of s_data that exist

    uint32_t m[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        *reinterpret_cast<uint4 *>(s_data + o4[i]) = v[i];
        m[i] = nl_mask16(v[i]);
        c[i] = __popc(m[i]);
    }
    if (w == 0) {
        // where the line that runs over the tile's end stops: the first newline of the extension
        uint32_t me = 0;
        if (tid < SG_EXT / 16) { *reinterpret_cast<uint4 *>(s_data + TILE + 16 * tid) = xe; me = nl_mask16(xe); }
        const unsigned long long hit = __ballot(me != 0u);
        uint32_t ext = 0xFFFFu;
        if (hit) {
            const int fl = __ffsll((long long)hit) - 1;
            ext = (uint32_t)(16 * fl) + ((uint32_t)__ffs(__shfl((int)me, fl)) - 1u);
        }
        if (l == 0) s_ext = ext;
    }
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
    const bool dense = total > (uint32_t)SLOT;
    if (!dense) {
        uint32_t rb = wbase;
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
        // this wave's entries: flags looked up (the byte behind a newline at the tile's last offset is the
        // extension's first), stored to the index, and left in the list
        uint16_t *gdst = a.ent + T * SLOT;
        for (uint32_t jj = (uint32_t)l; jj < wtot; jj += 64) {
            const uint32_t off = (uint32_t)s_list[wbase + jj];
            const uint32_t nb = (uint32_t)s_data[off + 1u];
            const uint32_t fl = (nb == a.at_char) ? (uint32_t)FL_AT : (nb == '+') ? (uint32_t)FL_PLUS : 0u;
            const uint16_t e = (uint16_t)(off | (fl << 14));
            s_list[wbase + jj] = e;
            __builtin_nontemporal_store(e, gdst + wbase + jj);
        }
    }
    if (tid == 0) {
        a.cnt[T] = total;
        if (dense) atomicOr(a.bad, FZ_BAD_SHAPE | FZ_BAD_INDEX);
    }
    __syncthreads();                                   // the list, flags included, is complete
    // ---- which lines are quality lines: [none][+] = S P, from the first 64 entries --------------------
    const int tot = dense ? 0 : (int)total;
    uint8_t phase = FZ_NOPHASE;
    {
        const uint32_t f1 = (uint32_t)s_list[min(l, max(tot - 1, 0))] >> 14;
        const uint32_t f2 = (uint32_t)s_list[min(l + 1, max(tot - 1, 0))] >> 14;
        const unsigned long long hit = __ballot(l + 1 < tot && f1 == 0u && f2 == (uint32_t)FL_PLUS);
        if (hit) phase = (uint8_t)((__ffsll((long long)hit) - 1 + 2) & 3);     // the line behind entry i is S: Q lines follow entries = i + 2 (mod 4)
    }
    if (phase == FZ_NOPHASE && tot > 0 && T > 0) {
        // RARE on short reads (a short last tile), EVERY tile on long ones (a tile of a few long lines -- which the pass
        // then refuses anyway: it must get there cheaply): no S P pair among the tile's own lines -- look for one among the
        // lines in front of it, in the 1024 bytes before the tile (round 5: the walk was one byte per memory round trip, up
        // to 4096 of them, and a first scan of 4 GiB of 30 kbp reads spent 58 ms here before it was refused).  Every wave
        // does the same: lane l of round k looks at the byte at distance 64 k + l + 1 and the one behind it; the newlines
        // then go by in the order of the walk, closest first, on the scalar side: newline -1, -2, ... with the class of the
        // byte behind each; [none][+] at entries (e, e + 1) makes the line behind e the sequence line.
        uint32_t fnext = (uint32_t)s_list[0] >> 14;            // flags of entry e + 1, starting with entry 0
        int e = -1;
        for (int k4 = 0; k4 < 4 && phase == FZ_NOPHASE; k4++) {
            unsigned long long NL[4], PL[4], AT[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int64_t pos = tbase - 1 - (64 * (4 * k4 + k) + l);
                uint32_t b0 = 0, b1 = 0;
                if (pos >= 0) { b0 = a.d[pos]; b1 = a.d[pos + 1]; }          // (pos + 1 <= tbase < n)
                NL[k] = __ballot(b0 == '\n');
                PL[k] = __ballot(b0 == '\n' && b1 == '+');
                AT[k] = __ballot(b0 == '\n' && b1 == a.at_char);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned long long N = NL[k];
                while (N != 0ull && phase == FZ_NOPHASE) {
                    const int i = __ffsll((long long)N) - 1;
                    N &= N - 1ull;
                    const uint32_t fl = ((AT[k] >> i) & 1ull) ? (uint32_t)FL_AT : ((PL[k] >> i) & 1ull) ? (uint32_t)FL_PLUS : 0u;
                    if (fl == 0u && fnext == (uint32_t)FL_PLUS) phase = (uint8_t)(((e + 2) % 4 + 4) % 4);
                    fnext = fl;
                    e--;
                }
            }
            if (tbase - 1 - 256 * (k4 + 1) < 0) break;
        }
    }
    // (a tile without any newline starts no line: nothing to decode, nothing to vouch for)
    bool shape_bad = dense || (phase == FZ_NOPHASE && tot > 0);
    int NQ = (phase != FZ_NOPHASE && tot - 1 >= (int)phase) ? (tot - 1 - (int)phase) / 4 + 1 : 0;
    if (NQ > SLOT / 4) { shape_bad = true; NQ = 0; }
    // ---- the tile's quality lines: those that start behind entries phase, phase + 4, ... -----------------
    int st = 0, len = 0;
    if (tid < NQ) {
        const int e = (int)phase + 4 * tid;
        st = (int)(s_list[e] & OFF_MASK) + 1;
        int en;
        if (e + 1 < tot) en = (int)(s_list[e + 1] & OFF_MASK);
        else {
            // the tile's last line: it ends at the extension's first newline; at the buffer's end if that comes
            // first; a line longer than the extension is more than a segment can take
            const uint32_t ext = s_ext;
            if (ext != 0xFFFFu) en = TILE + (int)ext;
            else if (nvalid < TILE + SG_EXT) en = nvalid;
            else { en = st; shape_bad = true; }
        }
        len = max(en - st, 0);
    }
    const uint32_t incl = wave_incl_scan((uint32_t)len);
    if (l == 63) s_wq[w] = incl;
    const bool any_bad = __syncthreads_or(shape_bad ? 1 : 0) != 0;
    uint32_t wpre = 0, qtot = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t t = s_wq[q];
        if (q < w) wpre += t;
        qtot += t;
    }
    const int qcount = (int)qtot;
    const bool bad_here = any_bad || qcount > SG_STRIDE || (T + 1) * (int64_t)SG_STRIDE > a.out_cap;
    // s_list is through (its last readers were the line table's lanes, in front of the barrier above): its memory now holds
    //   s_piece[p]  the quality line that holds byte 16 p of the segment (every lane of the gather looks its line up
    //               instead of searching the table of starts: interpolation + two loops were 30 instructions per piece)
    //   s_tab[h]    the 16-byte mask "bytes below h" (a piece that spans two lines is a select under it)
    uint8_t *s_piece = reinterpret_cast<uint8_t *>(s_list);
    uint32_t *s_tab = reinterpret_cast<uint32_t *>(s_list) + FZ_TAB_AT / 4;
    if (tid < NQ) {
        const int qs = (int)(wpre + incl - (uint32_t)len);
        s_qs[tid] = (uint16_t)qs; s_src[tid] = (uint16_t)st;
        if (!bad_here)
            for (int pz = (qs + 15) >> 4; 16 * pz < qs + len; pz++) s_piece[pz] = (uint8_t)tid;
    }
    if (tid >= 128 && tid < 128 + 17 * 4) s_tab[tid - 128] = lt_mask((tid - 128) >> 2, (tid - 128) & 3);
    if (tid == 0) {
        s_qs[NQ] = (uint16_t)min(qcount, 0xFFFF);
        a.qphase[T] = (any_bad || phase == FZ_NOPHASE) ? FZ_NOPHASE : phase;
        // (every tile of a buffer of long reads says the same: look before storing -- tens of thousands of atomics onto one
        // address are milliseconds)
        if (bad_here && !(__hip_atomic_load(a.bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FZ_BAD_SHAPE)) atomicOr(a.bad, FZ_BAD_SHAPE);
    }
    __syncthreads();
    if (bad_here || qcount <= 0) return;
    // ---- gather: every lane takes 16-byte pieces of the segment; whole aligned stores (the bytes behind the
    //      segment's last piece belong to nobody) ---------------------------------------------------------------
    const uint32_t vv = (uint32_t)(uint8_t)a.qadd * 0x01010101u;
    int8_t *seg = a.out + T * (int64_t)SG_STRIDE;
    for (int lo = 16 * tid; lo < qcount; lo += 16 * 256) {
        const int mm = (int)s_piece[lo >> 4];
        const int kend = min(16, qcount - lo);
        const int h0 = min((int)s_qs[mm + 1] - lo, kend);
        const uint4 A = lds_load16(s_data + (int)s_src[mm] + (lo - (int)s_qs[mm]));
        uint32_t y[4] = {A.x, A.y, A.z, A.w};
        if (h0 < kend) {
            int m2 = mm + 1;
            while ((int)s_qs[m2 + 1] == (int)s_qs[m2]) m2++;            // (empty quality lines)
            const int h1 = min((int)s_qs[m2 + 1] - lo, kend);
            const uint4 B = lds_load16(s_data + (int)s_src[m2] - h0);
            const uint32_t Bw[4] = {B.x, B.y, B.z, B.w};
            const uint4 mk4 = *reinterpret_cast<const uint4 *>(s_tab + 4 * h0);
            const uint32_t mk[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
#pragma unroll
            for (int q = 0; q < 4; q++) y[q] = (y[q] & mk[q]) | (Bw[q] & ~mk[q]);
            if (h1 < kend) {
                const uint4 t4 = fz_gather_tail(s_data, s_qs, s_src, make_uint4(y[0], y[1], y[2], y[3]), m2 + 1, h1, kend, lo);
                y[0] = t4.x; y[1] = t4.y; y[2] = t4.z; y[3] = t4.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) y[q] = addb4(y[q], vv);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 t; t.x = y[0]; t.y = y[1]; t.z = y[2]; t.w = y[3];
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(seg + lo));
    }
}


// ---- the same single pass with the output IN PLACE: decoded byte of input offset p at output offset p -----------------
// Nothing is packed: a tile writes the 16-byte pieces of its own bytes that touch a quality line, decoded, at the offsets
// they have in the input (a superset of the quality bytes: the rest of such a piece is some other line's, decoded too,
// and nobody's).  A record's bytes are out[pos4 : pos5] in the coordinates of d -- contiguous whatever tiles they cross,
// so there is no limit on the length of a line -- and qoff[i] is pos4 itself.  Which lines are quality lines is the same
// speculation as above ([none][+] among the tile's first 64 entries), verified the same way (k_rows4<2>); a tile that
// cannot tell (no such pair: a tile inside one long line, or one with a newline or two) writes ALL of its bytes and has
// nothing to be verified.  Output buffer: as many bytes as the input's tiles (a.out_cap >= ntiles * TILE).

// four bytes, each one of A C G T N in either case
__device__ __forceinline__ bool all_bases4(uint32_t x)
{
    const uint32_t y = x & 0xDFDFDFDFu;
    uint32_t hit = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t letter = (k == 0 ? 'A' : k == 1 ? 'C' : k == 2 ? 'G' : k == 3 ? 'T' : 'N') * 0x01010101u;
        const uint32_t v = y ^ letter;
        hit |= ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);          // 0x80 in every byte that is the letter
    }
    return hit == 0x80808080u;
}

__global__ __launch_bounds__(256, 8) void k_scan_ident(SegArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_data[TILE + 16];
    __shared__ __attribute__((aligned(16))) uint16_t s_list[SLOT];
    __shared__ uint32_t s_wtot[4];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int64_t T = blockIdx.x;
    const int64_t tbase = T << TILE_SHIFT;
    if (T == 0 && tid == 0 && a.d_L) *a.d_L = a.Lval;
    uint32_t o4[4];
    uint4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o4[i] = (uint32_t)(w * 4096 + i * 1024 + l * 16);
    if (tbase + TILE <= a.n) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.d + tbase + o4[i]));
            v[i] = make_uint4(t.x, t.y, t.z, t.w);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = load_tail16(a.d, a.n, tbase + o4[i]);
    }
    const uint32_t nxt = (tbase + TILE < a.n) ? (uint32_t)a.d[tbase + TILE] : 0u;   // the flags of a newline at the tile's last offset
    uint32_t m[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        *reinterpret_cast<uint4 *>(s_data + o4[i]) = v[i];
        m[i] = nl_mask16(v[i]);
        c[i] = __popc(m[i]);
    }
    if (tid == 0) s_data[TILE] = (uint8_t)nxt;
    const uint32_t s01 = wave_incl_scan(c[0] | (c[1] << 16));
    const uint32_t s23 = wave_incl_scan(c[2] | (c[3] << 16));
    const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)s01, 63), t23 = (uint32_t)__builtin_amdgcn_readlane((int)s23, 63);
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
    const bool no_room = (T + 1) * (int64_t)TILE > a.out_cap;
    if (tid == 0) {
        a.cnt[T] = total;
        if (dense) atomicOr(a.bad, FZ_BAD_SHAPE | FZ_BAD_INDEX);
        else if (no_room) atomicOr(a.bad, FZ_BAD_SHAPE);
    }
    if (dense) { if (tid == 0) a.qphase[T] = FZ_NOPHASE; return; }
    uint32_t j[4];                                       // entries of the tile in front of piece i
    {
        uint32_t rb = wbase;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t mm = m[i];
            uint32_t idx = rb + ex[i];
            j[i] = idx;
            while (mm) {
                const uint32_t p = (uint32_t)__ffs((int)mm) - 1u;
                mm &= mm - 1u;
                s_list[idx] = (uint16_t)(o4[i] + p);
                idx++;
            }
            rb += rowtot[i];
        }
        uint16_t *gdst = a.ent + T * SLOT;
        for (uint32_t jj = (uint32_t)l; jj < wtot; jj += 64) {
            const uint32_t off = (uint32_t)s_list[wbase + jj];
            const uint32_t nb = (uint32_t)s_data[off + 1u];
            const uint32_t fl = (nb == a.at_char) ? (uint32_t)FL_AT : (nb == '+') ? (uint32_t)FL_PLUS : 0u;
            const uint16_t e = (uint16_t)(off | (fl << 14));
            s_list[wbase + jj] = e;
            __builtin_nontemporal_store(e, gdst + wbase + jj);
        }
    }
    __syncthreads();                                   // the list, flags included, is complete
    const int tot = (int)total;
    uint32_t phase = FZ_ALL;
    {
        const uint32_t f1 = (uint32_t)s_list[min(l, max(tot - 1, 0))] >> 14;
        const uint32_t f2 = (uint32_t)s_list[min(l + 1, max(tot - 1, 0))] >> 14;
        const unsigned long long hit = __ballot(l + 1 < tot && f1 == 0u && f2 == (uint32_t)FL_PLUS);
        if (hit) phase = (uint32_t)((__ffsll((long long)hit) - 1 + 2) & 3);     // the line behind entry i is S: Q lines follow entries = i + 2 (mod 4)
    }
    if (phase == FZ_ALL) {
        // LONG LINES (a tile inside one line, or with a newline or two): no such pair.  Second-rate evidence, good enough
        // for a guess that is verified like the first kind: 64 bytes of nothing but bases at the tile's beginning (its end)
        // are a stretch of a sequence line -- the line behind entry -1 (behind the tile's last entry) is S.  A tile inside
        // a quality line finds neither and writes everything, which is what it would have written anyway.
        const int at = (l < 4) ? 16 * l : TILE - 64 + 16 * (l & 3);
        const uint4 x = *reinterpret_cast<const uint4 *>(s_data + at);
        const bool b = all_bases4(x.x) && all_bases4(x.y) && all_bases4(x.z) && all_bases4(x.w);
        const unsigned long long ok = __ballot(b);
        if ((ok & 0xFull) == 0xFull) phase = 1u;
        else if ((ok & 0xF0ull) == 0xF0ull) phase = (uint32_t)(tot + 1) & 3u;
    }
    if (tid == 0) a.qphase[T] = (uint8_t)phase;
    if (no_room) return;
    const uint32_t vv = (uint32_t)(uint8_t)a.qadd * 0x01010101u;
    int8_t *seg = a.out + tbase;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // the piece's first byte belongs to the line behind entry j - 1, its last to the one behind entry j - 1 + c
        const uint32_t r = (j[i] - 1u - phase) & 3u;
        bool q = phase == FZ_ALL || r == 0u || r + c[i] >= 4u;
        // whole 64-byte blocks (four lanes): at 150-base reads 2.84 TB/s against 1.93 with single pieces, 2.11 with 32-byte
        // pairs and 2.72 with 128 bytes -- partly written blocks cost more than the extra bytes
        // (profiles/r05_probes/single_pass_in_place.txt)
        q |= __shfl_xor((int)q, 1) != 0;
        q |= __shfl_xor((int)q, 2) != 0;
        if (q) {
            const uint4 x = *reinterpret_cast<const uint4 *>(s_data + o4[i]);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 t; t.x = addb4(x.x, vv); t.y = addb4(x.y, vv); t.z = addb4(x.z, vv); t.w = addb4(x.w, vv);
            __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(seg + o4[i]));
        }
    }
}

}  // namespace ffq
