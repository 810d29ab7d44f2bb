// ffq_pgz.h -- ONE gzip member inflated by several threads (host code; no device involved).
//
// What the reference hands readfastq_iter for a compressed file is `gzip.open(...)`
// (/root/reference/src/fastqandfurious.py:282-334; its benchmark input is a gzip FASTQ,
// /root/reference/doc/performance.rst:21-22), and a plain `gzip` file is ONE deflate stream: zlib
// inflates it at ~0.33 GB/s on one core, which is what the iterator over such a file was bound by
// (BGZF members carry their own length and were already inflated side by side: ffq_stream.h).
//
// A deflate stream can be entered at any BLOCK boundary if one accepts not knowing the 32 KiB in
// front of it: the compressed bytes of a batch are cut into chunks, chunk 0 starts where the last
// batch ended (window known, plain bytes out), every other chunk looks for the first bit position
// that parses as the header of a dynamic-Huffman block and inflates from there into 16-bit symbols --
// a literal, or 0x8000 + i for "byte i of the 32 KiB window I was not given".  A chunk runs up to the
// first block boundary at or behind its end that the same header test accepts, so that its end is
// the next chunk's start by construction; the stitch step checks exactly that (end bit == start
// bit), resolves the trailing windows chunk after chunk (32 KiB each: cheap), and the symbols of all
// chunks are then turned into bytes side by side (a 64 K-entry lookup per chunk) together with their
// CRC-32, which crc32_combine puts together for the member's trailer check.
// (The two-stage scheme is the published one of pugz / rapidgzip; the code here is this build's own.)
//
// zlib keeps the last word: the engine commits output batch by batch, and whatever it does not
// understand -- a chunk-0 decode error, no progress, a block too long for its buffers, the end of the
// file inside a block -- makes it give up AT THE LAST COMMITTED BLOCK BOUNDARY (bit position, window,
// running CRC and length), from where the serial zlib inflate of ffq_stream.h goes on (inflatePrime +
// inflateSetDictionary on a raw stream) and says what is wrong with the file in its own words.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <emmintrin.h>
#include <unistd.h>

namespace ffq {
namespace pgz {

constexpr int WSIZE = 32768;
constexpr uint32_t F_LIT = 0x10, F_LEN = 0x20, F_EOB = 0x40, F_SUB = 0x80;    // table entry: payload << 16 | extra bits << 8 | kind | code length
constexpr int LBITS = 10, DBITS = 8, PBITS = 7;                                // bits looked up at once: literal/length, distance, code-length code
constexpr int LT_CAP = (1 << LBITS) + 288 * 32, DT_CAP = (1 << DBITS) + 32 * 128;
constexpr int OUT_SLACK = 258 + 16;                                            // a match may be copied a word at a time past its end

// RFC 1951 section 3.2.5 / 3.2.7
static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }     // (x86-64: little endian)

static inline uint32_t bitrev(uint32_t c, int len)
{
    c = ((c & 0x5555u) << 1) | ((c >> 1) & 0x5555u);
    c = ((c & 0x3333u) << 2) | ((c >> 2) & 0x3333u);
    c = ((c & 0x0F0Fu) << 4) | ((c >> 4) & 0x0F0Fu);
    c = ((c & 0x00FFu) << 8) | ((c >> 8) & 0x00FFu);
    return c >> (16 - len);
}

enum TableKind { T_LITLEN, T_DIST, T_CODELEN };

static inline uint32_t sym_entry(TableKind kind, int sym)
{
    if (kind == T_CODELEN) return ((uint32_t)sym << 16) | F_LIT;
    if (kind == T_DIST) return sym < 30 ? ((uint32_t)DIST_BASE[sym] << 16) | ((uint32_t)DIST_EXTRA[sym] << 8) | F_LEN : 0u;
    if (sym < 256) return ((uint32_t)sym << 16) | F_LIT;
    if (sym == 256) return F_EOB;
    if (sym < 286) return ((uint32_t)LEN_BASE[sym - 257] << 16) | ((uint32_t)LEN_EXTRA[sym - 257] << 8) | F_LEN;
    return 0u;                                                 // 286, 287 / 30, 31: part of the code, an error when met
}

// Kraft sum of a set of code lengths, by zlib's rules (inftrees.c): over-subscribed -> -1; incomplete -> -1 unless
// it is a single code of length 1 (never for the code-length code); nothing at all -> 0 (distances only).
static inline int check_lengths(const int *count, TableKind kind)
{
    int maxlen = 15;
    while (maxlen > 0 && !count[maxlen]) maxlen--;
    if (maxlen == 0) return kind == T_DIST ? 0 : -1;
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= count[len];
        if (left < 0) return -1;
    }
    if (left > 0 && (kind == T_CODELEN || maxlen != 1)) return -1;
    return maxlen;
}

// Decoding table of a canonical Huffman code: 2^pbits direct entries, longer codes through sub-tables of
// 2^(maxlen - pbits) entries behind them.  -1: not a set of lengths zlib would take.
static int build_table(const uint8_t *lens, int n, int pbits, uint32_t *tab, int cap, TableKind kind)
{
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    const int maxlen = check_lengths(count, kind);
    if (maxlen < 0) return -1;
    const int psize = 1 << pbits;
    memset(tab, 0, (size_t)psize * sizeof(uint32_t));
    if (maxlen == 0) return 0;
    uint32_t next_code[16];
    uint32_t code = 0;
    for (int len = 1; len <= 15; len++) {
        code = (code + (uint32_t)count[len - 1]) << 1;
        next_code[len] = code;
    }
    const int sb = maxlen - pbits;
    int next_sub = psize;
    for (int sym = 0; sym < n; sym++) {
        const int len = lens[sym];
        if (!len) continue;
        const uint32_t rev = bitrev(next_code[len]++, len);
        const uint32_t e = sym_entry(kind, sym);
        if (len <= pbits) {
            for (uint32_t j = rev; j < (uint32_t)psize; j += 1u << len) tab[j] = e | (uint32_t)len;
        } else {
            const uint32_t prefix = rev & (uint32_t)(psize - 1);
            uint32_t link = tab[prefix];
            if (!(link & F_SUB)) {
                if (next_sub + (1 << sb) > cap) return -1;
                memset(tab + next_sub, 0, sizeof(uint32_t) << sb);
                link = ((uint32_t)next_sub << 16) | ((uint32_t)sb << 8) | F_SUB | (uint32_t)pbits;
                tab[prefix] = link;
                next_sub += 1 << sb;
            }
            uint32_t *sub = tab + (link >> 16);
            for (uint32_t j = rev >> pbits; j < (1u << sb); j += 1u << (len - pbits)) sub[j] = e | (uint32_t)(len - pbits);
        }
    }
    return 0;
}

struct FixedTables {
    uint32_t lt[LT_CAP], dt[DT_CAP];
    FixedTables()
    {
        uint8_t l[288], d[32];
        for (int i = 0; i < 288; i++) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        for (int i = 0; i < 32; i++) d[i] = 5;
        (void)build_table(l, 288, LBITS, lt, LT_CAP, T_LITLEN);
        (void)build_table(d, 32, DBITS, dt, DT_CAP, T_DIST);
    }
};
static const FixedTables &fixed_tables() { static const FixedTables t; return t; }

// Kraft sums (in 128ths) of four 3-bit code lengths at a time
static const uint16_t *kraft4()
{
    struct T { uint16_t k[4096]; T() { for (int x = 0; x < 4096; x++) { int s = 0; for (int q = 0; q < 4; q++) { const int l = (x >> (3 * q)) & 7; if (l) s += 128 >> l; } k[x] = (uint16_t)s; } } };
    static const T t;
    return t.k;
}

// Does the header of a non-final dynamic-Huffman block (RFC 1951 section 3.2.7) start at this bit?  All of it is
// checked: counts, a complete code-length code, the run-length coded lengths, an end-of-block code, complete
// literal/length and distance codes.  (`in` is readable 16 bytes past in_len.)
static bool is_candidate(const uint8_t *in, int64_t in_len, int64_t bit)
{
    const int64_t nbits = in_len * 8;
    if (bit + 17 + 12 > nbits) return false;
    uint64_t v = ld64(in + (bit >> 3)) >> (bit & 7);
    if ((v & 7u) != 4u) return false;                           // BFINAL = 0, BTYPE = 10
    const int hlit = (int)((v >> 3) & 31u), hdist = (int)((v >> 8) & 31u), hclen = (int)((v >> 13) & 15u) + 4;
    if (hlit > 29 || hdist > 29) return false;
    int64_t p = bit + 17;
    if (p + 3 * hclen > nbits) return false;
    v = ld64(in + (p >> 3)) >> (p & 7);                          // >= 57 bits = 19 x 3
    // the code-length code must be complete: Kraft sum of the 3-bit lengths, four of them per table look-up (nearly
    // every position that got this far is turned away here)
    const uint64_t fields = v & ((1ull << (3 * hclen)) - 1ull);
    const uint16_t *K = kraft4();
    if (K[fields & 4095u] + K[(fields >> 12) & 4095u] + K[(fields >> 24) & 4095u] + K[(fields >> 36) & 4095u] + K[(fields >> 48) & 4095u] != 128) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) cl[CL_ORDER[i]] = (uint8_t)((v >> (3 * i)) & 7u);
    p += 3 * hclen;
    uint32_t pt[1 << PBITS];
    if (build_table(cl, 19, PBITS, pt, 1 << PBITS, T_CODELEN) != 0) return false;
    const int nl = hlit + 257, total = nl + hdist + 1;
    uint8_t lens[320];
    int i = 0;
    while (i < total) {
        if (p + 14 > nbits + 64) return false;
        v = ld64(in + (p >> 3)) >> (p & 7);
        const uint32_t e = pt[v & ((1u << PBITS) - 1u)];
        if (!(e & F_LIT)) return false;
        const int sym = (int)(e >> 16), cb = (int)(e & 15u);
        v >>= cb;
        p += cb;
        if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
        int rep, val = 0;
        if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)(v & 3u); p += 2; }
        else if (sym == 17) { rep = 3 + (int)(v & 7u); p += 3; }
        else { rep = 11 + (int)(v & 127u); p += 7; }
        if (i + rep > total) return false;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (p > nbits) return false;
    if (!lens[256]) return false;
    int cnt[16] = {0};
    for (i = 0; i < nl; i++) cnt[lens[i]]++;
    cnt[0] = 0;
    if (check_lengths(cnt, T_LITLEN) < 0) return false;
    memset(cnt, 0, sizeof cnt);
    for (i = nl; i < total; i++) cnt[lens[i]]++;
    cnt[0] = 0;
    return check_lengths(cnt, T_DIST) >= 0;
}

// First candidate at or behind `from`, in front of `to`; -1: none.  (`to` may lie behind the input -- the last chunk of a
// batch the end of the file cut short: the scan itself stops where the input does, it reads a word per byte position.)
static int64_t find_candidate(const uint8_t *in, int64_t in_len, int64_t from, int64_t to)
{
    to = std::min(to, in_len * 8);
    for (int64_t byte = from >> 3; byte * 8 < to; byte++) {
        // eight bit positions from one load: BFINAL/BTYPE = 0b100 at bit r means bits r..r+2 of the word
        const uint64_t w = ld64(in + byte);
        const uint32_t lo = (uint32_t)(w & 0x3FFu);
        // positions r with bit r = 0, bit r+1 = 0, bit r+2 = 1
        uint32_t m = ~lo & ~(lo >> 1) & (lo >> 2) & 0xFFu;
        while (m) {
            const int r = __builtin_ctz(m);
            m &= m - 1;
            const int64_t bit = byte * 8 + r;
            if (bit < from || bit >= to) continue;
            if (is_candidate(in, in_len, bit)) return bit;
        }
    }
    return -1;
}

enum { R_BOUNDARY = 0, R_FINAL = 1, R_FULL = 2, R_INPUT_END = 3, R_ERROR = 4 };

// A resumable inflate over one buffer of compressed bytes.  T = uint8_t: the window in front of the output is known
// (win_valid bytes right in front of out_base); T = uint16_t: it is not, and the WSIZE elements in front of out_base
// hold 0x8000 + i, so that a copy from there carries the reference along.
template <typename T>
struct Inflater {
    const uint8_t *in = nullptr;
    int64_t in_len = 0;                 // in[in_len, in_len + 16) is readable (zeros)
    const uint8_t *ip = nullptr;
    uint64_t bb = 0;
    int bc = 0;
    T *out_base = nullptr, *op = nullptr, *out_end = nullptr;
    int64_t win_valid = 0;
    int64_t limit_bit = INT64_MAX;      // stop at the first accepted block boundary at or behind this bit
    int state = 0;                      // 0: at a block boundary, 1: in a stored block, 2: in a Huffman block
    bool final_block = false;
    uint32_t stored_left = 0;
    int64_t b_bit = 0, b_out = 0;       // the last block boundary: bit position, elements written up to it
    const uint32_t *lt = nullptr, *dt = nullptr;
    uint32_t ltab[LT_CAP], dtab[DT_CAP];

    inline void refill() { bb |= ld64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56; }
    inline void consume(int n) { bb >>= n; bc -= n; }
    inline int64_t bitpos() const { return (int64_t)(ip - in) * 8 - bc; }

    void start(const uint8_t *in_, int64_t in_len_, int64_t bit)
    {
        in = in_; in_len = in_len_;
        ip = in + (bit >> 3); bb = 0; bc = 0;
        refill();
        consume((int)(bit & 7));
        state = 0; final_block = false; stored_left = 0;
        b_bit = bit; b_out = op - out_base;
    }
    void set_out(T *base, int64_t pos, int64_t cap) { out_base = base; op = base + pos; out_end = base + cap - OUT_SLACK; }

    int input_end() { op = out_base + b_out; return R_INPUT_END; }
    int error() { return bitpos() > in_len * 8 ? input_end() : R_ERROR; }

    int read_dynamic()
    {
        refill();
        const int nl = (int)(bb & 31u) + 257, nd = (int)((bb >> 5) & 31u) + 1, ncl = (int)((bb >> 10) & 15u) + 4;
        consume(14);
        if (nl > 286 || nd > 30) return error();
        uint8_t cl[19] = {0};
        const uint8_t *const in_lim = in + in_len;
        for (int i = 0; i < ncl; i++) {
            if (bc < 3) {
                refill();
                if (__builtin_expect(ip >= in_lim, 0) && bitpos() > in_len * 8) return input_end();
            }
            cl[CL_ORDER[i]] = (uint8_t)(bb & 7u);
            consume(3);
        }
        uint32_t pt[1 << PBITS];
        if (build_table(cl, 19, PBITS, pt, 1 << PBITS, T_CODELEN) != 0) return error();
        uint8_t lens[320];
        const int total = nl + nd;
        int i = 0;
        while (i < total) {
            refill();
            // (every symbol: a header cut short by the end of the input decodes its zero padding as short lengths, and
            // only 16 bytes behind in_len are readable)
            if (__builtin_expect(ip >= in_lim, 0) && bitpos() > in_len * 8) return input_end();
            const uint32_t e = pt[bb & ((1u << PBITS) - 1u)];
            if (!(e & F_LIT)) return error();
            consume((int)(e & 15u));
            const int sym = (int)(e >> 16);
            if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
            int rep, val = 0;
            if (sym == 16) { if (i == 0) return error(); val = lens[i - 1]; rep = 3 + (int)(bb & 3u); consume(2); }
            else if (sym == 17) { rep = 3 + (int)(bb & 7u); consume(3); }
            else { rep = 11 + (int)(bb & 127u); consume(7); }
            if (i + rep > total) return error();
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (bitpos() > in_len * 8) return input_end();
        if (!lens[256]) return error();
        if (build_table(lens, nl, LBITS, ltab, LT_CAP, T_LITLEN) != 0) return error();
        if (build_table(lens + nl, nd, DBITS, dtab, DT_CAP, T_DIST) != 0) return error();
        lt = ltab; dt = dtab;
        return -1;
    }

    // symbols of the current block; -1: its end-of-block code was read
    int huff()
    {
        const uint8_t *const in_lim = in + in_len;
        const uint32_t *const L = lt, *const D = dt;
        for (;;) {
            if (op >= out_end) return R_FULL;
            refill();
            if (__builtin_expect(ip >= in_lim, 0) && bitpos() > in_len * 8) return input_end();
            uint32_t e = L[bb & ((1u << LBITS) - 1u)];
            if (__builtin_expect(e & F_SUB, 0)) {
                consume(LBITS);
                e = L[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))];
            }
            consume((int)(e & 15u));
            if (e & F_LIT) {
                *op++ = (T)(e >> 16);
                // a second literal from the same 56 bits (at most 30 are gone by then)
                e = L[bb & ((1u << LBITS) - 1u)];
                if (!(e & F_LIT)) continue;
                consume((int)(e & 15u));
                *op++ = (T)(e >> 16);
                continue;
            }
            if (!(e & F_LEN)) {
                if (e & F_EOB) return -1;
                return error();
            }
            const int xl = (int)((e >> 8) & 15u);
            int len = (int)(e >> 16) + (int)(bb & ((1u << xl) - 1u));
            consume(xl);
            uint32_t d = D[bb & ((1u << DBITS) - 1u)];
            if (__builtin_expect(d & F_SUB, 0)) {
                consume(DBITS);
                d = D[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))];
            }
            consume((int)(d & 15u));
            if (!(d & F_LEN)) return error();
            const int xd = (int)((d >> 8) & 15u);
            const int64_t dist = (int64_t)(d >> 16) + (int64_t)(bb & ((1u << xd) - 1u));
            consume(xd);
            if (dist > (op - out_base) + win_valid) return error();
            const T *src = op - dist;
            T *dst = op;
            op += len;
            constexpr int W = 8 / (int)sizeof(T), W2 = 16 / (int)sizeof(T);
            if (dist >= W2) {
                do {
                    _mm_storeu_si128(reinterpret_cast<__m128i *>(dst), _mm_loadu_si128(reinterpret_cast<const __m128i *>(src)));
                    dst += W2; src += W2; len -= W2;
                } while (len > 0);
            } else if (dist >= W) {
                do { memcpy(dst, src, 8); dst += W; src += W; len -= W; } while (len > 0);
            } else if (dist == 1) {
                const T c = *src;
                do { *dst++ = c; } while (--len > 0);
            } else {
                do { *dst++ = *src++; } while (--len > 0);
            }
        }
    }

    int run()
    {
        for (;;) {
            if (state == 0) {
                const int64_t bp = bitpos();
                if (bp > in_len * 8) return input_end();
                b_bit = bp; b_out = op - out_base;
                if (final_block) return R_FINAL;
                if (bp >= limit_bit && is_candidate(in, in_len, bp)) return R_BOUNDARY;
                if (bp + 3 > in_len * 8) return input_end();
                refill();
                final_block = bb & 1u;
                const int type = (int)((bb >> 1) & 3u);
                consume(3);
                if (type == 0) {
                    consume(bc & 7);
                    if (bc < 32) refill();
                    const uint32_t len = (uint32_t)(bb & 0xFFFFu), nlen = (uint32_t)((bb >> 16) & 0xFFFFu);
                    consume(32);
                    if (len != (~nlen & 0xFFFFu)) return error();
                    ip -= bc >> 3; bb = 0; bc = 0;              // the bytes behind LEN / NLEN are read from the buffer
                    if (ip > in + in_len) return input_end();
                    stored_left = len;
                    state = 1;
                } else if (type == 1) {
                    lt = fixed_tables().lt; dt = fixed_tables().dt;
                    state = 2;
                } else if (type == 2) {
                    const int r = read_dynamic();
                    if (r >= 0) return r;
                    state = 2;
                } else return error();
            }
            if (state == 1) {
                while (stored_left) {
                    const int64_t avail = (in + in_len) - ip;
                    if (avail <= 0) return input_end();
                    const int64_t room = out_end - op;
                    if (room <= 0) return R_FULL;
                    const int64_t n = std::min<int64_t>(std::min<int64_t>(avail, room), stored_left);
                    for (int64_t i = 0; i < n; i++) op[i] = (T)ip[i];
                    op += n; ip += n; stored_left -= (uint32_t)n;
                }
                state = 0;
                continue;
            }
            const int r = huff();
            if (r >= 0) return r;
            state = 0;
        }
    }
};

// ---- a few threads that run the chunks of a batch ---------------------------------------------------------------------
struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    const std::function<void(int)> *fn = nullptr;
    int njobs = 0, active = 0;
    std::atomic<int> next{0};
    uint64_t gen = 0;
    bool quit = false;

    void work() { for (;;) { const int j = next.fetch_add(1); if (j >= njobs) break; (*fn)(j); } }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv_go.wait(lk, [&] { return quit || gen != seen; });
                if (quit) break;
                seen = gen;
            }
            work();
            std::lock_guard<std::mutex> lk(m);
            if (--active == 0) cv_done.notify_one();
        }
    }
    bool start(int n)
    {
        try { for (int i = 0; i < n; i++) th.emplace_back(&Pool::worker, this); } catch (...) {}
        return (int)th.size() == n;
    }
    void run(int n, const std::function<void(int)> &f)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            fn = &f; njobs = n; next.store(0); active = (int)th.size(); gen++;
        }
        cv_go.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return active == 0; });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
};

struct Stats { std::atomic<int64_t> batches{0}, chunks{0}, rejected{0}, giveups{0}, members{0}, ns_find{0}, ns_exact{0}, ns_markers{0}, ns_stage1{0}, ns_stitch{0}, ns_stage2{0}, ns_read{0}; };
static inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static Stats &stats() { static Stats s; return s; }

struct Chunk {
    int64_t lo_bit = 0, limit_bit = 0;           // (bits relative to the batch's buffer)
    int64_t start_bit = -1, end_bit = -1, nout = 0;
    int status = R_ERROR;
    bool ok = false;
    uint16_t *b16 = nullptr; int64_t cap16 = 0;  // [WSIZE references][symbols]
    uint8_t *b8 = nullptr; int64_t cap8 = 0;     // chunk 0: [WSIZE window][bytes]
    uint8_t win[WSIZE];                          // the window BEHIND this chunk (set by the stitch step)
    int64_t out_at = 0;                          // where its bytes go in the batch's output
    uint32_t crc = 0;
    Inflater<uint16_t> *i16 = nullptr;
    Inflater<uint8_t> *i8 = nullptr;
    ~Chunk() { free(b16); free(b8); delete i16; delete i8; }
};

static inline int64_t env_i64(const char *name, int64_t dflt)
{
    const char *e = getenv(name);
    return e && atoll(e) > 0 ? atoll(e) : dflt;
}

// One member, from its header to its trailer.
struct Engine {
    int fd = -1;
    int threads = 1, nslots = 1;     // chunks of a batch: threads x FFQ_PGZ_CPT (two: a thread that drew a slow chunk is waited for less)
    int64_t chunk_bytes = 1 << 20;      // compressed bytes per chunk (FFQ_PGZ_CHUNK)
    int64_t max_out = 1ll << 24;        // elements a chunk may grow to before it stops at its last boundary
    int64_t giveup_after = 0;           // FFQ_PGZ_GIVEUP_AFTER = k: hand over to zlib after k batches (tests of the hand-over)
    // Host memory of the symbol buffers (two bytes per output byte of a chunk, up to max_out each): FFQ_PGZ_MEM bounds
    // their sum (default 1 GiB per open stream) -- fewer chunks per batch when the initial buffers alone would not fit, a
    // chunk that would grow past it stops at its last block boundary instead (as at max_out), and what grew is given
    // back when its member is through.
    int64_t mem_budget = 1ll << 30;
    std::atomic<int64_t> mem16{0};
    int64_t nbatches = 0;
    int64_t file_size = 0;
    Pool *pool = nullptr;
    std::vector<Chunk *> ck;
    std::vector<uint8_t> in;
    // ---- the commit point ----
    int64_t pos_bit = 0;                // absolute (file) bit position of the next block
    uint8_t window[WSIZE];              // the bytes in front of it (right-aligned)
    int64_t win_valid = 0;
    uint32_t crc = 0;
    uint64_t isize = 0;
    // ---- the batch whose bytes are being handed out ----
    int b_na = 0;                       // chunks taken
    int64_t b_total = 0, emit_pos = 0;  // bytes they hold, bytes handed out
    bool b_fin = false;                 // the last one ends with the member's final block
    bool member_done = false, gave_up = false, failed = false;
    int64_t end_off = 0;                // file offset behind the member's trailer
    std::string msg;

    ~Engine() { for (Chunk *c : ck) delete c; delete pool; }

    bool init(int fd_, int threads_, int64_t file_size_)
    {
        fd = fd_; threads = threads_; file_size = file_size_;
        chunk_bytes = env_i64("FFQ_PGZ_CHUNK", chunk_bytes);
        max_out = env_i64("FFQ_PGZ_MAX_OUT", max_out);
        giveup_after = env_i64("FFQ_PGZ_GIVEUP_AFTER", 0);
        pool = new (std::nothrow) Pool();
        if (!pool || !pool->start(threads - 1)) return false;
        nslots = threads * (int)std::min<int64_t>(env_i64("FFQ_PGZ_CPT", threads > 1 ? 2 : 1), 8);
        mem_budget = env_i64("FFQ_PGZ_MEM", mem_budget);
        const int64_t per_chunk = (int64_t)(WSIZE + cap0()) * (int64_t)sizeof(uint16_t);
        nslots = (int)std::max<int64_t>(std::min<int64_t>(nslots, (mem_budget * 3 / 4) / per_chunk), std::min(threads, 2));
        for (int i = 0; i < nslots; i++) {
            Chunk *c = new (std::nothrow) Chunk();
            if (!c) return false;
            ck.push_back(c);
        }
        return true;
    }

    int64_t pread_full(uint8_t *dst, int64_t n, int64_t off)
    {
        int64_t got = 0;
        while (got < n) {
            const ssize_t r = pread(fd, dst + got, (size_t)(n - got), (off_t)(off + got));
            if (r < 0) { if (errno == EINTR) continue; return -1; }
            if (r == 0) break;
            got += r;
        }
        return got;
    }

    // The gzip header (RFC 1952) at member_off; false: not one this engine takes (zlib looks at it then).
    bool begin(int64_t member_off)
    {
        member_done = gave_up = failed = false;
        b_na = 0; b_total = emit_pos = 0; b_fin = false;
        nbatches = 0;
        uint8_t h[4096];
        const int64_t n = pread_full(h, sizeof h, member_off);
        if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xE0)) return false;
        const int flg = h[3];
        int64_t p = 10;
        if (flg & 4) { if (p + 2 > n) return false; p += 2 + (h[p] | (h[p + 1] << 8)); }
        if (flg & 8) { while (p < n && h[p]) p++; p++; }
        if (flg & 16) { while (p < n && h[p]) p++; p++; }
        if (flg & 2) p += 2;
        if (p + 8 > n) return false;
        pos_bit = (member_off + p) * 8;
        win_valid = 0; crc = 0; isize = 0;
        return true;
    }

    int64_t cap0() const { return std::max<int64_t>(chunk_bytes * 6, 1 << 16) + OUT_SLACK; }      // elements a chunk's buffer starts with
    void grow16(Chunk *c, int64_t cap)
    {
        uint16_t *nb = static_cast<uint16_t *>(malloc((size_t)(WSIZE + cap) * sizeof(uint16_t)));
        if (!nb) throw std::bad_alloc();
        if (c->b16 && c->i16 && c->i16->op) memcpy(nb + WSIZE, c->b16 + WSIZE, (size_t)(c->i16->op - c->i16->out_base) * sizeof(uint16_t));
        for (int i = 0; i < WSIZE; i++) nb[i] = (uint16_t)(0x8000 + i);
        mem16 += (cap - (c->b16 ? c->cap16 : 0)) * (int64_t)sizeof(uint16_t);
        free(c->b16);
        c->b16 = nb; c->cap16 = cap;
    }
    // (a member is through: buffers that grew go back to the allocator -- the next member's chunks start small again)
    void shrink16()
    {
        for (Chunk *c : ck)
            if (c->b16 && c->cap16 > cap0()) {
                mem16 -= c->cap16 * (int64_t)sizeof(uint16_t);
                free(c->b16);
                c->b16 = nullptr; c->cap16 = 0;
                if (c->i16) c->i16->op = nullptr;
            }
        if ((int64_t)in.capacity() > (int64_t)nslots * chunk_bytes * 2 + (8 << 20)) std::vector<uint8_t>().swap(in);
    }
    static void grow8(Chunk *c, int64_t cap, int64_t keep)
    {
        uint8_t *nb = static_cast<uint8_t *>(malloc((size_t)(WSIZE + cap)));
        if (!nb) throw std::bad_alloc();
        if (c->b8) memcpy(nb, c->b8, (size_t)(WSIZE + keep));
        free(c->b8);
        c->b8 = nb; c->cap8 = cap;
    }

    void run_exact(Chunk *c, int64_t in_len, int64_t start_bit)
    {
        c->ok = false;
        if (!c->i8) c->i8 = new Inflater<uint8_t>();
        Inflater<uint8_t> &f = *c->i8;
        const int64_t cap0 = this->cap0();
        if (c->cap8 < cap0) { free(c->b8); c->b8 = nullptr; grow8(c, cap0, 0); }
        memcpy(c->b8, window, WSIZE);
        f.win_valid = win_valid;
        f.limit_bit = c->limit_bit;
        f.set_out(c->b8 + WSIZE, 0, c->cap8);
        f.start(in.data(), in_len, start_bit);
        int r;
        for (;;) {
            r = f.run();
            if (r != R_FULL) break;
            const int64_t have = f.op - f.out_base;
            if (c->cap8 >= max_out) { f.op = f.out_base + f.b_out; r = R_INPUT_END; break; }    // (stops at its last boundary)
            grow8(c, std::min(c->cap8 * 2, max_out), have);
            f.set_out(c->b8 + WSIZE, have, c->cap8);
        }
        c->status = r;
        c->start_bit = start_bit; c->end_bit = f.b_bit; c->nout = f.b_out;
        c->ok = r != R_ERROR && (c->end_bit > start_bit);
    }

    void run_markers(Chunk *c, int64_t in_len)
    {
        c->ok = false;
        if (!c->i16) c->i16 = new Inflater<uint16_t>();
        Inflater<uint16_t> &f = *c->i16;
        const int64_t cap0 = this->cap0();
        f.op = nullptr;
        if (c->cap16 < cap0) grow16(c, cap0);
        int64_t from = c->lo_bit;
        for (int tries = 0; tries < 16; tries++) {
            const int64_t tf = now_ns();
            const int64_t s = find_candidate(in.data(), in_len, from, c->limit_bit);
            stats().ns_find += now_ns() - tf;
            if (s < 0) return;
            f.win_valid = WSIZE;
            f.limit_bit = c->limit_bit;
            f.set_out(c->b16 + WSIZE, 0, c->cap16);
            f.start(in.data(), in_len, s);
            int r;
            for (;;) {
                r = f.run();
                if (r != R_FULL) break;
                const int64_t have = f.op - f.out_base;
                const int64_t ncap = std::min(c->cap16 * 2, max_out);
                if (c->cap16 >= max_out || mem16.load() + (ncap - c->cap16) * (int64_t)sizeof(uint16_t) > mem_budget) {
                    f.op = f.out_base + f.b_out; r = R_INPUT_END; break;            // (stops at its last boundary)
                }
                grow16(c, ncap);
                f.set_out(c->b16 + WSIZE, have, c->cap16);
            }
            if (r == R_ERROR) { from = s + 1; continue; }           // not a block after all: the next candidate
            c->status = r;
            c->start_bit = s; c->end_bit = f.b_bit; c->nout = f.b_out;
            c->ok = c->end_bit > s;
            return;
        }
    }

    // (without a branch: whether a symbol is a literal is as good as random)
    static inline uint8_t resolve1(uint16_t v, const uint8_t *w)
    {
        const uint32_t m = 0u - (uint32_t)(v >> 15);
        return (uint8_t)((v & ~m) | (w[v & 0x7FFFu] & m));
    }

    // n symbols -> bytes.  Literals sixteen at a time (SSE2 pack); a group that holds references -- they come in runs: the
    // part of a header every record copies from the one before -- is then patched element by element.
    static void resolve_run(const uint16_t *sy, uint8_t *o, int64_t n, const uint8_t *w)
    {
        int64_t i = 0;
        for (; i + 16 <= n; i += 16) {
            const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(sy + i));
            const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(sy + i + 8));
            _mm_storeu_si128(reinterpret_cast<__m128i *>(o + i), _mm_packus_epi16(a, b));      // (a reference is negative: packed to 0)
            uint32_t m = ((uint32_t)_mm_movemask_epi8(a) | ((uint32_t)_mm_movemask_epi8(b) << 16)) & 0xAAAAAAAAu;
            while (m) {
                const int j = __builtin_ctz(m) >> 1;
                m &= m - 1;
                o[i + j] = w[sy[i + j] & 0x7FFFu];
            }
        }
        for (; i < n; i++) o[i] = resolve1(sy[i], w);
    }

    // One batch: inflate, stitch.  0: committed (emit hands out its bytes); -1: nothing could be committed (give up here).
    int64_t batch()
    {
        if (giveup_after > 0 && nbatches >= giveup_after) return -1;
        nbatches++;
        const int64_t t00 = now_ns();
        const int64_t byte0 = pos_bit >> 3;
        // (a member's first batch is a small one -- a quarter of a chunk per thread: the first bytes are out sooner)
        const bool first = nbatches == 1 && chunk_bytes >= (1 << 18);
        const int64_t cb = first ? chunk_bytes / 4 : chunk_bytes;
        const int ns = first ? threads : nslots;
        const int64_t slack = std::max<int64_t>(cb, 1 << 20);
        const int64_t want = std::min<int64_t>((int64_t)ns * cb + slack, file_size - byte0);
        if (want <= 0) return -1;
        if ((int64_t)in.size() < want + 16) in.resize((size_t)want + 16);
        // (read side by side: from the page cache one thread copies 5-10 GB/s, a batch inflates at several)
        const int64_t rp = 4 << 20;
        const int nrp = (int)((want + rp - 1) / rp);
        std::vector<int64_t> rgot((size_t)nrp, 0);
        const std::function<void(int)> rd = [&](int q) { rgot[(size_t)q] = pread_full(in.data() + q * rp, std::min(rp, want - q * rp), byte0 + q * rp); };
        if (nrp > 1) pool->run(nrp, rd);
        else rd(0);
        int64_t in_len = 0;
        for (int q = 0; q < nrp; q++) {
            if (rgot[(size_t)q] < 0) return -1;
            in_len += rgot[(size_t)q];
            if (rgot[(size_t)q] < std::min(rp, want - q * rp)) break;          // (the file is shorter than fstat said)
        }
        if (in_len <= 0) return -1;
        memset(in.data() + in_len, 0, 16);
        const int nck = (int)std::min<int64_t>(ns, (in_len + cb - 1) / cb);
        for (int k = 0; k < nck; k++) {
            ck[k]->lo_bit = (int64_t)k * cb * 8;
            ck[k]->limit_bit = (int64_t)(k + 1) * cb * 8;
            ck[k]->ok = false;
        }
        const int64_t rel0 = pos_bit - byte0 * 8;
        std::atomic<int> oom{0};
        const std::function<void(int)> stage1 = [&](int k) {
            try {
                const int64_t t = now_ns();
                if (k == 0) run_exact(ck[0], in_len, rel0);
                else run_markers(ck[k], in_len);
                (k ? stats().ns_markers : stats().ns_exact) += now_ns() - t;
            } catch (const std::bad_alloc &) { ck[k]->ok = false; oom.store(1); }
        };
        const int64_t t1 = now_ns();
        stats().ns_read += t1 - t00;
        pool->run(nck, stage1);
        const int64_t t2 = now_ns();
        stats().ns_stage1 += t2 - t1;
        stats().batches++;
        if (oom.load() || !ck[0]->ok) return -1;
        // ---- stitch: a chunk is taken if it starts at the bit its predecessor ended at -------------------------------
        int na = 0;
        int64_t pos = rel0, total = 0;
        bool fin = false;
        const uint8_t *prev = window;
        int64_t prev_valid = win_valid;
        for (int k = 0; k < nck && !fin; k++) {
            Chunk *c = ck[k];
            if (!c->ok || c->start_bit != pos) break;
            if (k > 0 && prev_valid < WSIZE) break;              // (references into a window shorter than 32 KiB are not checked)
            // the window behind this chunk
            if (k == 0) memcpy(c->win, c->b8 + c->nout, WSIZE);
            else if (c->nout >= WSIZE) resolve_run(c->b16 + c->nout, c->win, WSIZE, prev);
            else {
                memmove(c->win, prev + c->nout, (size_t)(WSIZE - c->nout));
                resolve_run(c->b16 + WSIZE, c->win + (WSIZE - c->nout), c->nout, prev);
            }
            c->out_at = total;
            total += c->nout;
            pos = c->end_bit;
            prev_valid = std::min<int64_t>(WSIZE, prev_valid + c->nout);
            prev = c->win;
            fin = c->status == R_FINAL;
            na++;
        }
        stats().chunks += na;
        stats().rejected += nck - na;
        if (na == 0 || (total == 0 && !fin && pos == rel0)) return -1;
        // ---- committed: the bytes are made when they are asked for (emit) --------------------------------------------------
        stats().ns_stitch += now_ns() - t2;
        memcpy(window, ck[na - 1]->win, WSIZE);
        win_valid = prev_valid;
        pos_bit = byte0 * 8 + pos;
        b_na = na; b_total = total; b_fin = fin; emit_pos = 0;
        if (total == 0) finish_batch();
        return 0;
    }

    // The next min(n, what the batch still holds) bytes of the batch, straight into dst: cut into pieces, each piece
    // resolved (symbols -> bytes through the window in front of its chunk) and summed (CRC-32) by one thread, the sums
    // put together in order.
    int64_t emit(uint8_t *dst, int64_t n)
    {
        const int64_t t0 = now_ns();
        const int64_t m = std::min(n, b_total - emit_pos);
        const int P = (int)std::min<int64_t>(threads, std::max<int64_t>(1, m >> 18));
        std::vector<uint32_t> sums((size_t)P);
        const std::function<void(int)> fn = [&](int p) {
            const int64_t a = emit_pos + m * p / P, b = emit_pos + m * (p + 1) / P;
            uint32_t c32 = (uint32_t)crc32(0L, Z_NULL, 0);
            int k = 0;
            while (k + 1 < b_na && ck[k + 1]->out_at <= a) k++;
            for (; k < b_na && ck[k]->out_at < b; k++) {
                const Chunk *c = ck[k];
                const int64_t lo = std::max(a, c->out_at), hi = std::min(b, c->out_at + c->nout);
                if (hi <= lo) continue;
                uint8_t *o = dst + (lo - emit_pos);
                if (k == 0) memcpy(o, c->b8 + WSIZE + (lo - c->out_at), (size_t)(hi - lo));
                else {
                    const uint16_t *sy = c->b16 + WSIZE + (lo - c->out_at);
                    const uint8_t *w = ck[k - 1]->win;
                    resolve_run(sy, o, hi - lo, w);
                }
                for (int64_t q = 0; q < hi - lo; q += 1 << 30) c32 = (uint32_t)crc32(c32, o + q, (uInt)std::min<int64_t>(hi - lo - q, 1 << 30));
            }
            sums[(size_t)p] = c32;
        };
        if (P == 1) fn(0);
        else pool->run(P, fn);
        for (int p = 0; p < P; p++) {
            const int64_t len = (emit_pos + m * (p + 1) / P) - (emit_pos + m * p / P);
            if (len) crc = (uint32_t)crc32_combine(crc, sums[(size_t)p], (z_off_t)len);
        }
        isize += (uint64_t)m;
        emit_pos += m;
        if (emit_pos == b_total) finish_batch();
        stats().ns_stage2 += now_ns() - t0;
        return m;
    }

    // All of the batch is out: behind a final block comes the member's trailer -- CRC-32 and length (modulo 2^32) of
    // what the member inflates to (RFC 1952 section 2.3.1)
    void finish_batch()
    {
        if (!b_fin) return;
        b_fin = false;
        const int64_t t = (pos_bit + 7) >> 3;
        uint8_t tr[8];
        if (pread_full(tr, 8, t) != 8) { failed = true; msg = "compressed file ended before the end-of-stream marker was reached"; return; }
        const uint32_t fcrc = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        const uint32_t flen = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (fcrc != crc) { failed = true; msg = "incorrect data check"; }
        else if (flen != (uint32_t)isize) { failed = true; msg = "incorrect length check"; }
        member_done = true;
        end_off = t + 8;
        stats().members++;
        shrink16();
    }

    // Up to n bytes of the member.  Returns what was written; then look at member_done / gave_up / failed
    // (each only once `pending()` is 0).
    int64_t read(uint8_t *dst, int64_t n)
    {
        int64_t got = 0;
        while (got < n) {
            try {
                if (emit_pos < b_total) { got += emit(dst + got, n - got); continue; }
                if (member_done || gave_up || failed) break;
                if (batch() < 0) { gave_up = true; stats().giveups++; break; }
            } catch (const std::bad_alloc &) {
                // (out of memory in front of a batch: zlib goes on; while its bytes are handed out: nothing to go back to)
                if (emit_pos < b_total) { failed = true; msg = "out of host memory"; }
                else { gave_up = true; stats().giveups++; }
                break;
            }
        }
        return got;
    }
    int64_t pending() const { return b_total - emit_pos; }
};

}  // namespace pgz
}  // namespace ffq
