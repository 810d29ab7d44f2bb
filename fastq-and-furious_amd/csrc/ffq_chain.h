// ffq_chain.h -- the record chain over the line index, wave-centric kernels.
//
// What is computed: the chain of readfastq_iter
//   /root/reference/src/fastqandfurious.py:251-279
// (search "\n@" from offset; scanner call; offset = pos5 - 1; repeat) with the C
// extension's scanner (/root/reference/src/_fastqandfurious.c:25-153).
//
// A "group" is OWN_T consecutive line-index tiles.  ONE WAVE per group (64-lane
// workgroups: no workgroup barriers, every sync is wave-local):
//   1. loads the line index of a window [run-in tile | OWN_T own tiles |
//      look-ahead tile] into LDS (a few hundred entries);
//   2. makes every "\n@" match of the run-in tail and the own tiles a NODE and
//      computes that node's scanner call and successor (thread per node: the entries
//      after the candidate read independently, the successor from the positions of
//      the next three nodes; what does not fit goes, one node at a time, through the walkers'
//      wave-wide searches: node_wave);
//   3. follows the successor links from the window's earliest node: a chain is a
//      union of runs of consecutive nodes joined by jumps, walked run by run with
//      one LDS read per run (pointer doubling in the dense configuration);
//   4. stages the chain's records of the own tiles as 16-byte group-relative
//      tuples and writes a per-group summary (entry candidate, exit candidate,
//      record count, quality bytes).
//
// Speculation.  For group 0 the chain start is exact.  For g > 0 the chain is
// started at the earliest candidate of the run-in (the last RUNIN_BYTES of the
// previous tile): a chain started at a false '@' candidate (a quality line that
// begins with '@') re-synchronises with the true chain within a few records.
// k_resolve_* then checks y[g+1] == exit[g] for every group up to the one the
// chain ends in; with group 0 exact this proves every guess by induction.  A
// rejected guess is repaired: k_repair_mark gives the group its predecessor's exit
// as a forced entry, k_chain_wave re-runs just those groups, and everything is
// verified again (the first rejected group is exact after each round).  What does
// not fit the LDS budget goes to the dense configuration of the same kernels, then
// to the serial walker (still on the GPU).
//
//   k_chain_wave   steps 1-4                                  (VALU-bound)
//   k_resolve_a/b  verification + exclusive scan of counts    (tiny)
//   k_repair_mark  forced entries for the rejected groups
//   k_expand       staged tuples -> int64[n][6] rows (+ quality CSR offsets)
//                  48 B written per record, coalesced through LDS
#pragma once
#include "ffq_dev.h"

namespace ffq {

constexpr int OWN_T = 4;                   // own tiles per group
constexpr int NTW = OWN_T + 2;             // + run-in tile + look-ahead tile
constexpr int RUNIN_BYTES = 8192;          // tail of the previous tile used as run-in
// LDS entry word: window-relative position (18 bits) | flags << 18 | node id << 20
constexpr uint32_t WP_MASK = 0x3FFFFu;
constexpr int WF_SHIFT = 18;
constexpr int WN_SHIFT = 20;
static_assert(NTW * TILE + 1 <= (int)WP_MASK, "window positions must fit WP_MASK");
constexpr uint32_t WN_MASK = 0x3FFu;
constexpr uint32_t NO_NODE = 0x3FFu;
constexpr uint16_t NX_OUT = 0xFFFF, NX_NOCAND = 0xFFFE, NX_STOP = 0xFFFD;
constexpr uint16_t NM_EXT = 0xFFFF;
constexpr uint16_t UNMARKED = 0xFFFF;

constexpr int64_t Y_NOCAND = -1, X_END_TERM = -2, Y_UNRES = -3, X_END_FINAL = -4;
constexpr int64_t FORCE_NONE = -1;

constexpr int RES_BLOCK = 1024;            // groups per k_resolve_a workgroup
constexpr int DCHUNK = 8192;               // records per chunk of the walked groups' stage (64 KiB of 8-byte records)

struct StageRec { uint32_t p0, p1, p3, p4; };     // relative to the group's window origin
// The lean kernel's groups (ffq_lite.h) stage HALF of that: every record it takes starts in the four own tiles (64 KiB) and
// its call spans less than a tile, so pos0 fits 16 bits counted from the own tiles' first byte and pos1 / pos3 / pos4 fit 16
// bits counted from pos0 -- 8 bytes per record written by k_chain_lite and read back by k_expand instead of 16 (flag bit 4
// of the group says which layout its stage holds; a repair pass or the general kernel rewrites both)
struct StageRec8 { uint16_t d0, d1, d3, d4; };     // pos0 - (window origin + TILE + sentinel shift + 1); pos1 / pos3 / pos4 - pos0
constexpr uint32_t FLAG_STAGE8 = 16u;

struct GroupTerm {
    int64_t pos[6];
    int32_t status;
    int32_t pad;
};

struct ChainBufs {
    int64_t *y;          // [ng] entry candidate ('\n' buffer coordinate) or Y_*
    int64_t *exit;       // [ng] first chain candidate past the own tiles, or Y_NOCAND / X_END_*
    uint32_t *cnt;       // [ng] records of the own tiles
    uint32_t *flags;     // [ng] bit0: irregular (does not fit this kernel's LDS budget)
    uint32_t *lines;     // [ng] newlines in the own tiles
    int64_t *qb;         // [ng] quality bytes of those records
    GroupTerm *term;     // [ng] scanner status/posbuffer where the chain stops (only if it does)
    StageRec *stage;     // [ng][nmax]
    // groups that are WALKED (k_dense_walk: dense tiles, ffq_dense.h) hold up to DCHUNK records: each takes a chunk of a
    // second stage the first time it is walked (sbase[g] = its chunk + 1, 0: the group's records are in `stage`)
    int32_t *sbase;      // [ng]
    StageRec *dstage;    // [dchunks][DCHUNK]
    uint32_t *dhead;     // chunks handed out (and asked for: may exceed dchunks -> ERR_DSTAGE, the host grows the stage)
    int32_t dchunks;
    uint32_t *dlist;     // [ng] groups k_chain_lite (ffq_lite.h) declined, in no particular order
    uint32_t *dcnt;      //      how many
    uint32_t *ilist;     // [ng] groups that do not fit k_chain_wave (flag bit 0), as the first pass meets them: what
    uint32_t *icnt;      //      k_dense_walk looks at, instead of at every group's flags
    int64_t *rloc;       // [ng] exclusive prefix of cnt inside the resolve block
    int64_t *qloc;       // [ng] same for qb
    int64_t *part;       // [nblk][4] block totals (cnt, qb, lines, -) -> exclusive prefixes
    int32_t *mins;       // [4] first terminating group, first bad group, number of bad groups (from the fill value 0x7F7F7F7F up)
    int64_t *force;      // [ng] repair pass: the "\n@" the chain enters the group with (FORCE_NONE: leave alone)
    unsigned long long *prof;   // optional: per-phase cycle sums of k_chain_wave (diagnostics)
    int32_t nmax;
    int32_t ng;
};

struct DevRes {
    int64_t n_records, n_qual_bytes, n_lines, end_offset;
    int64_t last_pos[6];
    int32_t last_status, end_state, fallback, term_group;
    int32_t has_final;
    int32_t bad_group;   // first group whose guess was not confirmed (fallback only; 0x7F7F7F7F: none)
    int32_t bad_irregular;   // that group does not fit this configuration's LDS budget (vs a wrong guess)
    int32_t n_bad;       // groups whose guess was not confirmed (or that did not fit), all of them
    int64_t approx_records;   // records the groups counted, confirmed or not (how long the records are, roughly)
    int32_t fast4_hint;  // written by k_finalize4 only: 1 = the four-line fast path stood on this buffer (probe scans)
    int32_t fused_bad;   // written by k_finalize4 only: FZ_BAD_* of the single-pass decode (ffq_fused.h), 0 = it stood
    int32_t fast4_dense; // written by k_finalize4 only: the row kernel refused a DENSE tile (its DENSE instantiation takes those)
    int32_t n_declined;  // general path: groups k_chain_lite (ffq_lite.h) left to k_chain_wave (many: the host skips the lean kernel next time)
};

// Result hand-over.  The last result-writing kernel of a scan copies the result block and the
// control block into host-mapped pinned memory and leaves the control block zeroed for the
// context's next scan: no copy engine and no memset on the path between two scans.
struct Pub {
    Ctl *ctl;
    Ctl *h_ctl;        // nullptr: this kernel is not the publisher
    DevRes *h_res;
    unsigned long long *h_seq;     // host-mapped word the host polls (FFQ_F_POLL_RESULT), or nullptr
    unsigned long long seq;        // the value that says "this scan's result block is there"
};
__device__ __forceinline__ void publish(const Pub &pb, const DevRes *res)
{
    if (!pb.h_res) return;
    *pb.h_res = *res;
    unsigned long long need = 0;
    if (pb.ctl->pool_any) {
        // (dense tiles only: 16 loads in flight at a time)
        for (int b0 = 0; b0 < POOL_NB; b0 += 16) {
            unsigned long long v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = pb.ctl->pool_heads[b0 + i];
#pragma unroll
            for (int i = 0; i < 16; i++) { need = max(need, v[i]); pb.ctl->pool_heads[b0 + i] = 0; }
        }
        pb.ctl->pool_any = 0;
    }
    pb.h_ctl->err = pb.ctl->err;
    pb.h_ctl->pool_head = need * POOL_NB;
    pb.ctl->err = 0;
    // last, and after everything above has left for host memory
    if (pb.h_seq) __hip_atomic_store(pb.h_seq, pb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    return (uint32_t)__shfl((int)wave_incl_scan(v), 63);
}


// successor codes of a node (values below are node ids)
constexpr uint32_t SN_OUT = 0xFFFFu;      // successor candidate lies beyond the window
constexpr uint32_t SN_NOCAND = 0xFFFEu;   // no further "\n@" in the buffer
constexpr uint32_t SN_STOP = 0xFFFDu;     // the node's scanner call is not COMPLETE
constexpr uint32_t SN_AHEAD = 0xFFFCu;    // in the window, past the own tiles (position in sx)
constexpr int SEG_LIMIT = 128;

// value a[node >> 6] held by lane (node & 63), for a wave-uniform node id:
// v_readlane with scalar operands (a select over the slots would be turned into an
// indexed load from scratch by the compiler)
template <int PER>
__device__ __forceinline__ uint32_t read_node(const uint32_t (&a)[PER], int node)
{
    const int u = __builtin_amdgcn_readfirstlane(node >> 6);
    const int l = __builtin_amdgcn_readfirstlane(node & 63);
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (u == i) v = (uint32_t)__builtin_amdgcn_readlane((int)a[i], l);
    return v;
}

// first set bit at or after node id c over the per-slot masks; PER*64 if none
template <int PER>
__device__ __forceinline__ int first_set_from(const unsigned long long (&m)[PER], int c)
{
    int r = PER * 64;
#pragma unroll
    for (int u = PER - 1; u >= 0; u--) {
        unsigned long long x = m[u];
        if (c > u * 64) x = (c >= u * 64 + 64) ? 0ull : (x >> (c - u * 64)) << (c - u * 64);
        if (x) r = u * 64 + (__ffsll((long long)x) - 1);
    }
    return r;
}

// number of set bits strictly below node id c
template <int PER>
__device__ __forceinline__ int count_below(const unsigned long long (&m)[PER], int c)
{
    int n = 0;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        unsigned long long x = m[u];
        if (c < u * 64 + 64) x = (c <= u * 64) ? 0ull : (x & ((1ull << (c - u * 64)) - 1ull));
        n += __popcll(x);
    }
    return n;
}

// set bits of a wave-wide mask below this lane (v_mbcnt: two instructions, no 64-bit shift); this lane's own bit
__device__ __forceinline__ int bits_below_lane(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ bool bit_of_lane(unsigned long long m, int lane)
{
    return (((lane & 32) ? (uint32_t)(m >> 32) : (uint32_t)m) >> (lane & 31)) & 1u;
}

struct FollowOut {
    Rec r;
    int64_t after;
};

// ---- the generic scanner call of a window node, by the WHOLE WAVE (wave-uniform arguments) ---------
// The fast path of k_chain_wave covers records whose lines all lie within the next entries of the
// window; everything else (long wrapped records, window / buffer edges, records that continue beyond
// the window) comes here, one node at a time: the scanner call of the node whose "\n@" is index
// entry `hk` at coordinate Pk, and the candidate the chain continues with, through the wave-wide
// searches of the walkers (wv_find / wv_record, ffq_dev.h) -- a handful of memory round trips
// whatever the record's length.  (Until round 2 every lane walked its own node entry by entry:
// 13 k dependent loads, 20 ms, for ONE wrapped 1 MB record in a buffer of short reads; tools/cliffs.py.)
// Out of line on purpose (code size); the LineIndex through a pointer to its copy in global memory.
__device__ __noinline__ FollowOut node_wave(const LineIndex *Lg, H hk, int64_t Pk, int eof)
{
    FollowOut f;
    H hm1, hs;
    int64_t Ps;
    int fls;
    wv_record_t<false>(*Lg, hk, Pk, Lg->len(), eof, f.r, hm1);
    f.after = Y_NOCAND;
    if (f.r.status == ST_COMPLETE && wv_find_t<false>(*Lg, hm1, FL_AT, f.r.p5 - 1, hs, Ps, fls)) f.after = Ps;
    return f;
}

// first "\n@" match at coordinate >= X behind tile t1 - 1, by the whole wave (Y_NOCAND: none); out of line
// like node_wave (the search inlined into the kernel costs it registers it has no use for elsewhere)
__device__ __noinline__ int64_t cand_after_wave(const LineIndex *Lg, int t1, int64_t X)
{
    H hs; int64_t Ps; int fls;
    return wv_find_t<true>(*Lg, H{t1 - 1, 0x7FFFFFF0}, FL_AT, X, hs, Ps, fls) ? Ps : Y_NOCAND;
}

// PER: node slots per lane (NMAX = 64*PER nodes per group); EMAX: window entries;
// WPB: waves (= groups) per workgroup; DOUBLING: keep the pointer-doubling fallback for
// chains with more than SEG_LIMIT jumps (otherwise such a group is reported irregular
// and the host re-runs the stage with the DOUBLING configuration).
// (the body for ONE group g, by one wave; the kernel below calls it once per wave, or -- behind k_chain_lite -- for every
// group of the list that kernel declined)
template <int PER, int EMAX, int WPB, bool DOUBLING>
__device__ __forceinline__ void chain_wave_group(const LineIndex &L, const LineIndex *__restrict__ Lg, int64_t offset, int eof,
                                                 const ChainBufs &B, const int g, int only_deferred, int ablate)
{
    constexpr int NMAX = PER * 64;
    constexpr int ND = DOUBLING ? NMAX : 1;
    __shared__ uint32_t went_all[WPB][EMAX + 8];     // (+8: a lane's words past the last entry, see the window loop)
    __shared__ uint16_t nidx_all[WPB][NMAX];   // node -> window entry index of its "\n@"
    __shared__ uint16_t nx16_all[WPB][NMAX];   // node -> successor node / SN_*
    __shared__ uint32_t pk_all[WPB][NMAX];     // node -> run end | successor of the run end << 16
    __shared__ uint32_t bits_all[WPB][2 * (NMAX / 32)];   // run start bits, run end bits
    __shared__ uint16_t dS_all[WPB][ND];       // doubling fallback: node reached
    __shared__ uint16_t dC_all[WPB][ND];       //                    steps taken
    __shared__ uint16_t dD_all[WPB][ND];       //                    rank (UNMARKED: not on the chain)

    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (only_deferred == 1 && !(B.flags[g] & 4u)) return;   // second pass: only what the first one deferred
    if (only_deferred == 3) wave_sync();                    // (one group after another in the same LDS)
    // repair pass (only_deferred == 2): only the groups whose guess the verification rejected,
    // entered where the predecessor's chain says (no run-in speculation)
    const int64_t fpos = (only_deferred == 2) ? B.force[g] : FORCE_NONE;
    if (only_deferred == 2 && fpos == FORCE_NONE) return;
    uint32_t *went = went_all[wid];
    uint16_t *nidx = nidx_all[wid];
    uint32_t *npos = pk_all[wid];              // node -> window position of its "\n@"; the scanner phase
                                                // is through before the membership phase reuses the array

    const int own0 = g * OWN_T;
    const int own1 = min(own0 + OWN_T, L.ntiles);
    const bool has_runin = own0 > 0;
    const int wt0 = has_runin ? own0 - 1 : 0;
    const int wt1 = min(own1 + 1, L.ntiles);
    const int nwt = wt1 - wt0;
    const int sent = (wt0 == 0 && L.s) ? 1 : 0;
    const int64_t wpos0 = (int64_t)wt0 << TILE_SHIFT;
    const int64_t len = L.len();

    const bool prof = PROBES && B.prof != nullptr;
    long long ts[6] = {0, 0, 0, 0, 0, 0};
    if (prof) ts[0] = clock64();
    // ---- window directory + first FPE entries of every tile, one memory round trip ----
    // EPL entries per lane and pass: a pass costs ~25 instructions whatever it holds and ~20 per entry, and a tile of
    // wrapped reads has ~320 lines (80 columns, 50-300 bases: 51 bytes per line) -- with four entries per lane every
    // tile took a second pass for its last 64 entries (16 of 64 lanes busy)
    constexpr int EPL = 6, EPW = EPL / 2, PASS = 64 * EPL;
    constexpr int FP = 2, FPE = FP * PASS;        // 768 entries: lines of 22 bytes or more on average
    int tc[NTW];
    uint32_t ev[NTW][FP][EPW];
#pragma unroll
    for (int k = 0; k < NTW; k++) {
        tc[k] = (k < nwt) ? (int)L.cnt[wt0 + k] : 0;
#pragma unroll
        for (int p = 0; p < FP; p++) {
            typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
            u32x3 t = {0u, 0u, 0u};
            if (k < nwt)
                t = *reinterpret_cast<const u32x3 *>(L.ent + (int64_t)(wt0 + k) * SLOT + p * PASS + EPL * lane);
            ev[k][p][0] = t.x; ev[k][p][1] = t.y; ev[k][p][2] = t.z;
        }
    }
    int tb[NTW + 1];
    tb[0] = sent;
    bool irregular = false;
    uint32_t lines = 0;
#pragma unroll
    for (int k = 0; k < NTW; k++) {
        tb[k + 1] = tb[k] + tc[k];
        if (tc[k] > SLOT) irregular = true;
        if (k < nwt && wt0 + k >= own0 && wt0 + k < own1) lines += (uint32_t)tc[k];
    }
    const int nwin = tb[NTW];
    if (nwin > EMAX) irregular = true;
    if (irregular) {
        if (lane == 0) {
            B.y[g] = Y_UNRES; B.exit[g] = Y_UNRES; B.cnt[g] = 0; B.qb[g] = 0;
            B.flags[g] = 1; B.lines[g] = lines;      // (bit 2 cannot be pending: nothing was looked up)
            if (only_deferred != 2) B.ilist[atomicAdd(B.icnt, 1u)] = (uint32_t)g;      // (the first pass's walker works from this list)
        }
        return;
    }
    if (prof) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ts[1] = clock64(); }
    // own tiles are window tiles [kown0, kown1); tile 0 is the run-in tile when has_runin
    const int kown0 = has_runin ? 1 : 0, kown1 = own1 - wt0;
    // a candidate must lie at buffer coordinate >= offset: as a window-relative bound
    const int64_t offrel64 = offset - wpos0;
    const uint32_t offrel = offrel64 <= 0 ? 0u : (offrel64 > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)offrel64);
    int ncomp = 0, n_runin = 0;
    if (sent) {
        // the iterator's sentinel (fastqandfurious.py:245) is entry 0 of group 0
        const uint8_t b0 = L.n > 0 ? L.d[0] : 0;
        const uint32_t fl = (b0 == '@') ? FL_AT : (b0 == '+') ? FL_PLUS : 0;
        const bool isn = (fl & FL_AT) && offrel == 0;
        if (lane == 0) {
            went[0] = 0u | (fl << WF_SHIFT) | ((isn ? 0u : NO_NODE) << WN_SHIFT);
            if (isn) { nidx[0] = 0; npos[0] = 0u; }
        }
        ncomp = isn ? 1 : 0;
    }
    // ---- window entries -> LDS; the "\n@" matches of the run-in tail and the own tiles
    //      become nodes, numbered in entry order (wave prefix sum per tile) ---------------
    int maxc = 0;
#pragma unroll
    for (int k = 0; k < NTW; k++) {
        // first FPE entries of tile k (already in registers), EPL per lane and pass
        const int c = min(tc[k], FPE);
        maxc = max(maxc, tc[k]);
        const uint32_t relb = (uint32_t)(k << TILE_SHIFT) + (uint32_t)L.s;
        const bool runin_tile = has_runin && k == 0;
        const bool node_tile = runin_tile || (k >= kown0 && k < kown1);
#pragma unroll
        for (int p = 0; p < FP; p++) {
            if (p * PASS >= c) continue;          // wave-uniform
            uint32_t x[EPL];
#pragma unroll
            for (int i = 0; i < EPL; i++) x[i] = (i & 1) ? (ev[k][p][i >> 1] >> 16) : (ev[k][p][i >> 1] & 0xFFFFu);
            uint32_t isn = 0;      // bit i: entry i of this lane becomes a node
            if (node_tile && !runin_tile && relb >= offrel) {
                // an own tile that lies entirely at or behind `offset`: every "\n@" of it is a node
                const int nv = min(max(c - p * PASS - EPL * lane, 0), EPL);
                uint32_t at = 0;
#pragma unroll
                for (int h = 0; h < EPW; h++)
                    at |= (((ev[k][p][h] >> 14) & 1u) | ((ev[k][p][h] >> 29) & 2u)) << (2 * h);
                isn = at & ((1u << nv) - 1u);
            } else if (node_tile) {
#pragma unroll
                for (int i = 0; i < EPL; i++) {
                    const uint32_t off = x[i] & OFF_MASK;
                    if (p * PASS + EPL * lane + i < c && ((x[i] >> 14) & FL_AT) && relb + off >= offrel &&
                        (!runin_tile || off >= (uint32_t)(TILE - RUNIN_BYTES)))
                        isn |= 1u << i;
                }
            }
            uint32_t id = 0;
            if (node_tile) {
                const uint32_t nc = __popc(isn);
                const uint32_t incl = wave_incl_scan(nc);
                id = (uint32_t)ncomp + incl - nc;
                ncomp += (int)__shfl((int)incl, 63);
            }
            // One test per lane, no branch per entry: a lane whose first entry exists writes all its
            // window words (what lies past the tile's count -- at most EPL - 1 words, never a node -- is
            // overwritten by the next tile's entries, written later, or lies past nwin, within the
            // padding of the array, and is never read); the nodes among them, one in seven entries on
            // wrapped reads, are registered by a loop over the set bits.
            const int j0 = p * PASS + EPL * lane;
            if (j0 < c) {
                uint32_t *wdst = went + tb[k] + j0;
                // every word as "not a node" first: offset and flags moved into place with a shift, a
                // bit-field insert and a mask; the window base and the node field come in one add
                const uint32_t wconst = relb + (NO_NODE << WN_SHIFT);
#pragma unroll
                for (int i = 0; i < EPL; i++)
                    // offset stays, the two flag bits move up by WF_SHIFT - 14: x + flags * (2^18 - 2^14), one multiply-add
                    // (relb + offset stays below 2^18: no carry into the flags)
                    wdst[i] = __umul24(x[i] >> 14, (1u << WF_SHIFT) - (1u << 14)) + (x[i] + wconst);
                // the nodes among them (a "\n@" match: AT set, PLUS clear) get their word again, with the id
                uint32_t mrem = isn;
                while (mrem) {
                    const int i = __ffs((int)mrem) - 1;
                    mrem &= mrem - 1u;
                    uint32_t half = ev[k][p][0];
#pragma unroll
                    for (int h = 1; h < EPW; h++) if ((i >> 1) == h) half = ev[k][p][h];
                    const uint32_t off = ((i & 1) ? (half >> 16) : half) & OFF_MASK;
                    const uint32_t idw = min(id, (uint32_t)(NMAX - 1));     // (NMAX - 1 is reserved: such a group is given up below)
                    nidx[idw] = (uint16_t)(tb[k] + j0 + i);
                    npos[idw] = relb + off;
                    wdst[i] = (relb + off) | ((uint32_t)FL_AT << WF_SHIFT) | (idw << WN_SHIFT);
                    id++;
                }
            }
        }
        if (runin_tile && tc[k] <= FPE) n_runin = ncomp;
    }
    if (maxc > FPE) {
        // tiles with more than FPE lines (average line under 22 bytes).  Node ids must
        // follow entry order, so redo the numbering from the first such tile on.
        ncomp = 0; n_runin = 0;
        if (sent) ncomp = (((went[0] >> WN_SHIFT) & WN_MASK) != NO_NODE) ? 1 : 0;
        for (int k = 0; k < nwt; k++) {
            const int c = (int)L.cnt[wt0 + k];
            int base = sent;
            for (int q = 0; q < k; q++) base += (int)L.cnt[wt0 + q];
            const uint32_t relb = (uint32_t)(k << TILE_SHIFT) + (uint32_t)L.s;
            const bool runin_tile = has_runin && k == 0;
            const bool node_tile = runin_tile || (k >= kown0 && k < kown1);
            for (int j0 = 0; j0 < c; j0 += 256) {
                const uint2 v = *reinterpret_cast<const uint2 *>(L.ent + (int64_t)(wt0 + k) * SLOT + j0 + 4 * lane);
                const uint32_t x[4] = {v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16};
                uint32_t isn = 0;
                for (int i = 0; i < 4; i++) {
                    const uint32_t off = x[i] & OFF_MASK;
                    if (node_tile && j0 + 4 * lane + i < c && ((x[i] >> 14) & FL_AT) && relb + off >= offrel &&
                        (!runin_tile || off >= (uint32_t)(TILE - RUNIN_BYTES)))
                        isn |= 1u << i;
                }
                const uint32_t nc = __popc(isn);
                const uint32_t incl = wave_incl_scan(nc);
                uint32_t id = (uint32_t)ncomp + incl - nc;
                ncomp += (int)__shfl((int)incl, 63);
                for (int i = 0; i < 4; i++) {
                    const int j = j0 + 4 * lane + i;
                    if (j < c) {
                        uint32_t nid = NO_NODE;
                        if ((isn >> i) & 1u) {
                            if (id < (uint32_t)(NMAX - 1)) {
                                nid = id; nidx[id] = (uint16_t)(base + j); npos[id] = relb + (x[i] & OFF_MASK);
                            }
                            id++;
                        }
                        went[base + j] = (relb + (x[i] & OFF_MASK)) | ((x[i] >> 14) << WF_SHIFT) | (nid << WN_SHIFT);
                    }
                }
            }
            if (runin_tile) n_runin = ncomp;
        }
    }
    if (PROBES && ablate == 1) { if (lane == 0) B.lines[g] = lines + ncomp; return; }
    int own_hi = 0;                                    // entry index just past the own tiles
#pragma unroll
    for (int k = 0; k <= NTW; k++)
        if (k == kown1) own_hi = tb[k];
    if (prof) ts[2] = clock64();
    if (ncomp >= NMAX) {         // node id NMAX-1 == NO_NODE is reserved
        if (lane == 0) {
            B.y[g] = Y_UNRES; B.exit[g] = Y_UNRES; B.cnt[g] = 0; B.qb[g] = 0;
            B.flags[g] = 1; B.lines[g] = lines;      // (bit 2 cannot be pending: nothing was looked up)
            if (only_deferred != 2) B.ilist[atomicAdd(B.icnt, 1u)] = (uint32_t)g;      // (the first pass's walker works from this list)
        }
        return;
    }
    // behind the last node: three positions no successor test can accept (position + 1 >= qe never holds for 0)
    if (lane < 3 && ncomp + lane <= NMAX - 1) npos[ncomp + lane] = 0u;
    wave_sync();
    if (PROBES && ablate == 2) { if (lane == 0) B.lines[g] = lines + ncomp; return; }
    // the buffer's end as a window position (32-bit tests below)
    const uint32_t lenrel = (uint32_t)min(len - wpos0, (int64_t)0x7FFFFFF0);

    if (prof) ts[3] = clock64();
    // ---- one scanner call + successor search per node (node c = u*64 + lane) -------------
    // per node ONE register: successor (16 bits) | status (5 bits, biased by 1) << 16 |
    // mi << 21 (batch index of the "\n+" entry) | sj << 25 (batch index of the successor) |
    // fast << 29 (set: fields are re-read from the window when the record is staged)
    uint32_t info[PER];
    uint32_t pend = 0;          // slots of this lane that need the generic path
    // slots of this lane whose call lands EXACTLY on its successor: pos5 is the very newline in front of the next '@'.  Every
    // record of a well-formed file does; a call started at a quality line that happens to begin with '@' reads several
    // records as one and lands in the middle of a line.  Only used to choose WHERE the guessed chain starts (below): with
    // records of a kilobase the 8 KiB run-in holds four of them, too few for a chain started at a false candidate to fall
    // in with the true one every time (4 % of the groups were repaired, one pass per scan, at 1 kbp).
    uint32_t clean = 0;
    // successor of node c beyond the entries its call has read: the first "\n@" at >= qe - 1.  Every "\n@" of the own tiles
    // (and of the run-in tail) is a node, numbered in position order: a binary search of the nodes' positions finds it
    // wherever it lies -- until round 5 this looked at the eight entries behind qe - 1 only, and a call that "reads" several
    // records as one (a quality line of a long record that begins with '@': its successor is dozens of lines on) went through
    // the whole-wave call over the global index instead: 4-6 such nodes per group at 1-5 kbp, 55 000 cycles each, 70 % of the
    // kernel's time.  Behind the last node: any "\n@" of the look-ahead tile (SN_AHEAD; the summary finds its position), or
    // ~0u -- nothing in the window, the generic path decides.
    auto far_successor = [&](int c, int from, uint32_t qe) -> uint32_t {
        int lo = c + 1, hi = ncomp;
        while (lo < hi) {
            const int md = (lo + hi) >> 1;
            if (npos[md] + 1 >= qe) hi = md; else lo = md + 1;
        }
        while (lo < ncomp && (int)nidx[lo] < from) lo++;       // (behind the entries the call itself has looked at: the callers' rule)
        if (lo < ncomp) return (uint32_t)lo;
        lo = max(from, own_hi); hi = nwin;
        while (lo < hi) {
            const int md = (lo + hi) >> 1;
            if ((went[md] & WP_MASK) + 1 >= qe) hi = md; else lo = md + 1;
        }
        for (; lo < nwin; lo++)
            if ((went[lo] >> WF_SHIFT) & FL_AT) return SN_AHEAD;
        return 0xFFFFFFFFu;
    };
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int c = u * 64 + lane;
        info[u] = SN_STOP | ((uint32_t)(ST_HEAD_BEG + 1) << 16);
        if (c >= ncomp) continue;
        const int k = nidx[c];
        bool done = false;
        if (k + 12 < nwin) {
            // fast path: the entries after the candidate, read independently, and the positions
            // of the next three nodes (every "\n@" of the own range is a node, so the successor
            // -- the first "\n@" at >= qe - 1 -- is almost always one of them: three reads from
            // consecutive addresses instead of a search through nine entries)
            uint32_t w[10];
#pragma unroll
            for (int i = 0; i < 10; i++) w[i] = went[k + i];
            const uint32_t w12 = went[k + 12];
            uint32_t np[3];
#pragma unroll
            for (int q = 0; q < 3; q++) np[q] = npos[min(c + 1 + q, NMAX - 1)];      // (0 behind the last node: never >= qe - 1)
            const uint32_t r0 = w[0] & WP_MASK, r1 = w[1] & WP_MASK;
            if ((w12 & WP_MASK) + 4 < lenrel) {
                uint32_t plusmask = 0;
#pragma unroll
                for (int i = 2; i <= 9; i++)      // (entry 2 may end an EMPTY line right behind the header's; the later ones lie further on)
                    if (((w[i] >> WF_SHIFT) & FL_PLUS) && (i > 2 || (w[i] & WP_MASK) >= r1 + 2)) plusmask |= 1u << i;
                if (plusmask) {
                    const int mi = __ffs((int)plusmask) - 1;
                    const uint32_t r3 = went[k + mi] & WP_MASK, rm1 = went[k + mi + 1] & WP_MASK;
                    const bool invalid = (rm1 - r3 - 1 > 1) && (rm1 - r3 != r1 - r0);
                    const uint32_t qe = rm1 + 1 + r3 - r1 - 1;
                    int dn = 0;                       // successor = node c + dn (0: not among the next three)
                    if (np[2] + 1 >= qe) dn = 3;
                    if (np[1] + 1 >= qe) dn = 2;
                    if (np[0] + 1 >= qe) dn = 1;
                    if (invalid) {
                        info[u] = SN_STOP | ((uint32_t)(ST_INVALID + 1) << 16);
                        done = true;
                    } else if (dn) {
                        info[u] = (uint32_t)(c + dn) | ((uint32_t)(ST_COMPLETE + 1) << 16) | ((uint32_t)mi << 21) |
                                  (15u << 25) | (1u << 29);
                        if ((dn == 1 ? np[0] : dn == 2 ? np[1] : np[2]) == qe) clean |= 1u << u;
                        done = true;
                    } else {
                        // the successor is not an own node (it lies in the look-ahead tile: the last
                        // records of the group) or more than three candidates away
                        uint32_t atmask = 0;
                        uint32_t wj = 0;
#pragma unroll
                        for (int j = 12; j >= 4; j--) {
                            const uint32_t x = went[k + j];
                            if (((x >> WF_SHIFT) & FL_AT) && (x & WP_MASK) + 1 >= qe && j >= mi + 2) {
                                atmask |= 1u << j;
                                wj = x;                       // ends up as the lowest qualifying j
                            }
                        }
                        if (atmask) {
                            const int j = __ffs((int)atmask) - 1;
                            const uint32_t nid = (wj >> WN_SHIFT) & WN_MASK;
                            const uint32_t nx = (k + j < own_hi && nid != NO_NODE) ? nid : SN_AHEAD;
                            info[u] = nx | ((uint32_t)(ST_COMPLETE + 1) << 16) | ((uint32_t)mi << 21) |
                                      ((uint32_t)j << 25) | (1u << 29);
                            if ((wj & WP_MASK) == qe) clean |= 1u << u;
                            done = true;
                        } else if (qe + 2 < lenrel) {
                            // the record is COMPLETE but its successor lies beyond the batch (a
                            // candidate inside a wrapped quality block "reads" several records as
                            // one).  sj = 15: not encoded.
                            const uint32_t nx = far_successor(c, k + 13, qe);
                            if (nx != 0xFFFFFFFFu) {
                                info[u] = nx | ((uint32_t)(ST_COMPLETE + 1) << 16) | ((uint32_t)mi << 21) |
                                          (15u << 25) | (1u << 29);
                                if (nx < (uint32_t)NMAX && npos[nx] == qe) clean |= 1u << u;
                                done = true;
                            }
                        }
                    }
                } else {
                    // no "\n+" among the next nine entries: a record wrapped over many lines.  Look
                    // further, eight entries at a time (up to 255 entries, ~20 KB at 80 columns);
                    // bit 31 of the node word says that mi is the wide field (8 bits, no sj).
                    int mi = 0;
                    const int lim = min(nwin - k - 2, 256);          // (mi is an 8-bit field of the node word)
                    for (int bb = 10; bb < lim && !mi; bb += 8) {
                        uint32_t pm = 0;
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const uint32_t x = went[min(k + bb + i, nwin - 1)];
                            if (bb + i < lim && ((x >> WF_SHIFT) & FL_PLUS) && (x & WP_MASK) >= r1 + 2) pm |= 1u << i;
                        }
                        if (pm) mi = bb + (__ffs((int)pm) - 1);
                    }
                    if (mi) {
                        const uint32_t r3 = went[k + mi] & WP_MASK, rm1 = went[k + mi + 1] & WP_MASK;
                        if (rm1 + 2 <= lenrel) {      // the '+' line end lies inside the scanner's memchr range
                            const bool invalid = (rm1 - r3 - 1 > 1) && (rm1 - r3 != r1 - r0);
                            const uint32_t qe = rm1 + 1 + r3 - r1 - 1;
                            if (invalid) {
                                info[u] = SN_STOP | ((uint32_t)(ST_INVALID + 1) << 16);
                                done = true;
                            } else if (qe + 2 < lenrel) {
                                uint32_t nx = 0xFFFFFFFFu;
                                if (np[0] + 1 >= qe) nx = (uint32_t)(c + 1);
                                else if (np[1] + 1 >= qe) nx = (uint32_t)(c + 2);
                                else if (np[2] + 1 >= qe) nx = (uint32_t)(c + 3);
                                else nx = far_successor(c, k + mi + 2, qe);
                                if (nx != 0xFFFFFFFFu) {
                                    info[u] = nx | ((uint32_t)(ST_COMPLETE + 1) << 16) | ((uint32_t)mi << 21) |
                                              (1u << 29) | (1u << 31);
                                    if (nx < (uint32_t)NMAX && npos[nx] == qe) clean |= 1u << u;
                                    done = true;
                                }
                            }
                        }
                    }
                }
            }
        }
        if (!done) pend |= 1u << u;
    }
    // nodes the fast path could not finish (long wrapped records, window / buffer edges)
    if (prof) { const uint32_t np = wave_sum_u32((uint32_t)__popc(pend)); if (lane == 0) atomicAdd(&B.prof[7], (unsigned long long)np | ((unsigned long long)wave_max_u32((uint32_t)__popc(pend)) << 32)); }
    // (one node at a time, by the whole wave: node_wave)
    const int64_t own_end_pos = ((int64_t)own1 << TILE_SHIFT) + L.s;      // first coordinate past the own tiles
    const int64_t win_end_pos = ((int64_t)wt1 << TILE_SHIFT) + L.s;       //                   past the window
    auto node_handle = [&](int k) -> H {          // window entry index -> handle in the global index
        if (sent && k == 0) return H{-1, 0};
        int kk = 0;
#pragma unroll
        for (int q = 1; q < NTW; q++) if (k >= tb[q]) kk = q;
        return H{wt0 + kk, k - tb[kk]};
    };
#pragma unroll
    for (int u = 0; u < PER; u++) {
        unsigned long long pm = __ballot((pend >> u) & 1u);
        while (pm) {
            const int ln = __ffsll((long long)pm) - 1;
            pm &= pm - 1ull;
            const int c = u * 64 + ln;
            const int k = nidx[c];
            const int64_t Pk = wpos0 + (int64_t)(went[k] & WP_MASK);
            const FollowOut f = node_wave(Lg, node_handle(k), Pk, eof);
            uint32_t nxn = SN_STOP, ext = 0;
            if ((f.r.status == ST_COMPLETE || f.r.final_) && f.r.p4 - wpos0 > 0xFFFFFFF0ll) ext = 1;
            if (f.r.status == ST_COMPLETE) {
                if (f.after == Y_NOCAND) nxn = SN_NOCAND;
                else if (f.after >= win_end_pos) nxn = SN_OUT;
                else {
                    // inside the window: a node if it lies in front of the own tiles' end (every "\n@"
                    // there at or behind the offset is one), else a candidate of the look-ahead tile
                    nxn = SN_AHEAD;
                    if (f.after < own_end_pos) {
                        const uint32_t rel = (uint32_t)(f.after - wpos0);
#pragma unroll
                        for (int q = 0; q < PER; q++) {
                            const int cq = q * 64 + lane;
                            const unsigned long long hit = __ballot(cq < ncomp && npos[cq] == rel);
                            if (hit) nxn = (uint32_t)(q * 64 + (__ffsll((long long)hit) - 1));
                        }
                    }
                }
            }
            const uint32_t st = (uint32_t)(f.r.final_ ? ST_FINAL : f.r.status);
            const uint32_t v = nxn | ((st + 1u) << 16) | (ext ? (1u << 30) : 0u);
            if (lane == ln) { info[u] = v; if (f.r.status == ST_COMPLETE && f.after == f.r.p5) clean |= 1u << u; }
        }
    }
    pend = 0;
    if (PROBES && ablate == 3) { if (lane == 0) B.lines[g] = lines + info[0]; return; }
    int e_forced = -1;
    if (fpos != FORCE_NONE) {
        if (fpos >= ((int64_t)own1 << TILE_SHIFT) + L.s) {
            // the chain passes over this group (one record spans it): no members, same exit
            if (lane == 0) {
                B.y[g] = fpos; B.exit[g] = fpos; B.cnt[g] = 0; B.qb[g] = 0; B.flags[g] = 0; B.lines[g] = lines;
            }
            return;
        }
        const uint32_t frel = (uint32_t)(fpos - wpos0);
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int c = u * 64 + lane;
            const unsigned long long hit = __ballot(c < ncomp && npos[c] == frel);
            if (hit) e_forced = u * 64 + (__ffsll((long long)hit) - 1);
        }
        if (e_forced >= 0) n_runin = e_forced;       // nothing in front of it belongs to the chain
    }
    if (prof) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ts[4] = clock64(); }
    // ---- chain membership ---------------------------------------------------------------------
    // Node ids are in position order and a successor always lies further on, so a
    // chain is a union of runs c, c+1, ..., r of "simple" nodes (successor == c+1)
    // joined by jumps.  Walk run by run (wave-uniform); MB = member bit per node.
    unsigned long long NS[PER], MB[PER];
    uint16_t *nx16 = nx16_all[wid];
    uint32_t *pk = pk_all[wid];
    uint32_t *sbits = bits_all[wid], *ebits = bits_all[wid] + NMAX / 32;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int c = u * 64 + lane;
        NS[u] = __ballot(!(c < ncomp && (info[u] & 0xFFFFu) == (uint32_t)(c + 1)));
        if (c < ncomp) nx16[c] = (uint16_t)(info[u] & 0xFFFFu);
    }
    wave_sync();
    // pk[c] = end of the run that contains c | (successor of that run end) << 16
    {
        int later = NMAX;                 // first non-simple node in the slots behind u (uniform)
#pragma unroll
        for (int u = PER - 1; u >= 0; u--) {
            const int c = u * 64 + lane;
            const unsigned long long m = NS[u] >> lane;
            const int r = m ? c + (__ffsll((long long)m) - 1) : later;
            if (c < ncomp) pk[c] = (uint32_t)r | ((uint32_t)nx16[min(r, NMAX - 1)] << 16);
            if (NS[u]) later = u * 64 + (__ffsll((long long)NS[u]) - 1);
        }
    }
    int e0 = (e_forced >= 0) ? e_forced : 0, lastn = -1;
    if (e_forced < 0 && has_runin) {          // (the first group starts where the caller says: it is the induction's base)
        // the guessed chain starts at the first run-in node whose call lands exactly (see `clean`); none: at the first node
        // (better: whose SUCCESSOR lands exactly too -- one exact landing from a false candidate is a 1-in-10^4 event per
        // group, which at 65 536 groups still meant a repair pass per scan; two in a row is not seen)
        unsigned long long CL[PER], C2[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) CL[u] = __ballot(((clean >> u) & 1u) != 0u);
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t nx = info[u] & 0xFFFFu;
            bool ok2 = false;
            if (((clean >> u) & 1u) && nx < (uint32_t)NMAX) {
                unsigned long long wsel = 0ull;
#pragma unroll
                for (int q = 0; q < PER; q++) if ((int)(nx >> 6) == q) wsel = CL[q];
                ok2 = ((wsel >> (nx & 63u)) & 1ull) != 0ull;
            }
            C2[u] = __ballot(ok2);
        }
        // (a run-in without such a node -- records of several kilobases: the 8 KiB tail holds no header at all -- takes the
        // first one of the own tiles: the first true header there, which is where the predecessor's chain arrives)
        // (three in a row where the window shows that much: at 5 kbp -- six records per group, one false candidate per record --
        // two exact landings from a false start did turn up, once in 65 452 groups, and cost every scan a repair pass)
        unsigned long long C3[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t nx = info[u] & 0xFFFFu;
            bool ok3 = false;
            if (((clean >> u) & 1u) && nx < (uint32_t)NMAX) {
                unsigned long long wsel = 0ull;
#pragma unroll
                for (int q = 0; q < PER; q++) if ((int)(nx >> 6) == q) wsel = C2[q];
                ok3 = ((wsel >> (nx & 63u)) & 1ull) != 0ull;
            }
            C3[u] = __ballot(ok3);
        }
        int cs = first_set_from<PER>(C3, 0);
        if (cs >= ncomp) cs = first_set_from<PER>(C2, 0);
        if (cs >= ncomp) cs = first_set_from<PER>(CL, 0);
        if (cs < ncomp) e0 = cs;
#ifdef FFQ_PROBES
        if (ablate >= 1000 && g == ablate - 1000) {
            // (FFQ_ABLATE=1000+g: the nodes of one group as the start rule sees them)
            if (lane == 0) printf("[group %d] ncomp %d n_runin %d nwin %d own_hi %d start %d (C3 %d C2 %d CL %d) wpos0 %lld\n", g, ncomp, n_runin, nwin, own_hi, e0,
                                  first_set_from<PER>(C3, 0), first_set_from<PER>(C2, 0), first_set_from<PER>(CL, 0), (long long)wpos0);
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int c = u * 64 + lane;
                if (c < ncomp) printf("[group %d] node %3d entry %4d pos %lld succ %5u status %2d wide %d clean %d\n", g, c, (int)nidx[c], (long long)(wpos0 + (int64_t)(went[nidx[c]] & WP_MASK)),
                                      info[u] & 0xFFFFu, (int)((info[u] >> 16) & 31u) - 1, (int)(info[u] >> 31), (int)((clean >> u) & 1u));
            }
        }
#endif
    }
    bool unresolved = (fpos != FORCE_NONE && e_forced < 0), too_many_jumps = false;
    for (int attempt = 0; attempt < 4 && ncomp > 0 && !unresolved; attempt++) {
        if (lane < 2 * (NMAX / 32)) bits_all[wid][lane] = 0u;
        wave_sync();
        // serial part: one LDS read per run; run [cur, r] recorded as a start bit and an end bit
        int cur = e0, seg = 0;
        bool walked = false;
        for (; seg < SEG_LIMIT; seg++) {
            const uint32_t p = pk[cur];
            const int r = (int)(p & 0xFFFFu);
            const uint32_t nx = p >> 16;
            if (lane == 0) {
                atomicOr(&sbits[cur >> 5], 1u << (cur & 31));
                atomicOr(&ebits[r >> 5], 1u << (r & 31));
            }
            lastn = r;
            if (nx >= SN_AHEAD) { walked = true; break; }
            cur = (int)nx;
        }
        wave_sync();
        if (walked) {
            // member(c) = runs started at or before c outnumber runs ended before c
            int sb = 0, eb = 0;
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const unsigned long long sw = (unsigned long long)sbits[2 * u] | ((unsigned long long)sbits[2 * u + 1] << 32);
                const unsigned long long ew = (unsigned long long)ebits[2 * u] | ((unsigned long long)ebits[2 * u + 1] << 32);
                const int s_le = sb + bits_below_lane(sw) + (bit_of_lane(sw, lane) ? 1 : 0);
                const int e_lt = eb + bits_below_lane(ew);
                MB[u] = __ballot(s_le > e_lt);
                sb += __popcll(sw); eb += __popcll(ew);
            }
        }
        if (!walked && !DOUBLING) { unresolved = true; too_many_jumps = true; break; }
        if (!walked) {
            // too many jumps for the run walk: pointer doubling with bottom-up marking
            uint16_t *dS = dS_all[wid], *dC = dC_all[wid], *dD = dD_all[wid];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int c = u * 64 + lane;
                if (c < ncomp) {
                    const uint16_t s = ((info[u] & 0xFFFFu) < SN_AHEAD) ? (uint16_t)(info[u] & 0xFFFFu) : (uint16_t)c;
                    dS[c] = s; dC[c] = (s != c) ? 1 : 0; dD[c] = (c == e0) ? 0 : UNMARKED;
                }
            }
            wave_sync();
            int rounds = 1;
            while ((1 << rounds) < ncomp) rounds++;
            for (int k = 0; k < rounds; k++) {
                uint16_t s1[PER], s2[PER], c1[PER], c2[PER], dd[PER];
                bool mk[PER];
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int c = u * 64 + lane;
                    mk[u] = false;
                    if (c < ncomp) {
                        s1[u] = dS[c]; c1[u] = dC[c];
                        s2[u] = dS[s1[u]]; c2[u] = dC[s1[u]];
                        dd[u] = dD[c];
                        mk[u] = (dd[u] != UNMARKED) && (c1[u] == (uint16_t)(1u << k));
                    }
                }
                wave_sync();
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int c = u * 64 + lane;
                    if (c < ncomp) {
                        if (mk[u]) dD[s1[u]] = (uint16_t)(dd[u] + (1u << k));
                        dS[c] = s2[u];
                        dC[c] = (uint16_t)(c1[u] + c2[u]);
                    }
                }
                wave_sync();
            }
            uint32_t lastkey = 0;
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int c = u * 64 + lane;
                const bool m = (c < ncomp) && dD[c] != UNMARKED;
                MB[u] = __ballot(m);
                if (m) lastkey = max(lastkey, (uint32_t)c);
            }
            lastn = (int)wave_max_u32(lastkey);
        }
        const int lst = (int)((read_node<PER>(info, lastn) >> 16) & 31u) - 1;
        const bool died_in_runin = (lastn < n_runin) && (lst != ST_COMPLETE);
        if (!died_in_runin) break;
        // the chain stopped inside the run-in (it started at a false candidate):
        // restart from the next run-in node it did not visit
        unsigned long long free_[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) free_[u] = ~MB[u];
        const int nxt = first_set_from<PER>(free_, e0 + 1);
        if (nxt >= n_runin || attempt == 3) { unresolved = true; break; }
        e0 = nxt;
    }
    if (PROBES && ablate == 4) { if (lane == 0) B.lines[g] = lines + lastn; return; }

    if (prof) ts[5] = clock64();
    // ---- summary + staging of the own tiles' records ---------------------------------------------
    const int ynode = (ncomp > 0 && !unresolved) ? first_set_from<PER>(MB, n_runin) : NMAX;
    const bool have_y = ynode < ncomp;
    int64_t Y = Y_UNRES, EX = Y_UNRES;
    int32_t tstatus = 0;
    bool have_term = false;
    Rec tr;
    tr.p0 = tr.p1 = tr.p3 = tr.p4 = tr.p5 = -1; tr.status = 0; tr.final_ = false;
    if (!unresolved) {
        if (ncomp == 0) {
            // no candidate in the run-in tail / own tiles: the chain passes over this group.  The next
            // "\n@" behind the own tiles by the whole wave, 64 index entries per step (a lane walking
            // entry by entry took 20 ms over the 13 k lines of one wrapped 1 MB record; tools/cliffs.py)
            const int64_t Ps = cand_after_wave(Lg, own1, offset);
            if (lane == 0) {
                Y = Ps;
                EX = Y;
                if (Ps == Y_NOCAND) { have_term = true; tstatus = ST_HEAD_BEG; }
            }
        } else {
            const uint32_t li = read_node<PER>(info, lastn);
            const int st = (int)((li >> 16) & 31u) - 1;
            const uint32_t nx = li & 0xFFFFu;
            // what the chain does behind its last node of this group: known from the node word, or the
            // node's call once more by the whole wave (its successor lies beyond the window, the chain
            // stops there and the posbuffer of that call is wanted, or it is a generic node)
            const bool from_word = st == ST_COMPLETE &&
                                   (nx == SN_NOCAND || (nx == SN_AHEAD && (li >> 29 & 1u) && !(li >> 31) && ((li >> 25) & 15u) != 15u));
            // ... or an in-window call (bit 29) whose successor far_successor found in the look-ahead tile: the first "\n@"
            // there at >= pos5 - 1, by the whole wave over the window (the same entries, the same rule)
            const bool from_window = st == ST_COMPLETE && !from_word && nx == SN_AHEAD && (li >> 29 & 1u);
            int64_t after_win = Y_NOCAND;
            if (from_window) {
                const int kl = nidx[lastn];
                const int mi = (li >> 31) ? (int)((li >> 21) & 255u) : (int)((li >> 21) & 15u);
                const uint32_t r1 = went[kl + 1] & WP_MASK, r3 = went[kl + mi] & WP_MASK, rm1 = went[kl + mi + 1] & WP_MASK;
                const uint32_t qe = rm1 + 1 + r3 - r1 - 1;
                for (int j0 = own_hi; j0 < nwin && after_win == Y_NOCAND; j0 += 64) {
                    const int j = j0 + lane;
                    const uint32_t x = (j < nwin) ? went[j] : 0u;
                    const unsigned long long hit = __ballot(j < nwin && ((x >> WF_SHIFT) & FL_AT) && (x & WP_MASK) + 1 >= qe);
                    if (hit) after_win = wpos0 + (int64_t)(went[j0 + (__ffsll((long long)hit) - 1)] & WP_MASK);
                }
            }
            FollowOut fo;
            fo.after = Y_NOCAND;
            fo.r = tr;
            if (!from_word && !(from_window && after_win != Y_NOCAND)) {
                const int kl = nidx[lastn];
                fo = node_wave(Lg, node_handle(kl), wpos0 + (int64_t)(went[kl] & WP_MASK), eof);
            }
            if (lane == 0) {
                if (st == ST_COMPLETE) {
                    int64_t after;
                    if (nx == SN_NOCAND) after = Y_NOCAND;
                    else if (from_word) after = wpos0 + (int64_t)(went[nidx[lastn] + ((li >> 25) & 15u)] & WP_MASK);
                    else if (from_window && after_win != Y_NOCAND) after = after_win;
                    else after = fo.after;
                    EX = after;
                    if (after == Y_NOCAND) { have_term = true; tstatus = ST_HEAD_BEG; }
                } else {
                    // the chain stops at lastn: keep the scanner's posbuffer of that call
                    tr = fo.r;
                    EX = (st == ST_FINAL) ? X_END_FINAL : X_END_TERM;
                    have_term = true; tstatus = tr.status;
                }
                Y = have_y ? wpos0 + (int64_t)(went[nidx[ynode]] & WP_MASK) : EX;
            }
        }
    }
    // records of the own tiles, in chain order: rank = members in front (position order)
    uint32_t cnt = 0;
    unsigned long long qsum = 0;
    bool bad_range = false;
    unsigned long long RM[PER];              // own-tile members that are records (emit a row)
    const int d0 = (!unresolved && have_y) ? count_below<PER>(MB, ynode) : 0;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int c = u * 64 + lane;
        const int st = (int)((info[u] >> 16) & 31u) - 1;
        const bool rec = !unresolved && have_y && c < ncomp && c >= n_runin && bit_of_lane(MB[u], lane) &&
                         (st == ST_COMPLETE || st == ST_FINAL);
        if (rec && ((info[u] >> 30) & 1u)) bad_range = true;
        RM[u] = __ballot(rec && !((info[u] >> 30) & 1u));
        cnt += (uint32_t)__popcll(RM[u]);            // wave-uniform
    }
    {
        StageRec *stg = B.stage + (int64_t)g * B.nmax;
        int mbase = 0;                       // members in the slots in front (wave-uniform)
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int c = u * 64 + lane;
            const int rank = mbase + bits_below_lane(MB[u]) - d0;
            mbase += __popcll(MB[u]);
            if (!bit_of_lane(RM[u], lane)) continue;
            StageRec o;
            if ((info[u] >> 29) & 1u) {
                const int k = nidx[c];
                const int mi = (info[u] >> 31) ? (int)((info[u] >> 21) & 255u) : (int)((info[u] >> 21) & 15u);
                o.p0 = (went[k] & WP_MASK) + 1;
                o.p1 = went[k + 1] & WP_MASK;
                o.p3 = went[k + mi] & WP_MASK;
                o.p4 = (went[k + mi + 1] & WP_MASK) + 1;
            } else {
                pend |= 1u << u;       // fields through the generic path, below
                continue;
            }
            stg[rank] = o;
            qsum += (unsigned long long)(o.p3 - o.p1 - 1);
        }
#pragma unroll
        for (int u = 0; u < PER; u++) {
            unsigned long long pm = __ballot((pend >> u) & 1u);
            while (pm) {
                const int ln = __ffsll((long long)pm) - 1;
                pm &= pm - 1ull;
                const int c = u * 64 + ln;
                const int k = nidx[c];
                const FollowOut f = node_wave(Lg, node_handle(k), wpos0 + (int64_t)(went[k] & WP_MASK), eof);
                if (lane == ln) {
                    StageRec o;
                    o.p0 = (uint32_t)(f.r.p0 - wpos0); o.p1 = (uint32_t)(f.r.p1 - wpos0);
                    o.p3 = (uint32_t)(f.r.p3 - wpos0); o.p4 = (uint32_t)(f.r.p4 - wpos0);
                    stg[count_below<PER>(MB, c) - d0] = o;
                    qsum += (unsigned long long)(o.p3 - o.p1 - 1);
                }
            }
        }
        pend = 0;
    }
    const uint32_t qlo = wave_sum_u32((uint32_t)(qsum & 0xFFFFFu)), qhi = wave_sum_u32((uint32_t)(qsum >> 20));
    const bool anybad = __ballot(bad_range) != 0ull;
    if (lane == 0) {
        const bool bad = unresolved || anybad;
        B.y[g] = bad ? Y_UNRES : Y;
        B.exit[g] = bad ? Y_UNRES : EX;
        B.cnt[g] = cnt;
        B.qb[g] = ((int64_t)qhi << 20) + (int64_t)qlo;
        // bit 2 (deferred) may have been set by a lookup of this pass: keep it
        if (only_deferred) B.flags[g] = (anybad || too_many_jumps) ? 1u : 0u;
        else if (anybad || too_many_jumps) atomicOr(&B.flags[g], 1u);
        B.lines[g] = lines;
        if (have_term) {
            GroupTerm &t = B.term[g];
            t.status = tstatus;
            t.pos[0] = tr.p0; t.pos[1] = tr.p1; t.pos[2] = (tr.p1 >= 0) ? tr.p1 + 1 : -1;
            t.pos[3] = tr.p3; t.pos[4] = tr.p4; t.pos[5] = tr.p5;
        }
        if (prof) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const long long te = clock64();
            atomicAdd(&B.prof[0], (unsigned long long)(ts[1] - ts[0]));   // loads landed
            atomicAdd(&B.prof[1], (unsigned long long)(ts[2] - ts[1]));   // window written to LDS
            atomicAdd(&B.prof[2], (unsigned long long)(ts[3] - ts[2]));   // nodes numbered
            atomicAdd(&B.prof[3], (unsigned long long)(ts[4] - ts[3]));   // scanner calls
            atomicAdd(&B.prof[4], (unsigned long long)(ts[5] - ts[4]));   // membership
            atomicAdd(&B.prof[5], (unsigned long long)(te - ts[5]));      // summary + staging
            atomicAdd(&B.prof[6], 1ull);
        }
    }
}

template <int PER, int EMAX, int WPB, bool DOUBLING>
__global__ __launch_bounds__(WPB * 64) void k_chain_wave(LineIndex L, const LineIndex *__restrict__ Lg,
                                                         int64_t offset, int eof, ChainBufs B, int g0, int g1,
                                                         int only_deferred, int ablate)
{
    const int wid = threadIdx.x >> 6;
    const int g = g0 + blockIdx.x * WPB + wid;
    if (g >= g1) return;                 // no workgroup barrier is used below
    chain_wave_group<PER, EMAX, WPB, DOUBLING>(L, Lg, offset, eof, B, g, only_deferred, ablate);
}

// behind k_chain_lite (ffq_lite.h): the groups that kernel declined, from its list -- a grid of a few thousand waves takes
// them one after another (a launch over ALL groups, most of which return at once, cost 75 us per 10 GiB).  A kernel of its
// own: with both call sites in one kernel the register allocation of the direct one suffered (119 -> 201 VGPRs).
template <int PER, int EMAX, int WPB, bool DOUBLING>
__global__ __launch_bounds__(WPB * 64) void k_chain_wave_list(LineIndex L, const LineIndex *__restrict__ Lg,
                                                              int64_t offset, int eof, ChainBufs B)
{
    const int wid = threadIdx.x >> 6;
    const uint32_t nl = *B.dcnt;
    // (the verification's minima start at their fill value: set here, behind the lean kernel and in front of k_resolve_a,
    // instead of by a 16-byte fill of its own between two kernels)
    if (blockIdx.x == 0 && threadIdx.x < 4) B.mins[threadIdx.x] = 0x7F7F7F7F;
    for (uint32_t i = blockIdx.x * WPB + wid; i < nl; i += gridDim.x * WPB)
        chain_wave_group<PER, EMAX, WPB, DOUBLING>(L, Lg, offset, eof, B, (int)B.dlist[i], 3, 0);
}

// =========================================================================
// k_resolve_a: RES_BLOCK groups per workgroup.  Finds the first group the chain
// ends in and the first group whose guess is not confirmed by its predecessor's
// exit; exclusive-scans counts inside the block.
// k_resolve_b: one workgroup.  Scans the block totals, decides parallel vs
// serial, fills the result block (end state per fastqandfurious.py:256-279).
// =========================================================================
__global__ __launch_bounds__(RES_BLOCK) void k_resolve_a(ChainBufs B)
{
    __shared__ long long s_c[RES_BLOCK], s_q[RES_BLOCK];
    __shared__ int s_term, s_bad, s_nbad;
    __shared__ unsigned long long s_lines;
    const int tid = threadIdx.x;
    const int g = blockIdx.x * RES_BLOCK + tid;
    if (tid == 0) { s_term = 0x7FFFFFFF; s_bad = 0x7FFFFFFF; s_nbad = 0; s_lines = 0; }
    __syncthreads();
    long long c = 0, q = 0;
    if (g < B.ng) {
        const int64_t y = B.y[g], ex = B.exit[g];
        c = B.cnt[g]; q = B.qb[g];
        if (ex == Y_NOCAND || ex == X_END_TERM || ex == X_END_FINAL) atomicMin(&s_term, g);
        bool bad = (B.flags[g] & 5u) || (y == Y_UNRES);
        if (g > 0) {
            const int64_t pe = B.exit[g - 1];
            if (pe >= 0 && y != pe) bad = true;
        }
        if (bad) { atomicMin(&s_bad, g); atomicAdd(&s_nbad, 1); }
        atomicAdd(&s_lines, (unsigned long long)B.lines[g]);
    }
    s_c[tid] = c; s_q[tid] = q;
    __syncthreads();
    for (int d = 1; d < RES_BLOCK; d <<= 1) {
        long long v = 0, vq = 0;
        if (tid >= d) { v = s_c[tid - d]; vq = s_q[tid - d]; }
        __syncthreads();
        s_c[tid] += v; s_q[tid] += vq;
        __syncthreads();
    }
    if (g < B.ng) { B.rloc[g] = s_c[tid] - c; B.qloc[g] = s_q[tid] - q; }
    if (tid == RES_BLOCK - 1) {
        B.part[blockIdx.x * 4 + 0] = s_c[tid];
        B.part[blockIdx.x * 4 + 1] = s_q[tid];
    }
    if (tid == 0) {
        B.part[blockIdx.x * 4 + 2] = (long long)s_lines;
        if (s_term != 0x7FFFFFFF) atomicMin(&B.mins[0], s_term);
        if (s_bad != 0x7FFFFFFF) atomicMin(&B.mins[1], s_bad);
        if (s_nbad) atomicAdd(reinterpret_cast<unsigned int *>(&B.mins[2]), (unsigned int)s_nbad);     // (counts up from the fill value)
    }
}

// k_repair_mark: which groups a repair pass re-runs and where they are entered.  A group whose
// entry guess differs from its predecessor's exit (or that could not settle on one) is re-run
// from that exit.  The first such group is then exact (everything in front of it is verified);
// the later ones are exact if their predecessors were -- the next verification decides.
__global__ void k_repair_mark(ChainBufs B)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B.ng) return;
    int64_t f = FORCE_NONE;
    if (g > 0 && !(B.flags[g] & 4u)) {
        const int64_t pe = B.exit[g - 1], y = B.y[g];
        // (a group that does not fit k_chain_wave, bit 0, has no guess at all: k_group_walk enters
        // it at its predecessor's exit once that is known)
        if (pe >= 0 && (y != pe || (B.flags[g] & 1u))) f = pe;
    }
    B.force[g] = f;
}

__global__ __launch_bounds__(1024) void k_resolve_b(ChainBufs B, int nblk, int eof, int64_t offset,
                                                    int64_t add, DevRes *res)
{
    __shared__ long long s_c[1024], s_q[1024], s_l[1024];
    const int tid = threadIdx.x;
    // exclusive scan of the block totals (nblk <= 1024 per pass, carried across passes)
    long long carry_c = 0, carry_q = 0, carry_l = 0;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + tid;
        const long long c = (b < nblk) ? B.part[b * 4 + 0] : 0, q = (b < nblk) ? B.part[b * 4 + 1] : 0,
                        l = (b < nblk) ? B.part[b * 4 + 2] : 0;
        s_c[tid] = c; s_q[tid] = q; s_l[tid] = l;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            long long v = 0, vq = 0, vl = 0;
            if (tid >= d) { v = s_c[tid - d]; vq = s_q[tid - d]; vl = s_l[tid - d]; }
            __syncthreads();
            s_c[tid] += v; s_q[tid] += vq; s_l[tid] += vl;
            __syncthreads();
        }
        if (b < nblk) { B.part[b * 4 + 0] = carry_c + s_c[tid] - c; B.part[b * 4 + 1] = carry_q + s_q[tid] - q; }
        const long long tc = s_c[1023], tq = s_q[1023], tl = s_l[1023];
        __syncthreads();
        carry_c += tc; carry_q += tq; carry_l += tl;
    }
    if (tid == 0) {
        const int tterm = B.mins[0], tbad = B.mins[1];
        const bool fallback = (tterm >= B.ng) || (tbad <= tterm);     // mins start at 0x7F7F7F7F
        res->n_lines = carry_l;
        res->fallback = fallback ? 1 : 0;
        res->bad_group = tbad;
        res->n_bad = (int32_t)((unsigned int)B.mins[2] - 0x7F7F7F7Fu);
        res->approx_records = carry_c;
        res->n_declined = (int32_t)min(*B.dcnt, 0x7FFFFFFFu);
        res->bad_irregular = (tbad >= 0 && tbad < B.ng && (B.flags[tbad] & 5u)) ? 1 : 0;
        res->term_group = fallback ? -1 : tterm;
        res->end_offset = offset;
        res->has_final = 0;
        res->n_records = 0;
        res->n_qual_bytes = 0;
        if (!fallback) {
            res->n_records = B.part[(tterm / RES_BLOCK) * 4 + 0] + B.rloc[tterm] + B.cnt[tterm];
            res->n_qual_bytes = B.part[(tterm / RES_BLOCK) * 4 + 1] + B.qloc[tterm] + B.qb[tterm];
            const GroupTerm &tg = B.term[tterm];
            const int64_t ex = B.exit[tterm];
            const int st = tg.status;
            res->last_status = st;
            for (int i = 0; i < 6; i++) res->last_pos[i] = (tg.pos[i] >= 0) ? tg.pos[i] + add : -1;
            int end;
            if (ex == X_END_FINAL) { end = 0; res->has_final = 1; }
            else if (ex == Y_NOCAND) end = eof ? 0 : 1;
            else if (eof) end = (st == ST_QUAL_END) ? 2 : (st == ST_INVALID) ? 4 : 3;
            else end = (st == ST_INVALID) ? 4 : 1;
            res->end_state = end;
        }
    }
}

// ---- Phred decode of a batch of up to 64 records (one per lane) --------------------------------
// out[qo + b] = d[src + b] + qadd for b < len: arrayadd_b over each record's quality slice
// (/root/reference/src/_fastqandfurious.c:161-185; doc/user-guide.rst:130-141).
// Eight records at a time, one per 8-lane group, 2 x 16 bytes per lane and step; unaligned
// 16-byte loads/stores, SWAR byte add.
// =========================================================================
// k_expand: one wave per group.  Turns the staged group-relative tuples into the
// int64[n][6] rows (pos0..pos5 + add), rows written as whole 1 KiB lines through
// an LDS transpose; optional CSR offsets of the decoded qualities.
// Algorithmic traffic: 16 B read + 48 B (+ 8 B) written per record.
// =========================================================================
__global__ __launch_bounds__(64) void k_expand(ChainBufs B, const DevRes *__restrict__ res, int64_t add,
                                               int64_t *__restrict__ table, int64_t table_cap,
                                               int64_t *__restrict__ qoff, int64_t *__restrict__ qdir,
                                               int64_t qdir_cap, int64_t *__restrict__ p4s, int64_t p4_cap, int sshift, int in_place)
{
    __shared__ __attribute__((aligned(16))) int64_t s_rows[64 * 6];
    const int g = blockIdx.x, lane = threadIdx.x;
    if (res->fallback || g > res->term_group) return;
    const uint32_t cnt = B.cnt[g];
    if (cnt == 0) return;
    const int64_t r0 = B.part[(g / RES_BLOCK) * 4 + 0] + B.rloc[g];
    int64_t q0 = B.part[(g / RES_BLOCK) * 4 + 1] + B.qloc[g];
    const int own0 = g * OWN_T;
    const int64_t base = ((int64_t)((own0 > 0) ? own0 - 1 : 0) << TILE_SHIFT) + add;
    const int32_t sb = B.sbase[g];
    const StageRec *st = sb <= 0 ? B.stage + (int64_t)g * B.nmax : B.dstage + (int64_t)(sb - 1) * DCHUNK;
    const bool stage8 = (B.flags[g] & FLAG_STAGE8) != 0;              // (the lean kernel's 8-byte records: wave-uniform)
    const int64_t base8 = base + TILE + sshift + 1;
    for (uint32_t d0 = 0; d0 < cnt; d0 += 64) {
        const uint32_t dd = d0 + lane;
        const bool ok = dd < cnt;
        int64_t p0, p1, p3, p4;
        if (stage8) {
            uint2 w = make_uint2(0u, 0u);
            if (ok) w = reinterpret_cast<const uint2 *>(st)[dd];
            p0 = base8 + (w.x & 0xFFFFu); p1 = p0 + (w.x >> 16); p3 = p0 + (w.y & 0xFFFFu); p4 = p0 + (w.y >> 16);
        } else {
            StageRec r = {0, 0, 0, 0};
            if (ok) r = st[dd];
            p0 = base + r.p0; p1 = base + r.p1; p3 = base + r.p3; p4 = base + r.p4;
        }
        const int64_t p5 = p4 + p3 - p1 - 1;
        if (qoff && in_place) {
            // (the index pass decoded every byte at its own offset: a record's bytes start where pos4 lies in the buffer)
            if (ok && r0 + dd < table_cap) qoff[r0 + dd] = p4 - add - sshift;
        } else if (qoff) {
            const uint32_t ql = ok ? (uint32_t)(p5 - p4) : 0u;
            // quality lengths are < 2^31; chunk sums of 64 fit 64 bits via two 32-bit scans
            const uint32_t lo = wave_incl_scan(ql & 0xFFFFu), hi = wave_incl_scan(ql >> 16);
            const int64_t incl = ((int64_t)hi << 16) + (int64_t)lo;
            const int64_t myq = q0 + incl - ql;
            if (ok && r0 + dd < table_cap) { qoff[r0 + dd] = myq; qdir_mark(qdir, qdir_cap, myq, ql, r0 + dd); }
            if (ok && r0 + dd < p4_cap) p4s[r0 + dd] = p4;            // compact pos4 for the decode
            q0 += ((int64_t)__shfl((int)hi, 63) << 16) + (int64_t)(uint32_t)__shfl((int)lo, 63);
        }
        int64_t *mine = s_rows + lane * 6;
        mine[0] = p0; mine[1] = p1; mine[2] = p1 + 1; mine[3] = p3; mine[4] = p4; mine[5] = p5;
        __syncthreads();
        const uint32_t nrow = min(64u, cnt - d0);
        const int64_t rowbase = r0 + d0;
        // 48 B rows, 16-byte pieces: piece q of the chunk belongs to row q / 3
        const longlong2 *src = reinterpret_cast<const longlong2 *>(s_rows);
        longlong2 *dst = reinterpret_cast<longlong2 *>(table + rowbase * 6);
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const uint32_t q = lane + u * 64;
            if (q < nrow * 3 && rowbase + q / 3 < table_cap) dst[q] = src[q];
        }
        __syncthreads();
    }
}

}  // namespace ffq
