// vtx_sweep.hip — band_sweep_kernel: the band of bio 0.30.0's banded aligner for ANY task, eight lanes per task
// (the robust path behind band_diag_kernel / band_refine_kernel; round 5: no dp ring, 2.4 KB of LDS per task).
//
// What is computed: Band::create of banded::Aligner::local(read, haplotype) as the reference calls it
// (src/main.rs:898-901, K = 6, W = 20) — find_kmer_matches, sdpkpp, the anchor staircase with set_boundaries' lazy
// extension, the (2w + 1)-squares — restated in oracle/vtx_oracle.c (vtxo_band_create).  The output is the per-column row
// range [lo, hi) of every task, in the slot layout sw_banded_kernel reads; that kernel then scores the task exactly.
//
//   matches   Eq[c] = bit mask of the haplotype columns holding base c (five 256-bit masks; lane l of the task's eight
//             lanes owns word l = columns 32 l .. 32 l + 31, in registers).  The 6-mer matches of read row x are
//             M6(x) = AND_t Eq[x[x + t]] >> t — three funnel shifts per row over a sliding window (M2, M4, M6).
//             Bytes outside ACGTN, reads / haplotypes above 255 bases: the task is declined (the general kernel of
//             vtx_band.hip takes it).
//   sdpkpp    rows ascending; END events of row x (matches that started at row x - 6) enter C[ye] (LDS, ds_max_u32) and an
//             8-column block maximum with V << 16 | xq << 8 | yq (V = dp + xe + ye: the tuple order of the crate's max-tree);
//             START events: dp = max(6, prefixmax(y).V - (x + y) + 1, dp(x - 1, y - 1) + 1), the continuation wins ties.
//   sections  ROUND 5.  Along a run of continuing matches dp grows by exactly 1 per row: dp(x, y) = x + 6 - o with ONE deficit
//             o per SECTION — a run that starts at a match which does not continue its diagonal, or at one where a jump beats the
//             continuation.  Per DIAGONAL d = y - x the kernel keeps the latest section only, 16 bits (o, x0 = its first row):
//             512 diagonals = 1 KB per task instead of round 4's 2 KB of dp bytes (eight rows x 256 columns), and — what the
//             time was made of — a continuing match costs NOTHING at its START event (round 4: an LDS read of its partner's
//             dp and a write of its own, one trip of a serial per-lane loop per match).  It is looked at only when a jump
//             could beat it: lane-level test  pmax.V - 2x - 5 > G,  G = a lower bound of y - o over the lane's continuing
//             matches (carried from row to row: + 1 per row, the lower lane's bound for the match that crosses in).
//             The END event of the match that started at row xs reads dp = xs + 6 - o from its diagonal's entry.
//   stash     the one case the latest section does not cover: a jump beats the continuation at row r while matches of rows
//             r - 5 .. r - 1 of the OLD section have not ended.  Their (column, dp) go to a stash — eight buckets by end row,
//             seven entries each — and enter C when their row comes; the END loop skips a match whose row lies before its
//             diagonal's x0.  (Inside six rows a diagonal cannot break and resume, so x0 > xs means exactly this.)  A full
//             bucket declines the task (tandem repeats over two letters; 0 of 4 000 real-sequence tasks).
//   chain     a match that opens a section is logged (x, y, source) — in GLOBAL memory (a 4 KB slice per resident task, L2-
//             resident; LDS holds none of it: round 4's 1 KB log and its second pass over the log overflows are gone).  From
//             the best (dp, x, y): the section is the log entry of this diagonal with the largest x' <= x; go on from its source.
//   band      the staircase's anchors (lazy extension, sections, gaps) -> first / last anchor row per column (ds_min / ds_max)
//             -> lo / hi in closed form (vtx_band.hip's header).
// tests/sweepmodel/sweep_model.cpp (vtxs_band2) restates exactly this on the CPU; tests/test_sweep_model.py checks it against
// the oracle; tests/test_gpu_sweep.py checks the device's bands column by column against both.
//
// Machine mapping: one wavefront per workgroup = 8 tasks x 8 lanes, 612 words of LDS per task = 19.1 KB per wavefront: EIGHT
// wavefronts per CU (two per SIMD; round 4: 38.5 KB, one per SIMD — nothing hid a wave's LDS / SALU latency).  A persistent
// grid (a workgroup takes task groups blockIdx.x, + gridDim.x, ...) so that the log slices are per resident workgroup.
// Integer / LDS work, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "vtx_device.h"
#include "../../include/vtx_band_semantics.h"

namespace {

constexpr int K = VTX_REF_K, W = VTX_REF_W;
static_assert(K == 6 && W == 20 && VTX_REF_MATCH == 1 && VTX_REF_GAP_OPEN == -5 && VTX_REF_GAP_EXTEND == -1,
              "band_sweep_kernel's sdpkpp is written for k = 6, match 1, gap -5 / -1 (src/main.rs:33-38, :899)");
constexpr int SECCAP = 28;           // sections of the best chain (a 150-base read chains at most 25 six-mers end to end)
constexpr int MAXLEN = 255;          // read / haplotype bases (one byte per coordinate in the packed words)
constexpr int LOGCAP = 1024;         // sections a task may open (global memory), 128 per lane: lane l logs the sections that open in its
                                     // columns at entries l, l + 8, l + 16, ... with a counter of its own — no atomic, no wait (real
                                     // sequence: p99 55 per task; satellites: hundreds)
constexpr int STASH = 7;             // entries per stash bucket
constexpr int GRID_MAX = 1536;       // workgroups of the persistent grid: 256 CUs x 6.  Eight fit a CU's LDS (tools/residency_census.hip), but the
                                     // kernel saturates before: 100 k real-sequence loci take 998 / 525 / 284 / 195 / 208 ms with 256 / 512 / 1 024 /
                                     // 1 536 / 2 048 workgroups (profiles/r05_sweep_grid.txt) — and 6 x 19.1 KB leave LDS for other kernels' workgroups
constexpr int G_INF = 0x3fffffff;
// per-task LDS (32-bit words)
constexpr int O_OFF = 0;             // 512 diagonals x 16 bits: x0 << 8 | o; after the sweep: rmin[256]
constexpr int O_C = 256;             // C[ye], 256 words; after the sweep: rmax[256]
constexpr int O_BM = 512;            // block maxima of C, 8 columns each (32 words); after the sweep: the chain's sections
constexpr int O_STASH = 544;         // 8 buckets x (count, 7 entries: y << 8 | dp)
constexpr int O_MISC = 608;          // (spare)
constexpr int TASK_W = 612;          // = 4 (mod 32): the eight tasks of a wavefront start in eight different banks; 16-byte aligned
static_assert(TASK_W % 32 == 4 && TASK_W % 4 == 0 && O_MISC + 4 <= TASK_W && SECCAP <= 32, "LDS layout");

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP moves are never wrapped in a select: `cond ? dpp(x) : 0` compiles to an EXEC-masked DPP, and a source lane that EXEC
// disables reads as 0.  Masks are ANDed in.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {      // lanes without a source get 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_QUAD_XOR1 0xB1           // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E           // quad_perm:[2,3,0,1]
#define DPP_HALF_MIRROR 0x141        // row_half_mirror: lane i <-> 7 - i inside each group of eight
// maximum over the eight lanes of a task (every lane active)
__device__ __forceinline__ uint32_t group_max(uint32_t v) {
    v = max(v, dpp0<DPP_QUAD_XOR1>(v));
    v = max(v, dpp0<DPP_QUAD_XOR2>(v));
    return max(v, dpp0<DPP_HALF_MIRROR>(v));
}

// A: 0, C: 1, G: 2, T: 3, N: 4, anything else: 7
__device__ __forceinline__ uint32_t base_code(uint32_t b) {
    // 3 bits per entry, entry = b - 'A' (0 .. 20; 20 = everything else)
    constexpr uint64_t ALL7 = 0x7fffffffffffffffull;
    constexpr uint64_t HOLES = (7ull << (3 * 0)) | (7ull << (3 * 2)) | (7ull << (3 * 6)) | (7ull << (3 * 19)) | (7ull << (3 * 13));
    constexpr uint64_t LUT = (ALL7 & ~HOLES) | (0ull << (3 * 0)) | (1ull << (3 * 2)) | (2ull << (3 * 6)) | (3ull << (3 * 19)) | (4ull << (3 * 13));
    uint32_t i = b - 'A';
    i = i > 20u ? 20u : i;
    return (uint32_t)(LUT >> (3u * i)) & 7u;
}

}  // namespace

// tasks[n_tasks]: task = 2 * record + haplotype.  Every task either gets band slot h = atomicAdd(counters[0]) (hard_list[h] =
// task, lo at band + h * 2 * band_stride, hi at + band_stride) or, when declined, goes to overflow_list[atomicAdd(counters[1])].
// stat_counters != nullptr: stat_counters[reason] counts the declined tasks (1 bytes / lengths, 2 log full, 3 sections, 5 stash).
// n_dev != nullptr: the list length lives on the device (min(*n_dev, n_tasks); the grid is sized for n_tasks).
// glog: gridDim.x * 8 * LOGCAP words (vtxk_band_sweep_log_bytes).
__global__ __launch_bounds__(64) void band_sweep_kernel(
    const uint32_t* __restrict__ tasks, uint32_t n_tasks, const uint32_t* __restrict__ n_dev,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    uint16_t* __restrict__ band, uint32_t band_stride, uint32_t* __restrict__ hard_list,
    uint32_t* __restrict__ overflow_list, uint32_t* __restrict__ counters, uint32_t stats, uint32_t* __restrict__ stat_counters,
    uint8_t* __restrict__ stage, uint32_t* __restrict__ dbg, uint32_t* __restrict__ glog) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (n_dev) { const uint32_t nd = *n_dev; n_tasks = nd < n_tasks ? nd : n_tasks; }
    const int lane = (int)threadIdx.x;
    const int g = lane >> 3, l = lane & 7;
    uint32_t* T = lds + g * TASK_W;
    uint16_t* OFF = (uint16_t*)(T + O_OFF);
    uint32_t* Cw = T + O_C;
    uint32_t* BM = T + O_BM;
    uint32_t* ST = T + O_STASH;
    uint32_t* SEC = T + O_BM;
    uint32_t* LOG = glog + ((size_t)blockIdx.x * 8u + (uint32_t)g) * (size_t)LOGCAP;
    const int col0 = 32 * l;
    const uint32_t nbmask = l == 7 ? 0u : 0xffffffffu;            // lane 7's upper neighbour belongs to the next task
    const uint32_t ge1 = l >= 1 ? 0xffffffffu : 0u, ge2 = l >= 2 ? 0xffffffffu : 0u, ge4 = l >= 4 ? 0xffffffffu : 0u;
    const uint32_t ablate = VTX_ABLATE(stats >> 8);               // (profiling aid, libvtx_dev.so only; results are wrong by design)

#pragma unroll 1
    for (uint32_t grp = blockIdx.x; grp * 8u < n_tasks; grp += gridDim.x) {
    wave_sync();                                                  // (the previous group's LDS reads are done)
    const uint32_t slot = grp * 8u + (uint32_t)g;
    const bool have = slot < n_tasks;
    uint32_t task = 0, roff = 0, hoff = 0;
    int m = 0, n = 0;
    if (have) {
        task = tasks[slot];
        const uint32_t rid = task >> 1, hap = task & 1u;
        const vtx_record rec = records[rid];
        const vtx_locus loc = loci[rec_locus[rid]];
        m = (int)rec.read_len; roff = rec.read_off;
        n = (int)(hap ? loc.alt_len : loc.ref_len);
        hoff = hap ? loc.alt_off : loc.ref_off;
    }
    uint32_t decline = (have && (m > MAXLEN || n > MAXLEN)) ? 1u : 0u;
    if (decline) { m = 0; n = 0; }

    // ---- set-up: zero C / BM / stash counts / log count; read codes and Eq words to registers ----
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4* c4 = (uint4*)(Cw + col0);
#pragma unroll
        for (int i = 0; i < 8; ++i) c4[i] = z;
        *(uint4*)(BM + 4 * l) = z;
        ST[8 * l] = 0;
    }
    uint32_t bad = 0;
    uint32_t pk0 = 0, pk1 = 0, pk2 = 0, pk3 = 0;                  // codes of read rows 32 l .. 32 l + 31, a nibble each (7: no base)
    {
        // read bytes [32 l, 32 l + 32): two 16-byte loads (the arena is padded by 16 bytes)
        uint32_t wds[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (col0 + 16 * h < m) __builtin_memcpy(&v, read_arena + roff + col0 + 16 * h, 16);
            wds[4 * h] = v.x; wds[4 * h + 1] = v.y; wds[4 * h + 2] = v.z; wds[4 * h + 3] = v.w;
        }
        uint32_t packed[4] = {0, 0, 0, 0};
#pragma unroll
        for (int wi = 0; wi < 8; ++wi) {
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
                const int k = 4 * wi + bi;
                uint32_t c = 7u;
                if (col0 + k < m) {
                    c = base_code((wds[wi] >> (8 * bi)) & 0xffu);
                    bad |= (c == 7u) ? 1u : 0u;
                }
                packed[k >> 3] |= c << (4 * (k & 7));
            }
        }
        pk0 = packed[0]; pk1 = packed[1]; pk2 = packed[2]; pk3 = packed[3];
    }
    uint32_t eq0 = 0, eq1 = 0, eq2 = 0, eq3 = 0, eq4 = 0;
    {
        uint32_t wds[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (col0 + 16 * h < n) __builtin_memcpy(&v, hap_arena + hoff + col0 + 16 * h, 16);
            wds[4 * h] = v.x; wds[4 * h + 1] = v.y; wds[4 * h + 2] = v.z; wds[4 * h + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (col0 + j < n) {
                const uint32_t c = base_code((wds[j >> 2] >> (8 * (j & 3))) & 0xffu);
                const uint32_t bit = 1u << j;
                eq0 |= c == 0u ? bit : 0u; eq1 |= c == 1u ? bit : 0u; eq2 |= c == 2u ? bit : 0u;
                eq3 |= c == 3u ? bit : 0u; eq4 |= c == 4u ? bit : 0u;
                bad |= (c == 7u) ? 1u : 0u;
            }
        }
    }
    // a byte outside ACGTN anywhere in the task: declined (all eight lanes agree)
    bad = group_max(bad);
    if (bad) { decline = 1u; eq0 = eq1 = eq2 = eq3 = eq4 = 0; }       // (no matches: the sweep idles for this task)
    wave_sync();

    // ---- the sweep ----
    int mmax = m;
    mmax = max(mmax, __shfl_xor(mmax, 8)); mmax = max(mmax, __shfl_xor(mmax, 16)); mmax = max(mmax, __shfl_xor(mmax, 32));
    const int tmax = __builtin_amdgcn_readfirstlane(mmax) + K;    // feed step t = 0 .. m + 5: row r = t - 5 reaches m
    uint32_t m1p = 0;                                             // M1(t - 1)
    uint32_t m2a = 0, m2b = 0, m2c = 0, m2d = 0;                  // M2(t - 2) .. M2(t - 5)
    uint32_t h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0;      // M6(r - 1) .. M6(r - 6)
    uint32_t best = 0;
    uint32_t pmax = 0;                                            // see the END phase
    int G = G_INF;                                                // lower bound of y - o over this lane's matches of the previous row
    int stash_until = -1;                                         // the task's stash may hold entries for END rows <= this
    uint32_t stash_full = 0;
    uint32_t nlog = 0;                                            // sections this lane has logged: entry k at LOG[8 k + l]
    uint32_t lg0 = 0, lg1 = 0, lg2 = 0, lg3 = 0;                  // ... the first four stay in registers as well (the chain walk reads them there)
    uint32_t pre_e = 0xff00u;                                     // OFF entry of this lane's first END event of the NEXT row, requested a row ahead
    // the codes of eight rows per word, fetched from the lane that holds them (word j = rows 8 j .. 8 j + 7: register j & 3 of lane
    // j >> 2; selected by VALUE with the uniform j — a select between the variables themselves sends them through scratch memory)
#define FETCH_CODES(dst, jexpr)                                                                                  \
    {                                                                                                            \
        const int j_ = (jexpr);                                                                                  \
        uint32_t mine_ = pk0;                                                                                    \
        mine_ = (j_ & 3) == 1 ? pk1 : mine_; mine_ = (j_ & 3) == 2 ? pk2 : mine_; mine_ = (j_ & 3) == 3 ? pk3 : mine_;   \
        const uint32_t v_ = (uint32_t)__shfl((int)mine_, (lane & ~7) | ((j_ >> 2) & 7));                          \
        dst = j_ < 32 ? v_ : 0x77777777u;                                                                        \
    }
    uint32_t code_blk = 0, code_blk_next;
    FETCH_CODES(code_blk_next, 0)
#define LDS_DONE() __builtin_amdgcn_s_waitcnt(0xC07F)          /* s_waitcnt lgkmcnt(0) (gfx9 encoding: vmcnt / expcnt untouched) */
#define SHR_WORDS(v, s) __builtin_amdgcn_alignbit(dpp0<DPP_ROW_SHL(1)>(v) & nbmask, (v), (s))
    // one END event: the match that started at (r - 6, y) with dp enters C / BM; val_ = the inserted value
#define END_EVENT(y, dp, val_)                                                                        \
    {                                                                                                 \
        const uint32_t key_ = ((uint32_t)(dp) << 16) | ((uint32_t)(r - K) << 8) | (uint32_t)(y);      \
        best = max(best, key_);                                                                       \
        val_ = key_ + ((uint32_t)(r + K + (y)) << 16);   /* V = dp + xe + ye = dp + r + (y + 6) */     \
        atomicMax(&Cw[(y) + K], val_);                                                                \
        atomicMax(&BM[((y) + K) >> 3], val_);                                                         \
    }
#pragma unroll 1
    for (int t = 0; t < (ablate == 1 ? 0 : tmax); ++t) {
        if ((t & 7) == 0) { code_blk = code_blk_next; FETCH_CODES(code_blk_next, (t >> 3) + 1) }
        const uint32_t code = (code_blk >> (4 * (t & 7))) & 7u;
        // M1(t): the Eq word of this row's base
        uint32_t m1 = (code & 1u) ? ((code & 2u) ? eq3 : eq1) : ((code & 2u) ? eq2 : eq0);
        m1 = (code & 4u) ? (code == 4u ? eq4 : 0u) : m1;
        const uint32_t m2n = m1p & SHR_WORDS(m1, 1);              // M2(t - 1)
        const uint32_t m4 = m2d & SHR_WORDS(m2b, 2);              // M4(t - 5) = M2(t - 5) & M2(t - 3) >> 2
        const uint32_t w_start = m4 & SHR_WORDS(m2n, 4);          // M6(t - 5)
        m1p = m1; m2d = m2c; m2c = m2b; m2b = m2a; m2a = m2n;
        const int r = t - (K - 1);
        const uint32_t w_end = h6, w_prev = h1;                   // M6(r - 6), M6(r - 1)
        h6 = h5; h5 = h4; h4 = h3; h3 = h2; h2 = h1; h1 = w_start;
        if (r < 0) continue;                                      // (uniform)
        if (ablate == 2) { best |= w_start | w_end; continue; }   // (profiling aid) the match pipeline only
        const bool drain = r <= stash_until;
        // the OFF entry of this lane's first END event of the NEXT row (h6 = M6(r - 5) after the shift above), requested a row ahead
#define PREFETCH_END() pre_e = OFF[col0 + (h6 ? (int)__builtin_ctz(h6) : 0) - (r + 1 - K) + 256]
        if (!__any((w_start | w_end) != 0u || drain)) { G = G_INF; PREFETCH_END(); continue; }
        // ---- END events of row r: matches that started at row xs = r - 6; dp = xs + 6 - o of their diagonal's section ----
        if (__any(w_end != 0u)) {
            uint32_t ins_a = 0, ins_b = 0;                        // inserted into this lane's own 32 columns / into the next lane's
            uint32_t w = w_end;
            const int xs = r - K;
            bool first = true;
            while (__any(w != 0u)) {
                const bool on = w != 0u;
                const int y = col0 + (on ? (int)__builtin_ctz(w) : 0);
                w &= w - 1u;                                      // (0 stays 0)
                // x0 << 8 | o of the match's diagonal.  The lane's first match: requested at the end of the previous row (an entry
                // rewritten since — a jump beat the continuation in that row's START phase — carries x0 = r - 1 > xs: the old entry
                // then inserts what the stash inserts again, the same value twice)
                const uint32_t e = first ? pre_e : (uint32_t)OFF[y - xs + 256];
                first = false;
                if (on && xs >= (int)(e >> 8)) {                  // (x0 > xs: a match of the diagonal's previous section — its END is in the stash)
                    const uint32_t dp = (uint32_t)r - (e & 0xffu);
                    uint32_t val;
                    END_EVENT(y, dp, val)
                    if (((y + K) >> 5) == l) ins_a = max(ins_a, val); else ins_b = max(ins_b, val);
                }
            }
            // pmax = the largest value inserted so far at an end column below 32 (l + 1): an upper bound of every prefix maximum
            // this lane can ask for
            uint32_t sc = max(ins_a, dpp0<DPP_ROW_SHR(1)>(ins_b) & ge1);
            sc = max(sc, dpp0<DPP_ROW_SHR(1)>(sc) & ge1);
            sc = max(sc, dpp0<DPP_ROW_SHR(2)>(sc) & ge2);
            sc = max(sc, dpp0<DPP_ROW_SHR(4)>(sc) & ge4);
            pmax = max(pmax, sc);
        }
        if (__any(drain)) {                                       // stashed END events of this row (rare: a jump beat a continuation)
            wave_sync();
            const uint32_t* bk = ST + 8 * (r & 7);
            const uint32_t cnt = drain ? min(bk[0], (uint32_t)STASH) : 0u;     // (a bucket that overflowed has counted on: the task is declined below)
            uint32_t val = 0;
            if ((uint32_t)l < cnt) {
                const uint32_t e = bk[1 + l];
                const int y = (int)(e >> 8);
                END_EVENT(y, e & 0xffu, val)
            }
            pmax = max(pmax, group_max(val));                     // (whoever inserted it: every lane's bound takes it)
            wave_sync();
            if (drain && l == 0) ST[8 * (r & 7)] = 0;
        }
        wave_sync();
        if (ablate == 3) { G = G_INF; PREFETCH_END(); continue; } // (profiling aid) ... + the END events
        // ---- START events of row r ----
        int Gn = G_INF;
        if (__any(w_start != 0u)) {
            // is (r - 1, y - 1) a match?  bit b of the previous row's mask shifted up by one column (bit 31 of the lane below comes in at bit 0)
            const uint32_t left_prev = dpp0<DPP_ROW_SHR(1)>(w_prev) & ge1;
            const uint32_t pshift = __builtin_amdgcn_alignbit(w_prev, left_prev, 31);
            const uint32_t w_cont = w_start & pshift, w_new = w_start & ~pshift;
            // continuing matches: could a jump beat one of them?  y - o of a continuing match = its partner's + 1
            const int Glow = (int)(dpp0<DPP_ROW_SHR(1)>((uint32_t)G) & ge1) | (int)(~ge1 & (uint32_t)G_INF);
            int Gc = (w_cont & ~1u) ? G + 1 : G_INF;
            Gc = (w_cont & 1u) ? min(Gc, Glow + 1) : Gc;
            const int thr = (int)(pmax >> 16) - 2 * r - 5;        // a jump from q beats the continuation iff q.V - 2 r - 5 > y - o
            const bool chk = w_cont != 0u && pmax != 0u && thr > Gc;
            uint32_t todo = w_new | (chk ? w_cont : 0u);
            Gn = chk ? G_INF : Gc;                                // (a lane that looks at its continuing matches gets the exact minimum back)
            // the exclusive prefix maxima of the block maxima at this lane's four blocks (BM cannot change during the START phase):
            // once per row, requested together with the first trip's reads
            bool pbm_ready = false;
            uint32_t pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0;
            bool jumped = false;
            const bool qrow = __any(todo != 0u && pmax != 0u);        // somebody may have to ask for a prefix maximum in this row
            while (__any(todo != 0u)) {
                const bool on = todo != 0u;
                const int b = on ? (int)__builtin_ctz(todo) : 0;
                todo &= todo - 1u;
                const int y = col0 + b;
                const bool is_new = (w_new >> b) & 1u;
                const int di = y - r + 256;
                // ONE round trip to the LDS per trip: the diagonal's entry, the match's block of C, (first trip) the block maxima
                const uint32_t eo = OFF[di];
                const int blk = y >> 3, kk = y & 7, bi = blk & 3;
                uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0, bm = c0;
                if (qrow) {
                    c0 = *(const uint4*)(Cw + 8 * blk); c1 = *(const uint4*)(Cw + 8 * blk + 4);
                    if (!pbm_ready) bm = *(const uint4*)(BM + 4 * l);
                }
                LDS_DONE();
                if (qrow && !pbm_ready) {
                    const uint32_t p0 = bm.x, p1 = max(p0, bm.y), p2 = max(p1, bm.z), p3 = max(p2, bm.w);
                    uint32_t inc = p3;
                    inc = max(inc, dpp0<DPP_ROW_SHR(1)>(inc) & ge1);
                    inc = max(inc, dpp0<DPP_ROW_SHR(2)>(inc) & ge2);
                    inc = max(inc, dpp0<DPP_ROW_SHR(4)>(inc) & ge4);
                    const uint32_t exc = dpp0<DPP_ROW_SHR(1)>(inc) & ge1;
                    pb0 = exc; pb1 = max(exc, p0); pb2 = max(exc, p1); pb3 = max(exc, p2);
                    pbm_ready = true;
                }
                const int o_old = (int)(eo & 0xffu), x0_old = (int)(eo >> 8);
                int gy = y - o_old;
                const bool need_q = on && pmax != 0u && (is_new ? ((int)(pmax >> 16) - (r + y) + 1 >= K) : (thr > gy));
                uint32_t q = 0;
                if (need_q) {
                    q = (bi & 2) ? ((bi & 1) ? pb3 : pb2) : ((bi & 1) ? pb1 : pb0);
                    q = max(q, c0.x);
                    q = max(q, kk >= 1 ? c0.y : 0u); q = max(q, kk >= 2 ? c0.z : 0u); q = max(q, kk >= 3 ? c0.w : 0u);
                    q = max(q, kk >= 4 ? c1.x : 0u); q = max(q, kk >= 5 ? c1.y : 0u); q = max(q, kk >= 6 ? c1.z : 0u);
                    q = max(q, kk >= 7 ? c1.w : 0u);
                }
                if (on) {
                    const int cand = q ? (int)(q >> 16) - (r + y) + 1 : 0;
                    bool open = false;
                    int dv = K;
                    uint32_t src = 0xffffu;
                    if (is_new) {
                        open = true;
                        if (cand >= K) { dv = cand; src = q & 0xffffu; }
                    } else if (q && (int)(q >> 16) - 2 * r - 5 > gy) {
                        // a jump beats the continuation: the old section's matches of rows r - 5 .. r - 1 have not ended yet
                        open = true; dv = cand; src = q & 0xffffu; jumped = true;
                        for (int xp = max(x0_old, r - (K - 1)); xp <= r - 1; ++xp) {
                            uint32_t* bk = ST + 8 * ((xp + K) & 7);
                            const uint32_t pos = atomicAdd(&bk[0], 1u);
                            if (pos < (uint32_t)STASH) bk[1 + pos] = ((uint32_t)(y - (r - xp)) << 8) | (uint32_t)(xp + K - o_old);
                            else stash_full = 1u;
                        }
                    }
                    if (open) {
                        const int o_new = r + K - dv;
                        OFF[di] = (uint16_t)(((uint32_t)r << 8) | (uint32_t)o_new);
                        gy = y - o_new;
                        const uint32_t ent = ((uint32_t)r << 24) | ((uint32_t)y << 16) | src;
                        if (nlog < (uint32_t)(LOGCAP / 8)) LOG[8u * nlog + (uint32_t)l] = ent;
                        lg0 = nlog == 0u ? ent : lg0; lg1 = nlog == 1u ? ent : lg1; lg2 = nlog == 2u ? ent : lg2; lg3 = nlog == 3u ? ent : lg3;
                        ++nlog;
                    }
                    Gn = min(Gn, gy);
                }
            }
            if (__any(jumped)) stash_until = max(stash_until, (int)group_max(jumped ? (uint32_t)(r + K - 1) : 0u));
        }
        G = w_start ? Gn : G_INF;
        PREFETCH_END();                                          // (the wait is the barrier's)
        wave_sync();
    }
#undef SHR_WORDS
#undef END_EVENT
#undef LDS_DONE
#undef FETCH_CODES
#undef PREFETCH_END
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");            // the log lives in global memory: written by any lane, read by all eight
    wave_sync();
    best = group_max(best);
    stash_full = group_max(stash_full);
    const uint32_t nlog_max = group_max(nlog);
    if (!decline && nlog_max > (uint32_t)(LOGCAP / 8)) decline = 2u;
    if (!decline && stash_full) decline = 5u;
    const bool seeded = best != 0u && !decline && ablate != 4;     // (ablate 4: profiling aid — the sweep without the chain walk and the band)
    auto log_at = [&](uint32_t i) -> uint32_t { return __hip_atomic_load(&LOG[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    // ---- the chain's sections, last first ----
    // (log entry i = 8 k + l is lane l's k-th; every lane scans its own — the first four from its registers: a walk over a task
    // whose lanes logged four sections or fewer each touches no memory at all)
    int nsec = 0;
    {
        int cx = (int)((best >> 8) & 0xffu), cy = (int)(best & 0xffu);
        bool walking = seeded;
        while (__any(walking)) {
            const int d = cy - cx;
            uint32_t pick = 0;                                      // (x' + 1) << 12 | log index, maximum over the diagonal's entries with x' <= x
            if (walking) {
                auto look = [&](uint32_t e, uint32_t k) {
                    const int ex = (int)(e >> 24), ey = (int)((e >> 16) & 0xffu);
                    if (k < nlog && ey - ex == d && ex <= cx) pick = max(pick, ((uint32_t)(ex + 1) << 12) | (8u * k + (uint32_t)l));
                };
                look(lg0, 0u); look(lg1, 1u); look(lg2, 2u); look(lg3, 3u);
                for (uint32_t k = 4u; k < nlog; ++k) look(log_at(8u * k + (uint32_t)l), k);
            }
            pick = group_max(pick);
            // the picked entry: from its owner's registers (index < 32) or from memory
            const uint32_t pidx = pick & 0xfffu;
            const uint32_t psel = (pidx & 16u) ? ((pidx & 8u) ? lg3 : lg2) : ((pidx & 8u) ? lg1 : lg0);
            const uint32_t preg = (uint32_t)__shfl((int)psel, (lane & ~7) | (int)(pidx & 7u));
            if (walking) {
                if (pick == 0u || nsec >= SECCAP) { decline = pick == 0u ? 4u : 3u; walking = false; }
                else {
                    const uint32_t e = pidx < 32u ? preg : log_at(pidx);
                    const int ex = (int)(e >> 24), ey = (int)((e >> 16) & 0xffu);
                    if (l == 0) SEC[nsec] = ((uint32_t)ex << 16) | ((uint32_t)ey << 8) | (uint32_t)(cx - ex + 1);
                    ++nsec;
                    const uint32_t src = e & 0xffffu;
                    if (src == 0xffffu) walking = false;
                    else { cx = (int)(src >> 8); cy = (int)(src & 0xffu); }
                }
            }
        }
    }
    wave_sync();
    // ---- anchors -> first / last anchor row per column (the section store and C are dead: rmin / rmax take their place) ----
    uint32_t* rmin = T + O_OFF;
    uint32_t* rmax = T + O_C;
    {
        uint4* a = (uint4*)(rmin + col0);
        uint4* b4 = (uint4*)(rmax + col0);
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu); b4[i] = make_uint4(0, 0, 0, 0); }
    }
    wave_sync();
    int cA = 0, cB = -1;
    const bool banded = seeded && !decline;
    {
        constexpr int LAZY = VTX_BAND_LAZY_EXT(K) > 255 ? 255 : VTX_BAND_LAZY_EXT(K);
        constexpr int LAST = VTX_BAND_KMER_LAST_ANCHOR(K);
        // a diagonal run of cnt cells from (r0, c0); a horizontal run (dr = 0)
        auto run = [&](int r0, int c0, int cnt, int dr) {
            for (int i = l; i < cnt; i += 8) {
                const uint32_t rr = (uint32_t)(r0 + dr * i);
                atomicMin(&rmin[c0 + i], rr);
                atomicMax(&rmax[c0 + i], rr);
            }
        };
        int nloop = banded ? nsec : 0;
        int smax = nloop;
        smax = max(smax, __shfl_xor(smax, 8)); smax = max(smax, __shfl_xor(smax, 16)); smax = max(smax, __shfl_xor(smax, 32));
        int pr = 0, pc = 0;
        for (int s = 0; s < smax; ++s) {                          // sections in chain order = SEC[nsec - 1 - s]
            if (s >= nloop) continue;
            const uint32_t e = SEC[nsec - 1 - s];
            const int x0 = (int)(e >> 16), y0 = (int)((e >> 8) & 0xffu), len = (int)(e & 0xffu);
            if (s == 0) {
                const int d0 = min(min(x0, y0), LAZY);
                run(x0 - d0, y0 - d0, d0 + 1, 1);
                cA = y0 - d0;
            } else {
                const int dr = x0 - pr, dc = y0 - pc, dg = min(dr, dc);
                run(pr, pc, dg + 1, 1);
                if (dr > dc) {
                    if (l == 0) { atomicMin(&rmin[pc + dg], (uint32_t)(pr + dg)); atomicMax(&rmax[pc + dg], (uint32_t)x0); }
                } else {
                    run(pr + dg, pc + dg, dc - dg + 1, 0);
                }
            }
            run(x0, y0, len + LAST, 1);                            // anchors 0 .. len - 1 + LAST
            pr = x0 + len - 1 + K; pc = y0 + len - 1 + K;
            if (s == nloop - 1) {
                if (LAST < K && l == 0) { atomicMin(&rmin[pc], (uint32_t)pr); atomicMax(&rmax[pc], (uint32_t)pr); }   // add_gap's origin
                const int d1 = min(min(m - pr, n - pc), LAZY);
                run(pr, pc, d1 + 1, 1);
                cB = pc + d1;
            }
        }
    }
    wave_sync();
    if (dbg && have && l == 0) {                                   // (vtx_debug_bands with VTX_SWEEP_DBG=1: what the task's lanes agreed on)
        uint32_t* o = dbg + (size_t)slot * 64;
        o[0] = best; o[1] = nlog_max; o[2] = (uint32_t)nsec; o[3] = (uint32_t)cA; o[4] = (uint32_t)cB; o[5] = decline; o[6] = (uint32_t)m; o[7] = (uint32_t)n;
        for (int i = 0; i < 12; ++i) o[8 + i] = SEC[i];
        for (int i = 0; i < 24; ++i) o[20 + i] = log_at((uint32_t)i);       // (entry 8 k + lane: slots a lane did not write hold an earlier task's)
        for (int i = 0; i < 10; ++i) { o[44 + i] = rmin[i]; o[54 + i] = rmax[i]; }
    }
    // ---- slots and ranges ----
    const bool emit = have && !decline;
    const uint64_t em = __ballot(emit && l == 0), dm = __ballot(have && decline && l == 0);
    uint32_t hbase = 0, obase = 0;
    if (lane == 0) {
        if (em) hbase = atomicAdd(&counters[0], (uint32_t)__popcll(em));
        if (dm) obase = atomicAdd(&counters[1], (uint32_t)__popcll(dm));
    }
    hbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)hbase);
    obase = (uint32_t)__builtin_amdgcn_readfirstlane((int)obase);
    const uint64_t below = (1ull << (lane & ~7)) - 1ull;           // the l == 0 lanes of the tasks before this one
    if (emit) {
        const uint32_t h = hbase + (uint32_t)__popcll(em & below);
        if (l == 0) { hard_list[h] = task; if (stage) stage[task] = 4; }
        uint16_t* lo = band + (size_t)h * 2u * band_stride;
        uint16_t* hi = lo + band_stride;
        const int rows = m + 1;
        for (int j = l; j <= n; j += 8) {
            uint32_t lov = 0x7fffu, hiv = 0;
            if (!seeded) {                                         // no k-mer match at all: Band::full_matrix
                if (VTX_BAND_NO_SEED_FULL_MATRIX) { lov = 0; hiv = (uint32_t)rows; }
            } else if (j >= cA - W && j <= cB + W) {
                const int c0 = max(j - W, cA), c1 = min(j + W, cB);
                lov = (uint32_t)max((int)rmin[c0] - W, 0);
                hiv = (uint32_t)min((int)rmax[c1] + W + 1, rows);
            }
            lo[j] = (uint16_t)lov; hi[j] = (uint16_t)hiv;
        }
    } else if (have && l == 0) {
        overflow_list[obase + (uint32_t)__popcll(dm & below)] = task;
        if (stat_counters) atomicAdd(&stat_counters[min(decline, 7u)], 1u);
    }
    }   // task groups
}

extern "C" uint32_t vtxk_band_sweep_grid(uint32_t n_tasks) {
    static const uint32_t cap = VTX_DEV_ENV("VTX_SWEEP_GRID") ? (uint32_t)std::min(std::max(atoi(VTX_DEV_ENV("VTX_SWEEP_GRID")), 1), GRID_MAX) : (uint32_t)GRID_MAX;   // experiment knob (libvtx_dev.so)
    return (uint32_t)std::min<uint64_t>(((uint64_t)n_tasks + 7) / 8, (uint64_t)cap);
}
// the log slices of the largest grid (one buffer for the life of the context)
extern "C" size_t vtxk_band_sweep_log_bytes(void) { return (size_t)GRID_MAX * 8 * LOGCAP * sizeof(uint32_t); }

extern "C" hipError_t vtxk_launch_band_sweep(const uint32_t* tasks, uint32_t n_tasks, const uint32_t* n_dev,
                                             const vtx_record* records,
                                             const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                             const uint8_t* hap_arena, uint16_t* band, uint32_t band_stride, uint32_t* hard_list,
                                             uint32_t* overflow_list, uint32_t* counters, uint32_t* stat_counters, uint8_t* stage,
                                             uint32_t* dbg, uint32_t* glog, hipStream_t s) {
    if (!n_tasks) return hipSuccess;
    static const uint32_t ablate = VTX_DEV_ENV("VTX_SWEEP_ABLATE") ? (uint32_t)atoi(VTX_DEV_ENV("VTX_SWEEP_ABLATE")) << 8 : 0u;
    const size_t shmem = (size_t)8 * TASK_W * sizeof(uint32_t);
    hipLaunchKernelGGL(band_sweep_kernel, dim3(vtxk_band_sweep_grid(n_tasks)), dim3(64), shmem, s, tasks, n_tasks, n_dev, records,
                       rec_locus, loci, read_arena, hap_arena, band, band_stride, hard_list, overflow_list, counters,
                       ablate, stat_counters, stage, dbg, glog);
    return hipGetLastError();
}
// what the kernel holds: reads and haplotypes up to this many bases (longer ones are declined task by task; a batch whose
// haplotypes are all longer should not be sent here at all)
extern "C" uint32_t vtxk_band_sweep_max_len(void) { return (uint32_t)MAXLEN; }
