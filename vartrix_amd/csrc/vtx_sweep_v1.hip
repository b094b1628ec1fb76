// vtx_sweep_v1.hip — round 4's band_sweep_kernel (dp bytes of eight rows in LDS, 4.8 KB per task, one wavefront per SIMD), kept as
// the REFERENCE the round-5 kernel (vtx_sweep.hip) is compared with: linked into libvtx_dev.so only, selected there with
// VTX_SWEEP_V1=1 (tests/test_gpu_sweep.py: identical bands and scores; tools/gpu_campaign.sh: A/B timing).
//
// What is computed: Band::create of banded::Aligner::local(read, haplotype) as the reference calls it
// (src/main.rs:898-901, K = 6, W = 20) — find_kmer_matches, sdpkpp, the anchor staircase with set_boundaries' lazy
// extension, the (2w + 1)-squares — restated in oracle/vtx_oracle.c (vtxo_band_create).  The output is the per-column row
// range [lo, hi) of every task, in the slot layout sw_banded_kernel reads; that kernel then scores the task exactly.
//
// Why a new kernel: the certificate stages decide a task only when its alignment lives on one diagonal.  On loci drawn
// from real (repeat-rich) sequence 17-20 % of the tasks are left, and round 3 sent them through band_run_kernel (15-entry
// piece lists per lane), then a wavefront-per-task general kernel (band_coop_kernel, ~70 ns per task) — 580 ms per step
// against 18 ms on an iid genome.  This kernel has no lists to overflow and no per-match storage:
//
//   matches   Eq[c] = bit mask of the haplotype columns holding base c (five 256-bit masks; lane l of the task's eight
//             lanes owns word l = columns 32 l .. 32 l + 31, in registers).  The 6-mer matches of read row x are
//             M6(x) = AND_t Eq[x[x + t]] >> t — three funnel shifts per row over a sliding window (M2, M4, M6).
//             Bytes outside ACGTN, reads / haplotypes above 255 bases: the task is declined (overflow list: the general
//             kernel of vtx_band.hip takes it).
//   sdpkpp    rows ascending.  END events of row x (matches that started at row x - 6): their value
//             V << 16 | xq << 8 | yq (V = dp + xe + ye; the tuple order of the crate's max-tree, ties to the larger
//             match index) enters C[ye] (LDS, ds_max_u32), an 8-column block maximum and a per-task maximum.  START events
//             of row x: dp = max(6, prefixmax(y).V - (x + y) + 1, dp(x - 1, y - 1) + 1): the continuation wins ties, a jump
//             needs >= 6 (oracle: `cand > dp || cand == dp && larger index`; every jump source has a smaller index than the
//             continuation partner).  The prefix maximum is only looked up when the per-task maximum says a jump COULD beat
//             the continuation.  dp is final at the start event (the partner's was), kept one byte per (row mod 8, column).
//   chain     a match that does not continue its diagonal opens a section and is logged (x, y, source).  From the best
//             (dp, x, y): the section is the log entry of this diagonal with the largest x' <= x; go on from its source.
//   band      the staircase's anchors (lazy extension, sections, gaps) -> first / last anchor row per column (ds_min / ds_max)
//             -> lo / hi in closed form (vtx_band.hip's header).
// tests/sweepmodel/sweep_model.cpp restates exactly this on the CPU; tests/test_sweep_model.py checks it against the oracle.
//
// Machine mapping: one wavefront per workgroup = 8 tasks x 8 lanes, 4.1 KB of LDS per task.  Integer / LDS work, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "vtx_device.h"
#include "../../include/vtx_band_semantics.h"

namespace {

constexpr int K = VTX_REF_K, W = VTX_REF_W;
static_assert(K == 6 && W == 20 && VTX_REF_MATCH == 1 && VTX_REF_GAP_OPEN == -5 && VTX_REF_GAP_EXTEND == -1,
              "band_sweep_kernel's sdpkpp is written for k = 6, match 1, gap -5 / -1 (src/main.rs:33-38, :899)");
constexpr int SECCAP = 28;           // sections of the best chain (a 150-base read chains at most 25 six-mers end to end)
constexpr int MAXLEN = 255;          // read / haplotype bases (one byte per coordinate in the packed words)
// per-task LDS (32-bit words).  LOGCAP = sections a task may open: 256 for the first pass (real sequence: p99 55; 4.8 KB per task,
// four wavefronts per CU), 1024 for the second pass over what the first declined (satellites, tandem repeats over two letters:
// hundreds of dominated pieces whose every match opens a section; 7.9 KB per task, two wavefronts per CU).
template <int LOGCAP>
struct Lay {
    static constexpr int O_RING = 0;            // 8 rows x 256 dp bytes; after the sweep: rmin[256], rmax[256]
    static constexpr int O_C = 512;             // C[ye], 256 words
    static constexpr int O_BM = 768;            // block maxima of C, 8 columns each
    static constexpr int O_PBM = 800;           // exclusive prefix maxima of BM
    static constexpr int O_CODE = 832;          // read base codes, one NIBBLE per row (7: no base): 8 rows per word, 36 words
    static constexpr int O_SEC = 900;           // SECCAP sections: x0 << 16 | y0 << 8 | matches
    static constexpr int O_MISC = 928;          // [0] log entries
    static constexpr int O_LOG = 948;           // LOGCAP section records: x << 24 | y << 16 | source (0xffff: none)
    static constexpr int TASK_W = 948 + LOGCAP; // = 20 (mod 32) for both capacities: the eight tasks of a wavefront start in eight different banks
    static_assert(TASK_W % 32 == 20 && TASK_W % 4 == 0, "bank spread / 16-byte alignment of the task slices");
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP moves are never wrapped in a select: `cond ? dpp(x) : 0` compiles to an EXEC-masked DPP, and a source lane that EXEC
// disables reads as 0 (the first version of the block-maximum scan lost lane 0's blocks that way).  Masks are ANDed in.
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
// stat_counters != nullptr: stat_counters[reason] counts the declined tasks (1 bytes / lengths, 2 log full, 3 sections).
// n_dev != nullptr: the list length lives on the device (min(*n_dev, n_tasks); the grid is sized for n_tasks).
//
// The sweep's fast path: nearly every lane has at most ONE match per row, and it continues the lane's match of the row before.  The
// dp of a lane's first match of the last six rows rides in a register (one byte per row): the END event six rows later and the
// continuation test of the next row read it there; whether (x - 1, y - 1) is a match at all is a bit of the previous row's mask.
// The dp bytes in LDS are only READ for a lane's second and further matches of a row (repeats).
template <int LOGCAP>
__global__ __launch_bounds__(64) void band_sweep_v1_kernel(
    const uint32_t* __restrict__ tasks, uint32_t n_tasks, const uint32_t* __restrict__ n_dev,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    uint16_t* __restrict__ band, uint32_t band_stride, uint32_t* __restrict__ hard_list,
    uint32_t* __restrict__ overflow_list, uint32_t* __restrict__ counters, uint32_t stats, uint32_t* __restrict__ stat_counters,
    uint8_t* __restrict__ stage, uint32_t* __restrict__ dbg) {
    typedef Lay<LOGCAP> L;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    if (n_dev) { const uint32_t nd = *n_dev; n_tasks = nd < n_tasks ? nd : n_tasks; }
    if (blockIdx.x * 8u >= n_tasks) return;
    const int lane = (int)threadIdx.x;
    const int g = lane >> 3, l = lane & 7;
    uint32_t* T = lds + g * L::TASK_W;
    uint8_t* ring = (uint8_t*)(T + L::O_RING);
    uint32_t* Cw = T + L::O_C;
    uint32_t* BM = T + L::O_BM;
    uint32_t* PBM = T + L::O_PBM;
    uint32_t* LOG = T + L::O_LOG;
    uint32_t* codes32 = T + L::O_CODE;
    uint32_t* SEC = T + L::O_SEC;
    uint32_t* MISC = T + L::O_MISC;

    const uint32_t slot = blockIdx.x * 8u + (uint32_t)g;
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
    const int col0 = 32 * l;

    // ---- set-up: zero C / BM / counters, read codes to LDS, Eq words to registers ----
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        uint4* c4 = (uint4*)(Cw + col0);
#pragma unroll
        for (int i = 0; i < 8; ++i) c4[i] = z;
        *(uint4*)(BM + 4 * l) = z;
        *(uint4*)(PBM + 4 * l) = z;
        if (l == 0) MISC[0] = 0;
    }
    uint32_t bad = 0;
    {
        // read bytes [32 l, 32 l + 32): two 16-byte loads (the arena is padded by 16 bytes)
        uint32_t wds[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (col0 + 16 * h < m) __builtin_memcpy(&v, read_arena + roff + col0 + 16 * h, 16);
            wds[4 * h] = v.x; wds[4 * h + 1] = v.y; wds[4 * h + 2] = v.z; wds[4 * h + 3] = v.w;
        }
        uint32_t packed[4] = {0, 0, 0, 0};                        // rows 32 l .. 32 l + 31, a nibble each
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
        *(uint4*)(codes32 + 4 * l) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        if (l == 0) *(uint4*)(codes32 + 32) = make_uint4(0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u);
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
    const uint32_t nbmask = l == 7 ? 0u : 0xffffffffu;            // lane 7's upper neighbour belongs to the next task
    const uint32_t ge1 = l >= 1 ? 0xffffffffu : 0u, ge2 = l >= 2 ? 0xffffffffu : 0u, ge4 = l >= 4 ? 0xffffffffu : 0u;
    int mmax = m;
    mmax = max(mmax, __shfl_xor(mmax, 8)); mmax = max(mmax, __shfl_xor(mmax, 16)); mmax = max(mmax, __shfl_xor(mmax, 32));
    const int tmax = __builtin_amdgcn_readfirstlane(mmax) + K;    // feed step t = 0 .. m + 5: row r = t - 5 reaches m
    uint32_t m1p = 0;                                             // M1(t - 1)
    uint32_t m2a = 0, m2b = 0, m2c = 0, m2d = 0;                  // M2(t - 2) .. M2(t - 5)
    uint32_t h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0;      // M6(r - 1) .. M6(r - 6)
    uint32_t best = 0;
    uint64_t dq = 0;                                              // dp of this lane's FIRST match of rows r - 1 (bits 0-7) .. r - 6 (bits 40-47)
    uint32_t pmax = 0;                                            // see the END phase
    uint32_t code_blk = 0, code_blk_next = codes32[0];          // the codes of eight rows per word: one LDS read (and one wait) per eight rows
#define LDS_DONE() __builtin_amdgcn_s_waitcnt(0xC07F)          /* s_waitcnt lgkmcnt(0) (gfx9 encoding: vmcnt / expcnt untouched) */
#define SHR_WORDS(v, s) __builtin_amdgcn_alignbit(dpp0<DPP_ROW_SHL(1)>(v) & nbmask, (v), (s))
    // one END event: the match that started at (r - 6, y) with dp enters C / BM; returns the inserted value
#define END_EVENT(y, dp)                                                                              \
    {                                                                                                 \
        const uint32_t key_ = ((uint32_t)(dp) << 16) | ((uint32_t)(r - K) << 8) | (uint32_t)(y);      \
        best = max(best, key_);                                                                       \
        const uint32_t val_ = key_ + ((uint32_t)(r + K + (y)) << 16);   /* V = dp + xe + ye = dp + r + (y + 6) */ \
        atomicMax(&Cw[(y) + K], val_);                                                                \
        atomicMax(&BM[((y) + K) >> 3], val_);                                                         \
        if ((((y) + K) >> 5) == l) ins_a = max(ins_a, val_); else ins_b = max(ins_b, val_);           \
    }
    const uint32_t ablate = VTX_ABLATE(stats >> 8);                           // (profiling aid, tools/ablate_sweep.sh; results are wrong by design)
#pragma unroll 1
    for (int t = 0; t < (ablate == 1 ? 0 : tmax); ++t) {
        if ((t & 7) == 0) { code_blk = code_blk_next; code_blk_next = codes32[min((t >> 3) + 1, 35)]; }
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
        // (the dp bytes of row r live in ring slot r & 7, last used by row r - 8.  Nothing clears them: whether a cell holds a match is
        // a bit of that row's mask, the byte is only read where the bit is set)
        if (!__any((w_start | w_end) != 0u)) { dq <<= 8; continue; }
        // ---- END events of row r: matches that started at row r - 6.  The lane's first one has its dp in dq ----
        if (__any(w_end != 0u)) {
            uint32_t ins_a = 0, ins_b = 0;                        // inserted into this lane's own 32 columns / into the next lane's
            uint32_t w = w_end;
            if (w) {
                const int y = col0 + (int)__builtin_ctz(w);
                w &= w - 1u;
                const uint32_t dp = (uint32_t)(dq >> 40) & 0xffu;
                END_EVENT(y, dp)
            }
            if (__any(w != 0u)) {
                wave_sync();
                const uint8_t* rrow = ring + (((r - K) & 7) << 8);
                while (__any(w != 0u)) {
                    const bool on = w != 0u;
                    const int y = col0 + (on ? (int)__builtin_ctz(w) : 0);
                    w &= w - 1u;                                          // (0 stays 0)
                    const uint32_t dp = on ? (uint32_t)rrow[y] : 0u;
                    if (on) END_EVENT(y, dp)
                }
            }
            // pmax = the largest value inserted so far at an end column below 32 (l + 1): an upper bound of every prefix maximum
            // this lane can ask for.  (A bound over the whole task — the first version — made every main-diagonal match after a
            // chance match far to the RIGHT look up the prefix maximum for the next ~15 rows: 60 % of the kernel's time.)
            uint32_t sc = max(ins_a, dpp0<DPP_ROW_SHR(1)>(ins_b) & ge1);
            sc = max(sc, dpp0<DPP_ROW_SHR(1)>(sc) & ge1);
            sc = max(sc, dpp0<DPP_ROW_SHR(2)>(sc) & ge2);
            sc = max(sc, dpp0<DPP_ROW_SHR(4)>(sc) & ge4);
            pmax = max(pmax, sc);
        }
        wave_sync();
        if (ablate == 3) { dq <<= 8; continue; }                  // (profiling aid) ... + the END events
        // ---- START events of row r ----
        uint32_t dv_first = 0;
        if (__any(w_start != 0u)) {
            // is (r - 1, y - 1) a match?  bit b of the previous row's mask shifted up by one column (bit 31 of the lane below comes in
            // at bit 0); is it its lane's FIRST match of that row (then its dp rides in that lane's dq)?  the same with the lowest bits
            const uint32_t left_prev = dpp0<DPP_ROW_SHR(1)>(w_prev) & ge1;
            const uint32_t pshift = __builtin_amdgcn_alignbit(w_prev, left_prev, 31);
            const uint32_t lowp = w_prev & (0u - w_prev);
            const uint32_t fshift = __builtin_amdgcn_alignbit(lowp, dpp0<DPP_ROW_SHR(1)>(lowp) & ge1, 31);
            const uint32_t own_dp = (uint32_t)dq & 0xffu;
            const uint32_t left_dp = dpp0<DPP_ROW_SHR(1)>(own_dp) & ge1;
            uint32_t w = w_start;
            const uint8_t* prow = ring + (((r - 1) & 7) << 8);
            uint8_t* crow = ring + ((r & 7) << 8);
            bool pbm_ready = false;
            bool first = true;
            while (__any(w != 0u)) {
                const bool on = w != 0u;
                const int b = on ? (int)__builtin_ctz(w) : 0;
                w &= w - 1u;
                const int y = col0 + b;
                const bool exists = on && ((pshift >> b) & 1u);
                int dpc = !exists ? 0 : (((fshift >> b) & 1u) ? (int)(b ? own_dp : left_dp) : -1);
                if (__any(dpc < 0)) {
                    wave_sync();
                    if (dpc < 0) dpc = (int)prow[y - 1];
                    LDS_DONE();        // the wait belongs INSIDE the rare branch: at the join the compiler would wait on every path — for the
                }                      // END events' atomics the fast path has just issued (41 % of the kernel's wave cycles were such waits)
                // could a jump beat the continuation (or reach 6 where there is none)?  upper bound from pmax
                const int cand_ub = (int)(pmax >> 16) - (r + y) + 1;
                const bool need_q = on && pmax != 0u && cand_ub > (dpc ? dpc + 1 : K - 1);
                uint32_t q = 0;
                if (__any(need_q)) {
                    if (!pbm_ready) {
                        // exclusive prefix maxima of the block maxima (BM cannot change during the START phase)
                        wave_sync();
                        const uint4 bm = *(const uint4*)(BM + 4 * l);
                        const uint32_t p0 = bm.x, p1 = max(p0, bm.y), p2 = max(p1, bm.z), p3 = max(p2, bm.w);
                        uint32_t inc = p3;
                        inc = max(inc, dpp0<DPP_ROW_SHR(1)>(inc) & ge1);
                        inc = max(inc, dpp0<DPP_ROW_SHR(2)>(inc) & ge2);
                        inc = max(inc, dpp0<DPP_ROW_SHR(4)>(inc) & ge4);
                        const uint32_t exc = dpp0<DPP_ROW_SHR(1)>(inc) & ge1;
                        *(uint4*)(PBM + 4 * l) = make_uint4(exc, max(exc, p0), max(exc, p1), max(exc, p2));
                        pbm_ready = true;
                        wave_sync();
                    }
                    if (need_q) {
                        const int blk = y >> 3, kk = y & 7;
                        q = PBM[blk];
                        const uint4 c0 = *(const uint4*)(Cw + 8 * blk), c1 = *(const uint4*)(Cw + 8 * blk + 4);
                        q = max(q, c0.x);
                        q = max(q, kk >= 1 ? c0.y : 0u); q = max(q, kk >= 2 ? c0.z : 0u); q = max(q, kk >= 3 ? c0.w : 0u);
                        q = max(q, kk >= 4 ? c1.x : 0u); q = max(q, kk >= 5 ? c1.y : 0u); q = max(q, kk >= 6 ? c1.z : 0u);
                        q = max(q, kk >= 7 ? c1.w : 0u);
                    }
                    LDS_DONE();
                }
                if (on) {
                    int dv = K;
                    uint32_t src = 0xffffu;
                    if (q) {
                        const int cand = (int)(q >> 16) - (r + y) + 1;
                        if (cand >= K) { dv = cand; src = q & 0xffffu; }
                    }
                    bool cont = false;
                    if (dpc && dpc + 1 >= dv) { dv = dpc + 1; cont = true; }
                    crow[y] = (uint8_t)dv;
                    if (first) dv_first = (uint32_t)dv;
                    if (!cont) {
                        const uint32_t pos = atomicAdd(&MISC[0], 1u);
                        if (pos < (uint32_t)LOGCAP) LOG[pos] = ((uint32_t)r << 24) | ((uint32_t)y << 16) | src;
                    }
                }
                first = false;
            }
        }
        dq = (dq << 8) | dv_first;
        wave_sync();
    }
#undef SHR_WORDS
#undef END_EVENT
#undef LDS_DONE
    wave_sync();
    best = group_max(best);
    const uint32_t logn = MISC[0];
    if (!decline && logn > (uint32_t)LOGCAP) decline = 2u;
    const bool seeded = best != 0u && !decline && ablate != 4;     // (ablate 4: profiling aid — the sweep without the chain walk and the band)

    // ---- the chain's sections, last first ----
    int nsec = 0;
    {
        int cx = (int)((best >> 8) & 0xffu), cy = (int)(best & 0xffu);
        bool walking = seeded;
        while (__any(walking)) {
            const int d = cy - cx;
            uint32_t pick = 0;                                      // (x' + 1) << 12 | log index, maximum over the diagonal's entries with x' <= x
            if (walking) {
                for (uint32_t i = (uint32_t)l; i < logn; i += 8u) {
                    const uint32_t e = LOG[i];
                    const int ex = (int)(e >> 24), ey = (int)((e >> 16) & 0xffu);
                    if (ey - ex == d && ex <= cx) pick = max(pick, ((uint32_t)(ex + 1) << 12) | i);
                }
            }
            pick = group_max(pick);
            if (walking) {
                if (pick == 0u || nsec >= SECCAP) { decline = pick == 0u ? 4u : 3u; walking = false; }
                else {
                    const uint32_t e = LOG[pick & 0xfffu];
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
    // ---- anchors -> first / last anchor row per column (the ring is dead: rmin / rmax take its place) ----
    uint32_t* rmin = T + L::O_RING;
    uint32_t* rmax = rmin + 256;
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
        o[0] = best; o[1] = logn; o[2] = (uint32_t)nsec; o[3] = (uint32_t)cA; o[4] = (uint32_t)cB; o[5] = decline; o[6] = (uint32_t)m; o[7] = (uint32_t)n;
        for (int i = 0; i < 12; ++i) o[8 + i] = SEC[i];
        for (int i = 0; i < 24; ++i) o[20 + i] = (uint32_t)i < logn ? LOG[i] : 0u;
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
}

// tier 0: 256 sections per task (first pass: 4.8 KB of LDS per task, still four wavefronts per CU); tier 1: 1024 (second pass over
// the first's log overflows: 7.9 KB per task, two wavefronts per CU)
extern "C" hipError_t vtxk_launch_band_sweep_v1(int tier, const uint32_t* tasks, uint32_t n_tasks, const uint32_t* n_dev,
                                             const vtx_record* records,
                                             const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                             const uint8_t* hap_arena, uint16_t* band, uint32_t band_stride, uint32_t* hard_list,
                                             uint32_t* overflow_list, uint32_t* counters, uint32_t* stat_counters, uint8_t* stage,
                                             uint32_t* dbg, hipStream_t s) {
    if (!n_tasks) return hipSuccess;
    static const uint32_t ablate = VTX_DEV_ENV("VTX_SWEEP_ABLATE") ? (uint32_t)atoi(VTX_DEV_ENV("VTX_SWEEP_ABLATE")) << 8 : 0u;
#define LAUNCH_SWEEP(CAP)                                                                                          \
    {                                                                                                              \
        const size_t shmem = (size_t)8 * Lay<CAP>::TASK_W * sizeof(uint32_t);                                      \
        if (shmem > 48 * 1024) {                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void*)band_sweep_v1_kernel<CAP>,                               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);           \
            if (e != hipSuccess) return e;                                                                         \
        }                                                                                                          \
        hipLaunchKernelGGL(band_sweep_v1_kernel<CAP>, dim3((n_tasks + 7) / 8), dim3(64), shmem, s, tasks, n_tasks, n_dev, records, \
                           rec_locus, loci, read_arena, hap_arena, band, band_stride, hard_list, overflow_list, counters,      \
                           ablate, stat_counters, stage, dbg);                                                     \
    }
    if (tier == 0) LAUNCH_SWEEP(256) else LAUNCH_SWEEP(1024)
#undef LAUNCH_SWEEP
    return hipGetLastError();
}
