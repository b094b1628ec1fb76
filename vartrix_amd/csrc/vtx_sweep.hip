// vtx_sweep.hip — band_sweep_kernel: the band of bio 0.30.0's banded aligner for ANY task, eight lanes per task
// (round 4; the robust path behind band_diag_kernel / band_refine_kernel).
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
constexpr int LOGCAP = 128;          // sections a task may open (real sequence: p99 55, max 88; tandem repeats over 2 letters: 400+)
constexpr int SECCAP = 12;           // sections of the best chain
constexpr int MAXLEN = 255;          // read / haplotype bases (one byte per coordinate in the packed words)
// per-task LDS (32-bit words)
constexpr int O_RING = 0;            // 8 rows x 256 dp bytes; after the sweep: rmin[256], rmax[256]
constexpr int O_C = 512;             // C[ye], 256 words
constexpr int O_BM = 768;            // block maxima of C, 8 columns each
constexpr int O_PBM = 800;           // exclusive prefix maxima of BM
constexpr int O_LOG = 832;           // LOGCAP section records: x << 24 | y << 16 | source (0xffff: none)
constexpr int O_CODE = 960;          // read base codes, one byte per row (7: no base), 272 bytes
constexpr int O_SEC = 1028;          // SECCAP sections: x0 << 16 | y0 << 8 | matches
constexpr int O_MISC = 1040;         // [0] log entries, [1] maximum of every inserted value
constexpr int TASK_W = 1044;         // = 4 (mod 32) x 5: the eight tasks of a wavefront start in eight different banks

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {      // lanes without a source get 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_SHR(n) (0x110 + (n))

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
// stats != 0: counters[48 + reason] counts the declined tasks (1 bytes / lengths, 2 log full, 3 sections).
__global__ __launch_bounds__(64) void band_sweep_kernel(
    const uint32_t* __restrict__ tasks, uint32_t n_tasks,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    uint16_t* __restrict__ band, uint32_t band_stride, uint32_t* __restrict__ hard_list,
    uint32_t* __restrict__ overflow_list, uint32_t* __restrict__ counters, uint32_t stats, uint8_t* __restrict__ stage) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8 * TASK_W];
    const int lane = (int)threadIdx.x;
    const int g = lane >> 3, l = lane & 7;
    uint32_t* T = lds + g * TASK_W;
    uint8_t* ring = (uint8_t*)(T + O_RING);
    uint32_t* Cw = T + O_C;
    uint32_t* BM = T + O_BM;
    uint32_t* PBM = T + O_PBM;
    uint32_t* LOG = T + O_LOG;
    uint8_t* codes = (uint8_t*)(T + O_CODE);
    uint32_t* SEC = T + O_SEC;
    uint32_t* MISC = T + O_MISC;

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
        if (l == 0) { MISC[0] = 0; MISC[1] = 0; }
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
        uint32_t packed[8];
#pragma unroll
        for (int wi = 0; wi < 8; ++wi) {
            uint32_t pk = 0;
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
                const int idx = col0 + 4 * wi + bi;
                uint32_t c = 7u;
                if (idx < m) {
                    c = base_code((wds[wi] >> (8 * bi)) & 0xffu);
                    bad |= (c == 7u) ? 1u : 0u;
                }
                pk |= c << (8 * bi);
            }
            packed[wi] = pk;
        }
        uint4* cd = (uint4*)(codes + col0);
        cd[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        cd[1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
        if (l == 0) *(uint4*)(codes + 256) = make_uint4(0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u);
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
    bad |= (uint32_t)__shfl_xor((int)bad, 1); bad |= (uint32_t)__shfl_xor((int)bad, 2); bad |= (uint32_t)__shfl_xor((int)bad, 4);
    if (bad) { decline = 1u; eq0 = eq1 = eq2 = eq3 = eq4 = 0; }       // (no matches: the sweep idles for this task)
    wave_sync();

    // ---- the sweep ----
    const uint32_t nbmask = l == 7 ? 0u : 0xffffffffu;            // lane 7's upper neighbour belongs to the next task
    int mmax = m;
    mmax = max(mmax, __shfl_xor(mmax, 8)); mmax = max(mmax, __shfl_xor(mmax, 16)); mmax = max(mmax, __shfl_xor(mmax, 32));
    const int tmax = __builtin_amdgcn_readfirstlane(mmax) + K;    // feed step t = 0 .. m + 5: row r = t - 5 reaches m
    uint32_t m1p = 0;                                             // M1(t - 1)
    uint32_t m2a = 0, m2b = 0, m2c = 0, m2d = 0;                  // M2(t - 2) .. M2(t - 5)
    uint32_t h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0;      // M6(r - 1) .. M6(r - 6)
    uint32_t best = 0;
    uint32_t code_next = codes[0];
#define SHR_WORDS(v, s) __builtin_amdgcn_alignbit(dpp0<DPP_ROW_SHL(1)>(v) & nbmask, (v), (s))
#pragma unroll 1
    for (int t = 0; t < tmax; ++t) {
        const uint32_t code = code_next;
        code_next = codes[min(t + 1, 271)];
        // M1(t): the Eq word of this row's base
        uint32_t m1 = (code & 1u) ? ((code & 2u) ? eq3 : eq1) : ((code & 2u) ? eq2 : eq0);
        m1 = (code & 4u) ? (code == 4u ? eq4 : 0u) : m1;
        const uint32_t m2n = m1p & SHR_WORDS(m1, 1);              // M2(t - 1)
        const uint32_t m4 = m2d & SHR_WORDS(m2b, 2);              // M4(t - 5) = M2(t - 5) & M2(t - 3) >> 2
        const uint32_t w_start = m4 & SHR_WORDS(m2n, 4);          // M6(t - 5)
        m1p = m1; m2d = m2c; m2c = m2b; m2b = m2a; m2a = m2n;
        const int r = t - (K - 1);
        const uint32_t w_end = h6;
        h6 = h5; h5 = h4; h4 = h3; h3 = h2; h2 = h1; h1 = w_start;
        if (r < 0) continue;                                      // (uniform)
        // the dp bytes of row r live in ring slot r & 7 (last used by row r - 8): cleared even when the row has no match —
        // row r + 1 looks its continuation partners up here
        {
            uint4* rc = (uint4*)(ring + ((r & 7) << 8) + col0);
            rc[0] = make_uint4(0, 0, 0, 0); rc[1] = make_uint4(0, 0, 0, 0);
        }
        if (!__any((w_start | w_end) != 0u)) continue;
        wave_sync();
        // ---- END events of row r: matches that started at row r - 6 ----
        {
            uint32_t w = w_end;
            const uint8_t* rrow = ring + (((r - K) & 7) << 8);
            while (__any(w != 0u)) {
                const bool on = w != 0u;
                const int b = on ? (int)__builtin_ctz(w) : 0;
                w &= w - 1u;                                              // (0 stays 0)
                const int y = col0 + b;
                const uint32_t dp = on ? (uint32_t)rrow[y] : 0u;
                if (on) {
                    const uint32_t key = (dp << 16) | ((uint32_t)(r - K) << 8) | (uint32_t)y;
                    best = max(best, key);
                    const uint32_t val = key + ((uint32_t)(r + K + y) << 16);          // V = dp + xe + ye = dp + r + (y + 6); V << 16 | xq << 8 | yq
                    const int ye = y + K;
                    atomicMax(&Cw[ye], val);
                    atomicMax(&BM[ye >> 3], val);
                    atomicMax(&MISC[1], val);
                }
            }
        }
        wave_sync();
        // ---- START events of row r ----
        {
            uint32_t w = w_start;
            const uint8_t* prow = ring + (((r - 1) & 7) << 8);
            uint8_t* crow = ring + ((r & 7) << 8);
            const uint32_t gmax = MISC[1];
            bool pbm_ready = false;
            while (__any(w != 0u)) {
                const bool on = w != 0u;
                const int b = on ? (int)__builtin_ctz(w) : 0;
                w &= w - 1u;
                const int y = col0 + b;
                const int dpc = (on && r > 0 && y > 0) ? (int)prow[y - 1] : 0;
                // could a jump beat the continuation (or reach 6 where there is none)?  upper bound from the task's maximum
                const int cand_ub = (int)(gmax >> 16) - (r + y) + 1;
                const bool need_q = on && gmax != 0u && cand_ub > (dpc ? dpc + 1 : K - 1);
                uint32_t q = 0;
                if (__any(need_q)) {
                    if (!pbm_ready) {
                        // exclusive prefix maxima of the block maxima (BM cannot change during the START phase)
                        const uint4 bm = *(const uint4*)(BM + 4 * l);
                        const uint32_t p0 = bm.x, p1 = max(p0, bm.y), p2 = max(p1, bm.z), p3 = max(p2, bm.w);
                        uint32_t inc = p3;
                        inc = max(inc, l >= 1 ? dpp0<DPP_ROW_SHR(1)>(inc) : 0u);
                        inc = max(inc, l >= 2 ? dpp0<DPP_ROW_SHR(2)>(inc) : 0u);
                        inc = max(inc, l >= 4 ? dpp0<DPP_ROW_SHR(4)>(inc) : 0u);
                        const uint32_t exc = l >= 1 ? dpp0<DPP_ROW_SHR(1)>(inc) : 0u;
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
                    if (!cont) {
                        const uint32_t pos = atomicAdd(&MISC[0], 1u);
                        if (pos < (uint32_t)LOGCAP) LOG[pos] = ((uint32_t)r << 24) | ((uint32_t)y << 16) | src;
                    }
                }
            }
        }
        wave_sync();
    }
#undef SHR_WORDS
    wave_sync();
    best = max(best, (uint32_t)__shfl_xor((int)best, 1));
    best = max(best, (uint32_t)__shfl_xor((int)best, 2));
    best = max(best, (uint32_t)__shfl_xor((int)best, 4));
    const uint32_t logn = MISC[0];
    if (!decline && logn > (uint32_t)LOGCAP) decline = 2u;
    const bool seeded = best != 0u && !decline;

    // ---- the chain's sections, last first ----
    int nsec = 0;
    {
        int cx = (int)((best >> 8) & 0xffu), cy = (int)(best & 0xffu);
        bool walking = seeded;
        while (__any(walking)) {
            const int d = cy - cx;
            uint32_t pick = 0;                                      // (x' + 1) << 8 | log index, maximum over the diagonal's entries with x' <= x
            if (walking) {
                for (uint32_t i = (uint32_t)l; i < logn; i += 8u) {
                    const uint32_t e = LOG[i];
                    const int ex = (int)(e >> 24), ey = (int)((e >> 16) & 0xffu);
                    if (ey - ex == d && ex <= cx) pick = max(pick, ((uint32_t)(ex + 1) << 8) | i);
                }
            }
            pick = max(pick, (uint32_t)__shfl_xor((int)pick, 1));
            pick = max(pick, (uint32_t)__shfl_xor((int)pick, 2));
            pick = max(pick, (uint32_t)__shfl_xor((int)pick, 4));
            if (walking) {
                if (pick == 0u || nsec >= SECCAP) { decline = pick == 0u ? 4u : 3u; walking = false; }
                else {
                    const uint32_t e = LOG[pick & 0xffu];
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
    uint32_t* rmin = T + O_RING;
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
        if (stats) atomicAdd(&counters[48 + min(decline, 7u)], 1u);
    }
}

extern "C" hipError_t vtxk_launch_band_sweep(const uint32_t* tasks, uint32_t n_tasks, const vtx_record* records,
                                             const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                             const uint8_t* hap_arena, uint16_t* band, uint32_t band_stride, uint32_t* hard_list,
                                             uint32_t* overflow_list, uint32_t* counters, int stats, uint8_t* stage, hipStream_t s) {
    if (!n_tasks) return hipSuccess;
    hipLaunchKernelGGL(band_sweep_kernel, dim3((n_tasks + 7) / 8), dim3(64), 0, s, tasks, n_tasks, records, rec_locus, loci,
                       read_arena, hap_arena, band, band_stride, hard_list, overflow_list, counters, (uint32_t)stats, stage);
    return hipGetLastError();
}
// what the kernel holds: reads and haplotypes up to this many bases (longer ones are declined task by task; a batch whose
// haplotypes are all longer should not be sent here at all)
extern "C" uint32_t vtxk_band_sweep_max_len(void) { return (uint32_t)MAXLEN; }
