// vtx_band.hip — device side of the *banded* aligner flavour
// (bio 0.30.0 banded::Aligner::local as restated in oracle/vtx_oracle.c; reference call site
// src/main.rs:899-901 with K = 6, W = 20, src/main.rs:33-34).
//
// The banded flavour's pipeline (vtx_api.hip: vtx_run), round 4:
//   band_tables_kernel -> band_diag_kernel (+ band_refine_kernel): the certificate of a task whose alignment lives on ONE diagonal;
//   what they leave: (a) with every off-diagonal match harmless but bounds apart: its band IS one diagonal stretch — the band-masked
//   DP expands it from one word (sw_banded_kernel<.., 2>); (b) repeats (more than 40 off-diagonal matches spread over many
//   diagonals): band_sweep_kernel (vtx_sweep.hip: the band of ANY task) + band-masked DP; (c) the others (a read against the other
//   allele of an indel: two diagonals): band_run_kernel's general certificate below, its hard tasks to the masked DP, its list
//   overflows to band_sweep_kernel; (d) what band_sweep_kernel declines (bytes outside ACGTN, more than 255 bases, more than
//   1024 sections): band_coop_kernel / band_kernel, the literal general path.
//
//   band_run_kernel    one lane per (record, haplotype) task, k-mer tables of the haplotypes shared in LDS:
//                      seeding (exact 6-mer matches as diagonal pieces), sdpkpp chaining over the pieces,
//                      traceback to the anchor staircase, and the DP-FREE CERTIFICATE below.  Certified
//                      tasks get their score here; the others go to the hard list with their staircase.
//   band_kernel        general fallback (per-task global scratch, literal Fenwick-tree sdpkpp) for tasks that
//                      exceed band_run_kernel's capacities; every task it handles goes to the hard list.
//   band_expand_kernel staircase polyline -> per-column row ranges [lo, hi).
//   sw_banded_kernel   (vtx_kernels.hip) the systolic packed-i16 DP restricted to those ranges: the exact
//                      banded score of the hard tasks.
//
// Band in closed form.  Every cell the crate adds (set_boundaries' lazy extensions, add_kmer,
// add_entry for continued k-mers, add_gap) lies on ONE monotone, connected staircase from
// (first - d0) to (last_end + d1); each adds the (2w+1)-square around it and ranges only grow by
// min/max.  With rmin[c] / rmax[c] = first / last anchor row in anchored column c in [cA, cB]:
//     lo[j] = max(0, rmin[max(j - w, cA)] - w),  hi[j] = min(rows, rmax[min(j + w, cB)] + w + 1)
// for j in [cA - w, cB + w], empty elsewhere.  tests/ check this against the oracle's literal
// add_entry loops.
//
// Certificate (no DP).  cert <= banded <= full <= ub, so cert == ub decides the banded score:
//   cert  the anchor staircase is an in-band path; its local-alignment score (affine gaps along the
//         vertical / horizontal pieces, free restart at 0) is a lower bound of the banded score.
//   ub    upper bound of the FULL-matrix score from the exact-match runs of >= K bases (= the diagonal
//         pieces of k-mer matches the seeding found — all of them).  Write an alignment as maximal runs of
//         match columns separated by events (mismatch: 5; gap of length L: 5 + L).  A run of >= 6 columns
//         is a sub-run of a piece.  Between two consecutive long runs there are e >= 1 events and e - 1 short
//         runs (<= 5 columns each), which net <= 5 (e - 1) - 5 e - sum L = -5 - sum L, and sum L >= the
//         difference of the two diagonals; the stretches before the first / after the last long run net <= 0;
//         an alignment without a long run scores <= 5.  So
//             full <= max(5, max over chains of sub-runs of pieces [ sum len - sum J ]),  J = 5 + |d' - d|,
//         and between runs of the same diagonal, D >= 1 bases apart: J = 6 ceil((D + 5) / 6) - D (gap-free: e mismatches
//         cost 6 e - D and short runs need e >= ceil((D + 5) / 6); a stretch with gaps costs at least J_gap(D) >= that).
//         (band_diag_kernel, which has the match mask of its diagonal, uses min(6 e - D, J_gap(D)) with the true e.)
//         run_ub() maximises this over the piece list (one number per piece, fixpoint over ordered pairs);
//         oracle/vtx_certify.c restates it on the CPU with the proof in full, tests/ check ub >= full.
//   Measured on the synthetic workloads: cert == ub for 99.3 % of SNV tasks (0.5 % substitution errors),
//   97.5-98.7 % with indels <= 20 bp; the rest is scored exactly by the band-masked DP.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "vtx_device.h"
#include "vtx_fast_core.h"
#include "vtx_band_trim.h"
#include "../../include/vtx_band_semantics.h"

#define KMER 6
#define BANDW 20
// The certificate (run_ub / ub_join_same: "a mismatch costs 5 + 1", "short runs <= K - 1 = 5", "a gap of length L costs 5 + L",
// "J_gap") and the closed-form sdpkpp (k-mer = K matches, gap_open + d * gap_extend) are DERIVED for the
// reference's constants; vtx_create refuses any other configuration (vtx_api.hip), and this ties the two together:
static_assert(KMER == VTX_REF_K && BANDW == VTX_REF_W && VTX_REF_MATCH == 1 && VTX_REF_MISMATCH == -5 && VTX_REF_GAP_OPEN == -5 &&
              VTX_REF_GAP_EXTEND == -1, "vtx_band.hip's bounds are proved for K = 6, W = 20, +1 / -5, gap -5 / -1 (src/main.rs:33-38) only");
#define HASH_BITS 9
#define HASH_SIZE (1 << HASH_BITS)
// band slot markers (lo[0]): a staircase polyline follows / the whole matrix is in band
#define BAND_POLYLINE 0xffffu
#define BAND_FULL_MATRIX 0xfffeu

// Per-task slices of the workspace (all sized by the launch).  S = element stride: 1 = every task owns a contiguous slab
// (LDS variant, slow path); 64 = the slabs of the 64 lanes of a wavefront are interleaved element by element, so the
// lanes' accesses to the same index (the lanes of a wavefront run the serial algorithm nearly in step) share cache lines.
template <typename T, int S>
struct strided {
    T* p;
    __device__ __forceinline__ T& operator[](size_t i) const { return p[i * S]; }
};
template <int S>
struct band_scratch_t {
    strided<uint16_t, S> head;   // HASH_SIZE
    strided<uint16_t, S> next;   // n
    strided<uint32_t, S> mt;     // M_cap packed (x << 16 | y)
    strided<int32_t, S> dps;     // M_cap
    strided<int32_t, S> dpp;     // M_cap
    strided<int2, S> tree;       // n + KMER + 4 Fenwick nodes {v, match index}
    strided<int32_t, S> cont;    // M_cap: the match one step up the diagonal (-1: none)
    strided<uint16_t, S> rmin;   // n + 2
    strided<uint16_t, S> rmax;   // n + 2
    // carve the arrays out of a slab; `lane` = this task's position among the S interleaved ones
    __device__ __forceinline__ size_t carve(uint8_t* ws, uint32_t lane, uint32_t m_cap, uint32_t max_hap) {
        size_t o = 0;
#define CARVE(field, T, count) field.p = (T*)(ws + o * S) + lane; o += (size_t)(count) * sizeof(T);
        CARVE(head, uint16_t, HASH_SIZE)
        CARVE(next, uint16_t, (size_t)max_hap + 2)
        CARVE(rmin, uint16_t, (size_t)max_hap + 2)
        CARVE(rmax, uint16_t, (size_t)max_hap + 2)
        o = (o + 15) & ~(size_t)15;
        CARVE(tree, int2, (size_t)max_hap + KMER + 4)
        CARVE(mt, uint32_t, m_cap)
        CARVE(dps, int32_t, m_cap)
        CARVE(dpp, int32_t, m_cap)
        CARVE(cont, int32_t, m_cap)
#undef CARVE
        return o;
    }
};

__device__ __forceinline__ uint32_t kmer_hash_dev(const uint8_t* s) {
    uint32_t h = 2166136261u;
#pragma unroll
    for (int i = 0; i < KMER; ++i) { h ^= s[i]; h *= 16777619u; }
    return (h ^ (h >> 15)) & (HASH_SIZE - 1);
}
__device__ __forceinline__ bool kmer_eq(const uint8_t* a, const uint8_t* b) {
    bool eq = true;
#pragma unroll
    for (int i = 0; i < KMER; ++i) eq &= a[i] == b[i];
    return eq;
}
__device__ __forceinline__ bool ent_gt(int32_t av, int32_t ai, int32_t bv, int32_t bi) {
    return av > bv || (av == bv && ai > bi);
}

// local score of one step of the staircase walk
struct walk_state { int32_t s; int32_t gap; int32_t best; int dir; };   // dir: 0 none/diag, 1 vertical, 2 horizontal
__device__ __forceinline__ void walk_diag(walk_state& w, bool match) {
    int32_t v = (w.s > w.gap ? w.s : w.gap) + (match ? 1 : -5);
    w.s = v > 0 ? v : 0; w.gap = -100000; w.dir = 0;
    if (w.s > w.best) w.best = w.s;
}
__device__ __forceinline__ void walk_gap(walk_state& w, int dir) {
    // affine: opening from S costs -6, extending the same direction -1
    int32_t open = w.s - 6;
    int32_t ext = (w.dir == dir) ? w.gap - 1 : -100000;
    w.gap = open > ext ? open : ext; w.dir = dir;
    // S of the gap cell is max(gap state, 0): the walk may restart at 0 at any in-band cell
    w.s = w.gap > 0 ? w.gap : 0;
}

// Returns: 0 ok; 1 match capacity exceeded (needs a larger slab).  *cert_out = INT32_MAX when there is no
// k-mer match (Band::full_matrix: banded == full by construction).
template <int S>
__device__ int band_task(const uint8_t* x, int m, const uint8_t* y, int n, band_scratch_t<S> sc, uint32_t m_cap,
                         int32_t* cert_out, int* cA_out, int* cB_out) {
    *cert_out = 0;
    // ---- find_kmer_matches: chained hash of y's k-mers, probes in x order, j ascending ----
    uint32_t M = 0;
    if (m >= KMER && n >= KMER) {
        for (int i = 0; i < HASH_SIZE; ++i) sc.head[i] = 0xffff;
        for (int j = n - KMER; j >= 0; --j) {
            uint32_t h = kmer_hash_dev(y + j);
            sc.next[j] = sc.head[h]; sc.head[h] = (uint16_t)j;
        }
        for (int i = 0; i + KMER <= m; ++i) {
            uint32_t h = kmer_hash_dev(x + i);
            for (uint32_t j = sc.head[h]; j != 0xffff; j = sc.next[j]) {
                if (kmer_eq(x + i, y + j)) {
                    if (M < m_cap) sc.mt[M] = ((uint32_t)i << 16) | j;
                    ++M;
                }
            }
        }
    }
    if (M > m_cap) return 1;
    if (M == 0) { *cert_out = INT32_MAX; return 0; }   // Band::full_matrix
    // ---- sdpkpp ----
    // Fenwick queries and updates request all of their nodes before looking at any (a lane is alone with its task:
    // every dependent round trip to the slab is latency nobody hides)
    const int tn = n + KMER + 2;
    for (int i = 0; i <= tn; ++i) sc.tree[i] = make_int2(INT32_MIN, -1);
    int32_t best_v = KMER, best_i = 0;
    uint32_t s_ptr = 0, e_ptr = 0;    // next start / end event (both in match order)
    // continuation partner of a match = the match one step up its diagonal: the matches are sorted by (x, y), so a
    // pointer into the previous row follows the current row (merge scan)
    int row_x = -2;
    uint32_t row_begin = 0, prev_hi = 0, scan = 0;
    // the next start and end matches stay in registers (requested when their pointer moves, not when they are needed)
    uint32_t ms = sc.mt[0], me = ms;
    while (e_ptr < M) {
        // next event: start (xs, ys, s+M) vs end (xe+K, ye+K, e); start ids sort after end ids
        bool take_start = false;
        if (s_ptr < M) {
            const uint32_t sx = ms >> 16, sy = ms & 0xffff, ex = (me >> 16) + KMER, ey = (me & 0xffff) + KMER;
            take_start = (sx < ex) || (sx == ex && sy < ey);   // equal coordinates: end first
        }
        if (take_start) {
            const uint32_t p = s_ptr++;
            const int32_t px = (int32_t)(ms >> 16), py = (int32_t)(ms & 0xffff);
            if (s_ptr < M) ms = sc.mt[s_ptr];
            int32_t dv = KMER, dp = -1;
            int32_t bv = INT32_MIN, bi = -1;
            for (int i = py + 1; i > 0;) {
                int2 e[6]; bool on[6];
#pragma unroll
                for (int l = 0; l < 6; ++l) { on[l] = i > 0; e[l] = sc.tree[on[l] ? i : 1]; i -= i & (-i); }
#pragma unroll
                for (int l = 0; l < 6; ++l) if (on[l] && ent_gt(e[l].x, e[l].y, bv, bi)) { bv = e[l].x; bi = e[l].y; }
            }
            if (bi >= 0) {
                const int32_t cand = bv - 5 - (px + py) + KMER;      // stored v = dp + (xe + ye); gap_open -5, extend -1
                if (cand > dv || (cand == dv && bi > dp)) { dv = cand; dp = bi; }
            }
            if (px != row_x) {
                if (px == row_x + 1) { scan = row_begin; prev_hi = p; } else { scan = prev_hi = p; }
                row_x = px; row_begin = p;
            }
            int32_t c = -1;
            if (py > 0) {
                while (scan < prev_hi && (int32_t)(sc.mt[scan] & 0xffff) < py - 1) ++scan;
                if (scan < prev_hi && (int32_t)(sc.mt[scan] & 0xffff) == py - 1) c = (int32_t)scan;
            }
            sc.dps[p] = dv; sc.dpp[p] = dp; sc.cont[p] = c;
        } else {
            const uint32_t p = e_ptr++;
            const int32_t px = (int32_t)(me >> 16), py = (int32_t)(me & 0xffff);
            if (e_ptr < M) me = sc.mt[e_ptr];
            const int32_t c = sc.cont[p];
            int32_t dv = sc.dps[p];
            if (c >= 0) {
                const int32_t cand = sc.dps[c] + 1;
                if (cand > dv || (cand == dv && c > sc.dpp[p])) { dv = cand; sc.dps[p] = cand; sc.dpp[p] = c; }
            }
            const int32_t v = dv + (px + KMER) + (py + KMER);
            for (int i = py + KMER + 1; i <= tn;) {
                int2 e[6]; int at[6];
#pragma unroll
                for (int l = 0; l < 6; ++l) { at[l] = i <= tn ? i : 0; e[l] = sc.tree[at[l] ? at[l] : 1]; i += i & (-i); }
#pragma unroll
                for (int l = 0; l < 6; ++l) if (at[l] && ent_gt(v, (int32_t)p, e[l].x, e[l].y)) sc.tree[at[l]] = make_int2(v, (int32_t)p);
            }
            if (ent_gt(dv, (int32_t)p, best_v, best_i)) { best_v = dv; best_i = (int32_t)p; }
        }
    }
    // ---- traceback: reverse the prev links in place so the chain can be walked forward ----
    int32_t cur = best_i, nxt = -1;
    while (cur >= 0) { const int32_t pv = sc.dpp[cur]; sc.dpp[cur] = nxt; nxt = cur; cur = pv; }
    const int32_t first = nxt;                       // dpp[] now holds the successor on the chain
    // ---- anchor staircase -> rmin / rmax per anchored column, and the certificate walk ----
    const int fx = (int)(sc.mt[first] >> 16), fy = (int)(sc.mt[first] & 0xffff);
    int d0 = fx < fy ? fx : fy; if (d0 > VTX_BAND_LAZY_EXT(KMER)) d0 = VTX_BAND_LAZY_EXT(KMER);
    int r = fx - d0, c = fy - d0;                    // current anchor (DP coordinates: cell (r, c))
    const int cA = c;
    walk_state w = {0, -100000, 0, 0};
    sc.rmin[c] = (uint16_t)r; sc.rmax[c] = (uint16_t)r;
    // helpers: move diagonally / vertically / horizontally to a target, recording anchors.
    // A move INTO cell (r, c) by a diagonal step scores x[r-1] vs y[c-1].
#define STEP_DIAG()  { ++r; ++c; sc.rmin[c] = (uint16_t)r; sc.rmax[c] = (uint16_t)r; walk_diag(w, x[r - 1] == y[c - 1]); }
#define STEP_DOWN()  { ++r; sc.rmax[c] = (uint16_t)r; walk_gap(w, 1); }
#define STEP_RIGHT() { ++c; sc.rmin[c] = (uint16_t)r; sc.rmax[c] = (uint16_t)r; walk_gap(w, 2); }
    for (int i = 0; i < d0; ++i) STEP_DIAG()
    int32_t p = first;
    while (p >= 0) {
        const int px = (int)(sc.mt[p] >> 16), py = (int)(sc.mt[p] & 0xffff);
        // add_gap(prev_end -> (px, py)): diagonal run of min(dr, dc), then the straight remainder
        int dr = px - r, dc = py - c;
        int dg = dr < dc ? dr : dc;
        for (int i = 0; i < dg; ++i) STEP_DIAG()
        dr = px - r; dc = py - c;
        for (int i = 0; i < dr; ++i) STEP_DOWN()
        for (int i = 0; i < dc; ++i) STEP_RIGHT()
        // the k-mer itself (add_kmer, or add_entry for a continued k-mer: same cells)
        const int32_t nx = sc.dpp[p];
        // add_kmer anchors the cells d = 0 .. VTX_BAND_KMER_LAST_ANCHOR(k); the cell d = k is an anchor either way (add_gap's origin /
        // set_boundaries' extension start from it: oracle/vtx_oracle.c), so the walkers of this file always go on to d = k: the constant
        // cannot move a band (tests/test_band_kat.py::test_last_anchor_alternative_never_moves_a_band; libvtx_anchor5.so agrees)
        int steps = KMER;
        if (nx >= 0) {
            const int qx = (int)(sc.mt[nx] >> 16), qy = (int)(sc.mt[nx] & 0xffff);
            if (qx == px + 1 && qy == py + 1) steps = 1;     // next match continues: advance one cell only
        }
        for (int i = 0; i < steps; ++i) STEP_DIAG()
        p = nx;
    }
    int d1 = (m - r) < (n - c) ? (m - r) : (n - c); if (d1 > VTX_BAND_LAZY_EXT(KMER)) d1 = VTX_BAND_LAZY_EXT(KMER);
    for (int i = 0; i < d1; ++i) STEP_DIAG()
#undef STEP_DIAG
#undef STEP_DOWN
#undef STEP_RIGHT
    *cA_out = cA; *cB_out = c;
    *cert_out = w.best;
    return 0;
}

// per-column ranges from the staircase (closed form, see the file header)
template <int S>
__device__ void band_ranges(const band_scratch_t<S>& sc, int cA, int cB, int m, int n, uint16_t* lo, uint16_t* hi) {
    const int rows = m + 1;
    for (int j = 0; j <= n; ++j) {
        if (j < cA - BANDW || j > cB + BANDW) { lo[j] = 0x7fff; hi[j] = 0; continue; }
        const int c0 = j - BANDW > cA ? j - BANDW : cA;
        const int c1 = j + BANDW < cB ? j + BANDW : cB;
        const int l = (int)sc.rmin[c0] - BANDW;
        const int h = (int)sc.rmax[c1] + BANDW + 1;
        lo[j] = (uint16_t)(l > 0 ? l : 0);
        hi[j] = (uint16_t)(h < rows ? h : rows);
    }
}

// One lane per task (task = 2 * record + hap).  tasks == nullptr: task = task_base + slot.
// Every task is appended to hard_list with its ranges in band[] (or the full-matrix marker) and gets its
// exact score from sw_banded_kernel.  counters[0] = hard tasks, counters[1] = capacity overflows.
// IN_LDS: the task's slab (and a copy of its read and haplotype) lives in LDS, `lds_tasks` tasks per workgroup (one
// lane each, the other lanes idle): a serial lane pays ~30 ns per dependent access instead of a round trip to HBM —
// for a FEW tasks (shallow data: a few hundred overflows per run, whose 2 ms of latency were a third of the step);
// many tasks keep the global slabs, where every lane of the chip works.
template <bool IN_LDS, int SG = 64>
__global__ __launch_bounds__(64) void band_kernel(
    const uint32_t* __restrict__ tasks, uint32_t n_tasks, uint32_t task_base,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    uint8_t* __restrict__ workspace, uint64_t ws_stride, uint32_t m_cap, uint32_t max_hap,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score,
    uint16_t* __restrict__ band, uint32_t band_stride, uint32_t* __restrict__ hard_list,
    uint32_t* __restrict__ overflow_list, uint32_t* __restrict__ counters, uint32_t lds_tasks, uint32_t max_read) {
    // lds_tasks: IN_LDS: tasks per workgroup; global slabs: ACTIVE lanes per wavefront (a power of two <= 64).  The lanes of a
    // wavefront run different serial loops (event order, Fenwick depths, chain lengths all differ): the wavefront pays for the
    // union of their paths.  A short list (config 3: 9 k tasks on a chip that holds 260 k lanes) spreads thin instead.
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_slab[];
    uint32_t slot;
    if constexpr (IN_LDS) {
        if (threadIdx.x >= lds_tasks) return;
        slot = blockIdx.x * lds_tasks + threadIdx.x;
    } else {
        if (threadIdx.x >= lds_tasks) return;
        slot = blockIdx.x * lds_tasks + threadIdx.x;
    }
    if (slot >= n_tasks) return;
    const uint32_t task = tasks ? tasks[slot] : task_base + slot;
    const uint32_t rid = task >> 1, hap = task & 1;
    const vtx_record rec = records[rid];
    const vtx_locus loc = loci[rec_locus[rid]];
    const uint8_t* x = read_arena + rec.read_off;
    const uint8_t* y = hap_arena + (hap ? loc.alt_off : loc.ref_off);
    const int m = (int)rec.read_len, n = (int)(hap ? loc.alt_len : loc.ref_len);
    if (m == 0 || n == 0) { (hap ? alt_score : ref_score)[rid] = 0; return; }
    // IN_LDS: a contiguous slab per task; global: the slabs of a wavefront's 64 tasks interleaved element by element
    constexpr int S = IN_LDS ? 1 : SG;
    band_scratch_t<S> sc;
    if constexpr (IN_LDS) {
        uint8_t* ws = lds_slab + (size_t)threadIdx.x * ws_stride;
        size_t o = sc.carve(ws, 0, m_cap, max_hap);
        // the task's read and haplotype next to its slab (every k-mer probe compares bytes of both)
        uint8_t* xl = ws + o; o += ((size_t)max_read + 3) & ~(size_t)3;
        uint8_t* yl = ws + o;
        for (int i = 0; i < m; ++i) xl[i] = x[i];
        for (int j = 0; j < n; ++j) yl[j] = y[j];
        x = xl; y = yl;
    } else {
        if constexpr (SG == 1) sc.carve(workspace + (uint64_t)slot * ws_stride, 0, m_cap, max_hap);      // a contiguous slab per task
        else sc.carve(workspace + (uint64_t)blockIdx.x * 64u * ws_stride, threadIdx.x, m_cap, max_hap);
    }
    int32_t cert = 0;
    int cA = 0, cB = 0;
    const int rc = band_task(x, m, y, n, sc, m_cap, &cert, &cA, &cB);
    if (rc) { overflow_list[atomicAdd(&counters[1], 1u)] = task; return; }
    const uint32_t h = atomicAdd(&counters[0], 1u);
    hard_list[h] = task;
    uint16_t* lo = band + (size_t)h * 2 * band_stride;
    if (cert == INT32_MAX) { lo[0] = BAND_FULL_MATRIX; return; }   // no k-mer match: Band::full_matrix
    band_ranges(sc, cA, cB, m, n, lo, lo + band_stride);
}

// =============================================================================================
// band_coop_kernel — the general path for ONE task per WAVEFRONT (round 3; band_kernel above is one task per lane).
// Tasks reach the general path because their piece lists overflow: repeats (satellites, poly-A, microsatellites) with hundreds
// of k-mer matches.  On loci drawn from real sequence that is 12 % of the tasks, and one serial lane per task with its slab in
// global memory took 650 of the step's 800 ms.  Here the 64 lanes share the task, everything lives in LDS:
//   A  k-mer matches: a lane per read row compares its 6 bytes with every haplotype 6-mer (an LDS broadcast per column) —
//      count pass, prefix sum over the rows, write pass: the list comes out in (x, y) order, as find_kmer_matches gives it.
//   B  sdpkpp row by row.  The reference processes events in (x, y) order, END events (x + k, y + k) before START events at
//      equal coordinates.  All END events of row X (the matches of row X - k) first, then all START events of row X, gives
//      every START the same query result: an END of the same row with a larger column is outside its prefix either way.  The
//      max-Fenwick tree holds v << 16 | match index (v = dp + xe + ye < 2^16, index < 2^16: ent_gt is the unsigned order), so
//      an update is ds_max_u32 along the tree path and the lanes of a row need no other ordering.  The continuation partner
//      of a START (the match one step up its diagonal) is a binary search in the previous row.
//   C  the staircase: traceback by one lane, then a lane per chain link writes the anchor rows of its columns (consecutive
//      links own consecutive column ranges; a vertical run raises rmax of ONE column, possibly the previous link's last: an
//      atomic max in a second pass), then a lane per column turns rmin / rmax into the band's row range (band_ranges).
// Same outputs as band_kernel: hard_list + band[] (or the full-matrix marker), overflow_list for tasks beyond CAP matches
// or the LDS shapes (they take band_kernel).  counters[0] = hard tasks, counters[1] = overflows.
// =============================================================================================
__device__ __forceinline__ void wave_sync();
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v = max(v, __shfl_xor(v, d));
    return v;
}
#define COOP_MAX_READ 256u
#define COOP_NONE 0xffffu
// bytes of LDS per task for MC matches
extern "C" size_t vtxk_band_coop_lds(uint32_t max_hap, uint32_t mc) {
    const size_t cols = ((size_t)max_hap + KMER + 9) & ~(size_t)1;
    // mt, dpv: 4 B per match; cont: 2 B; row_off: u16 x (COOP_MAX_READ + 4); path: u16 x COOP_MAX_READ; tree: u32 x cols;
    // rmin, rmax: u32 x cols each (the 8-byte haplotype words of phase A alias them: 8-byte aligned)
    // lastm (phase B: the match of the last two rows at every column — the continuation partner without a search) aliases them too
    const size_t o = (size_t)mc * 10 + 2 * (COOP_MAX_READ + 4) + 2 * COOP_MAX_READ + 4 * cols + 8 * cols;
    return (o + 63) & ~(size_t)63;
}

// G lanes per task (64 / G tasks per wavefront, in lockstep), MC matches per task
template <int G, int MC>
__global__ __launch_bounds__(64) void band_coop_kernel(
    const uint32_t* __restrict__ tasks, uint32_t n_tasks,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena, uint32_t max_hap, uint32_t task_bytes,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score,
    uint16_t* __restrict__ band, uint32_t band_stride, uint32_t* __restrict__ hard_list,
    uint32_t* __restrict__ overflow_list, uint32_t* __restrict__ counters, uint32_t ablate) {
    extern __shared__ __attribute__((aligned(16))) uint8_t coop_lds[];
    constexpr int TPW = 64 / G;                                       // tasks per wavefront
    const int grp = threadIdx.x / G, l = threadIdx.x % G;
    const uint32_t cols = (max_hap + KMER + 9u) & ~1u;
    uint8_t* base = coop_lds + (size_t)grp * task_bytes;
    uint32_t* mt = (uint32_t*)base;                                   // x << 16 | y
    uint32_t* dpv = mt + MC;                                          // dp << 16 | predecessor (COOP_NONE: none)
    uint16_t* cont = (uint16_t*)(dpv + MC);                           // the match one step up the diagonal
    uint16_t* row_off = cont + MC;                                    // first match of row i; [rows] = M
    uint16_t* path = row_off + (COOP_MAX_READ + 4);
    uint32_t* tree = (uint32_t*)(path + COOP_MAX_READ);
    uint32_t* rmin = tree + cols;
    uint32_t* rmax = rmin + cols;
    uint32_t* lastm = rmin;                                           // phase B only: [2][cols]: (row + 1) << 16 | match index, by row parity
    uint64_t* y6 = (uint64_t*)rmin;                                   // phase A only
    __shared__ uint32_t s_len[TPW], s_slot[TPW];

    const uint32_t slot = blockIdx.x * TPW + (uint32_t)grp;
    bool live = slot < n_tasks;
    uint32_t task = 0, rid = 0, hap = 0;
    const uint8_t *x = read_arena, *y = hap_arena;
    int m = 0, n = 0;
    if (live) {
        task = tasks[slot];
        rid = task >> 1; hap = task & 1;
        const vtx_record rec = records[rid];
        const vtx_locus loc = loci[rec_locus[rid]];
        x = read_arena + rec.read_off;
        y = hap_arena + (hap ? loc.alt_off : loc.ref_off);
        m = (int)rec.read_len; n = (int)(hap ? loc.alt_len : loc.ref_len);
        if (m == 0 || n == 0) { if (l == 0) (hap ? alt_score : ref_score)[rid] = 0; live = false; }
        else if ((uint32_t)m > COOP_MAX_READ || (uint32_t)n > max_hap) {   // beyond the LDS shapes: band_kernel
            if (l == 0) overflow_list[atomicAdd(&counters[1], 1u)] = task;
            live = false;
        }
    }
    if (!live) { m = 0; n = 0; }
    // ---- A: matches ----
    const int rows = m >= KMER ? m - KMER + 1 : 0, ycols = (n >= KMER && rows > 0) ? n - KMER + 1 : 0;
    auto six = [](const uint8_t* p) -> uint64_t {
        return (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24) | ((uint64_t)p[4] << 32) | ((uint64_t)p[5] << 40);
    };
    for (int j = l; j < ycols; j += G) y6[j] = six(y + j);
    constexpr int RPL = COOP_MAX_READ / G;                            // rows per lane: row = l + G k
    uint64_t x6[RPL];
    uint32_t cnt[RPL];
#pragma unroll
    for (int k = 0; k < RPL; ++k) { const int i = l + G * k; x6[k] = i < rows ? six(x + i) : ~0ull; cnt[k] = 0; }
    wave_sync();
    const int ycols_w = wave_max_i32(ycols);
    // (eight columns per trip: their LDS loads go out together — one load per trip was a round trip per column)
    for (int j0 = 0; j0 < ycols_w; j0 += 8) {
        uint64_t w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = j0 + u < ycols ? y6[j0 + u] : ~1ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < RPL; ++k) cnt[k] += (x6[k] == w[u]) ? 1u : 0u;
        }
    }
    // exclusive prefix over the rows of the task (the rounds k in order, the group's lanes in order within a round)
    uint32_t off[RPL];
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        uint32_t inc = cnt[k];
#pragma unroll
        for (int d = 1; d < G; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, G); if (l >= d) inc += o; }
        off[k] = run + inc - cnt[k];
        run += (uint32_t)__shfl((int)inc, G - 1, G);
    }
    const uint32_t M = run;
    if (live && M > (uint32_t)MC) { if (l == 0) overflow_list[atomicAdd(&counters[1], 1u)] = task; live = false; }
    if (live && M == 0) {                                              // Band::full_matrix
        if (l == 0) { const uint32_t h = atomicAdd(&counters[0], 1u); hard_list[h] = task; band[(size_t)h * 2 * band_stride] = BAND_FULL_MATRIX; }
        live = false;
    }
    if (!__any(live)) return;
    const int rows_l = live ? rows : 0;                                // (a task that left keeps its lanes idle)
    const int m_l = live ? m : -1;
    if (live) {
#pragma unroll
        for (int k = 0; k < RPL; ++k) { const int i = l + G * k; if (i < rows) row_off[i] = (uint16_t)off[k]; }
        if (l == 0) row_off[rows] = (uint16_t)M;
    }
    for (int j0 = 0; j0 < ycols_w; j0 += 8) {
        uint64_t w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = (live && j0 + u < ycols) ? y6[j0 + u] : ~1ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < RPL; ++k) if (x6[k] == w[u]) { mt[off[k]] = ((uint32_t)(l + G * k) << 16) | (uint32_t)(j0 + u); ++off[k]; }
        }
    }
    wave_sync();                                                       // (y6 is dead: rmin / rmax may be written)
    if (VTX_ABLATE(ablate) == 1) { if (mt[0] == 0x7fffffffu) counters[2] = 1; return; }     // (profiling aid: phase A only; results are wrong)
    // ---- B: sdpkpp ----
    const int tn = n + KMER + 2;                                      // < 1024 (the host launches this kernel for haplotypes <= 1000 bases): tree paths of <= 10 nodes
    if (live) for (int i = l; i <= tn; i += G) tree[i] = 0;
    if (live) for (int i = l; i < 2 * (int)cols; i += G) lastm[i] = 0;
    uint32_t best = 0;                                                 // dp << 16 | index of the best END so far (this lane's)
    wave_sync();
    const int m_w = wave_max_i32(m_l);
    // One iteration = the START events of row X - 1 (upper half of the lanes) and the END events of row X - k (lower half), in ONE
    // instruction stream: two rounds of LDS loads, then the writes and the tree updates.  START(X - 1) must see the END events up
    // to row X - 1 (earlier iterations) and none of row X: its tree loads are issued before this iteration's atomics, and the LDS
    // keeps a wavefront's order.  END(X) reads the dp of its continuation partner, settled by END(X - 1).  Rows k apart never
    // touch the same match.  (END(X) then START(X) as two passes with a fence each was twice the chain of round trips.)
    constexpr uint32_t HALF = G / 2;
    const bool is_start = (uint32_t)l >= HALF;
    const uint32_t hl = (uint32_t)l & (HALF - 1);
    auto do_start = [&](uint32_t p, int sr, uint32_t yv, const uint32_t* tv, uint32_t lm) {
        uint32_t bq = 0;
#pragma unroll
        for (int u = 0; u < 10; ++u) bq = tv[u] > bq ? tv[u] : bq;
        uint32_t dv = KMER, pr = COOP_NONE;
        if (bq) {
            const int cand = (int)(bq >> 16) - 5 - (sr + (int)yv) + KMER;            // gap_open -5, extend -1 per unit of (x + y)
            if (cand >= (int)dv) { dv = (uint32_t)cand; pr = bq & 0xffffu; }          // (a tie goes to the tree's entry: its index > -1)
        }
        const uint32_t c = (sr > 0 && (lm >> 16) == (uint32_t)sr) ? (lm & 0xffffu) : COOP_NONE;   // tag = row + 1 of the writer
        dpv[p] = (dv << 16) | pr;
        cont[p] = (uint16_t)c;
        lastm[(sr & 1) * cols + yv] = ((uint32_t)(sr + 1) << 16) | p;
    };
    auto do_end = [&](uint32_t p, int X, uint32_t yv, uint32_t d0, uint32_t c, uint32_t dc) {
        uint32_t dv = d0 >> 16, pr = d0 & 0xffffu;
        if (c != COOP_NONE) {
            const uint32_t cand = (dc >> 16) + 1u;
            if (cand > dv || (cand == dv && (pr == COOP_NONE || c > pr))) { dv = cand; pr = c; dpv[p] = (dv << 16) | pr; }
        }
        const uint32_t v = dv + (uint32_t)X + yv + KMER;                             // dp + (x + k) + (y + k)
        const uint32_t packed = (v << 16) | p;
        int i = (int)yv + KMER + 1;
#pragma unroll
        for (int u = 0; u < 10; ++u) { if (i <= tn) atomicMax(&tree[i], packed); i += i & (-i); }     // (nine nodes at most for tn < 512)
        const uint32_t me = (dv << 16) | p;
        best = me > best ? me : best;
    };
    for (int X = 0; X <= m_w; ++X) {
        const int sr = X - 1, er = X - KMER;
        uint32_t sb = 0, se = 0, eb = 0, ee = 0;
        if (sr >= 0 && sr < rows_l) { sb = row_off[sr]; se = row_off[sr + 1]; }
        if (er >= 0 && er < rows_l) { eb = row_off[er]; ee = row_off[er + 1]; }
        if (!__any(se - sb > HALF || ee - eb > HALF)) {
            const uint32_t p = (is_start ? sb : eb) + hl;
            const bool act = p < (is_start ? se : ee);
            uint32_t w = 0, d0 = 0, c = COOP_NONE;
            if (act) { w = mt[p]; d0 = dpv[p]; c = cont[p]; }                        // (a START lane does not use d0 and c)
            const uint32_t yv = w & 0xffffu;
            uint32_t tv[10], x0 = 0;
            {
                const bool st = act && is_start;
                int i = (int)yv + 1;
#pragma unroll
                for (int u = 0; u < 10; ++u) { tv[u] = (st && i > 0) ? tree[i] : 0u; i -= i & (-i); }
                if (st) { if (yv > 0) x0 = lastm[((sr + 1) & 1) * cols + yv - 1]; }   // row sr - 1, column y - 1
                else if (act && c != COOP_NONE) x0 = dpv[c];
            }
            if (act && is_start) do_start(p, sr, yv, tv, x0);
            if (act && !is_start) do_end(p, X, yv, d0, c, x0);
        } else {
            // a row with more events than half a wavefront holds (satellites): the two kinds in turn, START first
            for (uint32_t p = sb + (uint32_t)l; p < se; p += G) {
                const uint32_t yv = mt[p] & 0xffffu;
                uint32_t tv[10];
                int i = (int)yv + 1;
#pragma unroll
                for (int u = 0; u < 10; ++u) { tv[u] = i > 0 ? tree[i] : 0u; i -= i & (-i); }
                const uint32_t lm = yv > 0 ? lastm[((sr + 1) & 1) * cols + yv - 1] : 0u;
                do_start(p, sr, yv, tv, lm);
            }
            wave_sync();
            for (uint32_t p = eb + (uint32_t)l; p < ee; p += G) {
                const uint32_t yv = mt[p] & 0xffffu, d0 = dpv[p], c = cont[p];
                do_end(p, X, yv, d0, c, c != COOP_NONE ? dpv[c] : 0u);
            }
        }
        wave_sync();
    }
#pragma unroll
    for (int d = 1; d < G; d <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)best, d, G); best = o > best ? o : best; }
    // (the reference starts `best` at (k, match 0): an END beats it iff (dp, index) > (k, 0) — dp >= k always, so the maximum
    //  over (dp, index) is the same entry)
    if (VTX_ABLATE(ablate) == 2) { if (best == 0x7fffffffu) counters[2] = 1; return; }      // (profiling aid: phases A + B)
    // ---- traceback (one lane per task), the chain reversed in path[] ----
    if (live && l == 0) {
        uint32_t cur = best & 0xffffu, len = 0;
        while (cur != COOP_NONE && len < COOP_MAX_READ) { path[len++] = (uint16_t)cur; cur = dpv[cur] & 0xffffu; }
        s_len[grp] = len;
        s_slot[grp] = atomicAdd(&counters[0], 1u);
    }
    wave_sync();
    if (!live) return;                                                 // (no wave_sync below this point is needed by other groups... see pass 2)
    const int L = (int)s_len[grp];
    const uint32_t hslot = s_slot[grp];
    auto link = [&](int t) -> uint32_t { return mt[path[L - 1 - t]]; };  // link t of the chain, forward
    const int fx = (int)(link(0) >> 16), fy = (int)(link(0) & 0xffffu);
    const int d0 = min(min(fx, fy), (int)VTX_BAND_LAZY_EXT(KMER));
    const int cA = fy - d0;
    int lx, ly;                                                        // cell behind the last chained k-mer
    {
        const uint32_t w = link(L - 1);
        lx = (int)(w >> 16) + KMER; ly = (int)(w & 0xffffu) + KMER;
    }
    const int d1 = min(min(m - lx, n - ly), (int)VTX_BAND_LAZY_EXT(KMER));
    const int cB = ly + d1;
    // pass 1: every link writes the anchor row of its own columns
    for (int t = l; t < L + 1; t += G) {
        if (t == L) {                                                  // the lazy extension behind the chain
            for (int i = 1; i <= d1; ++i) { rmin[ly + i] = (uint32_t)(lx + i); rmax[ly + i] = (uint32_t)(lx + i); }
            continue;
        }
        const uint32_t w = link(t);
        const int px = (int)(w >> 16), py = (int)(w & 0xffffu);
        int ax, ay;
        if (t == 0) { ax = fx - d0; ay = fy - d0; rmin[ay] = (uint32_t)ax; rmax[ay] = (uint32_t)ax; }
        else {
            const uint32_t q = link(t - 1);
            const int qx = (int)(q >> 16), qy = (int)(q & 0xffffu);
            const int sq = (px == qx + 1 && py == qy + 1) ? 1 : KMER;
            ax = qx + sq; ay = qy + sq;
        }
        const int dr = px - ax, dc = py - ay, dg = min(dr, dc);
        for (int i = 1; i <= dg; ++i) { rmin[ay + i] = (uint32_t)(ax + i); rmax[ay + i] = (uint32_t)(ax + i); }
        for (int c = ay + dg + 1; c <= py; ++c) { rmin[c] = (uint32_t)px; rmax[c] = (uint32_t)px; }       // horizontal remainder (dc > dr)
        int st = KMER;
        if (t + 1 < L) { const uint32_t nx = link(t + 1); if ((int)(nx >> 16) == px + 1 && (int)(nx & 0xffffu) == py + 1) st = 1; }
        for (int i = 1; i <= st; ++i) { rmin[py + i] = (uint32_t)(px + i); rmax[py + i] = (uint32_t)(px + i); }
    }
    wave_sync();
    // pass 2: vertical remainders (dr > dc) raise rmax of the column the run happens in
    for (int t = l; t < L; t += G) {
        const uint32_t w = link(t);
        const int px = (int)(w >> 16), py = (int)(w & 0xffffu);
        int ax, ay;
        if (t == 0) { ax = fx - d0; ay = fy - d0; }
        else {
            const uint32_t q = link(t - 1);
            const int qx = (int)(q >> 16), qy = (int)(q & 0xffffu);
            const int sq = (px == qx + 1 && py == qy + 1) ? 1 : KMER;
            ax = qx + sq; ay = qy + sq;
        }
        const int dr = px - ax, dc = py - ay;
        if (dr > dc) atomicMax(&rmax[ay + dc], (uint32_t)px);
    }
    wave_sync();
    // ---- per-column ranges (band_ranges) ----
    if (l == 0) hard_list[hslot] = task;
    uint16_t* lo = band + (size_t)hslot * 2 * band_stride;
    uint16_t* hi = lo + band_stride;
    const int nrows = m + 1;
    for (int j = l; j <= n; j += G) {
        if (j < cA - BANDW || j > cB + BANDW) { lo[j] = 0x7fff; hi[j] = 0; continue; }
        const int c0 = j - BANDW > cA ? j - BANDW : cA;
        const int c1 = j + BANDW < cB ? j + BANDW : cB;
        const int lw = (int)rmin[c0] - BANDW;
        const int hw = (int)rmax[c1] + BANDW + 1;
        lo[j] = (uint16_t)(lw > 0 ? lw : 0);
        hi[j] = (uint16_t)(hw < nrows ? hw : nrows);
    }
}

// a wavefront per task; tier 0: 512 matches (8.7 KB of LDS: 18 tasks per CU), tier 1: 1024 (13.8 KB: 11), tier 2: 4096 (3 per CU).
// The kernel is bound by its chain of LDS round trips per read row times the tasks a CU's LDS holds, not by instruction issue:
// on the real-sequence workload at 30 k loci (tools/coop_ablate.sh) one 1024-match tier with END and START as two fenced
// passes per row took 157 ms for 1.75 M tasks (matches 39, sdpkpp 110, staircase 8); 32 or 16 lanes per task (two or four tasks
// per wavefront in lockstep, the template's G) were SLOWER, 171 and 258 ms — the same number of tasks in flight, longer rows;
// START and END of neighbouring rows in one instruction stream plus the 512-match tier: 90 + 11 ms (the step: 257 ms with the
// serial kernel alone, 234 ms with the first version, 200 ms now; full size 800 -> 590 ms).
extern "C" hipError_t vtxk_launch_band_coop(int tier, const uint32_t* tasks, uint32_t n_tasks, const vtx_record* records,
                                            const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                            const uint8_t* hap_arena, uint32_t max_hap, int32_t* ref_score, int32_t* alt_score,
                                            uint16_t* band, uint32_t band_stride, uint32_t* hard_list, uint32_t* overflow_list,
                                            uint32_t* counters, hipStream_t s) {
    if (!n_tasks) return hipSuccess;
    const uint32_t mc = tier == 0 ? 512u : (tier == 1 ? 1024u : 4096u), tpw = 1u;
    static const uint32_t ablate = VTX_DEV_ENV("VTX_COOP_ABLATE") ? (uint32_t)atoi(VTX_DEV_ENV("VTX_COOP_ABLATE")) : 0u;     // profiling aid
    const size_t task_bytes = vtxk_band_coop_lds(max_hap, mc), shmem = task_bytes * tpw;
    if (shmem > 64 * 1024) return hipErrorInvalidValue;
#define LAUNCH_COOP(G, MC)                                                                                               \
    {                                                                                                                    \
        if (shmem > 48 * 1024) {                                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)band_coop_kernel<G, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
            if (e != hipSuccess) return e;                                                                               \
        }                                                                                                                \
        hipLaunchKernelGGL((band_coop_kernel<G, MC>), dim3((n_tasks + tpw - 1) / tpw), dim3(64), shmem, s, tasks, n_tasks, records, rec_locus, \
                           loci, read_arena, hap_arena, max_hap, (uint32_t)task_bytes, ref_score, alt_score, band, band_stride,   \
                           hard_list, overflow_list, counters, ablate);                                                  \
    }
    if (tier == 0) LAUNCH_COOP(64, 512) else if (tier == 1) LAUNCH_COOP(64, 1024) else LAUNCH_COOP(64, 4096)
#undef LAUNCH_COOP
    return hipGetLastError();
}

// active lanes per wavefront of band_kernel<false> for a list of n_tasks (the workspace holds 64 slabs per wavefront either way)
extern "C" uint32_t vtxk_band_lanes(uint32_t n_tasks) {
    static const int forced = VTX_DEV_ENV("VTX_BAND_LANES") ? atoi(VTX_DEV_ENV("VTX_BAND_LANES")) : 0;       // experiment knob
    if (forced > 0) {                                         // a power of two in [4, 64]: anything else would leave tasks unscored
        uint32_t lanes = 4;
        while (lanes < 64 && lanes < (uint32_t)forced) lanes <<= 1;
        return lanes;
    }
    uint32_t lanes = 64;
    while (lanes > 4 && (n_tasks + lanes - 1) / lanes < 4096u) lanes >>= 1;        // >= 4096 wavefronts: 16 per CU
    return lanes;
}
extern "C" size_t vtxk_band_ws_stride(uint32_t m_cap, uint32_t max_hap) {
    size_t o = HASH_SIZE * 2 + 3 * ((size_t)max_hap + 2) * 2;
    o = (o + 15) & ~(size_t)15;
    o += 2 * ((size_t)max_hap + KMER + 4) * 4 + 4 * (size_t)m_cap * 4;
    return (o + 63) & ~(size_t)63;
}

// lds_stride: bytes per task of the in-LDS variant (slab for m_cap matches + read + haplotype)
extern "C" size_t vtxk_band_lds_stride(uint32_t m_cap, uint32_t max_hap, uint32_t max_read) {
    return (vtxk_band_ws_stride(m_cap, max_hap) + (((size_t)max_read + 3) & ~(size_t)3) + max_hap + 15) & ~(size_t)15;
}

// in_lds != 0: the in-LDS variant (workspace unused); the caller has checked that at least one task fits 160 KiB
extern "C" hipError_t vtxk_launch_band(const uint32_t* tasks, uint32_t n_tasks, uint32_t task_base,
                                       const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                       const uint8_t* read_arena, const uint8_t* hap_arena, uint8_t* workspace,
                                       uint64_t ws_stride, uint32_t m_cap, uint32_t max_hap, int32_t* ref_score,
                                       int32_t* alt_score, uint16_t* band, uint32_t band_stride, uint32_t* hard_list,
                                       uint32_t* overflow_list, uint32_t* counters, int in_lds, uint32_t max_read,
                                       hipStream_t s) {
    if (!n_tasks) return hipSuccess;
    if (in_lds) {
        const size_t stride = vtxk_band_lds_stride(m_cap, max_hap, max_read);
        const uint32_t per_wg = (uint32_t)std::min<size_t>(64, (160 * 1024 - 512) / stride);
        if (!per_wg) return hipErrorInvalidValue;
        const size_t shmem = (size_t)per_wg * stride;
        hipError_t e = hipFuncSetAttribute((const void*)band_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(band_kernel<true>, dim3((n_tasks + per_wg - 1) / per_wg), dim3(64), shmem, s, tasks, n_tasks,
                           task_base, records, rec_locus, loci, read_arena, hap_arena, workspace, (uint64_t)stride, m_cap,
                           max_hap, ref_score, alt_score, band, band_stride, hard_list, overflow_list, counters, per_wg, max_read);
        return hipGetLastError();
    }
    const uint32_t lanes = vtxk_band_lanes(n_tasks);
    if (lanes < 64)
        hipLaunchKernelGGL((band_kernel<false, 1>), dim3((n_tasks + lanes - 1) / lanes), dim3(64), 0, s, tasks, n_tasks, task_base, records,
                           rec_locus, loci, read_arena, hap_arena, workspace, ws_stride, m_cap, max_hap, ref_score, alt_score,
                           band, band_stride, hard_list, overflow_list, counters, lanes, 0u);
    else
        hipLaunchKernelGGL((band_kernel<false, 64>), dim3((n_tasks + lanes - 1) / lanes), dim3(64), 0, s, tasks, n_tasks, task_base, records,
                           rec_locus, loci, read_arena, hap_arena, workspace, ws_stride, m_cap, max_hap, ref_score, alt_score,
                           band, band_stride, hard_list, overflow_list, counters, lanes, 0u);
    return hipGetLastError();
}

#ifndef VTX_PS
#define VTX_PS 15
#endif
#ifndef VTX_WPE
#define VTX_WPE 4   // wavefronts per SIMD band_run_kernel is compiled and launched for
#endif
// PSV (template parameter of band_run_kernel and its list functions): per-lane LDS entries, pieces + segments.
// 15 at five wavefronts per SIMD (7.7 KiB per wavefront), 12 at six (6.1 KiB) — VTX_PS is the default / LDS-table value.
#define LG 64       // jump-log entries per task (global)
// pieces run_compact may drop from a task's list and still leave it to the pending kernel: list + spill = the 32 lanes of
// band_pending_kernel
#define SPILL_OF(psv) (32 - (psv))
#define PS_MIN 12
#define TASK_WORDS (LG * 2 + SPILL_OF(PS_MIN) * 2)   // scratch of one RESIDENT lane (global, reused block after block): jump log, spilled pieces
#define PEND_WORDS (2 + 2 * 32 + (4 * SG + 6))       // record of a pending task: header, cert, 32 pieces (list entries, then spilled ones), staircase
#define SG 10       // chain segments per task
static_assert(VTX_PS >= PS_MIN && VTX_PS <= 24, "list + spill entries of a task are the 32 lanes of band_pending_kernel");
#define NONE_ID 0xffffffffu
#define CH_END 0xffffu     // end of a k-mer chain / empty bucket

// position part of a tagged head word (build_tables: bits 12-15 carry a tag for band_diag_kernel); CH_END stays CH_END
__device__ __forceinline__ uint32_t head_pos(uint32_t raw) { return raw == 0xffffu ? 0xffffu : (raw & 0xfffu); }
__device__ __forceinline__ uint32_t kw_hash(uint32_t lo, uint32_t hi, uint32_t head_mask) {
    // one 32-bit multiply (a quarter-rate instruction): the two bytes of `hi` are folded in with a rotation first
    const uint32_t h = (lo ^ (hi << 11) ^ (hi >> 3)) * 0x9E3779B1u;
    return (h >> 18) & head_mask;
}

// LDS table of one haplotype (band_run_kernel): ent[y] = {k-mer bytes 0-3, bytes 4-5 | next y << 16} (one 8-byte load per
// chain step), head[n_heads] u16, bytes[max_hap + 8] raw haplotype bytes (staircase walk), fb[max_hap + 8] flag bytes:
// fb[y] = (byte y & 0x7f) | 0x80 if the k-mer that ENDS at y is unique in the haplotype (continuation shortcut).
// ... and uq[] (vtx_fast_core.h: tab_uq_off): one bit per position, set if the k-mer STARTING there is unique, behind
// UQ_PAD_WORDS zero words (band_diag_kernel reads 192 bits of it at the bit offset of its diagonal).
// (in_lds: a table band_run_kernel builds for itself in LDS ends behind pb[] — the twin list and the three-row sets are band_diag_kernel's)
static size_t band_table_stride(uint32_t max_hap, uint32_t n_heads, bool in_lds = false) {
    return in_lds ? vtxf::tab_stride_lds(max_hap, n_heads) : vtxf::tab_stride(max_hap, n_heads);
}
#define TB_ENT(tb) ((uint2*)(tb))
#define TB_HEAD(tb) ((uint16_t*)((tb) + (size_t)max_hap * 8))
#define TB_BYTES(tb) ((uint8_t*)(TB_HEAD(tb) + n_heads))
#define TB_FB(tb) (TB_BYTES(tb) + max_hap + 8)
#define TB_UQ(tb) ((uint32_t*)((tb) + vtxf::tab_uq_off(max_hap, n_heads)))
#define TB_PB(tb) ((uint32_t*)((tb) + vtxf::tab_pb_off(max_hap, n_heads)))
#define TB_TW(tb) ((uint8_t*)((tb) + vtxf::tab_tw_off(max_hap, n_heads)))
#define TB_T3(tb) ((uint32_t*)((tb) + vtxf::tab_t3_off(max_hap, n_heads)))

// LDS exchanged between the lanes of ONE wavefront (its instructions reach the LDS in program order): a compiler-level
// fence is all the ordering it needs
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef VTX_DEVTOOLS        // (behind round 3's table kernel, the reference of band_tables_kernel — which builds both inside its flag loop)
// The three-row presence sets of a table in LDS (vtx_fast_core.h: Tab t3[]; round 6), by one wavefront; t3[] was zeroed with the bitmaps.
__device__ __forceinline__ void build_t3_wave(uint8_t* tb, uint32_t hn, uint32_t max_hap, uint32_t n_heads, int tid) {
    const uint2* ent = TB_ENT(tb);
    uint32_t* t3 = TB_T3(tb);
    const int nk = hn >= (uint32_t)KMER ? (int)hn - KMER + 1 : 0;
    for (int y = tid; y < nk && vtxf::T3_BYTES != 0; y += 64) {
        const uint32_t c = vtxf::kw_code(ent[y].x, ent[y].y & 0xffffu);
        atomicOr(&t3[vtxf::t3_word_a(c)], 1u << vtxf::t3_bit_a(c));
        atomicOr(&t3[vtxf::t3_word_b(c)], 1u << vtxf::t3_bit_b(c));
        atomicOr(&t3[vtxf::t3_word_c(c)], 1u << vtxf::t3_bit_c(c));
    }
}

// The twin list of a finished table in LDS (vtx_fast_core.h: Tab tw[]; round 6), by ONE wavefront (lane = tid): the pairs (y, y'),
// y != y', of positions with the same k-mer in (y, y') order — rounds of 64 positions, a prefix sum of the twin counts per round, the
// chains ascend so every position writes its twins in order.  hib: the haplotype holds a byte >= 0x80 (no uniqueness flags, no list).
// The tw area was zeroed with the bitmaps; the head words carry their tags already.
__device__ __forceinline__ void build_twins_wave(uint8_t* tb, uint32_t hn, bool hib, uint32_t max_hap, uint32_t n_heads, int tid) {
    const uint2* ent = TB_ENT(tb);
    const uint16_t* head = TB_HEAD(tb);
    const uint32_t* uq = TB_UQ(tb) + vtxf::UQ_PAD_WORDS;
    uint8_t* tw = TB_TW(tb);
    const int nk = hn >= (uint32_t)KMER ? (int)hn - KMER + 1 : 0;
    bool ok = !hib && hn <= 256u;
    uint32_t total = 0;
    for (int base = 0; base < nk && ok; base += 64) {
        const int y = base + tid;
        uint32_t c = 0, first = CH_END;
        uint2 k = make_uint2(0, 0);
        if (y < nk && !((uq[y >> 5] >> (y & 31)) & 1u)) {
            k = ent[y];
            first = head_pos(head[kw_hash(k.x, k.y & 0xffffu, n_heads - 1)]);
            for (uint32_t e = first; e != CH_END; e = ent[e].y >> 16) c += (ent[e].x == k.x && ((ent[e].y ^ k.y) & 0xffffu) == 0);
            c -= 1u;                                                      // (the position itself)
        }
        uint32_t inc = c;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, sft); if (tid >= sft) inc += t; }
        uint32_t off = total + inc - c;
        total += (uint32_t)__shfl((int)inc, 63);
        if (total > vtxf::TW_MAX) { ok = false; break; }                  // (wave-uniform)
        if (c)
            for (uint32_t e = first; e != CH_END; e = ent[e].y >> 16)
                if (e != (uint32_t)y && ent[e].x == k.x && ((ent[e].y ^ k.y) & 0xffffu) == 0) {
                    tw[8 + 2 * off] = (uint8_t)y; tw[9 + 2 * off] = (uint8_t)e; ++off;
                }
    }
    wave_sync();
    if (!ok) for (uint32_t i = tid; i < vtxf::TW_BYTES / 4; i += 64) ((uint32_t*)tw)[i] = 0;
    wave_sync();
    if (tid == 0) tw[0] = ok ? (uint8_t)total : (uint8_t)vtxf::TW_NONE;
}
#endif

// Tail of band_run_kernel: traceback through the jump log (chain = a few diagonal segments), walk of
// the anchor staircase (its local score = the lower bound `cert`), polyline for hard tasks.
// Returns 0: certified (cert == ub, so banded == full == ub); 1: hard (verts / *nv_out filled);
// 2: capacity exceeded.
__device__ int band_finish(const uint32_t* mylog, uint32_t ls, uint32_t lg_n, uint32_t best_id, const uint8_t* x,
                           const uint8_t* yb, uint32_t m, uint32_t n, int32_t ub, uint32_t* verts_out,
                           uint32_t* nv_out, int32_t* cert_out) {
        // ---- traceback through the jump log: chain = diagonal segments (x0, y0, len), last first ----
        uint32_t seg_xy[SG], seg_len[SG];
        uint32_t n_seg = 0;
        uint32_t cur = best_id;
        bool bad = false;
        while (true) {
            const int32_t cx = (int32_t)(cur >> 16), cy = (int32_t)(cur & 0xffff);
            int32_t bx = -1; uint32_t bprev = NONE_ID, bid = 0;
            for (uint32_t i = 0; i < lg_n; ++i) {
                const uint32_t p = mylog[i * ls];
                const int32_t px = (int32_t)(p >> 16), py = (int32_t)(p & 0xffff);
                if (py - px == cy - cx && px <= cx && px > bx) { bx = px; bprev = mylog[i * ls + 1]; bid = p; }
            }
            if (bx < 0 || n_seg == SG) { bad = true; break; }
            seg_xy[n_seg] = bid; seg_len[n_seg] = (uint32_t)(cx - bx + 1); ++n_seg;
            if (bprev == NONE_ID) break;
            cur = bprev;
        }
        if (bad) return 2;
        // ---- staircase walk: certificate, and the polyline if the task turns out hard ----
        // One vertex after every non-empty piece (pure diagonal / vertical / horizontal).
        uint32_t* verts = verts_out;
        uint32_t nv = 0;
        const uint32_t fst = seg_xy[n_seg - 1];
        const int fx = (int)(fst >> 16), fy = (int)(fst & 0xffff);
        int d0 = fx < fy ? fx : fy; if (d0 > VTX_BAND_LAZY_EXT(KMER)) d0 = VTX_BAND_LAZY_EXT(KMER);
        int r = fx - d0, c = fy - d0;
        walk_state w = {0, -100000, 0, 0};
#define EMIT() verts[nv++] = ((uint32_t)r << 16) | (uint32_t)c
        EMIT();
        {
            // the (2w+1)-square around the first anchor is in band, so the path may start up to w cells further up its
            // diagonal (not part of the staircase: no vertex) — a mismatch a few bases inside the read's end no longer
            // costs the certificate the bases beyond it
            const int t0 = min(min(r, c), BANDW);
            int rr = r - t0, cc = c - t0;
            for (int i = 0; i < t0; ++i) { ++rr; ++cc; walk_diag(w, x[rr - 1] == yb[cc - 1]); }
        }
        for (int i = 0; i < d0; ++i) { ++r; ++c; walk_diag(w, x[r - 1] == yb[c - 1]); }
        if (d0 > 0) EMIT();
        for (uint32_t sgi = n_seg; sgi-- > 0;) {
            const int px = (int)(seg_xy[sgi] >> 16), py = (int)(seg_xy[sgi] & 0xffff);
            int dr = px - r, dc = py - c;
            const int dg = dr < dc ? dr : dc;
            for (int i = 0; i < dg; ++i) { ++r; ++c; walk_diag(w, x[r - 1] == yb[c - 1]); }
            if (dg > 0) EMIT();
            dr = px - r; dc = py - c;
            for (int i = 0; i < dr; ++i) { ++r; walk_gap(w, 1); }
            if (dr > 0) EMIT();
            for (int i = 0; i < dc; ++i) { ++c; walk_gap(w, 2); }
            if (dc > 0) EMIT();
            // the segment's k-mer cells: len + K - 1 diagonal steps, every one an exact match
            const int run = (int)seg_len[sgi] - 1 + KMER;
            const int32_t v = (w.s > w.gap ? w.s : w.gap) + 1;      // s >= 0, so v >= 1
            w.s = v + (run - 1); w.gap = -100000; w.dir = 0;
            if (w.s > w.best) w.best = w.s;
            r += run; c += run;
            EMIT();
        }
        int d1 = ((int)m - r) < ((int)n - c) ? ((int)m - r) : ((int)n - c); if (d1 > VTX_BAND_LAZY_EXT(KMER)) d1 = VTX_BAND_LAZY_EXT(KMER);
        for (int i = 0; i < d1; ++i) { ++r; ++c; walk_diag(w, x[r - 1] == yb[c - 1]); }
        if (d1 > 0) EMIT();
        {
            const int t1 = min(min((int)m - r, (int)n - c), BANDW);      // ... and up to w cells past the last anchor
            for (int i = 0; i < t1; ++i) { ++r; ++c; walk_diag(w, x[r - 1] == yb[c - 1]); }
        }
#undef EMIT
        *cert_out = w.best;
        if (w.best == ub) return 0;                          // cert == ub: banded == full == ub
        *nv_out = nv;
        return 1;
}

// ---- band_run_kernel's chain DP over pieces (see the kernel for the overall scheme) ----------------
// LDS entry j of lane tid: e_id[j*256+tid] = x0 << 16 | y0, e_dl[j*256+tid] = dp0 << 16 | len
// (dp0 == 0: piece collected by phase 1 but not started yet).  A started entry is a SEGMENT (linear:
// dp = dp0 + u).  Pieces start in list order (= match order); a breakpoint inside a segment can only
// happen at the ENTRY row of another segment (after its entry a linear segment's candidate minus the
// row index never grows, while the run's own value never shrinks), so every ordered pair
// (segment -> segment) yields at most one candidate match, queued as an event and re-examined with a
// full query when the sweep reaches it.
struct run_state {
    uint32_t n_ent, i_next, ev0, ev1, lg_n;
    int32_t best_v; uint32_t best_id;
    bool overflow; uint32_t why;
    bool ub_ok;            // every piece is still known: in the list, or (n_sp of them) spilled to the task's global area
    uint32_t n_sp;
};

__device__ __forceinline__ void run_segq(uint32_t id0, uint32_t dp0, uint32_t len, int32_t qx, int32_t qy, int32_t& bV,
                                         uint32_t& bid) {
    const int32_t sx = (int32_t)(id0 >> 16), sy = (int32_t)(id0 & 0xffff);
    int32_t u = (int32_t)len - 1;
    u = min(u, qx - sx - KMER);
    u = min(u, qy - sy - KMER);
    if (u >= 0) {
        const int32_t v = (int32_t)dp0 + sx + sy + 2 * KMER + 3 * u;
        const uint32_t qid = id0 + (uint32_t)u * 0x10001u;
        if (v > bV || (v == bV && (bid == NONE_ID || qid > bid))) { bV = v; bid = qid; }
    }
}

// segment s entering segment e: the first row of e at which an element of s is visible.  e_len is the
// length to test against (0xffff for a piece that is still growing).
__device__ __forceinline__ void run_gen_event(run_state& st, uint32_t s_id, uint32_t s_dp, uint32_t s_len, uint32_t e_id,
                                              uint32_t e_dp, uint32_t e_len) {
    const int32_t ga = (int32_t)(e_id >> 16) - (int32_t)(s_id >> 16) - KMER;
    const int32_t gb = (int32_t)(e_id & 0xffff) - (int32_t)(s_id & 0xffff) - KMER;
    const int32_t ts = max(max(-ga, -gb), 0);
    if (ts < 1 || ts >= (int32_t)e_len) return;
    const int32_t gu = min(min(ga + ts, gb + ts), (int32_t)s_len - 1);
    const int32_t gc = (int32_t)s_dp + gu + 1 - ((ga + ts - gu) + (gb + ts - gu));
    if (gc <= (int32_t)e_dp + ts) return;
    const uint32_t mid = e_id + (uint32_t)ts * 0x10001u;
    if (mid == st.ev0 || mid == st.ev1) return;
    if (st.ev0 == NONE_ID) st.ev0 = mid;
    else if (st.ev1 == NONE_ID) st.ev1 = mid;
    else { st.overflow = true; st.why = 4; }
}

// events both ways between the (new) segment at index ni and every other started segment
template <int NT>
__device__ void run_pair_events(run_state& st, const uint32_t* e_id, const uint32_t* e_dl, int tid, uint32_t ni,
                                uint32_t open_a, uint32_t open_b) {
    const uint32_t nid = e_id[ni * NT + tid], ndl = e_dl[ni * NT + tid];
    const uint32_t nd = ndl >> 16, nl = ndl & 0xffff;
    const bool n_open = ni == open_a || ni == open_b;
    if (!(nl >= 2 || n_open || nd + nl - 1 > KMER)) return;     // an isolated k-mer with dp = K: no events either way
    for (uint32_t j = 0; j < st.n_ent && !st.overflow; ++j) {
        if (j == ni) continue;
        const uint32_t dl = e_dl[j * NT + tid];
        if ((dl >> 16) == 0) continue;
        const uint32_t sid = e_id[j * NT + tid];
        const bool j_open = j == open_a || j == open_b;
        if (nl >= 2 || n_open) run_gen_event(st, sid, dl >> 16, dl & 0xffff, nid, nd, n_open ? 0xffffu : nl);
        if ((dl & 0xffff) >= 2 || j_open) run_gen_event(st, nid, nd, nl, sid, dl >> 16, j_open ? 0xffffu : (dl & 0xffff));
    }
}

// Process, in match order, every event and every piece start with id < limit.
template <int NT, int PSV>
__device__ void run_advance(run_state& st, uint32_t* e_id, uint32_t* e_dl, int tid, uint32_t limit, uint32_t& open_a,
                            uint32_t& open_b, uint32_t* mylog) {
    while (!st.overflow) {
        // next unstarted piece
        uint32_t i = st.i_next;
        while (i < st.n_ent && (e_dl[i * NT + tid] >> 16) != 0) ++i;
        st.i_next = i;
        uint32_t start_id = i < st.n_ent ? e_id[i * NT + tid] : NONE_ID;
        if (start_id >= limit) start_id = limit;
        // ---- events (possible breakpoints) due before that, in match order ----
        while (!st.overflow) {
            const uint32_t mid = st.ev0 < st.ev1 ? st.ev0 : st.ev1;
            if (mid == NONE_ID || mid >= start_id) break;
            if (mid == st.ev0) st.ev0 = NONE_ID; else st.ev1 = NONE_ID;
            const int32_t mx = (int32_t)(mid >> 16), my = (int32_t)(mid & 0xffff);
            int32_t bV = INT32_MIN; uint32_t bid = NONE_ID;
            uint32_t e = NONE_ID, eid = 0, edp = 0, elen = 0;
            for (uint32_t j = 0; j < st.n_ent; ++j) {
                const uint32_t dl = e_dl[j * NT + tid];
                if ((dl >> 16) == 0) continue;
                const uint32_t sid = e_id[j * NT + tid];
                const int32_t sx = (int32_t)(sid >> 16);
                if ((int32_t)(sid & 0xffff) - sx == my - mx && sx <= mx && mx < sx + (int32_t)(dl & 0xffff)) {
                    e = j; eid = sid; edp = dl >> 16; elen = dl & 0xffff;
                }
                run_segq(sid, dl >> 16, dl & 0xffff, mx, my, bV, bid);
            }
            if (e == NONE_ID) continue;                        // beyond the end of its piece
            const uint32_t t = (uint32_t)(mx - (int32_t)(eid >> 16));
            if (t == 0 || bid == NONE_ID) continue;
            const int32_t cand = bV - 5 - (mx + my) + KMER;
            if (cand <= (int32_t)(edp + t)) continue;          // continuation wins (ties included)
            // breakpoint: split segment e at t; the tail becomes a new segment (keeps e's open status if it was open)
            if (st.n_ent == PSV || st.lg_n == LG) { st.overflow = true; st.why = st.lg_n == LG ? 5 : 3; break; }
            e_dl[e * NT + tid] = (edp << 16) | t;
            e_id[st.n_ent * NT + tid] = mid;
            e_dl[st.n_ent * NT + tid] = ((uint32_t)cand << 16) | (elen - t);
            mylog[st.lg_n * (2 * NT)] = mid; mylog[st.lg_n * (2 * NT) + 1] = bid; ++st.lg_n;
            const uint32_t ni = st.n_ent++;
            if (open_a == e) open_a = ni;                     // the growing end of an open piece is its tail
            if (open_b == e) open_b = ni;
            run_pair_events<NT>(st, e_id, e_dl, tid, ni, open_a, open_b);
        }
        if (st.overflow || start_id >= limit || i >= st.n_ent) break;
        // ---- start of piece i ----
        const int32_t px = (int32_t)(start_id >> 16), py = (int32_t)(start_id & 0xffff);
        const uint32_t plen = e_dl[i * NT + tid] & 0xffff;
        int32_t bV = INT32_MIN; uint32_t bid = NONE_ID;
        int32_t cdp = -1;
        for (uint32_t j = 0; j < st.n_ent; ++j) {
            const uint32_t dl = e_dl[j * NT + tid];
            if ((dl >> 16) == 0) continue;
            const uint32_t sid = e_id[j * NT + tid];
            if (sid + (dl & 0xffff) * 0x10001u == start_id) cdp = (int32_t)((dl >> 16) + (dl & 0xffff));
            run_segq(sid, dl >> 16, dl & 0xffff, px, py, bV, bid);
        }
        int32_t dp = KMER; uint32_t prev = NONE_ID;
        if (bid != NONE_ID) {
            const int32_t cand = bV - 5 - (px + py) + KMER;
            if (cand >= dp) { dp = cand; prev = bid; }
        }
        if (cdp >= dp) {
            dp = cdp;                                          // adjacent piece: LCSk++ continuation, no log entry
        } else {
            if (st.lg_n == LG) { st.overflow = true; st.why = 5; break; }
            mylog[st.lg_n * (2 * NT)] = start_id; mylog[st.lg_n * (2 * NT) + 1] = prev; ++st.lg_n;
        }
        e_dl[i * NT + tid] = ((uint32_t)dp << 16) | plen;
        st.i_next = i + 1;
        run_pair_events<NT>(st, e_id, e_dl, tid, i, open_a, open_b);
    }
}

// Drop started, closed segments none of whose elements can win a query any more (2*(len-1) < x - x0 - dp0 - 1,
// or dominated by an element of another segment);
// their ends are folded into the running best first.  Updates the indices of the two open pieces.
template <int NT, int PSV>
__device__ void run_compact(run_state& st, uint32_t* e_id, uint32_t* e_dl, int tid, uint32_t xr, uint32_t& a_idx,
                            uint32_t& b_idx, uint32_t* spill) {
    uint32_t w = 0, na = NONE_ID, nb = NONE_ID, ni = NONE_ID;
    for (uint32_t j = 0; j < st.n_ent; ++j) {
        const uint32_t sid = e_id[j * NT + tid], dl = e_dl[j * NT + tid];
        const int32_t dp0 = (int32_t)(dl >> 16), len = (int32_t)(dl & 0xffff);
        const bool open = j == a_idx || j == b_idx;
        // an event still queued for one of its matches keeps a segment
        bool has_ev = false;
        for (int k = 0; k < 2; ++k) {
            const uint32_t mid = k ? st.ev1 : st.ev0;
            if (mid == NONE_ID) continue;
            const int32_t mx = (int32_t)(mid >> 16), sx = (int32_t)(sid >> 16);
            if ((int32_t)(mid & 0xffff) - mx == (int32_t)(sid & 0xffff) - sx && mx >= sx) has_ev = true;
        }
        bool dead = dp0 != 0 && !open && !has_ev && 2 * (len - 1) < (int32_t)xr - (int32_t)(sid >> 16) - dp0 - 1;
        if (!dead && dp0 != 0 && !open && !has_ev && (int32_t)xr > (int32_t)(sid >> 16) + len) {
            // Dominance: a query takes the element with the largest V = dp + xe + ye among those ending at or before
            // its start (ties: larger id).  If another started segment has an element that ends no later than this
            // segment's FIRST element and beats the V of its LAST element strictly, no element of this segment can
            // ever win a query.  The row after its last match is behind the sweep, so no piece can still continue
            // it — unless that piece is already waiting in the list.
            const int32_t qx = (int32_t)(sid >> 16) + KMER, qy = (int32_t)(sid & 0xffff) + KMER;
            const int32_t v_last = dp0 + (len - 1) + (qx + len - 1) + (qy + len - 1);
            const uint32_t next_id = sid + (uint32_t)len * 0x10001u;
            int32_t bV = INT32_MIN; uint32_t bid = NONE_ID;
            bool continued = false;
            for (uint32_t k = 0; k < st.n_ent; ++k) {
                if (k == j) continue;
                const uint32_t kid = e_id[k * NT + tid], kdl = e_dl[k * NT + tid];
                if ((kdl >> 16) == 0) { continued |= kid == next_id; continue; }
                run_segq(kid, kdl >> 16, kdl & 0xffff, qx, qy, bV, bid);
            }
            dead = !continued && bid != NONE_ID && bV > v_last;
        }
        if (dead) {
            const int32_t v = dp0 + len - 1;
            const uint32_t eid = sid + (uint32_t)(len - 1) * 0x10001u;
            if (v > st.best_v || (v == st.best_v && eid > st.best_id)) { st.best_v = v; st.best_id = eid; }
            // the upper bound needs every piece: park the dropped one (start, k-mers) behind the jump log
            if (st.n_sp < SPILL_OF(PSV)) { spill[st.n_sp * (2 * NT)] = sid; spill[st.n_sp * (2 * NT) + 1] = (uint32_t)len; ++st.n_sp; }
            else st.ub_ok = false;
            continue;
        }
        if (j == a_idx) na = w;
        if (j == b_idx) nb = w;
        if (ni == NONE_ID && dp0 == 0) ni = w;
        if (w != j) { e_id[w * NT + tid] = sid; e_dl[w * NT + tid] = dl; }
        ++w;
    }
    st.n_ent = w;
    st.i_next = ni == NONE_ID ? w : ni;
    a_idx = na; b_idx = nb;
}

// ---- upper bound of the full-matrix score from the piece list (see the file header) --------------------
// Entry j covers bases [x0, x0 + len + K - 1) of its diagonal (len k-mers).  G (kept in the high half of
// e_dl, which held dp0 — dead after phase 2) = best value a chain brings into the entry minus the entry
// offset; the value of a chain ending with the whole entry is len_bases + G.  A predecessor q is used up to
// its last base that precedes the entry point in both coordinates, and the entry point is the first base
// of p from which all of q is usable (later entries lose a base of p per base gained, earlier ones a base
// of q), so one candidate per ordered pair is enough.  Entries that continue each other on a diagonal (a
// piece split by a breakpoint or by the two-register window of phase 1) join at cost 0 through D == 0.
// Fixpoint over ordered pairs; list order (= start order) settles in one pass plus a confirming one.
__device__ __forceinline__ int32_t ub_join_same(int32_t D) {
    return 6 * ((D + 10) / 6) - D;                         // gap-free with ceil((D + 5) / 6) mismatches; a stretch with gaps costs more (J_gap, oracle/vtx_certify.c)
}

template <int NT>
__device__ int32_t run_ub(const uint32_t* e_id, uint32_t* e_dl, int tid, uint32_t n_ent) {
    for (uint32_t j = 0; j < n_ent; ++j) e_dl[j * NT + tid] &= 0xffffu;
    bool changed = true;
    // Pass 0 walks the entries in list order, so an edge q -> p with q before p sees q's settled value; pass 1 only
    // has to re-examine the BACK edges (q after p), whose q was still unsettled in pass 0.  If that changes nothing
    // the forward edges are still right; otherwise full passes until nothing moves.
    for (int pass = 0; pass < 5 && changed; ++pass) {
        changed = false;
        for (uint32_t p = 0; p < n_ent; ++p) {
            const uint32_t idp = e_id[p * NT + tid], dlp = e_dl[p * NT + tid];
            const int32_t xp = (int32_t)(idp >> 16), yp = (int32_t)(idp & 0xffff);
            const int32_t lp = (int32_t)(dlp & 0xffff) + KMER - 1;
            const int32_t g0 = (int32_t)(dlp >> 16);
            int32_t g = g0;
            for (uint32_t q = pass == 1 ? p + 1 : 0; q < n_ent; ++q) {
                if (q == p) continue;
                const uint32_t idq = e_id[q * NT + tid], dlq = e_dl[q * NT + tid];
                const int32_t xq = (int32_t)(idq >> 16), yq = (int32_t)(idq & 0xffff);
                const int32_t lq = (int32_t)(dlq & 0xffff) + KMER - 1, gq = (int32_t)(dlq >> 16);
                int32_t s = max(xq + lq - xp, yq + lq - yp);
                s = min(max(s, 0), lp - 1);
                const int32_t t = min(lq - 1, min(xp - xq, yp - yq) + s - 1);
                if (t < 0) continue;
                const int32_t dd = (yp - xp) - (yq - xq);
                int32_t J = 5 + abs(dd);
                if (dd == 0) { const int32_t D = xp + s - xq - t - 1; J = D == 0 ? 0 : ub_join_same(D); }
                g = max(g, t + 1 + gq - J - s);
            }
            if (g != g0) { e_dl[p * NT + tid] = (dlp & 0xffffu) | ((uint32_t)g << 16); changed = true; }
        }
    }
    if (changed) return INT32_MAX;                         // not settled: leave the task to the DP
    int32_t ub = KMER - 1;
    for (uint32_t j = 0; j < n_ent; ++j) {
        const uint32_t dl = e_dl[j * NT + tid];
        ub = max(ub, (int32_t)(dl & 0xffff) + KMER - 1 + (int32_t)(dl >> 16));
    }
    return ub;
}

// Builds the k-mer tables of loci [lbase, lbase + n_tab / 2) in LDS (layout: band_table_stride); called by every
// thread of the workgroup; the tables are complete after the last barrier inside.
template <int NT>
__device__ __forceinline__ void build_tables(uint8_t* tables, uint32_t n_tab, uint32_t lbase,
                                             const vtx_locus* __restrict__ loci, const uint8_t* __restrict__ hap_arena,
                                             uint32_t max_hap, uint32_t table_stride, uint32_t n_heads, int tid,
                                             uint32_t& s_hibyte) {
        __syncthreads();
        // ---- build the haplotype tables of loci [lbase, lbase + loci_per_pass) ----
        for (uint32_t t = 0; t < n_tab; ++t) {
            const vtx_locus loc = loci[lbase + (t >> 1)];
            const uint32_t hn = max(loc.ref_len, loc.alt_len) > max_hap ? 0u : ((t & 1) ? loc.alt_len : loc.ref_len);   // (a locus of the slow list: no table)
            const uint8_t* hy = hap_arena + ((t & 1) ? loc.alt_off : loc.ref_off);
            uint8_t* tb = tables + (size_t)t * table_stride;
            uint2* ent = TB_ENT(tb);
            uint16_t* head = TB_HEAD(tb);
            uint8_t* bytes = TB_BYTES(tb);
            uint8_t* fb = TB_FB(tb);
            if (tid == 0 && t == 0) s_hibyte = 0;
            for (uint32_t y = tid; y < hn; y += NT) {
                bytes[y] = hy[y];
                fb[y] = hy[y] & 0x7f;
                if (y + KMER <= hn) {
                    // (the bucket rides in the next-pointer field until the chains are linked: the serial loop below is a quarter of
                    //  this kernel's instructions, and the hash was most of an iteration)
                    const uint32_t lo = (uint32_t)hy[y] | ((uint32_t)hy[y + 1] << 8) | ((uint32_t)hy[y + 2] << 16) | ((uint32_t)hy[y + 3] << 24);
                    const uint32_t hi = (uint32_t)hy[y + 4] | ((uint32_t)hy[y + 5] << 8);
                    ent[y].x = lo;
                    ent[y].y = hi | (kw_hash(lo, hi, n_heads - 1) << 16);
                }
            }
            for (uint32_t i = tid; i < n_heads; i += NT) head[i] = CH_END;
            uint32_t* uq = TB_UQ(tb);
            for (uint32_t i = tid; i < (table_stride - vtxf::tab_uq_off(max_hap, n_heads)) / 4; i += NT) uq[i] = 0;     // uq[] and, behind it, pb[128] (and tw[], t3[] of a table that goes to global memory)
        }
        __syncthreads();
        if ((uint32_t)tid < n_tab) {     // one lane per table: sequential head insertion, descending y => ascending chains
            const vtx_locus loc = loci[lbase + ((uint32_t)tid >> 1)];
            const uint32_t hn = max(loc.ref_len, loc.alt_len) > max_hap ? 0u : ((tid & 1) ? loc.alt_len : loc.ref_len);
            uint8_t* tb = tables + (size_t)tid * table_stride;
            uint2* ent = TB_ENT(tb);
            uint16_t* head = TB_HEAD(tb);
            if (hn >= KMER)
                for (int y = (int)hn - KMER; y >= 0; --y) {
                    const uint32_t ey = ent[y].y;
                    const uint32_t h = ey >> 16;
                    ent[y].y = (ey & 0xffffu) | ((uint32_t)head[h] << 16);
                    head[h] = (uint16_t)y;
                }
        }
        __syncthreads();
        // uniqueness flags: a read k-mer that continues an open piece onto a UNIQUE haplotype k-mer has no other match
        // in this haplotype, so phase 1 may extend the piece without probing the table.  The flag rides in bit 7 of the
        // flag byte of the k-mer's LAST base; a haplotype with a byte >= 0x80 gets no flags (shortcut off).
        for (uint32_t t = 0; t < n_tab; ++t) {
            const vtx_locus loc = loci[lbase + (t >> 1)];
            const uint32_t hn = max(loc.ref_len, loc.alt_len) > max_hap ? 0u : ((t & 1) ? loc.alt_len : loc.ref_len);   // (a locus of the slow list: no table)
            uint8_t* tb = tables + (size_t)t * table_stride;
            const uint8_t* bytes = TB_BYTES(tb);
            for (uint32_t y = tid; y < hn; y += NT) if (bytes[y] & 0x80) atomicOr(&s_hibyte, 1u << t);
        }
        __syncthreads();
        for (uint32_t t = 0; t < n_tab; ++t) {
            if ((s_hibyte >> t) & 1u) continue;
            const vtx_locus loc = loci[lbase + (t >> 1)];
            const uint32_t hn = max(loc.ref_len, loc.alt_len) > max_hap ? 0u : ((t & 1) ? loc.alt_len : loc.ref_len);   // (a locus of the slow list: no table)
            uint8_t* tb = tables + (size_t)t * table_stride;
            const uint2* ent = TB_ENT(tb);
            const uint16_t* head = TB_HEAD(tb);
            uint8_t* fb = TB_FB(tb);
            uint32_t* uq = TB_UQ(tb) + vtxf::UQ_PAD_WORDS;
            for (uint32_t y = tid; y + KMER <= hn; y += NT) {
                const uint2 k = ent[y];
                uint32_t same = 0;
                for (uint32_t e = head[kw_hash(k.x, k.y & 0xffffu, n_heads - 1)]; e != CH_END; e = ent[e].y >> 16)
                    same += (ent[e].x == k.x && ((ent[e].y ^ k.y) & 0xffffu) == 0);
                if (same == 1) {
                    fb[y + KMER - 1] |= 0x80;      // only this thread touches that byte
                    atomicOr(&uq[y >> 5], 1u << (y & 31u));
                }
            }
        }
        __syncthreads();
        // bucket tags (band_diag_kernel, vtx_fast_core.h: walk_bucket): a bucket with ONE entry carries 4 hash bits of its k-mer
        // in bits 12-15 of its head word, a bucket with more HEAD_MULTI — most probes of a k-mer that is not in the haplotype end at
        // the head word.  Readers that want the plain position strip the tag (head_pos).
        for (uint32_t t = 0; t < n_tab; ++t) {
            uint8_t* tb = tables + (size_t)t * table_stride;
            const uint2* ent = TB_ENT(tb);
            uint16_t* head = TB_HEAD(tb);
            // presence bitmap: one bit per 12-bit k-mer code (vtx_fast_core.h: kw_code)
            {
                const vtx_locus loc = loci[lbase + (t >> 1)];
                const uint32_t hn = max(loc.ref_len, loc.alt_len) > max_hap ? 0u : ((t & 1) ? loc.alt_len : loc.ref_len);
                uint32_t* pb = TB_PB(tb);
                for (uint32_t y = tid; y + KMER <= hn; y += NT) {
                    const uint32_t code = vtxf::kw_code(ent[y].x, ent[y].y & 0xffffu);
                    atomicOr(&pb[code >> 5], 1u << (code & 31u));
                }
            }
            for (uint32_t h = tid; h < n_heads; h += NT) {
                const uint32_t y0 = head[h];
                if (y0 == CH_END) continue;
                const uint2 k = ent[y0];
                const uint32_t tag = (k.y >> 16) == CH_END ? vtxf::kw_tag(vtxf::kw_mix(k.x, k.y & 0xffffu)) : vtxf::HEAD_MULTI;
                head[h] = (uint16_t)(y0 | (tag << 12));
            }
        }
        __syncthreads();
}

#ifdef VTX_DEVTOOLS        // (developer build only: the production library carries one table kernel)
// Round 3's table kernel, kept as the REFERENCE of band_tables_kernel (further down) under VTX_BAND_TABLES_V1=1: one workgroup
// (one wavefront) per locus of [l0, l0 + n_loci): both tables in LDS (build_tables), then copied to
// gtables[(locus - l0) * 2 * table_stride].
__global__ __launch_bounds__(64) void band_tables_v1_kernel(const vtx_locus* __restrict__ loci, uint32_t l0, uint32_t n_loci,
                                                         const uint8_t* __restrict__ hap_arena, uint32_t max_hap,
                                                         uint32_t table_stride, uint32_t n_heads, uint8_t* __restrict__ gtables, uint32_t with_t3) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t s_hibyte;
    const int tid = threadIdx.x;
    for (uint32_t l = blockIdx.x; l < n_loci; l += gridDim.x) {
        build_tables<64>((uint8_t*)smem, 2, l0 + l, loci, hap_arena, max_hap, table_stride, n_heads, tid, s_hibyte);
        for (uint32_t t = 0; t < 2; ++t) {
            const vtx_locus loc = loci[l0 + l];
            const uint32_t hn = max(loc.ref_len, loc.alt_len) > max_hap ? 0u : (t ? loc.alt_len : loc.ref_len);
            build_twins_wave((uint8_t*)smem + (size_t)t * table_stride, hn, ((s_hibyte >> t) & 1u) != 0, max_hap, n_heads, tid);
            if (with_t3) build_t3_wave((uint8_t*)smem + (size_t)t * table_stride, hn, max_hap, n_heads, tid);
        }
        wave_sync();
        const uint4* src = (const uint4*)smem;
        uint4* dst = (uint4*)(gtables + (size_t)l * 2 * table_stride);
        for (uint32_t i = tid; i < 2 * table_stride / 16; i += 64) dst[i] = src[i];
    }
}
#endif

template <int NT, bool GT, int WPE, int PSV>
__global__ __launch_bounds__(NT, WPE) void band_run_kernel(
    uint32_t n_tasks, uint32_t task_base,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    uint32_t max_hap, uint32_t min_hap, uint32_t tables_per_pass, uint32_t table_stride,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score,
    uint32_t* __restrict__ logbuf, uint16_t* __restrict__ band, uint32_t band_stride,
    uint32_t* __restrict__ hard_list, uint32_t* __restrict__ overflow_list, uint32_t* __restrict__ pending_list,
    uint32_t* __restrict__ pend_buf, uint32_t hard_cap, uint32_t pend_cap,
    uint32_t* __restrict__ counters, uint32_t ablate, uint32_t n_heads, const uint8_t* __restrict__ gtables,
    uint32_t gt_l0, uint32_t xcd_claim, const uint32_t* __restrict__ task_list) {
    // task_list != nullptr (GT only): the tasks are task_list[0 .. n_tasks) instead of task_base + 0 .. n_tasks — the second
    // chance of the tasks whose 12-entry lists overflowed, in the 15-entry variant
    // GT: the tables of every locus were built by band_tables_kernel in global memory (shallow loci: a wavefront's 64
    // tasks span many loci, tables in LDS cost the occupancy); !GT: built here, per block, in LDS
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t s_hibyte, s_blk;              // bit t: the haplotype of table t holds a byte >= 0x80 (no continuation shortcut)
    const int tid = threadIdx.x;
    // per-lane LDS arrays, element i of lane tid at [i * NT + tid]
    // staircase of ended matches, stored as RUNS: elements (ye0 + t, V0 + 3t, id0 + t*(1,1)), t < len
    // (a continued k-mer adds +1 to ye and, with dp + 1, +3 to V = dp + xe + ye)
    uint32_t* pm_a = smem;                     // parked segments: id0 = x0 << 16 | y0
    uint32_t* pm_id = pm_a + PSV * NT;        //                  dp0 << 16 | len
    uint8_t* tables = (uint8_t*)(pm_id + PSV * NT);
#define PM_A(i) pm_a[(i) * NT + tid]
#define PM_ID(i) pm_id[(i) * NT + tid]

    // Persistent workgroups: the grid is what the chip holds at once; a workgroup claims blocks of NT consecutive tasks
    // from a counter until none is left.  The per-lane global scratch (jump log, spilled pieces) therefore belongs to the
    // RESIDENT lane — a few hundred MB that stay in the caches instead of 656 B x every task of the batch.
    // jump log + spill area, entry-major within the workgroup: entry e of lane t at (e * NT + t) * 2 words, so the
    // entries the lanes of a wave write at about the same time share cache lines
    uint32_t* mylog = logbuf + (size_t)blockIdx.x * NT * TASK_WORDS + tid * 2;
    const uint32_t n_blocks = (n_tasks + NT - 1) / NT;
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8); each XCD claims blocks from its own eighth of the
    // batch first (counters[16 + xcd]), so the ~8 wavefronts that share a locus' tables and reads meet in one L2, and
    // helps the other eighths out when its own is done.
    // (xcd_claim == 0: one range for all)
    const bool by_xcd = xcd_claim != 0;
    const uint32_t xcd = by_xcd ? (blockIdx.x & 7u) : 0u;
    const uint32_t per_xcd = by_xcd ? (n_blocks + 7) / 8 : n_blocks;
    uint32_t turn = 0;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
        uint32_t b = 0xffffffffu;
        while (turn < 8) {
            const uint32_t r = (xcd + turn) & 7u;
            const uint32_t lo = r * per_xcd, hi = min(n_blocks, lo + per_xcd);
            const uint32_t k = lo + atomicAdd(&counters[16 + r], 1u);
            if (lo < hi && k < hi) { b = k; break; }
            ++turn;
        }
        s_blk = b;
    }
    __syncthreads();
    const uint32_t blk = s_blk;
    if (blk == 0xffffffffu) break;
    const uint32_t slot = blk * NT + tid;
    const bool have = slot < n_tasks;
    const uint32_t task = (GT && task_list) ? (have ? task_list[slot] : 0u) : task_base + slot;
    uint32_t rid = 0, hap = 0, my_locus = 0, m = 0, n = 0;
    const uint8_t* x = nullptr;
    if (have) {
        rid = task >> 1; hap = task & 1;
        const vtx_record rec = records[rid];
        my_locus = rec_locus[rid];
        m = rec.read_len;
        x = read_arena + rec.read_off;
        n = hap ? loci[my_locus].alt_len : loci[my_locus].ref_len;
    }
    // locus range of this workgroup (tasks are in record order, records in locus order)
    const uint32_t first_task = task_base + blk * NT;
    const uint32_t last_task = min(task_base + n_tasks - 1, first_task + NT - 1);
    const uint32_t l_first = rec_locus[first_task >> 1], l_last = rec_locus[last_task >> 1];
    const uint32_t loci_per_pass = GT ? 0x40000000u : tables_per_pass / 2;
    int32_t* my_score = (hap ? alt_score : ref_score) + rid;
    bool done = !have;
    if (have && (m == 0 || n == 0)) { *my_score = 0; done = true; }     // empty read / haplotype: score 0
    // reads / haplotypes beyond what these tables and lists hold are scored by slow_align_kernel (the host lists them)
    // (min_hap: the pass over the batch's long-haplotype loci — vtx_run's second pass — leaves the loci the first pass scored alone)
    if (have && (m > VTX_FAST_READ_LEN || max(loci[my_locus].ref_len, loci[my_locus].alt_len) > max_hap ||
                 max(loci[my_locus].ref_len, loci[my_locus].alt_len) <= min_hap)) done = true;
    // a task without any k-mer match has the whole matrix in band (Band::full_matrix): hard list, marker slot
    // (hard slots beyond the capacity of the band buffer go to the general kernel's list, which makes them hard in slices)
#define PUSH_FULL_MATRIX() { const uint32_t h_ = atomicAdd(&counters[0], 1u);                                    \
                             if (h_ < hard_cap) { hard_list[h_] = task; band[(size_t)h_ * 2 * band_stride] = BAND_FULL_MATRIX; } \
                             else overflow_list[atomicAdd(&counters[1], 1u)] = task; }

    for (uint32_t lbase = GT ? 0u : l_first; lbase <= (GT ? 0u : l_last); lbase += loci_per_pass) {     // GT: one pass over every locus
        const uint32_t n_tab = GT ? 0u : min(loci_per_pass, l_last - lbase + 1) * 2;
        if constexpr (!GT) build_tables<NT>(tables, n_tab, lbase, loci, hap_arena, max_hap, table_stride, n_heads, tid, s_hibyte);
        if (done || my_locus < lbase || my_locus >= lbase + loci_per_pass) continue;
        done = true;
        // ---- this lane's table ----
        const uint8_t* tb;
        if constexpr (GT) tb = gtables + ((size_t)(my_locus - gt_l0) * 2 + hap) * table_stride;
        else tb = tables + (size_t)((my_locus - lbase) * 2 + hap) * table_stride;
        const uint2* ent = TB_ENT(tb);
        const uint16_t* head = TB_HEAD(tb);
        const uint8_t* yb = TB_BYTES(tb);
        const uint8_t* fb = TB_FB(tb);
        // GT: one uniform base (gtables, in scalar registers) + a 32-bit per-lane offset — the loads take the
        // scalar-base addressing form and the lanes do 32-bit arithmetic (tables <= 4 GB: vtxk_band_gtables_bytes)
        const uint32_t t_ent = GT ? (uint32_t)(((size_t)(my_locus - gt_l0) * 2 + hap) * table_stride) : 0u;
        const uint32_t t_head = t_ent + max_hap * 8u;
        const uint32_t t_fb = t_head + n_heads * 2u + max_hap + 8u;
        auto ENT = [&](uint32_t i) -> uint2 {
            if constexpr (GT) return *(const uint2*)(gtables + (size_t)(t_ent + i * 8u)); else return ent[i];
        };
        auto HEAD = [&](uint32_t i) -> uint32_t {
            if constexpr (GT) return head_pos(*(const uint16_t*)(gtables + (size_t)(t_head + i * 2u))); else return head_pos(head[i]);
        };
        auto FB = [&](uint32_t i) -> uint32_t {
            if constexpr (GT) return gtables[(size_t)(t_fb + i)]; else return fb[i];
        };
        if (m < KMER || n < KMER) { PUSH_FULL_MATRIX() continue; }    // no k-mer: Band::full_matrix
        if (VTX_ABLATE(ablate) == 1) continue;                           // (profiling aid) table build only
        if (VTX_ABLATE(ablate) == 2) {                                   // (profiling aid) probe loop only
            uint32_t wl = (uint32_t)x[0] | ((uint32_t)x[1] << 8) | ((uint32_t)x[2] << 16) | ((uint32_t)x[3] << 24);
            uint32_t wh = (uint32_t)x[4] | ((uint32_t)x[5] << 8), cntm = 0;
            for (uint32_t xr = 0; xr + KMER <= m; ++xr) {
                for (uint32_t y = head_pos(head[kw_hash(wl, wh, n_heads - 1)]); y != CH_END; y = ent[y].y >> 16) cntm += (ent[y].x == wl && (ent[y].y & 0xffffu) == wh);
                const uint32_t nb = (xr + KMER < m) ? x[xr + KMER] : 0;
                wl = (wl >> 8) | (wh << 24); wh = ((wh >> 8) & 0xff) | (nb << 8);
            }
            if (cntm == 0xffffffffu) counters[7] = cntm;
            continue;
        }

        // ================= phase 1: geometric diagonal pieces (+ incremental phase 2 when the list fills) ======
        // Every k-mer match either extends one of the two pieces held in registers or opens a new piece
        // (appended to the LDS list).  A piece that really continues a parked piece is simply a new,
        // adjacent piece: phase 2 treats adjacency as the LCSk++ continuation.  No queries here.
        run_state st;
        st.n_ent = 0; st.i_next = 0; st.ev0 = NONE_ID; st.ev1 = NONE_ID; st.lg_n = 0;
        st.best_v = -1; st.best_id = 0; st.overflow = false; st.why = 0; st.ub_ok = true; st.n_sp = 0;
        // the two open pieces: list index, k-mers so far, and the id (x << 16 | y) of the match that would continue them
        // (kept incrementally: a multiply-add per unit costs four issue slots on this target)
        uint32_t a_idx = NONE_ID, a_nx = 0, a_len = 0, b_idx = NONE_ID, b_nx = 0, b_len = 0;
        {
            // The probe is a flat loop of WORK UNITS, every lane at its own row: a unit is either one step along the
            // k-mer chain of the lane's current row, or the advance to its next row.  A wavefront therefore pays the
            // maximum over its lanes of the SUM of their chain steps, not the sum over rows of the per-row maximum
            // (with 64 lanes nearly every row has one lane with a 3-entry bucket).  The advance first tries the
            // continuation shortcut: if the piece in `a` continues at (row, y), its last 5 bases are already known to
            // match, so one byte compare decides the k-mer; and if the haplotype k-mer at y is unique in its
            // haplotype (CH_UNIQ) there can be no other match in this row — the piece is extended without hashing or
            // walking a chain.  On clean reads 4 of 5 rows take the shortcut.
            // The read streams through three 8-byte register blocks A, B, C (bytes [base, base + 24) of the read).  A
            // lane takes at most one byte per unit, so every 8th unit — a wave-uniform point — each lane that has moved
            // into B shifts (A = B, B = C) and requests the next C with one 8-byte global load.  That load is only
            // touched 8 units later: the loop never waits on global memory (a per-lane refill "when needed" makes some
            // lane issue a load in nearly every iteration, and vmcnt cannot tell the loads of different lanes apart).
            // The arena is padded; bytes at or beyond m are never used.
            const uint32_t hmask = n_heads - 1;
            uint64_t wA, wB, wC;
            __builtin_memcpy(&wA, x, 8);
            __builtin_memcpy(&wB, x + 8, 8);
            __builtin_memcpy(&wC, x + 16, 8);
            uint32_t wbase = 0;                                              // read index of A's first byte
            uint32_t unit = 0;
            uint32_t wlo = (uint32_t)wA;                                     // k-mer word of row xr
            uint32_t whi = (uint32_t)(wA >> 32) & 0xffffu;
            uint32_t xr = 0;
            uint32_t ycur = HEAD(kw_hash(wlo, whi, hmask));                  // chain cursor of row xr
            bool live = true;
            bool service;
            do {
                // ---- hot loop: kept free of the (rare, large) list-full handling, which sits after it ----
                service = false;
                uint32_t pend_id = 0;
                while (live && !service) {
                    // (written as two predicated blocks, not if / else with `continue`: with back edges inside the
                    // branches the structurizer turns one of them into an inner loop, and the lanes of the other kind
                    // then wait for whole bursts — measured 3x slower than the row-lockstep loop)
                    if ((++unit & 7u) == 0) {                                        // wave-uniform: window refill point
                        if (xr + KMER - wbase >= 8) { wA = wB; wB = wC; wbase += 8; }  // the next base lies in B: shift
                        // every lane (re)requests its C block: unconditional, so the result is not merged with the old
                        // value (a merge would be a use, and a use waits for the load)
                        __builtin_memcpy(&wC, x + (wbase + 16 < m ? wbase + 16 : 0u), 8);   // (a block past the read is never used)
                    }
                    const bool chain = ycur != CH_END;
                    // ---- everything either kind of unit reads from LDS, requested together: one round trip per unit.
                    //      (The kernel waits on LDS latency, not on VALU issue: the lanes of the other kind compute a
                    //      few addresses they do not need.) ----
                    const uint2 e = ENT(chain ? ycur : 0u);                          // chain step: the entry
                    const uint32_t xn = xr + 1;                                       // advance: next row ...
                    const uint32_t bi = xn + KMER - 1;                                // ... the base that enters its k-mer
                    const uint32_t woff = bi - wbase;                                 // 0 .. 15 by the refill rule
                    const uint64_t wsrc = (woff & 8u) ? wB : wA;
                    const uint32_t nb = (uint32_t)(wsrc >> (8 * (woff & 7u))) & 0xffu;
                    const uint32_t nwlo = (wlo >> 8) | (whi << 24);
                    const uint32_t nwhi = ((whi >> 8) & 0xff) | (nb << 8);
                    const uint32_t hd = HEAD(kw_hash(nwlo, nwhi, hmask));             // ... its bucket
                    const uint32_t aid = a_nx;                                        // ... where piece a would continue
                    const uint32_t ay = aid & 0xffffu;
                    const bool a_ok = a_idx != NONE_ID && (aid >> 16) == xn && ay + KMER <= n;
                    const uint32_t fbv = FB(a_ok ? ay + KMER - 1 : 0u);               // ... flag byte of that k-mer's last base
                    asm volatile("" ::"v"(e.x), "v"(e.y), "v"(hd), "v"(fbv));        // all three loads before either branch
                    if (chain) {
                        // ---- unit: one chain entry of row xr ----
                        const uint32_t y = ycur;
                        ycur = e.y >> 16;
                        if (e.x == wlo && (e.y & 0xffffu) == whi) {
                            const uint32_t id = (xr << 16) | y;
                            if (a_idx != NONE_ID && id == a_nx) {
                                ++a_len; a_nx += 0x10001u;
                            } else if (b_idx != NONE_ID && id == b_nx) {
                                ++b_len; b_nx += 0x10001u;
                                uint32_t t;
                                t = a_idx; a_idx = b_idx; b_idx = t;
                                t = a_nx; a_nx = b_nx; b_nx = t;
                                t = a_len; a_len = b_len; b_len = t;
                            } else if (st.n_ent == PSV) {
                                service = true; pend_id = id;
                            } else {
                                if (b_idx != NONE_ID) pm_id[b_idx * NT + tid] = (pm_id[b_idx * NT + tid] & 0xffff0000u) | b_len;
                                b_idx = a_idx; b_nx = a_nx; b_len = a_len;
                                a_idx = st.n_ent; a_nx = id + 0x10001u; a_len = 1;
                                pm_a[st.n_ent * NT + tid] = id; pm_id[st.n_ent * NT + tid] = 1; ++st.n_ent;
                            }
                        }
                    }
                    if (!chain) {
                        // ---- unit: advance to row xr + 1 ----
                        xr = xn;
                        live = xn + KMER <= m;
                        if (live) {
                            wlo = nwlo; whi = nwhi;
                            // piece a continues onto a k-mer that is unique in the haplotype: the only match of this row
                            if (a_ok && fbv == nb + 0x80u) { ++a_len; a_nx += 0x10001u; }
                            else ycur = hd;
                        }
                    }
                }
                if (service) {
                    // list full: run the chain DP up to this match, drop segments that cannot matter any more,
                    // then open the pending piece and resume the probe where it stopped
                    if (a_idx != NONE_ID) pm_id[a_idx * NT + tid] = (pm_id[a_idx * NT + tid] & 0xffff0000u) | a_len;
                    if (b_idx != NONE_ID) pm_id[b_idx * NT + tid] = (pm_id[b_idx * NT + tid] & 0xffff0000u) | b_len;
                    run_advance<NT, PSV>(st, pm_a, pm_id, tid, pend_id, a_idx, b_idx, mylog);
                    if (!st.overflow) run_compact<NT, PSV>(st, pm_a, pm_id, tid, xr, a_idx, b_idx, mylog + LG * (2 * NT));
                    if (st.n_ent == PSV && !st.overflow) { st.overflow = true; st.why = 2; }
                    if (!st.overflow) {
                        // a breakpoint may have split an open piece and compaction renumbers: reload the register copies
                        if (a_idx != NONE_ID) { a_len = pm_id[a_idx * NT + tid] & 0xffff; a_nx = pm_a[a_idx * NT + tid] + a_len * 0x10001u; }
                        if (b_idx != NONE_ID) {
                            b_len = pm_id[b_idx * NT + tid] & 0xffff; b_nx = pm_a[b_idx * NT + tid] + b_len * 0x10001u;
                            pm_id[b_idx * NT + tid] = (pm_id[b_idx * NT + tid] & 0xffff0000u) | b_len;
                        }
                        b_idx = a_idx; b_nx = a_nx; b_len = a_len;
                        a_idx = st.n_ent; a_nx = pend_id + 0x10001u; a_len = 1;
                        pm_a[st.n_ent * NT + tid] = pend_id; pm_id[st.n_ent * NT + tid] = 1; ++st.n_ent;
                    }
                }
            } while (service && !st.overflow);
            if (a_idx != NONE_ID) pm_id[a_idx * NT + tid] = (pm_id[a_idx * NT + tid] & 0xffff0000u) | a_len;
            if (b_idx != NONE_ID) pm_id[b_idx * NT + tid] = (pm_id[b_idx * NT + tid] & 0xffff0000u) | b_len;
        }
        bool overflow = st.overflow;
        uint32_t why = st.why;
        if (!overflow && st.n_ent == 0 && st.best_v < 0) { PUSH_FULL_MATRIX() continue; }   // no k-mer match
        if (VTX_ABLATE(ablate) == 3) { if (st.n_ent == 0xffffu) counters[7] = st.n_ent; continue; }   // (profiling aid) phase 1 only
        // ================= phase 2: the rest of the chain DP =================
        if (!overflow) {
            uint32_t no_a = NONE_ID, no_b = NONE_ID;
            run_advance<NT, PSV>(st, pm_a, pm_id, tid, NONE_ID, no_a, no_b, mylog);
            overflow = st.overflow; why = st.why;
            // best match = end of the segment with the largest (dp, index)
            for (uint32_t j = 0; j < st.n_ent && !overflow; ++j) {
                const uint32_t dl = pm_id[j * NT + tid];
                const int32_t v = (int32_t)((dl >> 16) + (dl & 0xffff)) - 1;
                const uint32_t eid = pm_a[j * NT + tid] + ((dl & 0xffff) - 1) * 0x10001u;
                if (v > st.best_v || (v == st.best_v && eid > st.best_id)) { st.best_v = v; st.best_id = eid; }
            }
        }
        const uint32_t lg_n = st.lg_n;
        const uint32_t best_id = st.best_id;
        if (overflow) { overflow_list[atomicAdd(&counters[1], 1u)] = task; atomicAdd(&counters[2 + why], 1u); continue; }
        // ================= certificate: chain-of-runs upper bound vs the staircase's own score =================
        // Tasks whose list lost pieces to run_compact cannot bound here: their pieces (list + spill area) go to
        // band_pending_kernel, which computes the same bound with a wavefront's lanes over up to 32 pieces.
        const bool pending = st.ub_ok && st.n_sp > 0;
        const int32_t ub = (st.ub_ok && !pending) ? run_ub<NT>(pm_a, pm_id, tid, st.n_ent) : INT32_MAX;
        if (VTX_ABLATE(ablate) == 4) { if (ub == -1) counters[7] = 1; continue; }   // (profiling aid) everything but the staircase walk
        uint32_t verts[4 * SG + 6];
        uint32_t nv = 0;
        int32_t cert = 0;
        {
            const int fr = band_finish(mylog, 2 * NT, lg_n, best_id, x, yb, m, n, ub, verts, &nv, &cert);
            if (fr == 0) { *my_score = ub; continue; }
            if (fr == 2) { overflow_list[atomicAdd(&counters[1], 1u)] = task; atomicAdd(&counters[7], 1u); continue; }
        }
        if (pending) {
            // a record for band_pending_kernel: header, certificate, list entries, staircase, spilled pieces
            const uint32_t pi = atomicAdd(&counters[11], 1u);
            if (pi >= pend_cap) { overflow_list[atomicAdd(&counters[1], 1u)] = task; continue; }
            uint32_t* prec = pend_buf + (size_t)pi * PEND_WORDS;
            prec[0] = st.n_ent | (st.n_sp << 8) | (nv << 16);
            prec[1] = (uint32_t)cert;
            for (uint32_t j = 0; j < st.n_ent; ++j) {
                prec[2 + 2 * j] = pm_a[j * NT + tid];
                prec[3 + 2 * j] = pm_id[j * NT + tid] & 0xffffu;
            }
            for (uint32_t i = 0; i < 2 * st.n_sp; ++i) prec[2 + 2 * st.n_ent + i] = mylog[(LG + (i >> 1)) * (2 * NT) + (i & 1)];
            for (uint32_t i = 0; i < nv; ++i) prec[2 + 2 * 32 + i] = verts[i];
            pending_list[pi] = task;
            continue;
        }
        if (!st.ub_ok) atomicAdd(&counters[10], 1u);         // statistics: hard only because too many pieces were dropped
        const uint32_t h = atomicAdd(&counters[0], 1u);
        if (h >= hard_cap) { overflow_list[atomicAdd(&counters[1], 1u)] = task; continue; }
        hard_list[h] = task;
        uint16_t* lo = band + (size_t)h * 2 * band_stride;
        lo[0] = BAND_POLYLINE; lo[1] = (uint16_t)nv;
        uint32_t* vout = (uint32_t*)(lo + 2);
        for (uint32_t i = 0; i < nv; ++i) vout[i] = verts[i];
    }
  }   // next block of tasks
#undef PM_A
#undef PM_ID
#undef PUSH_FULL_MATRIX
}


// =============================================================================================
// band_pending_kernel — the run bound (see run_ub / the file header) for tasks whose piece list overflowed its LDS
// slots: their pieces were kept in the task's global area (list entries + spilled ones), together with the
// certificate and the staircase.  32 lanes per task, one piece per lane, G iterated Jacobi-style (every lane
// re-evaluates all its in-edges from the values of the previous round) until no lane of the wave moves.
// cert == ub: the score is written; otherwise the task joins the hard list with its staircase.
// =============================================================================================
__global__ __launch_bounds__(256) void band_pending_kernel(
    const uint32_t* __restrict__ pending, uint32_t n_pending, const uint32_t* __restrict__ pend_buf,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score, uint16_t* __restrict__ band, uint32_t band_stride,
    uint32_t* __restrict__ hard_list, uint32_t* __restrict__ counters) {
    __shared__ uint32_t s_id[8][32], s_lg[8][32];      // per task: piece start, (bases << 16 | G)
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const uint32_t pi = blockIdx.x * 8 + grp;
    const bool have = pi < n_pending;
    const uint32_t task = have ? pending[pi] : 0;
    const uint32_t* area = pend_buf + (size_t)(have ? pi : 0) * PEND_WORDS;
    uint32_t hdr = 0; int32_t cert = 0;
    if (have) { hdr = area[0]; cert = (int32_t)area[1]; }
    const uint32_t n_ent = hdr & 0xff, n_sp = (hdr >> 8) & 0xff, nv = hdr >> 16;
    const uint32_t n = n_ent + n_sp;                    // <= 32: list entries, then spilled pieces
    uint32_t id = 0; int32_t len = 0;
    if ((uint32_t)l < n) {
        const uint32_t* e = area + 2 + 2 * l;
        id = e[0]; len = (int32_t)e[1] + KMER - 1;
    }
    const int32_t xp = (int32_t)(id >> 16), yp = (int32_t)(id & 0xffff);
    int32_t g = 0;
    s_id[grp][l] = id;
    s_lg[grp][l] = (uint32_t)len;
    __builtin_amdgcn_wave_barrier();
    // The geometry of an ordered pair (q -> this piece) never changes between rounds, only G[q] does: the pair's
    // constant  t + 1 - J - s  is computed once (NO_EDGE: q cannot precede this piece), a round is then one add and one
    // max per predecessor.
    constexpr int32_t NO_EDGE = -30000;
    int32_t cq[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        int32_t c = NO_EDGE;
        if ((uint32_t)l < n && (uint32_t)q < n && q != l) {
            const uint32_t idq = s_id[grp][q];
            const int32_t lq = (int32_t)s_lg[grp][q];
            const int32_t xq = (int32_t)(idq >> 16), yq = (int32_t)(idq & 0xffff);
            int32_t sdx = max(xq + lq - xp, yq + lq - yp);
            sdx = min(max(sdx, 0), len - 1);
            const int32_t t = min(lq - 1, min(xp - xq, yp - yq) + sdx - 1);
            if (t >= 0) {
                const int32_t dd = (yp - xp) - (yq - xq);
                int32_t J = 5 + abs(dd);
                if (dd == 0) { const int32_t D = xp + sdx - xq - t - 1; J = D == 0 ? 0 : ub_join_same(D); }
                c = t + 1 - J - sdx;
            }
        }
        cq[q] = c;
    }
    __builtin_amdgcn_wave_barrier();
    bool moving = true;
    for (int round = 0; round < 40 && moving; ++round) {
        s_lg[grp][l] = (uint32_t)g;
        __builtin_amdgcn_wave_barrier();
        int32_t gn = g;
#pragma unroll
        for (int q = 0; q < 32; ++q) gn = max(gn, cq[q] + (int32_t)s_lg[grp][q]);
        moving = __any(gn != g);
        g = gn;
        __builtin_amdgcn_wave_barrier();
    }
    int32_t ub = (uint32_t)l < n ? len + g : KMER - 1;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) ub = max(ub, __shfl_xor(ub, o));
    if (moving) ub = INT32_MAX;                         // not settled: leave the task to the DP
    if (!have) return;
    if (ub == cert) {
        if (l == 0) ((task & 1) ? alt_score : ref_score)[task >> 1] = cert;
        return;
    }
    uint32_t h = 0;
    if (l == 0) { h = atomicAdd(&counters[0], 1u); hard_list[h] = task; }
    h = (uint32_t)__shfl((int)h, 0, 32);
    uint16_t* lo = band + (size_t)h * 2 * band_stride;
    if (l == 0) { lo[0] = BAND_POLYLINE; lo[1] = (uint16_t)nv; }
    uint32_t* vout = (uint32_t*)(lo + 2);
    for (uint32_t i = l; i < nv; i += 32) vout[i] = area[2 + 2 * 32 + i];
}

extern "C" uint32_t vtxk_band_task_words(void) { return TASK_WORDS; }

extern "C" uint32_t vtxk_band_pend_words(void) { return PEND_WORDS; }

extern "C" hipError_t vtxk_launch_band_pending(const uint32_t* pending, uint32_t n_pending, const uint32_t* pend_buf,
                                               int32_t* ref_score, int32_t* alt_score, uint16_t* band,
                                               uint32_t band_stride, uint32_t* hard_list, uint32_t* counters, hipStream_t s) {
    if (!n_pending) return hipSuccess;
    hipLaunchKernelGGL(band_pending_kernel, dim3((n_pending + 7) / 8), dim3(256), 0, s, pending, n_pending, pend_buf,
                       ref_score, alt_score, band, band_stride, hard_list, counters);
    return hipGetLastError();
}

// =============================================================================================
// slow_align_kernel — the exact path for records the fast kernels cannot hold (reads above VTX_FAST_READ_LEN bases,
// haplotypes above VTX_FAST_HAP_LEN: a long sequence-resolved indel, a large --padding, long-read data).  One lane per
// alignment, everything in a per-task global slab: the literal seeding / chaining / band of band_task (banded flavour) or
// the whole matrix (full flavour), then the affine local DP over the column ranges in 32-bit arithmetic.  Throughput is
// not a goal here — the reference aligns any length, so must this library, instead of rejecting the batch.
// =============================================================================================
__global__ __launch_bounds__(64) void slow_align_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ tasks, uint32_t n_tasks, int banded,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    uint8_t* __restrict__ workspace, uint64_t ws_stride, uint32_t m_cap, uint32_t max_hap, uint32_t max_read,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score, uint32_t* __restrict__ retry_list,
    uint32_t* __restrict__ counters) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_tasks) return;
    const uint32_t task = tasks ? tasks[slot] : 2u * recs[slot >> 1] + (slot & 1u);
    const uint32_t rid = task >> 1, hap = task & 1;
    const vtx_record rec = records[rid];
    const vtx_locus loc = loci[rec_locus[rid]];
    const uint8_t* x = read_arena + rec.read_off;
    const uint8_t* y = hap_arena + (hap ? loc.alt_off : loc.ref_off);
    const int m = (int)rec.read_len, n = (int)(hap ? loc.alt_len : loc.ref_len);
    int32_t* out = (hap ? alt_score : ref_score) + rid;
    if (m == 0 || n == 0) { *out = 0; return; }
    // slab layout: the band scratch (vtxk_band_ws_stride's arrays), then lo / hi and the four DP columns
    uint8_t* ws = workspace + (uint64_t)slot * ws_stride;
    band_scratch_t<1> sc;
    size_t o = sc.carve(ws, 0, m_cap, max_hap);
    o = (o + 15) & ~(size_t)15;
    uint16_t* lo = (uint16_t*)(ws + o); o += ((size_t)max_hap + 2) * 2;
    uint16_t* hi = (uint16_t*)(ws + o); o += ((size_t)max_hap + 2) * 2;
    o = (o + 15) & ~(size_t)15;
    int32_t* Sp = (int32_t*)(ws + o); o += ((size_t)max_read + 2) * 4;
    int32_t* Dp = (int32_t*)(ws + o); o += ((size_t)max_read + 2) * 4;
    int32_t* Sc = (int32_t*)(ws + o); o += ((size_t)max_read + 2) * 4;
    int32_t* Dc = (int32_t*)(ws + o); o += ((size_t)max_read + 2) * 4;
    bool whole = !banded;
    if (banded) {
        int32_t cert = 0;
        int cA = 0, cB = 0;
        if (band_task(x, m, y, n, sc, m_cap, &cert, &cA, &cB)) { retry_list[atomicAdd(&counters[0], 1u)] = task; return; }
        if (cert == INT32_MAX) whole = true;                       // no k-mer match: Band::full_matrix
        else band_ranges(sc, cA, cB, m, n, lo, hi);
    }
    // banded::Aligner::compute_alignment in local mode (oracle/vtx_oracle.c:vtxo_sw_ranges): cells outside the ranges
    // hold MIN; every in-band cell may start at 0; row 0 / column 0 cells are 0 when in band
    const int32_t MINS = -858993459;
    for (int i = 0; i <= m; ++i) { Sp[i] = MINS; Dp[i] = MINS; Sc[i] = MINS; Dc[i] = MINS; }
    {
        const int l0 = whole ? 0 : (int)lo[0], h0 = whole ? m + 1 : (int)hi[0];
        for (int i = l0; i < h0; ++i) Sp[i] = 0;
    }
    int32_t best = 0;
    int plo = 0, phi = 0;
    int qlo = whole ? 0 : (int)lo[0], qhi = whole ? m + 1 : max((int)hi[0], qlo);
    for (int j = 1; j <= n; ++j) {
        for (int i = plo; i < phi; ++i) { Sc[i] = MINS; Dc[i] = MINS; }
        const int lj = whole ? 0 : (int)lo[j], hj = whole ? m + 1 : (int)hi[j];
        const uint8_t q = y[j - 1];
        int32_t up_i = MINS;
        for (int i = lj; i < hj; ++i) {
            if (i == 0) { Sc[0] = 0; up_i = MINS; continue; }
            const int32_t d = max(Dp[i] - 1, Sp[i] - 6);
            const int32_t ii = max(up_i - 1, Sc[i - 1] - 6);
            int32_t sv = Sp[i - 1] + (x[i - 1] == q ? 1 : -5);
            sv = max(max(sv, max(d, ii)), 0);
            Sc[i] = sv; Dc[i] = d; up_i = ii;
            best = max(best, sv);
        }
        plo = qlo; phi = qhi;
        qlo = lj; qhi = hj > lj ? hj : lj;
        int32_t* t;
        t = Sp; Sp = Sc; Sc = t;
        t = Dp; Dp = Dc; Dc = t;
    }
    *out = best;
}

extern "C" size_t vtxk_slow_ws_stride(uint32_t m_cap, uint32_t max_hap, uint32_t max_read) {
    size_t o = vtxk_band_ws_stride(m_cap, max_hap);            // (a multiple of 64)
    o += 2 * ((size_t)max_hap + 2) * 2;
    o = (o + 15) & ~(size_t)15;
    o += 4 * ((size_t)max_read + 2) * 4;
    return (o + 63) & ~(size_t)63;
}

extern "C" hipError_t vtxk_launch_slow_align(const uint32_t* recs, const uint32_t* tasks, uint32_t n_tasks, int banded,
                                             const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                             const uint8_t* read_arena, const uint8_t* hap_arena, uint8_t* workspace,
                                             uint64_t ws_stride, uint32_t m_cap, uint32_t max_hap, uint32_t max_read,
                                             int32_t* ref_score, int32_t* alt_score, uint32_t* retry_list, uint32_t* counters,
                                             hipStream_t s) {
    if (!n_tasks) return hipSuccess;
    hipLaunchKernelGGL(slow_align_kernel, dim3((n_tasks + 63) / 64), dim3(64), 0, s, recs, tasks, n_tasks, banded, records,
                       rec_locus, loci, read_arena, hap_arena, workspace, ws_stride, m_cap, max_hap, max_read, ref_score,
                       alt_score, retry_list, counters);
    return hipGetLastError();
}

// Polyline -> lo / hi arrays, one 16-lane group per hard slot (slots written by band_kernel already
// hold arrays and are skipped).  Vertices are joined by pure diagonal / vertical / horizontal pieces.
__global__ __launch_bounds__(256) void band_expand_kernel(const uint32_t* __restrict__ hard_list, uint32_t n_hard,
                                                          const vtx_record* __restrict__ records,
                                                          const uint32_t* __restrict__ rec_locus,
                                                          const vtx_locus* __restrict__ loci,
                                                          const uint16_t* src, uint32_t src_stride,
                                                          uint16_t* band, uint32_t band_stride) {
    // src: one record of src_stride u16 per hard task (marker, vertex count, vertices) — the compact polyline records
    // of band_run_kernel / band_pending_kernel, or the band slots themselves (general kernel: arrays or a marker)
    __shared__ uint32_t sv[16][4 * SG + 8];
    const int grp = threadIdx.x / 16, l = threadIdx.x % 16;
    const uint32_t h = blockIdx.x * 16 + grp;
    const uint16_t* in = src + (size_t)(h < n_hard ? h : 0) * src_stride;
    uint16_t* lo = band + (size_t)(h < n_hard ? h : 0) * 2 * band_stride;
    uint16_t* hi = lo + band_stride;
    const bool poly = h < n_hard && in[0] == BAND_POLYLINE;
    const bool whole = h < n_hard && in[0] == BAND_FULL_MATRIX;
    uint32_t nv = 0;
    if (poly) {
        nv = in[1];
        const uint32_t* vin = (const uint32_t*)(in + 2);
        for (uint32_t i = l; i < nv; i += 16) sv[grp][i] = vin[i];
    }
    __syncthreads();
    if (!poly && !whole) return;
    const uint32_t task = hard_list[h];
    const vtx_record rec = records[task >> 1];
    const vtx_locus loc = loci[rec_locus[task >> 1]];
    const int m = (int)rec.read_len, n = (int)((task & 1) ? loc.alt_len : loc.ref_len);
    const int rows = m + 1;
    if (whole) {                                             // no k-mer match: Band::full_matrix (the alternative of include/vtx_band_semantics.h: an empty band)
        for (int j = l; j <= n; j += 16) { lo[j] = VTX_BAND_NO_SEED_FULL_MATRIX ? 0 : 0x7fff; hi[j] = VTX_BAND_NO_SEED_FULL_MATRIX ? (uint16_t)rows : 0; }
        return;
    }
    const uint32_t* v = sv[grp];
    const int cA = (int)(v[0] & 0xffff), cB = (int)(v[nv - 1] & 0xffff);
    for (int j = l; j <= n; j += 16) {
        uint16_t lj = 0x7fff, hj = 0;
        if (j >= cA - BANDW && j <= cB + BANDW) {
            const int c0 = j - BANDW > cA ? j - BANDW : cA;
            const int c1 = j + BANDW < cB ? j + BANDW : cB;
            // first anchor row in column c0: first piece that covers c0
            int rmin = 0, rmax = 0;
            for (uint32_t i = 0; i + 1 < nv; ++i) {
                const int ra = (int)(v[i] >> 16), ca = (int)(v[i] & 0xffff), cb = (int)(v[i + 1] & 0xffff);
                if (c0 >= ca && c0 <= cb) { rmin = (ca == cb) ? ra : ((int)(v[i + 1] >> 16) == ra ? ra : ra + (c0 - ca)); break; }
            }
            // last anchor row in column c1: last piece that covers c1
            for (uint32_t i = nv - 1; i-- > 0;) {
                const int ra = (int)(v[i] >> 16), ca = (int)(v[i] & 0xffff), rb = (int)(v[i + 1] >> 16), cb = (int)(v[i + 1] & 0xffff);
                if (c1 >= ca && c1 <= cb) { rmax = (ca == cb) ? rb : (rb == ra ? ra : ra + (c1 - ca)); break; }
            }
            const int a = rmin - BANDW, b = rmax + BANDW + 1;
            lj = (uint16_t)(a > 0 ? a : 0);
            hj = (uint16_t)(b < rows ? b : rows);
        }
        lo[j] = lj; hi[j] = hj;     // vertices were copied to LDS above: safe to overwrite in place
    }
}

// =============================================================================================
// band_diag_kernel — first stage of the banded flavour when the k-mer tables live in global memory: one lane per task,
// the per-task logic of vtx_fast_core.h (main-diagonal match mask, probes of the few rows that can hold an off-diagonal
// k-mer match, closed-form sdpkpp on the diagonal, certificate = best local score of the mask, run bound over the generic
// pieces).  Tasks it decides get their score here; the others are appended to fail_list for band_run_kernel (task-list
// mode), which handles every shape.  One wavefront per workgroup, 64 consecutive tasks (lanes 2i / 2i + 1 = the two
// haplotypes of one record: their read loads coalesce); the blocks are dealt so that each XCD works through one
// contiguous eighth of the batch (workgroups go to the XCDs round-robin): the ~8 wavefronts that share a locus' tables
// and reads meet in one L2.
// counters[12] = tasks left to band_run_kernel; counters[32 + why] = reasons (stats != 0).
// =============================================================================================

// =============================================================================================
// band_tables_kernel (round 4) — the k-mer tables of loci [l0, l0 + n_loci) in global memory, the same bytes build_tables leaves
// (layout: vtx_fast_core.h, Tab), ONE TABLE per wavefront instead of a locus: half the LDS per wavefront (twice the tables in
// flight per CU: the kernel is a chain of dependent LDS / global round trips per table, and the wavefronts in flight are what
// hides them), and no phase that runs on two lanes:
//   fill    one unaligned 8-byte load per position (bytes 0-5 = the k-mer) instead of six byte loads;
//   link    build_tables inserts the positions one by one in descending order (so that every chain ascends): ~185 dependent LDS
//           round trips on one lane.  Here: rounds of 64 consecutive positions, last round first; inside a round only k-mers of
//           the SAME bucket have to keep their order, so on every trip the largest pending position of each bucket goes (ds_max of
//           y + 1 on a slot; the slots are the 128 words of the still-empty presence bitmap; buckets that share a slot wait for
//           each other, which costs a trip and changes nothing) and clears its slot.  Same insertion order, same chains;
//   flags   uniqueness (chain walk), presence bit and — after one more fence — the head tags, all per position (build_tables
//           walks the 1024 head words for the tags).
// One wavefront per workgroup: its LDS instructions execute in program order, wave_sync() is all the ordering needed.
// =============================================================================================
__global__ __launch_bounds__(64) void band_tables_kernel(const vtx_locus* __restrict__ loci, uint32_t l0, uint32_t n_loci,
                                                         const uint8_t* __restrict__ hap_arena, uint32_t max_hap, uint32_t min_hap,
                                                         uint32_t table_stride, uint32_t n_heads, uint8_t* __restrict__ gtables, uint32_t with_t3) {
    // with_t3 == 0: t3[] is neither built nor written, and the launch leaves its 2 KB out of the workgroup's LDS (band_diag_kernel will not
    // read it: vtxk_launch_band_diag decides per batch, band_use_t3) — the kernel's time is the bytes it writes and the tables in flight per
    // CU.  (The twin lists stay: at four reads per locus they cost the table kernel 0.04 ms and save band_diag_kernel 0.1.)
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint8_t* tb = (uint8_t*)smem;
    const int tid = threadIdx.x;
    uint2* ent = TB_ENT(tb);
    uint16_t* head = TB_HEAD(tb);
    uint8_t* bytes = TB_BYTES(tb);
    uint8_t* fb = TB_FB(tb);
    uint32_t* uq0 = TB_UQ(tb);
    uint32_t* pb = TB_PB(tb);
    const uint32_t hmask = n_heads - 1;
    const uint32_t used_bytes = with_t3 ? table_stride : ((vtxf::tab_t3_off(max_hap, n_heads) + 15u) & ~15u);
    const uint32_t zero_words = (used_bytes - vtxf::tab_uq_off(max_hap, n_heads)) / 4;     // uq[] and, behind it, pb[128], tw[] and t3[]
    for (uint32_t t = blockIdx.x; t < 2u * n_loci; t += gridDim.x) {
        const vtx_locus loc = loci[l0 + (t >> 1)];
        if (max(loc.ref_len, loc.alt_len) <= min_hap) continue;               // (a locus of the other pass: its table is never read; wave-uniform)
        const uint32_t hn = max(loc.ref_len, loc.alt_len) > max_hap ? 0u : ((t & 1) ? loc.alt_len : loc.ref_len);   // (a locus of the slow list: no table)
        const uint8_t* hy = hap_arena + ((t & 1) ? loc.alt_off : loc.ref_off);
        const int nk = hn >= (uint32_t)KMER ? (int)hn - KMER + 1 : 0;
        wave_sync();                                                          // (the copy of the table before has read everything)
        for (uint32_t i = tid; i < n_heads / 4; i += 64) ((uint2*)head)[i] = make_uint2(0xffffffffu, 0xffffffffu);   // CH_END x 4
        for (uint32_t i = tid; i < zero_words; i += 64) uq0[i] = 0;      // (uq[] starts at a multiple of 4 only: no wider stores)
        bool hib = false;
        for (uint32_t y = tid; y < hn; y += 64) {
            const uint64_t w = vtxf::ld8(hy + y);                             // (the arena is padded: bytes beyond the haplotype are never USED)
            const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32) & 0xffffu;
            bytes[y] = (uint8_t)lo;
            fb[y] = (uint8_t)(lo & 0x7fu);
            hib |= (lo & 0x80u) != 0;
            if ((int)y < nk) ent[y] = make_uint2(lo, hi | (kw_hash(lo, hi, hmask) << 16));   // (the bucket rides in the next-pointer field until the link)
        }
        hib = __any(hib);
        wave_sync();
        for (int base = nk > 0 ? ((nk - 1) / 64) * 64 : -1; base >= 0; base -= 64) {
            const int y = base + tid;
            bool pend = y < nk;
            const uint32_t ey = pend ? ent[y].y : 0u;
            const uint32_t h = ey >> 16;
            uint32_t* slot = pb + (h & 127u);
            while (__any(pend)) {
                if (pend) atomicMax(slot, (uint32_t)y + 1u);
                wave_sync();
                const bool win = pend && *slot == (uint32_t)y + 1u;
                wave_sync();
                if (win) {
                    ent[y].y = (ey & 0xffffu) | ((uint32_t)head[h] << 16);
                    head[h] = (uint16_t)y;
                    *slot = 0;
                    pend = false;
                }
                wave_sync();
            }
        }
        // (t3[] built on its own in the LDS the entries take afterwards and copied out first — 2 KB less LDS per wavefront, 30 tables in
        //  flight per CU instead of 22 — was measured: 0.68 against 0.69 ms.  The kernel's time is the bytes it writes now: 1.44 GB.)
        // uniqueness flags (build_tables: a haplotype with a byte >= 0x80 gets none), the presence bitmap, and (round 6) the three-row
        // sets and the twin list — the pairs (y, y') of positions with the same k-mer in (y, y') order: rounds of 64 positions, the
        // twins of a position counted by the walk that decides its uniqueness, a prefix sum per round, the chains ascend (the same
        // list build_twins_wave leaves behind round 3's kernel)
        uint8_t* tw = TB_TW(tb);
        uint32_t* t3 = TB_T3(tb);
        bool tw_ok = !hib && hn <= 256u;
        uint32_t tw_total = 0;
        for (int base = 0; base < nk; base += 64) {
            const int y = base + tid;
            uint32_t c = 0, first = CH_END, klo = 0, khi = 0;
            if (y < nk) {
                const uint2 k = ent[y];
                klo = k.x; khi = k.y & 0xffffu;
                if (!hib) {
                    uint32_t same = 0;
                    first = head[kw_hash(klo, khi, hmask)];
                    for (uint32_t e = first; e != CH_END; e = ent[e].y >> 16)
                        same += (ent[e].x == klo && (ent[e].y & 0xffffu) == khi);
                    if (same == 1) {
                        fb[y + KMER - 1] |= 0x80;                                 // only this lane touches that byte
                        atomicOr(&uq0[vtxf::UQ_PAD_WORDS + ((uint32_t)y >> 5)], 1u << (y & 31));
                    }
                    c = same - 1u;
                }
                const uint32_t code = vtxf::kw_code(klo, khi);
                atomicOr(&pb[code >> 5], 1u << (code & 31u));
                if (vtxf::T3_BYTES != 0 && with_t3) {
                    atomicOr(&t3[vtxf::t3_word_a(code)], 1u << vtxf::t3_bit_a(code));
                    atomicOr(&t3[vtxf::t3_word_b(code)], 1u << vtxf::t3_bit_b(code));
                    atomicOr(&t3[vtxf::t3_word_c(code)], 1u << vtxf::t3_bit_c(code));
                }
            }
            const uint64_t nz = __ballot(c > 0);
            if (tw_ok && nz) {                                                    // (wave-uniform)
                uint32_t off, sum;
                if (__any(c > 1)) {
                    uint32_t inc = c;
#pragma unroll
                    for (int sft = 1; sft < 64; sft <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, sft); if (tid >= sft) inc += v; }
                    off = tw_total + inc - c;
                    sum = (uint32_t)__shfl((int)inc, 63);
                } else {
                    off = tw_total + (uint32_t)__popcll(nz & ((1ull << tid) - 1ull));
                    sum = (uint32_t)__popcll(nz);
                }
                tw_total += sum;
                if (tw_total > vtxf::TW_MAX) tw_ok = false;
                else if (c)
                    for (uint32_t e = first; e != CH_END; e = ent[e].y >> 16)
                        if (e != (uint32_t)y && ent[e].x == klo && (ent[e].y & 0xffffu) == khi) {
                            tw[8 + 2 * off] = (uint8_t)y; tw[9 + 2 * off] = (uint8_t)e; ++off;
                        }
            }
        }
        wave_sync();
        if (!tw_ok) for (uint32_t i = tid; i < vtxf::TW_BYTES / 4; i += 64) ((uint32_t*)tw)[i] = 0;
        wave_sync();
        if (tid == 0) tw[0] = tw_ok ? (uint8_t)tw_total : (uint8_t)vtxf::TW_NONE;
        // head tags: the first position of a chain writes its bucket's head word (a lane that reads a tagged word compares it
        // with its own position, which is not the chain's first: no match either way)
        for (int y = tid; y < nk; y += 64) {
            const uint2 k = ent[y];
            const uint32_t mix = vtxf::kw_mix(k.x, k.y & 0xffffu);
            const uint32_t h = vtxf::kw_bucket(mix, hmask);
            if (head[h] == (uint16_t)y) {
                const uint32_t tag = (k.y >> 16) == CH_END ? vtxf::kw_tag(mix) : vtxf::HEAD_MULTI;
                head[h] = (uint16_t)((uint32_t)y | (tag << 12));
            }
        }
        wave_sync();
        const uint4* src = (const uint4*)smem;
        uint4* dst = (uint4*)(gtables + (size_t)t * table_stride);
        for (uint32_t i = tid; i < used_bytes / 16; i += 64) dst[i] = src[i];
    }
}

// (developer build, VTX_DIAG_PHASES=1) where a wavefront of band_diag_kernel spends its cycles: s_memtime between the phases, summed
// over the wavefronts — phase i in g_diag_phase[i], the wavefronts in [15] (vtxk_diag_phases reads and clears them; tools/diag_phases.py)
// (256 copies, one per workgroup number mod 256: 760 k wavefronts adding to twelve words made the kernel six times slower)
__device__ unsigned long long g_diag_phase[256][16];
extern "C" hipError_t vtxk_diag_phases(unsigned long long* out) {
    static unsigned long long h[256][16];
    hipError_t e = hipMemcpyFromSymbol(h, HIP_SYMBOL(g_diag_phase), sizeof h);
    if (e != hipSuccess) return e;
    for (int i = 0; i < 16; ++i) { out[i] = 0; for (int k = 0; k < 256; ++k) out[i] += h[k][i]; }
    memset(h, 0, sizeof h);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_diag_phase), h, sizeof h);
}
#define REFINE_WORDS 12u       // record of band_refine_kernel: task, d, r | cert << 4 | far matches << 16, zc, RM piece words
extern "C" uint32_t vtxk_band_refine_words(void) { return REFINE_WORDS; }
// ST: type of an off-diagonal match entry — uint16_t (x << 8 | y: 40 entries per task in the same LDS) when every haplotype of
// the batch has <= 255 bases, else uint32_t (20 entries).
// A: mask words in use — 3 when no read of the batch exceeds 192 bases (the instruction count of rounds 3 - 5), else vtxf::NW = 4.
template <int WPE, class ST, int A>
__global__ __launch_bounds__(256, WPE) void band_diag_kernel(
    uint32_t n_tasks, uint32_t task_base, uint32_t n_blocks,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, uint32_t max_hap, uint32_t table_stride, uint32_t n_heads,
    const uint8_t* __restrict__ gtables, uint32_t gt_l0, int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score,
    uint32_t* __restrict__ fail_list, uint32_t* __restrict__ refine_rec, uint32_t refine_cap, uint32_t* __restrict__ counters,
    uint32_t stats, uint32_t* __restrict__ tight_list, uint32_t* __restrict__ tight_pack, uint8_t* __restrict__ stage,
    uint32_t* __restrict__ dense_list, uint32_t dense_mask, uint32_t min_hap) {
    // min_hap: tasks of loci whose longer haplotype has <= min_hap bases are left alone (vtx_run's second pass over a batch that mixes
    // haplotypes of <= 255 bases with a few longer ones: the first pass, with two-byte entries and the sweep behind it, scored them).
    // dense_list != nullptr (round 4): a task left for a reason in dense_mask (bit = vtxf::Why; by default W_MATCHES: more than 40
    // off-diagonal k-mer matches — repeats) goes there (counters[13]): band_run_kernel's piece lists would overflow on it, it takes
    // band_sweep_kernel directly.
    // tight_list != nullptr (round 4): a task whose off-diagonal matches are all harmless but whose bounds do not meet (or whose
    // generic set overflows) HAS a certificate — the chain is the closed form's, cert <= banded — so it leaves with
    // its band (the (2w + 1)-squares along ONE diagonal stretch: tight_pack[i] = vtxf::band_pack) on tight_list (counters[15]):
    // the band-masked DP scores it from that word, without band_sweep_kernel; score = cert meanwhile, a PROVISIONAL score
    // (the optional full-matrix check sw_banded_kernel<.., 1> decides it where full == cert).  Not on fail_list.  stage != nullptr: stage[task] = 1 for every task decided here (vtx_fetch_stage).
    // refine_rec != nullptr: a task with main pieces only whose bounds do not meet leaves a 12-word record for band_refine_kernel
    // (counters[14]; REFINE_WORDS) instead of going to band_run_kernel's list: that kernel prices the stretches of >= 3 errors
    // within a few bases from the real neighbour diagonals and needs nothing else of what this one found.
    // Four wavefronts per workgroup, each on its own 64 consecutive tasks and its own slice of the LDS (no workgroup barrier
    // anywhere): the four share their loci's tables in the CU's L1.
    constexpr int QROWS = 12;                                  // rows a lane contributes to the pool per round
    static_assert(64 * QROWS * 2 >= vtxf::GM * 64 * 4, "back() borrows the probe queue for its generic pieces");
    __shared__ uint32_t lane_mem_[4][vtxf::LANE_WORDS * 64];
    __shared__ uint16_t q_ent_[4][64 * QROWS];                 // pooled probes of one round: owner lane << 8 | row
    __shared__ uint32_t o_read_[4][64], o_tab_[4][64], s_cnt_[4][64];      // per owner lane: read offset, table offset, matches found
    __shared__ int32_t o_diag_[4][64];
    constexpr int WALK_CAP = 224;                              // entries of the walk list (the survivors of pass 1, across rounds)
    __shared__ uint16_t q_walk_[4][WALK_CAP];
    __shared__ uint32_t q_count_[4][2];
    const int wv = threadIdx.x >> 6, tid = threadIdx.x & 63;
    const bool ph_on = VTX_DEVTOOLS_ON && (stats & 0x80000u);
    unsigned long long ph_t = ph_on ? __builtin_readcyclecounter() : 0ull;
    auto PH = [&](int i) {
        if (VTX_DEVTOOLS_ON && ph_on) {
            const unsigned long long now = __builtin_readcyclecounter();
            if (tid == 0) atomicAdd(&g_diag_phase[blockIdx.x & 255u][i], now - ph_t);
            ph_t = now;
        }
    };
    uint32_t* lane_mem = lane_mem_[wv];
    uint16_t* q_ent = q_ent_[wv];
    uint16_t* q_walk = q_walk_[wv];
    uint32_t *o_read = o_read_[wv], *o_tab = o_tab_[wv], *s_cnt = s_cnt_[wv], *q_count = q_count_[wv];
    int32_t* o_diag = o_diag_[wv];
    const uint32_t per_xcd = (n_blocks + 7) / 8;
    const uint32_t blk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (blk >= n_blocks) return;
    const uint32_t slot = blk * 256 + threadIdx.x;
    const bool have = slot < n_tasks;
    const uint32_t task = task_base + slot;
    bool fail = false, live = false, whole = false, twins = false;
    uint32_t why = 0;
    // (where a task's score goes is recomputed at each of the four stores: a pointer held across the kernel is two registers of 128)
    auto my_score = [&]() -> int32_t* { return ((task & 1u) ? alt_score : ref_score) + (task >> 1); };
    vtxf::Front fr;
    fr.why = vtxf::W_SHAPE; fr.d = 0; fr.need = vtxf::m_zero();
    typedef vtxf::LaneS<ST, vtxf::S_WORDS, (A <= 3)> LaneT;
    const LaneT ln{lane_mem + vtxf::S_WORDS * 64 + tid, 64, (ST*)lane_mem + tid, 64};
    vtxf::Tab tb;
    tb.gt = gtables; tb.ent = tb.head = tb.bytes = tb.uq = tb.pb = 0; tb.hmask = n_heads - 1;
    s_cnt[tid] = 0;
    o_read[tid] = 0;
    const uint8_t* x = read_arena;
    int m = 0, n = 0;
    if (have) {
        const uint32_t rid = task >> 1, hap = task & 1;
        const vtx_record rec = records[rid];
        const uint32_t my_locus = rec_locus[rid];
        const vtx_locus loc = loci[my_locus];
        m = (int)rec.read_len; n = (int)(hap ? loc.alt_len : loc.ref_len);
        o_read[tid] = rec.read_off;                                      // (also where this lane drops out: its partner may probe the read)
        if (m == 0 || n == 0) {
            *my_score() = 0;                                             // empty read / haplotype: score 0
            if (stage) stage[task] = 1;
        } else if ((uint32_t)m > VTX_FAST_READ_LEN || max(loc.ref_len, loc.alt_len) > max_hap || max(loc.ref_len, loc.alt_len) <= min_hap) {
            // beyond the fast kernels: slow_align_kernel scores it (the host lists these records); or a locus of the other pass
        } else if (m < vtxf::K || n < vtxf::K || m > 64 * A) {
            fail = true; why = vtxf::W_SHAPE;
        } else {
            tb.ent = (uint32_t)(((size_t)(my_locus - gt_l0) * 2 + hap) * table_stride);
            tb.head = tb.ent + max_hap * 8u;
            tb.bytes = tb.ent + vtxf::tab_bytes_off(max_hap, n_heads);
            tb.uq = tb.ent + vtxf::tab_uq_off(max_hap, n_heads);
            tb.pb = tb.ent + vtxf::tab_pb_off(max_hap, n_heads);
            x = read_arena + rec.read_off;
            live = true;
            // (round 6) a haplotype with a twin list (two-byte entries only: positions are bytes there): the matches of the rows whose
            // main-diagonal k-mer is intact come from the list, the probes look at the other rows only.  (Asked here: the byte is on
            // its way while the read's words and the diagonal are.)
            if constexpr (sizeof(ST) == 2) twins = !(stats & 0x20000u) && vtxf::tab_has_twins(tb);
        }
    }
    if (VTX_ABLATE((stats >> 8) & 0xffu) == 4) { if (live && m == 0x7fffffff) counters[40] = 1; return; }           // (profiling aid) task set-up only
    // ---- the read, once: lanes 2i / 2i + 1 hold the two haplotypes of ONE record, so each loads half of its 8-byte words (16-byte
    //      loads) and the two swap halves — a quarter of the load instructions and of the L2 lines 8-byte loads per lane cost ----
    PH(0);                                                              // task set-up
    vtxf::ReadWords rw;
    {
        constexpr int HW = vtxf::RW / 2;                               // words per lane of a pair (12; 16 with four mask words)
        static_assert(vtxf::RW % 4 == 0, "16-byte loads");
        const int half = tid & 1;
        uint64_t mine[HW];
#pragma unroll
        for (int k = 0; k < HW / 2; ++k) {                             // words [HW half + 2k, + 2): bytes [8 HW half + 16 k, + 16)
            const int off = 8 * HW * half + 16 * k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (live && off < m) __builtin_memcpy(&v, x + off, 16);    // (the arena is padded by 16 bytes)
            mine[2 * k] = (uint64_t)v.x | ((uint64_t)v.y << 32);
            mine[2 * k + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
        }
#pragma unroll
        for (int k = 0; k < HW; ++k) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)mine[k], 1), hi = (uint32_t)__shfl_xor((int)(uint32_t)(mine[k] >> 32), 1);
            const uint64_t theirs = (uint64_t)lo | ((uint64_t)hi << 32);
            rw.w[k] = half ? theirs : mine[k];
            rw.w[HW + k] = half ? mine[k] : theirs;
        }
    }
    // ---- the main diagonal.  Lanes 2i / 2i + 1 hold the two haplotypes of one record: each looks ONE sample row up in its own
    //      table per round and the two exchange their candidates (a candidate is only ever a candidate: the lane keeps it if
    //      ITS mask has >= 20 matching bases) ----
    {
        // rounds of candidate search (cheap: a bucket lookup and an 8-base check each), THEN one mask per lane — a mask per
        // candidate inside the rounds ran the expensive part up to six times per wavefront for a handful of lanes
        bool have_d = !live;
        int d = 0;
        PH(1);                                                          // the read's words
        // (two sample rows per lane and trip — the loads of the two lookups going out together — was measured: 11.9 k -> 13.7 k cycles
        //  of a wavefront's 131 k in this phase; not kept)
#pragma unroll 1
        for (int round = 0; round < vtxf::N_SAMPLES / 2; ++round) {
            if (!__any(!have_d)) break;
            const int c_own = have_d ? vtxf::NO_DIAG : vtxf::cand_diag(x, vtxf::sample_row(2 * round + (tid & 1), m), tb);
            const int c_par = __shfl_xor(c_own, 1);
#pragma unroll 1
            for (int u = 0; u < 2; ++u) {
                const int dc = u ? c_par : c_own;
                if (have_d || dc == vtxf::NO_DIAG || dc < -(m - vtxf::K) || dc > n - vtxf::K) continue;
                if (vtxf::verify_diag(x, m, tb, n, dc)) { d = dc; have_d = true; }
            }
        }
        PH(2);                                                          // the search for the diagonal
        vtxf::M192 M = vtxf::m_zero();
        if (live && have_d) {
            M = vtxf::diag_mask<A>(rw, x, m, tb, n, d);
            if (vtxf::m_pop(M) < 20) have_d = false;
        }
        if (live && !have_d) { live = false; fail = true; why = vtxf::W_NO_DIAG; }
        PH(3);                                                          // the mask
        if (VTX_ABLATE((stats >> 8) & 0xffu) == 3) { if (live && d == 0x7fffffff) counters[40] = 1; return; }       // (profiling aid) up to the diagonal and its mask
        if (live) {
            vtxf::TwinHead th;
            if (twins) th = vtxf::twin_head(tb);
            fr = vtxf::front_rest<LaneT, A>(x, m, tb, n, ln, d, M, twins);
            if (fr.why != vtxf::W_OK) { live = false; fail = true; why = fr.why; }
            else if (VTX_ABLATE((stats >> 8) & 0xffu) != 10 && vtxf::whole_read(fr, m)) {      // (developer build, VTX_DIAG_ABLATE=10: the shortcut off — the A/B switch of include/vtx_band_semantics.h's fifth item; results stay right)
                // the read matches base for base: full <= m = cert, and the reference's chain is a perfect diagonal whatever else
                // matches (vtx_fast_core.h: whole_read) — decided here, before any probe: a fifth of the tasks of a clean workload
                *my_score() = m;
                if (stage) stage[task] = 1;
                live = false; whole = true;
            }
            // (the upper half: how many of the lane's matches are the list's — in order already, the sort starts behind them)
            if (live && twins) s_cnt[tid] = (uint32_t)vtxf::twin_matches(tb, fr, m, ln, th) * 0x10001u;
        }
    }
    PH(4);                                                              // pieces, chain, certificate, rows, twins
    constexpr uint32_t NO_TAB = 0xffffffffu;
    // (bit 0 — table offsets are multiples of 16: the lane took its twin list, so a row ANOTHER lane asked for may be one whose matches
    //  it holds already; the walks below drop those)
    o_tab[tid] = tb.head && !whole ? (tb.ent | (twins ? 1u : 0u)) : NO_TAB;     // (head offset 0: the lane never got as far as its table; a whole read: nobody needs to probe for it)
    o_diag[tid] = (int32_t)(((uint32_t)fr.d & 0xffffu) | ((uint32_t)n << 16));     // the diagonal and the haplotype's length
    // ---- pooled probes: the rows the lanes still have to look up differ a lot from lane to lane (a read that hangs over
    //      the padded window has up to 49 rows without a main-diagonal k-mer), so the wavefront's rows go through one queue
    //      and every lane probes for whoever owns the row.  Pass 1: the presence bitmap (one word of 512 bytes per
    //      haplotype) — most k-mers are not in the haplotype at all; the survivors are compacted in place.  Pass 2: bucket
    //      walk for the survivors; matches land in the owner's list. ----
    // The two lanes of a pair hold ONE read against the two haplotypes of its locus, and the rows they have to look up are nearly
    // the same (they differ around the variant): the pair pools the UNION of its rows — even rows from the even lane, odd rows
    // from the odd one — and an entry (pair, row) is probed in BOTH tables: one load of the read's bytes instead of two, half the
    // entries, half the trips of dependent loads below.  (A row one of the two did not ask for has no off-diagonal match in that
    // haplotype — that is why it was not asked for: looking it up finds nothing.)
    // Round 6, blocks of three rows: the rows a task asks for come in runs (the bases that hang over the haplotype's window, the six
    // rows around an error), and eight read bases hold three consecutive k-mers — the table's t3[] answers "is it in the haplotype" for
    // all three with one 8-byte load (vtx_fast_core.h).  The rows are cut into blocks that END at rows e = m - K (mod 3) (so no block
    // reaches beyond the read's last k-mer; the first may start before row 0: those rows are skipped); a queue entry is
    // pair << 11 | the block's three membership bits << 8 | e, the blocks go to the pair's two lanes in turn.  (Off for shallow batches — band_use_t3 —: a queue entry per row, even rows from the even lane, and pb[] — rounds 3 - 5.)
    const bool t3_mode = vtxf::T3_BYTES != 0 && !(stats & 0x40000u);                // (per batch: vtxk_launch_band_diag, band_use_t3)
    vtxf::M192 nrows = live ? fr.need : vtxf::m_zero();             // (t3_mode: the pair's rows, kept for the membership bits)
    vtxf::M192 nd;
    {
        auto other = [&](uint64_t v) {
            return (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)v, 1) | ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), 1) << 32);
        };
#pragma unroll
        for (int k = 0; k < A; ++k) nrows.w[k] |= other(nrows.w[k]);
        if (t3_mode) {
            const vtxf::M192 s1 = vtxf::m_shl<1>(nrows), s2 = vtxf::m_shl<2>(nrows);
            // the blocks alternate between the pair's lanes (a read hangs over the window on ONE side as a rule: the halves of the read
            // gave one lane all of that run, and a second round of the queue for it alone): block ends e = t (mod 6) to the even lane,
            // e = t + 3 (mod 6) to the odd one, t = m - K
            const uint32_t t = ((uint32_t)(m + 6 * 43 - vtxf::K) + ((tid & 1) ? 3u : 0u)) % 6u;       // (m >= 0)
#pragma unroll
            for (int k = 0; k < vtxf::NW; ++k) {
                // rows 64 k + b with (64 k + b) % 6 == t: 64 = 4 (mod 6), so b % 6 == (t - 4 k) % 6
                const uint32_t j = (t + 24u - 4u * (uint32_t)k) % 6u;
                const uint64_t pat = 0x1041041041041041ull << j;
                nd.w[k] = k < A ? (nrows.w[k] | s1.w[k] | s2.w[k]) & pat : 0ull;
            }
        } else {
            const uint64_t par = (tid & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
#pragma unroll
            for (int k = 0; k < vtxf::NW; ++k) nd.w[k] = k < A ? nrows.w[k] & par : 0ull;
        }
    }
    vtxf::MIter need_it = vtxf::m_iter(nd);
    if (VTX_ABLATE((stats >> 8) & 0xffu) == 1) { if (live && fr.cert == 0x7fffffff) counters[40] = 1; return; }      // (profiling aid) front only
    const uint32_t pb_rel = vtxf::tab_pb_off(max_hap, n_heads), head_rel = max_hap * 8u, bytes_rel = vtxf::tab_bytes_off(max_hap, n_heads);
    const uint32_t t3_rel = vtxf::tab_t3_off(max_hap, n_heads);
    // pass 2 over the first n_walk entries of the walk list (every lane calls it; 0xffff: a slot reserved by a lane that did not fit)
    auto walk_list = [&](uint32_t n_walk) {
        if (VTX_ABLATE((stats >> 8) & 0xffu) == 9) return;                                // (profiling aid) pass 1 without the bucket walks
        // two entries per lane and trip: a walk is three DEPENDENT loads (the read's bytes, the head word of their bucket, the
        // chain's first entry) — the two entries' loads go out together, level by level
        constexpr int WPL = 2;
        for (uint32_t i0 = 0; i0 < n_walk; i0 += 64 * WPL) {
            uint32_t own[WPL], row[WPL], hh[WPL], raw[WPL], tent[WPL];
            uint64_t w8[WPL], e0[WPL], ym8[WPL];
            bool go[WPL], chk[WPL];
            int od[WPL];
#pragma unroll
            for (int u = 0; u < WPL; ++u) {
                const uint32_t i = i0 + 64u * u + tid;
                const uint32_t e = i < n_walk ? q_walk[i] : 0xffffu;
                go[u] = e != 0xffffu;
                own[u] = go[u] ? e >> 8 : 0u; row[u] = go[u] ? e & 0xffu : 0u;       // (an idle slot reads lane 0's first bytes: inside the arena whatever its padding)
                w8[u] = vtxf::ld8(read_arena + o_read[own[u]] + row[u]);
                tent[u] = o_tab[own[u]];
                go[u] = go[u] && tent[u] != NO_TAB;
                if (!go[u]) tent[u] = 0;
                // a lane that took its twin list did not ask for the rows whose main-diagonal k-mer is intact — its pair's other lane
                // may have: the row's matches are in its list already.  Intact = the haplotype's six bytes at (row, row + d) are the read's.
                const uint32_t dn = (uint32_t)o_diag[own[u]];
                od[u] = (int)(int16_t)(uint16_t)dn;
                const int ym = (int)row[u] + od[u];
                chk[u] = go[u] && (tent[u] & 1u) && ym >= 0 && ym + vtxf::K <= (int)(dn >> 16);
                tent[u] &= ~1u;
                ym8[u] = vtxf::ld8(gtables + tent[u] + bytes_rel + (uint32_t)(chk[u] ? ym : 0));
            }
#pragma unroll
            for (int u = 0; u < WPL; ++u) {
                hh[u] = vtxf::kw_mix((uint32_t)w8[u], (uint32_t)(w8[u] >> 32) & 0xffffu);
                raw[u] = vtxf::ld2(gtables + tent[u] + head_rel + 2u * vtxf::kw_bucket(hh[u], n_heads - 1));
            }
#pragma unroll
            for (int u = 0; u < WPL; ++u) {
                const uint32_t tag = raw[u] >> 12;
                go[u] = go[u] && raw[u] != vtxf::HEAD_END && (tag == vtxf::HEAD_MULTI || tag == vtxf::kw_tag(hh[u]));   // (else: the bucket's only k-mer is another one)
                if (chk[u] && ((ym8[u] ^ w8[u]) & 0xffffffffffffull) == 0) go[u] = false;
                e0[u] = vtxf::ld8(gtables + tent[u] + 8u * (go[u] ? raw[u] & 0xfffu : 0u));
            }
#pragma unroll
            for (int u = 0; u < WPL; ++u) {
                if (!go[u]) continue;
                const uint32_t lo = (uint32_t)w8[u], hi = (uint32_t)(w8[u] >> 32) & 0xffffu;
                uint32_t yc = raw[u] & 0xfffu;
                uint64_t e = e0[u];
                for (;;) {
                    if ((uint32_t)e == lo && ((uint32_t)(e >> 32) & 0xffffu) == hi && (int)yc - (int)row[u] != od[u]) {
                        const uint32_t pos = atomicAdd(&s_cnt[own[u]], 1u) & 0xffffu;
                        if (pos < (uint32_t)LaneT::SMAX) ((ST*)lane_mem)[pos * 64 + own[u]] = (ST)((row[u] << LaneT::XS) | yc);
                    }
                    yc = (uint32_t)(e >> 48);
                    if (yc == vtxf::HEAD_END) break;
                    e = vtxf::ld8(gtables + tent[u] + 8u * yc);
                }
            }
        }
    };
    if (tid == 0) q_count[1] = 0;                                   // length of the walk list: it lives across the rounds
    for (;;) {
        if (tid == 0) q_count[0] = 0;
        wave_sync();
        const int cnt = min(QROWS, need_it.left());
        if (!__any(cnt > 0)) break;
        if (cnt > 0) {
            const uint32_t base = atomicAdd(&q_count[0], (uint32_t)cnt);
            if (t3_mode) {
                for (int t = 0; t < cnt; ++t) {
                    const int e = vtxf::m_next(need_it), s0 = e - 2;
                    // the pair's rows s0 .. e as three bits (rows below 0 are no rows)
                    const int kk = s0 >> 6, b = s0 & 63;                                      // (s0 < 0: kk = -1)
                    const uint64_t wl = kk <= 0 ? nrows.w[0] : (kk == 1 ? nrows.w[1] : (A > 3 && kk == 3 ? nrows.w[vtxf::NW - 1] : nrows.w[2]));
                    const uint64_t wh = kk <= 0 ? nrows.w[1] : (kk == 1 ? nrows.w[2] : (A > 3 && kk == 2 ? nrows.w[vtxf::NW - 1] : 0ull));
                    const uint32_t m3 = s0 < 0 ? ((uint32_t)nrows.w[0] << (uint32_t)(-s0)) & 7u
                                               : (uint32_t)((wl >> b) | (b > 61 ? wh << (64 - b) : 0ull)) & 7u;
                    q_ent[base + t] = (uint16_t)(((uint32_t)(tid >> 1) << 11) | (m3 << 8) | (uint32_t)e);
                }
            } else {
                for (int t = 0; t < cnt; ++t) q_ent[base + t] = (uint16_t)(((uint32_t)(tid >> 1) << 8) | (uint32_t)vtxf::m_next(need_it));
            }
        }
        wave_sync();
        PH(5);                                                          // queue fill
        if (VTX_ABLATE((stats >> 8) & 0xffu) == 8) continue;                              // (profiling aid) the rounds' queue fill only
        const uint32_t total = q_count[0];
        constexpr int EPL = 4;                                        // queue entries per lane and trip: their loads go out together
        if (t3_mode) {
          constexpr int EPL = A > 3 ? 2 : 6;                          // (the four-word build has no registers for four: a spilled dword costs every launch its scratch set-up)
          for (uint32_t i0 = 0; i0 < total; i0 += 64 * EPL) {
            uint32_t ent2[EPL], c16[EPL];
            uint64_t w8[EPL], ea[EPL], eb[EPL];
            bool on[EPL];
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const uint32_t i = i0 + 64u * u + tid;
                on[u] = i < total;
                ent2[u] = q_ent[on[u] ? i : 0];
                const int s0 = (int)(ent2[u] & 0xffu) - 2;              // the block's first row (below 0: the read's first bytes, shifted)
                const uint64_t raw8 = vtxf::ld8(read_arena + o_read[(ent2[u] >> 11) * 2u] + (uint32_t)max(s0, 0));
                w8[u] = s0 < 0 ? raw8 << (8 * -s0) : raw8;
            }
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                c16[u] = vtxf::kw_code8(w8[u]);
                const uint32_t ta = o_tab[(ent2[u] >> 11) * 2u], tb_ = o_tab[(ent2[u] >> 11) * 2u + 1u];   // (NO_TAB: a lane without a table)
                const uint32_t so = t3_rel + 8u * ((c16[u] >> 4) & 0xffu);
                ea[u] = ta == NO_TAB ? 0ull : vtxf::ld8(gtables + (ta & ~1u) + so);
                eb[u] = tb_ == NO_TAB ? 0ull : vtxf::ld8(gtables + (tb_ & ~1u) + so);
            }
            uint32_t nh = 0, hits[EPL];                                 // hits: bits 0-2 rows of the block in the even lane's haplotype, 3-5 in the odd lane's
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const uint32_t b0 = c16[u] & 15u, b1 = 16u + (((c16[u] >> 2) & 3u) | (((c16[u] >> 12) & 3u) << 2)), b2 = 32u + (c16[u] >> 12);
                const uint32_t ha = ((uint32_t)(ea[u] >> b0) & 1u) | (((uint32_t)(ea[u] >> b1) & 1u) << 1) | (((uint32_t)(ea[u] >> b2) & 1u) << 2);
                const uint32_t hb = ((uint32_t)(eb[u] >> b0) & 1u) | (((uint32_t)(eb[u] >> b1) & 1u) << 1) | (((uint32_t)(eb[u] >> b2) & 1u) << 2);
                const uint32_t m3 = on[u] ? (ent2[u] >> 8) & 7u : 0u;
                hits[u] = (ha & m3) | ((hb & m3) << 3);
                nh += (uint32_t)__builtin_popcount(hits[u]);
            }
            bool todo = nh > 0;
            for (;;) {
                if (todo) {
                    uint32_t pos = atomicAdd(&q_count[1], nh);
                    if (pos + nh <= (uint32_t)WALK_CAP) {
#pragma unroll
                        for (int u = 0; u < EPL; ++u) {                 // (walk entries name the OWNER lane: pair * 2 + haplotype)
                            const uint32_t pr9 = (ent2[u] >> 11) << 9, e = ent2[u] & 0xffu;
#pragma unroll
                            for (int j = 0; j < 6; ++j)                  // (a row that hit is a row of the block's membership bits: e + j % 3 >= 2)
                                if ((hits[u] >> j) & 1u) q_walk[pos++] = (uint16_t)(pr9 | (j >= 3 ? 0x100u : 0u) | (e + (uint32_t)(j % 3) - 2u));
                        }
                        todo = false;
                    } else {
                        for (; pos < (uint32_t)WALK_CAP; ++pos) q_walk[pos] = 0xffffu;
                    }
                }
                wave_sync();
                const uint32_t wc = q_count[1];
                if (wc <= (uint32_t)WALK_CAP) break;                   // (everybody fitted)
                walk_list((uint32_t)WALK_CAP);
                wave_sync();
                if (tid == 0) q_count[1] = 0;
                wave_sync();
            }
          }
        } else
        for (uint32_t i0 = 0; i0 < total; i0 += 64 * EPL) {
            uint32_t ent2[EPL], code[EPL], bits_a[EPL], bits_b[EPL];
            uint64_t w8[EPL];
            bool on[EPL];
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                const uint32_t i = i0 + 64u * u + tid;
                on[u] = i < total;
                ent2[u] = q_ent[on[u] ? i : 0];
                w8[u] = vtxf::ld8(read_arena + o_read[(ent2[u] >> 8) * 2u] + (ent2[u] & 0xffu));
            }
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                code[u] = vtxf::kw_code((uint32_t)w8[u], (uint32_t)(w8[u] >> 32) & 0xffffu);
                const uint32_t ta = o_tab[(ent2[u] >> 8) * 2u], tb_ = o_tab[(ent2[u] >> 8) * 2u + 1u];   // (NO_TAB: a lane without a table)
                bits_a[u] = ta == NO_TAB ? 0u : *(const uint32_t*)(gtables + (ta & ~1u) + pb_rel + 4u * (code[u] >> 5));
                bits_b[u] = tb_ == NO_TAB ? 0u : *(const uint32_t*)(gtables + (tb_ & ~1u) + pb_rel + 4u * (code[u] >> 5));
            }
            // the survivors (5 % of the probes) join the walk list: one LDS add per lane that has any, no ballots.  The list is
            // short (WALK_CAP entries — the LDS this kernel has left): a lane that does not fit marks what it reserved inside the
            // list as empty, the list is walked and emptied, and the lane tries again.
            uint32_t nh = 0;
            bool hit_a[EPL], hit_b[EPL];
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                hit_a[u] = on[u] && ((bits_a[u] >> (code[u] & 31u)) & 1u);
                hit_b[u] = on[u] && ((bits_b[u] >> (code[u] & 31u)) & 1u);
                nh += (hit_a[u] ? 1u : 0u) + (hit_b[u] ? 1u : 0u);
            }
            bool todo = nh > 0;
            for (;;) {
                if (todo) {
                    uint32_t pos = atomicAdd(&q_count[1], nh);
                    if (pos + nh <= (uint32_t)WALK_CAP) {
#pragma unroll
                        for (int u = 0; u < EPL; ++u) {                 // (walk entries name the OWNER lane: pair * 2 + haplotype)
                            const uint32_t e2 = ((ent2[u] >> 8) << 9) | (ent2[u] & 0xffu);
                            if (hit_a[u]) q_walk[pos++] = (uint16_t)e2;
                            if (hit_b[u]) q_walk[pos++] = (uint16_t)(e2 | 0x100u);
                        }
                        todo = false;
                    } else {
                        for (; pos < (uint32_t)WALK_CAP; ++pos) q_walk[pos] = 0xffffu;
                    }
                }
                wave_sync();
                const uint32_t wc = q_count[1];
                if (wc <= (uint32_t)WALK_CAP) break;                   // (everybody fitted)
                walk_list((uint32_t)WALK_CAP);
                wave_sync();
                if (tid == 0) q_count[1] = 0;
                wave_sync();
            }
        }
        PH(6);                                                          // pass 1 (and the walks of a full walk list)
    }
    PH(5);
    walk_list(q_count[1]);
    wave_sync();
    PH(7);                                                              // the walks
    if (VTX_ABLATE((stats >> 8) & 0xffu) == 2) { if (live && s_cnt[tid] == 0x7fffffffu) counters[40] = 1; return; }   // (profiling aid) front + probes
    uint32_t aux = 0xffffffffu;
    bool tight = false;
    // ---- the last phase, every lane for itself: sort, harmless tests, closure, run bound (vtx_fast_core.h).  (Pooling the harmless
    //      tests over the wavefront like the probes — one queue entry per match, A | T << 8 left in the entry, a short recurrence per
    //      owner — was built and measured: 1.67 ms pooled against 1.7 ms per lane, 18.24 against 18.33 ms per step: not kept.) ----
    const int ns = (int)min(s_cnt[tid] & 0xffffu, (uint32_t)LaneT::SMAX + 1u);
    bool spread = false;
    if (live && ns > LaneT::SMAX) {
        live = false; fail = true; why = vtxf::W_MATCHES;
        // More off-diagonal matches than a lane holds.  Two very different tasks end here: a read against the OTHER allele of an
        // indel (its second half lies on ONE other diagonal: band_run_kernel's two-piece case, decided by its certificate) and
        // repeats (matches on many diagonals: band_run_kernel's piece lists overflow, band_sweep_kernel takes them).  Eight of the
        // matches found so far tell them apart: no diagonal holds four of them -> spread.
        if (dense_list) {
            int dg[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const uint32_t w = (uint32_t)ln.s(5 * i); dg[i] = (int)(w & LaneT::YM) - (int)(w >> LaneT::XS); }
            int mode = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) cnt += dg[j] == dg[i] ? 1 : 0;
                mode = max(mode, cnt);
            }
            spread = mode <= 3;
        }
    }
    if (live) vtxf::back_sort(ns, ln, (int)(s_cnt[tid] >> 16));
    PH(8);                                                              // sort
    if (VTX_ABLATE((stats >> 8) & 0xffu) == 7) { if (live && ns == 0x7fffffff) counters[40] = 1; return; }            // (profiling aid) ... + the sort
    if (live && !vtxf::back_harmless(fr, ns, ln)) { live = false; fail = true; why = vtxf::W_NOT_HARMLESS; }
    PH(9);                                                              // harmless tests
    if (live) {
        const vtxf::Lane gl{(uint32_t*)q_ent + tid, 64};                 // (the queue is dead by now)
        const int32_t sc = vtxf::back_rest(fr, ns, ln, gl, &why, (int)VTX_ABLATE((stats >> 8) & 0xffu), nullptr, &aux);
        if (sc >= 0) { *my_score() = sc; if (stage) stage[task] = 1; }
        else { fail = true; tight = tight_list != nullptr; }
    }
    // Round 6, the sweep path (a tight list): EVERY task that leaves the last phase with its certificate — harmless matches only, bounds
    // apart or the generic set full — leaves a record for band_corridor_kernel (the task, its one-diagonal band, the rows of its
    // matches far out: vtx_fast_core.h, "the corridor certificate") instead of taking the masked DP; what that kernel does not decide
    // goes on to the tight list.  (stats bit 16 — libvtx_dev.so, VTX_BAND_NO_CORRIDOR=1 — and batches with haplotypes above 255 bases:
    // round 3's records for band_refine_kernel, main pieces only.)
    PH(10);                                                             // closure, run bound
    const bool corr_mode = tight_list != nullptr && refine_rec != nullptr && !(stats & 0x10000u);
    bool again = corr_mode ? (fail && tight) : (fail && why == vtxf::W_NOT_TIGHT && refine_rec != nullptr && aux != 0xffffffffu);
    const uint64_t am = __ballot(again);
    if (am) {
        uint32_t base = 0;
        const int leader = __ffsll((long long)am) - 1;
        if (tid == leader) base = atomicAdd(&counters[14], (uint32_t)__popcll(am));
        base = (uint32_t)__shfl((int)base, leader);
        const uint32_t pos = base + (uint32_t)__popcll(am & ((1ull << tid) - 1ull));
        if (again && pos >= refine_cap) again = false;              // (the record buffer is full: band_run_kernel takes it)
        if (again && corr_mode) {
            static_assert(REFINE_WORDS >= 4 + 2 * vtxf::NW, "the far rows fit the record");
            const vtxf::M192 far = vtxf::far_rows(fr.d, ns, ln);
            uint32_t* rec = refine_rec + (size_t)pos * REFINE_WORDS;
            rec[0] = task; rec[1] = vtxf::band_pack(fr);
            rec[2] = (uint32_t)fr.cert; rec[3] = 0;
#pragma unroll
            for (int i = 0; i < vtxf::NW; ++i) { rec[4 + 2 * i] = (uint32_t)far.w[i]; rec[5 + 2 * i] = (uint32_t)(far.w[i] >> 32); }
        } else if (again) {
            uint32_t* rec = refine_rec + (size_t)pos * REFINE_WORDS;
            rec[0] = task; rec[1] = vtxf::band_pack(fr);               // (the diagonal is its upper half)
            rec[2] = (uint32_t)fr.r | ((uint32_t)fr.cert << 4) | (aux << 16);
            rec[3] = fr.zc;
            for (int i = 0; i < vtxf::RM; ++i) rec[4 + i] = i < fr.r ? ln.at(i) : 0u;
        }
    }
    tight = tight && fail && !again;
    const uint64_t tm = __ballot(tight);
    if (tm) {
        uint32_t base = 0;
        const int leader = __ffsll((long long)tm) - 1;
        if (tid == leader) base = atomicAdd(&counters[15], (uint32_t)__popcll(tm));
        base = (uint32_t)__shfl((int)base, leader);
        if (tight) {
            const uint32_t pos = base + (uint32_t)__popcll(tm & ((1ull << tid) - 1ull));
            tight_list[pos] = task;
            tight_pack[pos] = vtxf::band_pack(fr);
            *my_score() = fr.cert;                                    // provisional: a lower bound of the banded score
        }
    }
    const bool dense = fail && !again && !tight && dense_list != nullptr && ((dense_mask >> why) & 1u) && (why != vtxf::W_MATCHES || spread);
    const uint64_t dm = __ballot(dense);
    if (dm) {
        uint32_t base = 0;
        const int leader = __ffsll((long long)dm) - 1;
        if (tid == leader) base = atomicAdd(&counters[13], (uint32_t)__popcll(dm));
        base = (uint32_t)__shfl((int)base, leader);
        if (dense) dense_list[base + (uint32_t)__popcll(dm & ((1ull << tid) - 1ull))] = task;
    }
    const uint64_t fm = __ballot(fail && !again && !tight && !dense);
    if (fm) {
        uint32_t base = 0;
        const int leader = __ffsll((long long)fm) - 1;
        if (tid == leader) base = atomicAdd(&counters[12], (uint32_t)__popcll(fm));
        base = (uint32_t)__shfl((int)base, leader);
        if (fail && !again && !tight && !dense) fail_list[base + (uint32_t)__popcll(fm & ((1ull << tid) - 1ull))] = task;
    }
    if (fail && !again && (stats & 0xffu)) atomicAdd(&counters[32 + why], 1u);
    PH(11);                                                             // lists
    if (VTX_DEVTOOLS_ON && ph_on && tid == 0) atomicAdd(&g_diag_phase[blockIdx.x & 255u][15], 1ull);
}

// =============================================================================================
// band_refine_kernel — second chance of the tasks band_diag_kernel left because cert != ub (round 3).  One lane per record
// (task, diagonal, certificate, far matches, the main pieces and the mismatches between them: everything the main-only run bound
// needs) runs that bound again (vtxf::main_pieces_ub, the function band_diag_kernel ran), this time with the corridor refinement
// of the same-diagonal joins (vtx_fast_core.h: corridor_cost): where >= 3 errors fall within a few bases the gap-free cost
// 6 e - D is above J_gap(D), the price of a hypothetical excursion over perfectly matching neighbour diagonals; an affine DP
// over the five diagonals around the stretch (~25 rows) prices the real ones.  At 3 % substitution errors that decides 55 % of
// the records (hard tasks 6.5 -> 3.3 % of all); what is still undecided goes to band_run_kernel's list.  A compact list instead
// of a branch inside band_diag_kernel: there the DP would run in nearly every wavefront for 6 % of the lanes.
// =============================================================================================
__global__ __launch_bounds__(256) void band_refine_kernel(
    const uint32_t* __restrict__ recs, uint32_t n_recs,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, uint32_t max_hap, uint32_t table_stride, uint32_t n_heads,
    const uint8_t* __restrict__ gtables, uint32_t gt_l0, int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score,
    uint32_t* __restrict__ fail_list, uint32_t* __restrict__ counters, uint32_t stats, uint32_t* __restrict__ tight_list,
    uint32_t* __restrict__ tight_pack, uint8_t* __restrict__ stage, const uint32_t* __restrict__ n_dev) {
    // tight_list != nullptr (round 4): an undecided record still has its certificate: provisional score + tight_list (counters[15])
    // instead of fail_list (see band_diag_kernel).  n_dev: the record count lives on the device (min(*n_dev, n_recs)).
    __shared__ uint32_t piece_mem_[4][vtxf::RM * 64];
    if (n_dev) { const uint32_t nd = *n_dev; n_recs = nd < n_recs ? nd : n_recs; }
    const int wv = threadIdx.x >> 6, tid = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    bool fail = false;
    uint32_t task = 0, pack = 0;
    if (slot < n_recs) {
        const uint4* rp = (const uint4*)(recs + (size_t)slot * REFINE_WORDS);
        const uint4 h = rp[0], p0 = rp[1], p1 = rp[2];
        task = h.x;
        const int d = (int)(h.y >> 16) - 256, r = (int)(h.z & 15u), cert = (int)((h.z >> 4) & 0xfffu), far_e = (int)(h.z >> 16);
        pack = h.y;
        const uint32_t rid = task >> 1, hap = task & 1;
        const vtx_record rec = records[rid];
        const uint32_t my_locus = rec_locus[rid];
        const vtx_locus loc = loci[my_locus];
        const vtxf::Lane pl{piece_mem_[wv] + tid, 64};
        pl.at(0) = p0.x; pl.at(1) = p0.y; pl.at(2) = p0.z; pl.at(3) = p0.w;
        pl.at(4) = p1.x; pl.at(5) = p1.y; pl.at(6) = p1.z; pl.at(7) = p1.w;
        const uint8_t* yb = gtables + ((size_t)(my_locus - gt_l0) * 2 + hap) * table_stride + vtxf::tab_bytes_off(max_hap, n_heads);
        const vtxf::Refine rf{read_arena + rec.read_off, yb, (int)rec.read_len, (int)(hap ? loc.alt_len : loc.ref_len)};
        // Round 6 (vtx_band_trim.h): next to the refined bound of the FULL score, the same bound over the pieces TRIMMED to the rows whose
        // main-diagonal cell lies in the band — a bound of the BANDED score (a path inside the band touches in-band cells only).  Where
        // the band cuts end pieces off (banded < full: half of what the refinement leaves on noisy reads) no bound of the full score can
        // meet the certificate; this one does.  Both in one pass: the corridor DPs of the joins are shared.  A tight list means every
        // haplotype has <= 255 bases: ca / cb are the bytes of the pack.
        const int base_ub = max(vtxf::K - 1, far_e > 0 ? far_e + 5 : 0);
        int ub, ubb = -1;
        if (tight_list) {
            const int ca = (int)((h.y >> 8) & 0xffu), cb = (int)(h.y & 0xffu);
            const int lo = max(0, ca - vtxf::W - 1 - d), hi = min((int)rec.read_len - 1, cb + vtxf::W - 1 - d);
            int band_part = 0;
            ub = max(base_ub, vtxf::main_pieces_ub_both(pl, r, h.w, d, &rf, far_e, lo, hi, &band_part));
            ubb = max(base_ub, band_part);
        } else ub = max(base_ub, vtxf::main_pieces_ub(pl, r, h.w, d, &rf, far_e));
        (hap ? alt_score : ref_score)[rid] = cert;                     // final when a bound meets it, provisional otherwise
        if (ub == cert) { if (stage) stage[task] = VTX_STAGE_REFINE_CERT; }
        else if (ubb == cert) { if (stage) stage[task] = VTX_STAGE_BAND_CERT; }      // a stage of its own: banded < full is allowed here
        else fail = true;
    }
    const uint64_t fm = __ballot(fail);
    if (fm) {
        uint32_t base = 0;
        const int leader = __ffsll((long long)fm) - 1;
        if (tid == leader) base = atomicAdd(&counters[tight_list ? 15 : 12], (uint32_t)__popcll(fm));
        base = (uint32_t)__shfl((int)base, leader);
        if (fail) {
            const uint32_t pos = base + (uint32_t)__popcll(fm & ((1ull << tid) - 1ull));
            (tight_list ? tight_list : fail_list)[pos] = task;
            if (tight_list) tight_pack[pos] = pack;
            if (stats & 0xffu) atomicAdd(&counters[32 + vtxf::W_NOT_TIGHT], 1u);
        }
    }
}

// =============================================================================================
// band_corridor_kernel (round 6) — the tasks band_diag_kernel leaves WITH a certificate and a one-diagonal band (every off-diagonal
// match harmless; bounds apart): one lane per record runs the corridor certificate (vtx_fast_core.h: corridor_bound — an exact
// affine-gap DP over the cells of the band within CC = 8 diagonals of the main one, 17 values per row in registers, plus "excursion"
// edges that bound whatever a path of the band can do outside the corridor; a maximum reached without an excursion edge IS the banded
// score).  ~45 k lane-instructions per task where the masked DP over the whole band (sw_banded_kernel<., ., 2>) spends 2 600 wave-
// instructions per task, and it decides 99.7 % of these tasks at 8 % substitution errors: the masked DP, 115 ms of that workload's
// step, and band_refine_kernel (36 ms) are replaced by this kernel.  What it does not decide keeps its certificate as a provisional
// score and goes on to the tight list (masked DP), as before.
// =============================================================================================
__global__ __launch_bounds__(256) void band_corridor_kernel(
    const uint32_t* __restrict__ recs, uint32_t n_recs,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score, uint32_t* __restrict__ counters, uint32_t stats,
    uint32_t* __restrict__ tight_list, uint32_t* __restrict__ tight_pack, uint8_t* __restrict__ stage, const uint32_t* __restrict__ n_dev) {
    if (n_dev) { const uint32_t nd = *n_dev; n_recs = nd < n_recs ? nd : n_recs; }
    const int tid = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    bool fail = false;
    uint32_t task = 0, pack = 0;
    if (slot < n_recs) {
        const uint4* rp = (const uint4*)(recs + (size_t)slot * REFINE_WORDS);
        const uint4 h = rp[0], p0 = rp[1], p1 = rp[2];
        task = h.x; pack = h.y;
        const int d = (int)(pack >> 16) - 256, ca = (int)((pack >> 8) & 0xffu), cb = (int)(pack & 0xffu), cert = (int)h.z;
        vtxf::M192 far = vtxf::m_zero();
        far.w[0] = (uint64_t)p0.x | ((uint64_t)p0.y << 32); far.w[1] = (uint64_t)p0.z | ((uint64_t)p0.w << 32);
        far.w[2] = (uint64_t)p1.x | ((uint64_t)p1.y << 32);
        if constexpr (vtxf::NW > 3) far.w[vtxf::NW - 1] = (uint64_t)p1.z | ((uint64_t)p1.w << 32);
        const uint32_t rid = task >> 1, hap = task & 1;
        const vtx_record rec = records[rid];
        const vtx_locus loc = loci[rec_locus[rid]];
        const int sc = vtxf::corridor_bound(read_arena + rec.read_off, (int)rec.read_len, hap_arena + (hap ? loc.alt_off : loc.ref_off),
                                            (int)(hap ? loc.alt_len : loc.ref_len), d, ca, cb, far);
        (hap ? alt_score : ref_score)[rid] = sc >= 0 ? sc : cert;     // final, or the certificate as a provisional score
        if (sc >= 0) { if (stage) stage[task] = VTX_STAGE_CORRIDOR_CERT; }
        else fail = true;
    }
    const uint64_t fm = __ballot(fail);
    if (fm) {
        uint32_t base = 0;
        const int leader = __ffsll((long long)fm) - 1;
        if (tid == leader) base = atomicAdd(&counters[15], (uint32_t)__popcll(fm));
        base = (uint32_t)__shfl((int)base, leader);
        if (fail) {
            const uint32_t pos = base + (uint32_t)__popcll(fm & ((1ull << tid) - 1ull));
            tight_list[pos] = task;
            tight_pack[pos] = pack;
            if (stats & 0xffu) atomicAdd(&counters[32 + vtxf::W_NOT_TIGHT], 1u);
        }
    }
}
extern "C" hipError_t vtxk_launch_band_corridor(const uint32_t* recs, uint32_t n_recs, const vtx_record* records, const uint32_t* rec_locus,
                                                const vtx_locus* loci, const uint8_t* read_arena, const uint8_t* hap_arena,
                                                int32_t* ref_score, int32_t* alt_score, uint32_t* counters, int stats,
                                                uint32_t* tight_list, uint32_t* tight_pack, uint8_t* stage, const uint32_t* n_dev, hipStream_t s) {
    if (!n_recs) return hipSuccess;
    hipLaunchKernelGGL(band_corridor_kernel, dim3((n_recs + 255) / 256), dim3(256), 0, s, recs, n_recs, records, rec_locus, loci, read_arena,
                       hap_arena, ref_score, alt_score, counters, (uint32_t)stats, tight_list, tight_pack, stage, n_dev);
    return hipGetLastError();
}

// Resident workgroups of band_run_kernel (an upper bound: 256 CUs x the most workgroups a CU can hold for that block
// size); the per-lane scratch is sized from it.
// tables in global memory below this many tasks per locus (experiment knob VTX_BAND_GT_MAX_TPL; 0: never)
static uint32_t gt_max_tpl() {
    static const uint32_t v = VTX_DEV_ENV("VTX_BAND_GT_MAX_TPL") ? (uint32_t)atoi(VTX_DEV_ENV("VTX_BAND_GT_MAX_TPL")) : 0x7fffffffu;
    return v;
}
// the tables of n_loci loci in global memory: band_tables_kernel (a table per wavefront); VTX_BAND_TABLES_V1=1: round 3's kernel
// (a locus per wavefront, serial chain insertion) — the reference the new one is compared with byte for byte (tests)
static void launch_band_tables(const vtx_locus* loci, uint32_t gt_l0, uint32_t n_loci, const uint8_t* hap_arena, uint32_t max_hap, uint32_t min_hap,
                               size_t tstride, uint32_t n_heads, uint8_t* gtables, bool with_t3, hipStream_t s) {
#ifdef VTX_DEVTOOLS
    if (VTX_DEV_ENV("VTX_BAND_TABLES_V1")) {
        hipLaunchKernelGGL(band_tables_v1_kernel, dim3(std::min(n_loci, 256u * 16u)), dim3(64), 2 * tstride, s, loci, gt_l0, n_loci,
                           hap_arena, max_hap, (uint32_t)tstride, n_heads, gtables, with_t3 ? 1u : 0u);
        return;
    }
#endif
    const size_t lds = with_t3 ? tstride : (size_t)((vtxf::tab_t3_off(max_hap, n_heads) + 15u) & ~15u);     // (the kernel's used_bytes)
    hipLaunchKernelGGL(band_tables_kernel, dim3(std::min(2u * n_loci, 256u * 32u)), dim3(64), lds, s, loci, gt_l0, n_loci,
                           hap_arena, max_hap, min_hap, (uint32_t)tstride, n_heads, gtables, with_t3 ? 1u : 0u);
}
// Blocks of three rows and t3[] (vtx_fast_core.h): 2 KB more to write per table, whatever the depth.  Headline 14.67 ms per step without
// them, 13.61 with (the blocks halve the queue's entries and balance them over a pair's lanes: one round instead of two); 250-base
// reads 16.9 -> 15.7; at four reads per locus, where the tables are a third of the step, 1.57 ms without against 1.67 with, level at
// sixteen.  Per batch: on from 16 tasks per locus.  (libvtx_dev.so: VTX_DIAG_T3=1 / VTX_DIAG_NO_T3=1 force it on / off.)
static bool band_use_t3(uint32_t tasks_per_locus) {
    if (vtxf::T3_BYTES == 0 || VTX_DEV_ENV("VTX_DIAG_NO_T3")) return false;
    if (VTX_DEV_ENV("VTX_DIAG_T3")) return true;
    return tasks_per_locus >= 16;
}

// buckets of a table's hash (power of two; experiment knob VTX_BAND_HEADS)
static uint32_t pick_heads(uint32_t tasks_per_locus, bool global_tables) {
    if (VTX_DEV_ENV("VTX_BAND_HEADS")) return (uint32_t)atoi(VTX_DEV_ENV("VTX_BAND_HEADS"));
    if (global_tables) return 1024;          // no LDS to fit: short chains, and band_diag_kernel's bucket tags want single-entry buckets
    if (tasks_per_locus < 48) return 256;
    return 512;
}

// Global table buffer for this shape of data: bytes to reserve (0: tables live in LDS) and how many loci they hold.  The
// kernel addresses the tables with 32-bit offsets: at most 4 GB, i.e. ~400 k deep / ~580 k shallow loci at padding 100 —
// a batch with more loci runs the banded stage in chunks of tasks whose loci fit (vtx_run).
extern "C" size_t vtxk_band_gtables_bytes(uint32_t n_loci, uint32_t max_hap, uint32_t tasks_per_locus, uint32_t* loci_cap) {
    if (loci_cap) *loci_cap = 0;
    if (tasks_per_locus >= gt_max_tpl()) return 0;
    const size_t per_locus = 2 * band_table_stride(max_hap, pick_heads(tasks_per_locus, true));
    size_t cap = ((size_t)4 << 30) - 65536;
    if (VTX_DEV_ENV("VTX_BAND_GT_BYTES")) cap = std::max<size_t>(per_locus, strtoull(VTX_DEV_ENV("VTX_BAND_GT_BYTES"), nullptr, 10));   // test hook
    const size_t hold = std::min<size_t>(n_loci, cap / per_locus);
    if (loci_cap) *loci_cap = (uint32_t)hold;
    return hold * per_locus;
}

// 1: vtxk_launch_band_run picks the six-wavefront / 12-entry variant for this shape — its overflow list deserves a second
// chance in the 15-entry variant (task_list mode) before the general kernel
extern "C" int vtxk_band_second_chance(uint32_t tasks_per_locus, int long_lists) {
    if (long_lists) return 0;
    const uint32_t w6 = VTX_DEV_ENV("VTX_BAND_W6_MIN_TPL") ? (uint32_t)atoi(VTX_DEV_ENV("VTX_BAND_W6_MIN_TPL")) : 80u;
    return tasks_per_locus >= w6 && tasks_per_locus < gt_max_tpl() && !VTX_DEV_ENV("VTX_BAND_NO_SECOND_CHANCE");
}

// persistent grid of band_run_kernel<nt, ., wpe>: what the chip holds (wavefronts per SIMD x 4 SIMDs x 256 CUs)
extern "C" uint32_t vtxk_band_run_grid(uint32_t nt, uint32_t wpe) { return 256u * (nt == 64 ? 4u * wpe : wpe); }
extern "C" uint32_t vtxk_band_run_lanes(void) { return 256u * 256u * (uint32_t)std::max(VTX_WPE, 6); }     // max over both block sizes of grid x nt

extern "C" hipError_t vtxk_launch_band_run(uint32_t n_tasks, uint32_t task_base, const vtx_record* records,
                                           const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                           const uint8_t* hap_arena, uint32_t max_hap, uint32_t min_hap, int32_t* ref_score,
                                           int32_t* alt_score, uint32_t* logbuf, uint16_t* band,
                                           uint32_t band_stride, uint32_t* hard_list, uint32_t* overflow_list,
                                           uint32_t* pending_list, uint32_t* pend_buf, uint32_t hard_cap, uint32_t pend_cap,
                                           uint32_t* counters, uint32_t tasks_per_locus, uint32_t gt_l0, uint32_t n_loci,
                                           uint8_t* gtables, size_t gtables_bytes, const uint32_t* task_list, int long_lists,
                                           hipStream_t s) {
    // long_lists: the caller saw many tasks overflow the 12-entry lists (noisy reads): 15-entry lists from the start
    // gtables != nullptr: room for the tables of loci [gt_l0, gt_l0 + n_loci) — the loci of THIS range of tasks; they are
    // built here, then read by the kernel
    if (!n_tasks) return hipSuccess;
    // (LDS-table variants, the fallback:) deep data (>= 64 tasks per locus): 256-task workgroups.  A workgroup processes its loci in passes of `tables / 2`
    // loci (the k-mer tables live in LDS); in a pass only the lanes of those loci work.  512-entry head arrays: two loci
    // fit next to the 28 KiB of lane arrays at 4 workgroups per CU, one pass per workgroup nearly always (2048-entry heads
    // fit one locus: 109 ms instead of 91 ms on config 3; 256-entry heads: 101 ms, the chains get longer).
    // Shallow data: one WAVEFRONT per workgroup (64 tasks), all of its loci resident at once when they fit — with
    // 256-task workgroups the wavefronts of a workgroup took turns (a pass holds the loci of one wavefront's tasks and
    // the other three wait at the barrier), so a CU had two working wavefronts; now every resident wavefront works.
    // Tables.  With a table buffer from the caller every locus' tables are built once per run in global memory
    // (band_tables_kernel) and the workgroups keep only their lane arrays in LDS: no table passes with idle lanes, five
    // wavefronts per SIMD, and nothing ties the lanes of a workgroup together any more — one wavefront per workgroup
    // (no barrier per block: config 3 51.1 -> 50.6 ms, 64 reads per locus +4 %).  Without the buffer (or when the
    // tables would not fit it) they live in LDS: 256-task workgroups for deep loci (two tables per pass next to the lane
    // arrays, four workgroups per CU), one-wavefront workgroups with up to 32 tables below 64 tasks per locus.
    const bool want_global = tasks_per_locus < gt_max_tpl() && gtables;
    uint32_t n_heads = pick_heads(tasks_per_locus, want_global);
    size_t tstride = band_table_stride(max_hap, n_heads, !want_global);
    if (want_global && (size_t)n_loci * 2 * tstride > gtables_bytes) {       // the buffer is too small: tables in LDS
        n_heads = pick_heads(tasks_per_locus, false);
        tstride = band_table_stride(max_hap, n_heads, true);
    }
    const bool global_tables = want_global && (size_t)n_loci * 2 * tstride <= gtables_bytes;
    const bool wave_wg = global_tables || tasks_per_locus < 64;
    const uint32_t nt = wave_wg ? 64 : 256;
    // wavefronts per SIMD and list entries per lane.  Tables in LDS: 4 x 15 (111 VGPRs; 5 -> 93 VGPRs cost more than the
    // occupancy gave, 3 less still).  Tables in global memory: the lanes wait on L2 / HBM instead of LDS and LDS holds
    // only the lists — 6 x 12 for deeper loci (config 3: band_run 45.0 -> 39.9 ms; the shorter lists send 2.7x the
    // tasks to the general kernel, still -2.5 ms per step), 5 x 15 for shallower ones (16 reads per locus: 6.9 vs 7.7 ms),
    // 4 x 15 when the grid does not fill the chip anyway.
    static const uint32_t w6_min_tpl = VTX_DEV_ENV("VTX_BAND_W6_MIN_TPL") ? (uint32_t)atoi(VTX_DEV_ENV("VTX_BAND_W6_MIN_TPL")) : 80u;   // experiment knob (crossover between 32 and 48 reads per locus)
    const int variant = !global_tables ? 0 : ((task_list || long_lists) ? (tasks_per_locus < 16 && !task_list ? 1 : 2) : (tasks_per_locus < 16 ? 1 : (tasks_per_locus < w6_min_tpl ? 2 : 3)));
    const uint32_t psv = variant == 3 ? 12u : (uint32_t)VTX_PS;
    const size_t lane_bytes = (size_t)(2 * psv) * nt * 4;
    uint32_t tables;
    if (global_tables) {
        tables = 2;
    } else if (wave_wg) {
        // loci of 64 consecutive tasks (+1 for the straddling locus), at most what 52 KiB hold (3 workgroups per CU)
        const uint32_t want = 2 * (64 / std::max(tasks_per_locus, 1u) + 2);
        const uint32_t fit = (uint32_t)((52 * 1024 - lane_bytes) / tstride) & ~1u;
        tables = std::min(want, fit);
    } else {
        const size_t budget = (tasks_per_locus < 192 ? 78 : 160 / VTX_WPE) * 1024 - 64;   // lane arrays + haplotype tables (+ 128 B static)
        tables = (uint32_t)((budget - std::min(budget, lane_bytes)) / tstride) & ~1u;
    }
    if (tables < 2) tables = 2;
    if (tables > 32) tables = 32;
    const size_t shmem = lane_bytes + (global_tables ? 0 : (size_t)tables * tstride);
    if (shmem > 160 * 1024 - 256) return hipErrorInvalidValue;
    if (task_list && !global_tables) return hipErrorInvalidValue;    // (list mode reads the tables the first pass built)
    if (global_tables && !task_list)
        launch_band_tables(loci, gt_l0, n_loci, hap_arena, max_hap, min_hap, tstride, n_heads, gtables, false, s);
    const uint32_t ablate = (uint32_t)(VTX_DEV_ENV("VTX_BAND_ABLATE") ? atoi(VTX_DEV_ENV("VTX_BAND_ABLATE")) : 0);
    const uint32_t xcd_claim = ((tasks_per_locus >= 24 || VTX_DEV_ENV("VTX_BAND_XCD")) && !VTX_DEV_ENV("VTX_BAND_NO_XCD")) ? 1u : 0u;   // (measured: config 3 -3 %, 64 / 32 reads per locus -4.5 %, 16: -1 %, 4: +2 %)
#define LAUNCH_RUN(NTV, GTV, WV, PV)                                                                                 \
    {                                                                                                                \
        if (shmem > 48 * 1024) {                                                                                     \
            hipError_t e = hipFuncSetAttribute((const void*)band_run_kernel<NTV, GTV, WV, PV>,                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);              \
            if (e != hipSuccess) return e;                                                                           \
        }                                                                                                            \
        hipLaunchKernelGGL((band_run_kernel<NTV, GTV, WV, PV>),                                                      \
                           dim3(std::min((n_tasks + NTV - 1) / NTV, vtxk_band_run_grid(NTV, WV))),                    \
                           dim3(NTV), shmem, s, n_tasks,                                                             \
                           task_base, records, rec_locus, loci, read_arena, hap_arena, max_hap, min_hap, tables,     \
                           (uint32_t)tstride, ref_score, alt_score, logbuf, band, band_stride, hard_list,            \
                           overflow_list, pending_list, pend_buf, hard_cap, pend_cap, counters, ablate, n_heads,     \
                           (const uint8_t*)gtables, gt_l0, task_list ? 0u : xcd_claim, task_list);                   \
    }
    if (variant == 1) LAUNCH_RUN(64, true, 4, VTX_PS)
    else if (variant == 2) LAUNCH_RUN(64, true, 5, VTX_PS)
    else if (variant == 3) LAUNCH_RUN(64, true, 6, 12)
    else if (wave_wg) LAUNCH_RUN(64, false, VTX_WPE, VTX_PS) else LAUNCH_RUN(256, false, VTX_WPE, VTX_PS)
#undef LAUNCH_RUN
    return hipGetLastError();
}

// Tables of loci [gt_l0, gt_l0 + n_loci) into gtables (the same layout and bucket count vtxk_launch_band_run picks for this
// shape of data), then band_diag_kernel over tasks [task_base, task_base + n_tasks).  Returns hipErrorInvalidValue when the
// tables do not fit the buffer (the caller then runs band_run_kernel alone, which falls back to tables in LDS).
extern "C" hipError_t vtxk_launch_band_diag(uint32_t n_tasks, uint32_t task_base, const vtx_record* records,
                                            const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                            const uint8_t* hap_arena, uint32_t max_hap, uint32_t min_hap, int32_t* ref_score, int32_t* alt_score,
                                            uint32_t* fail_list, uint32_t* refine_rec, uint32_t refine_cap, uint32_t* counters,
                                            uint32_t tasks_per_locus, uint32_t gt_l0, uint32_t n_loci, uint8_t* gtables,
                                            size_t gtables_bytes, int stats, uint32_t* tight_list, uint32_t* tight_pack, uint8_t* stage,
                                            uint32_t* dense_list, uint32_t dense_mask, uint32_t max_read, hipStream_t s) {
    // max_read: the longest read of the batch (the fast kernels' records) — up to 192 bases the kernel is built with three mask words
    if (!n_tasks) return hipSuccess;
    const uint32_t n_heads = pick_heads(tasks_per_locus, true);
    const size_t tstride = band_table_stride(max_hap, n_heads);
    if (!gtables || (size_t)n_loci * 2 * tstride > gtables_bytes) return hipErrorInvalidValue;
    const bool use_t3 = band_use_t3(tasks_per_locus);
    launch_band_tables(loci, gt_l0, n_loci, hap_arena, max_hap, min_hap, tstride, n_heads, gtables, use_t3, s);
    const uint32_t n_blocks = (n_tasks + 255) / 256;
    const uint32_t st = (uint32_t)stats | (VTX_DEV_ENV("VTX_DIAG_ABLATE") ? (uint32_t)atoi(VTX_DEV_ENV("VTX_DIAG_ABLATE")) << 8 : 0u) |
                        (VTX_DEV_ENV("VTX_BAND_NO_CORRIDOR") ? 0x10000u : 0u) |         // (A/B hook: round 5's records for band_refine_kernel)
                        (VTX_DEV_ENV("VTX_DIAG_NO_TWINS") ? 0x20000u : 0u) |            // (A/B hook: every row that is not intact and unique is probed, the twin lists unused)
                        (VTX_DEV_ENV("VTX_DIAG_PHASES") ? 0x80000u : 0u) |
                        (use_t3 ? 0u : 0x40000u);                                       // (a queue entry and a presence-bitmap word per row instead of t3[] and blocks of three rows)
    // two-byte match entries (40 per task) whenever a haplotype position fits a byte; VTX_DIAG_WIDE=1 forces the four-byte variant (tests)
    static const bool force_wide = VTX_DEV_ENV("VTX_DIAG_WIDE") != nullptr;
    static const bool force_four = VTX_DEV_ENV("VTX_DIAG_FOUR_WORDS") != nullptr;      // (tests: the four-word build on short reads)
#define LAUNCH_DIAG(STV, AV)                                                                                                             \
    hipLaunchKernelGGL((band_diag_kernel<4, STV, AV>), dim3(((n_blocks + 7) / 8) * 8), dim3(256), 0, s, n_tasks, task_base, n_blocks, records, \
                       rec_locus, loci, read_arena, max_hap, (uint32_t)tstride, n_heads, (const uint8_t*)gtables, gt_l0, ref_score,        \
                       alt_score, fail_list, refine_rec, refine_cap, counters, st, tight_list, tight_pack, stage, dense_list, dense_mask, min_hap)
    const bool three = max_read <= 192 && !force_four && vtxf::NW >= 3;
    if (max_hap <= 255 && !force_wide) { if (three) LAUNCH_DIAG(uint16_t, 3); else LAUNCH_DIAG(uint16_t, vtxf::NW); }
    else { if (three) LAUNCH_DIAG(uint32_t, 3); else LAUNCH_DIAG(uint32_t, vtxf::NW); }
#undef LAUNCH_DIAG
    return hipGetLastError();
}

// band_refine_kernel over n_recs records of REFINE_WORDS words: the tables are the ones vtxk_launch_band_diag built
extern "C" hipError_t vtxk_launch_band_refine(const uint32_t* recs, uint32_t n_recs, const vtx_record* records,
                                              const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                              uint32_t max_hap, int32_t* ref_score, int32_t* alt_score, uint32_t* fail_list,
                                              uint32_t* counters, uint32_t tasks_per_locus, uint32_t gt_l0, const uint8_t* gtables,
                                              int stats, uint32_t* tight_list, uint32_t* tight_pack, uint8_t* stage, const uint32_t* n_dev,
                                              hipStream_t s) {
    if (!n_recs) return hipSuccess;
    const uint32_t n_heads = pick_heads(tasks_per_locus, true);
    const size_t tstride = band_table_stride(max_hap, n_heads);
    hipLaunchKernelGGL(band_refine_kernel, dim3((n_recs + 255) / 256), dim3(256), 0, s, recs, n_recs, records, rec_locus, loci, read_arena,
                       max_hap, (uint32_t)tstride, n_heads, gtables, gt_l0, ref_score, alt_score, fail_list, counters, (uint32_t)stats,
                       tight_list, tight_pack, stage, n_dev);
    return hipGetLastError();
}

// ---- the SECOND stage (round 5): band_diag2_kernel ------------------------------------------------------------------------------
// What band_diag_kernel leaves because a task's off-diagonal matches do not fit its 40-entry list (W_MATCHES: loci in repeat-rich
// sequence; 14 % of the real-sequence workload) used to take band_sweep_kernel + the masked DP, 38 ns per task.  Most of these tasks
// still have their alignment on ONE diagonal: with a list of 64 entries and the harmless test bounding a match's dp from the matches
// that can really precede it (vtx_fast_core.h: back_harmless, LN::TIGHT) the same per-task logic decides 21 % of them outright and
// proves for another 62 % (band_stream_kernel included) that every off-diagonal match is harmless — the reference's chain lies on the main diagonal, the band is
// one diagonal stretch (band_pack), and the masked DP needs no sweep.  One lane per task, the plain per-lane form of the logic
// (vtxf::fast_task2: its own probes, no pooling): 7 M tasks, not 49 M.  Per lane 62 words of LDS (32 list, 8 pieces, 16 bound bytes,
// 6 generic pieces): 15.9 KB per wavefront.
// Output: T2_SCORE -> the score (stage 1); T2_TIGHT -> tight_list / tight_pack at counters[1] (provisional score = the certificate);
// T2_SWEEP -> sweep_list at counters[0]; T2_STREAM (more than 64 matches) -> stream_list / stream_diag (the diagonal) at *stream_cnt: band_stream_kernel.
constexpr int D2_LANE_WORDS = vtxf::S2_WORDS + vtxf::RM + vtxf::LaneS2::SMAX / 4 + vtxf::GM;
__global__ __launch_bounds__(64) void band_diag2_kernel(
    const uint32_t* __restrict__ tasks, uint32_t n_tasks,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, uint32_t max_hap, uint32_t table_stride, uint32_t n_heads,
    const uint8_t* __restrict__ gtables, uint32_t gt_l0, int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score,
    uint32_t* __restrict__ sweep_list, uint32_t* __restrict__ tight_list, uint32_t* __restrict__ tight_pack,
    uint32_t* __restrict__ stream_list, uint32_t* __restrict__ stream_diag, uint32_t* __restrict__ stream_cnt,
    uint32_t* __restrict__ counters, uint8_t* __restrict__ stage) {
    __shared__ uint32_t mem[D2_LANE_WORDS * 64];
    const int tid = (int)threadIdx.x;
    const uint32_t slot = blockIdx.x * 64u + (uint32_t)tid;
    uint32_t verdict = 0xffu, task = 0, pack = 0;          // (a lane without a task: no list)
    if (slot < n_tasks) {
        task = tasks[slot];
        const uint32_t rid = task >> 1, hap = task & 1u;
        const vtx_record rec = records[rid];
        const uint32_t my_locus = rec_locus[rid];
        const vtx_locus loc = loci[my_locus];
        const int m = (int)rec.read_len, n = (int)(hap ? loc.alt_len : loc.ref_len);
        vtxf::Tab tb;
        tb.gt = gtables; tb.hmask = n_heads - 1;
        tb.ent = (uint32_t)(((size_t)(my_locus - gt_l0) * 2 + hap) * table_stride);
        tb.head = tb.ent + max_hap * 8u;
        tb.bytes = tb.ent + vtxf::tab_bytes_off(max_hap, n_heads);
        tb.uq = tb.ent + vtxf::tab_uq_off(max_hap, n_heads);
        tb.pb = tb.ent + vtxf::tab_pb_off(max_hap, n_heads);
        // lane scratch, the 64 lanes interleaved: list (two-byte entries), pieces, bound bytes, generic pieces
        uint32_t* base = mem + tid;
        const vtxf::LaneS2 ln{base + vtxf::S2_WORDS * 64, 64, (uint16_t*)mem + tid, 64,
                              (uint8_t*)(mem + (vtxf::S2_WORDS + vtxf::RM) * 64) + tid, 64};
        const vtxf::Lane gl{base + (vtxf::S2_WORDS + vtxf::RM + vtxf::LaneS2::SMAX / 4) * 64, 64};
        vtxf::Result2 r = vtxf::fast_task2_list(read_arena + rec.read_off, m, tb, n, ln, gl);
        if (r.verdict == vtxf::T2_STREAM && !stream_list) r.verdict = vtxf::T2_SWEEP;
        verdict = r.verdict; pack = r.pack;
        if (verdict <= vtxf::T2_TIGHT) (hap ? alt_score : ref_score)[rid] = r.score;     // final (T2_SCORE) or provisional: the certificate
        if (verdict == vtxf::T2_SCORE && stage) stage[task] = 1;
    }
    const uint64_t sm = __ballot(verdict == vtxf::T2_SWEEP), tm = __ballot(verdict == vtxf::T2_TIGHT), rm = __ballot(verdict == vtxf::T2_STREAM);
    uint32_t sbase = 0, tbase = 0, rbase = 0;
    if (tid == 0) {
        if (sm) sbase = atomicAdd(&counters[0], (uint32_t)__popcll(sm));
        if (tm) tbase = atomicAdd(&counters[1], (uint32_t)__popcll(tm));
        if (rm) rbase = atomicAdd(stream_cnt, (uint32_t)__popcll(rm));
    }
    sbase = (uint32_t)__shfl((int)sbase, 0); tbase = (uint32_t)__shfl((int)tbase, 0); rbase = (uint32_t)__shfl((int)rbase, 0);
    const uint64_t below = (1ull << tid) - 1ull;
    if (verdict == vtxf::T2_SWEEP) sweep_list[sbase + (uint32_t)__popcll(sm & below)] = task;
    else if (verdict == vtxf::T2_TIGHT) { const uint32_t pos = tbase + (uint32_t)__popcll(tm & below); tight_list[pos] = task; tight_pack[pos] = pack; }
    else if (verdict == vtxf::T2_STREAM) { const uint32_t pos = rbase + (uint32_t)__popcll(rm & below); stream_list[pos] = task; stream_diag[pos] = pack; }
}

// band_stream_kernel: the tasks band_diag2_kernel's list could not hold (heavy repeats: hundreds of off-diagonal matches).  Whether
// every one of them is harmless does not need the list: vtxf::probe_harmless_stream keeps the matches of two rows and a running
// maximum.  One lane per task; per lane 40 words of LDS (64 window entries, 8 pieces): 10 KB per wavefront, so that the dependent
// loads of the bucket walks overlap across many wavefronts.  T2_TIGHT -> tight_list / tight_pack at counters[1] (after
// band_diag2_kernel's entries), T2_SWEEP -> sweep_list at counters[0].  *n_dev tasks (a device count: no host round trip between the two).
constexpr int ST_LANE_WORDS = vtxf::WIN_WORDS + vtxf::RM;
#ifndef VTX_STREAM_WAVES
#define VTX_STREAM_WAVES 4       // four wavefronts per SIMD (108 VGPRs; the LDS allows as many): what overlaps the bucket walks
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(VTX_STREAM_WAVES, VTX_STREAM_WAVES))) void band_stream_kernel(
    const uint32_t* __restrict__ tasks, const uint32_t* __restrict__ task_diag, const uint32_t* __restrict__ n_dev,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, uint32_t max_hap, uint32_t table_stride, uint32_t n_heads,
    const uint8_t* __restrict__ gtables, uint32_t gt_l0, int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score,
    uint32_t* __restrict__ sweep_list, uint32_t* __restrict__ tight_list, uint32_t* __restrict__ tight_pack,
    uint32_t* __restrict__ counters) {
    __shared__ uint32_t mem[ST_LANE_WORDS * 64];
    const int tid = (int)threadIdx.x;
    const uint32_t slot = blockIdx.x * 64u + (uint32_t)tid;
    const uint32_t n_tasks = *n_dev;
    if (blockIdx.x * 64u >= n_tasks) return;
    uint32_t verdict = 0xffu, task = 0, pack = 0;          // (a lane without a task: no list)
    if (slot < n_tasks) {
        task = tasks[slot];
        const uint32_t rid = task >> 1, hap = task & 1u;
        const vtx_record rec = records[rid];
        const uint32_t my_locus = rec_locus[rid];
        const vtx_locus loc = loci[my_locus];
        const int m = (int)rec.read_len, n = (int)(hap ? loc.alt_len : loc.ref_len);
        vtxf::Tab tb;
        tb.gt = gtables; tb.hmask = n_heads - 1;
        tb.ent = (uint32_t)(((size_t)(my_locus - gt_l0) * 2 + hap) * table_stride);
        tb.head = tb.ent + max_hap * 8u;
        tb.bytes = tb.ent + vtxf::tab_bytes_off(max_hap, n_heads);
        tb.uq = tb.ent + vtxf::tab_uq_off(max_hap, n_heads);
        tb.pb = tb.ent + vtxf::tab_pb_off(max_hap, n_heads);
        const vtxf::LaneW wl{mem + vtxf::WIN_WORDS * 64 + tid, 64, (uint16_t*)mem + tid, 64, nullptr, 0};      // window entries, pieces
        const vtxf::Result2 r = vtxf::fast_task2_stream(read_arena + rec.read_off, m, tb, n, wl, (int)task_diag[slot]);
        verdict = r.verdict; pack = r.pack;
        if (verdict == vtxf::T2_TIGHT) (hap ? alt_score : ref_score)[rid] = r.score;     // provisional: the certificate
    }
    const uint64_t sm = __ballot(verdict == vtxf::T2_SWEEP), tm = __ballot(verdict == vtxf::T2_TIGHT);
    uint32_t sbase = 0, tbase = 0;
    if (tid == 0) {
        if (sm) sbase = atomicAdd(&counters[0], (uint32_t)__popcll(sm));
        if (tm) tbase = atomicAdd(&counters[1], (uint32_t)__popcll(tm));
    }
    sbase = (uint32_t)__shfl((int)sbase, 0); tbase = (uint32_t)__shfl((int)tbase, 0);
    const uint64_t below = (1ull << tid) - 1ull;
    if (verdict == vtxf::T2_SWEEP) sweep_list[sbase + (uint32_t)__popcll(sm & below)] = task;
    else if (verdict == vtxf::T2_TIGHT) { const uint32_t pos = tbase + (uint32_t)__popcll(tm & below); tight_list[pos] = task; tight_pack[pos] = pack; }
}

// band_diag2_kernel (+ band_stream_kernel when stream_list != nullptr) over a task list (the tables are the ones vtxk_launch_band_diag
// built for the chunk); counters[0] / [1]: sweep / tight; *stream_cnt: the tasks that went through band_stream_kernel
extern "C" hipError_t vtxk_launch_band_diag2(const uint32_t* tasks, uint32_t n_tasks, const vtx_record* records, const uint32_t* rec_locus,
                                             const vtx_locus* loci, const uint8_t* read_arena, uint32_t max_hap, int32_t* ref_score,
                                             int32_t* alt_score, uint32_t tasks_per_locus, uint32_t gt_l0, const uint8_t* gtables,
                                             uint32_t* sweep_list, uint32_t* tight_list, uint32_t* tight_pack, uint32_t* counters,
                                             uint32_t* stream_list, uint32_t* stream_diag, uint32_t* stream_cnt, uint8_t* stage,
                                             hipStream_t s) {
    if (!n_tasks) return hipSuccess;
    if (max_hap > 255) return hipErrorInvalidValue;                  // (two-byte list entries)
    const uint32_t n_heads = pick_heads(tasks_per_locus, true);
    const size_t tstride = band_table_stride(max_hap, n_heads);
    hipLaunchKernelGGL(band_diag2_kernel, dim3((n_tasks + 63) / 64), dim3(64), 0, s, tasks, n_tasks, records, rec_locus, loci, read_arena,
                       max_hap, (uint32_t)tstride, n_heads, gtables, gt_l0, ref_score, alt_score, sweep_list, tight_list, tight_pack,
                       stream_list, stream_diag, stream_cnt, counters, stage);
    if (stream_list) {
        // (*stream_cnt <= n_tasks tasks: workgroups past the count leave at once)
        hipLaunchKernelGGL(band_stream_kernel, dim3((n_tasks + 63) / 64), dim3(64), 0, s, stream_list, stream_diag, stream_cnt, records, rec_locus, loci,
                           read_arena, max_hap, (uint32_t)tstride, n_heads, gtables, gt_l0, ref_score, alt_score, sweep_list, tight_list,
                           tight_pack, counters);
    }
    return hipGetLastError();
}

extern "C" uint32_t vtxk_band_poly_stride(void) { return (2 + 2 * (4 * SG + 6) + 7) & ~7u; }   // u16 per polyline record

extern "C" hipError_t vtxk_launch_band_expand(const uint32_t* hard_list, uint32_t n_hard, const vtx_record* records,
                                              const uint32_t* rec_locus, const vtx_locus* loci, const uint16_t* src,
                                              uint32_t src_stride, uint16_t* band, uint32_t band_stride, hipStream_t s) {
    if (!n_hard) return hipSuccess;
    hipLaunchKernelGGL(band_expand_kernel, dim3((n_hard + 15) / 16), dim3(256), 0, s, hard_list, n_hard, records,
                       rec_locus, loci, src, src_stride, band, band_stride);
    return hipGetLastError();
}
