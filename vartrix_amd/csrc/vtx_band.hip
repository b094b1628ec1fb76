// vtx_band.hip — device side of the *banded* aligner flavour
// (bio 0.30.0 banded::Aligner::local as restated in oracle/vtx_oracle.c; reference call site
// src/main.rs:899-901 with K = 6, W = 20, src/main.rs:33-34).
//
//   band_kernel        one lane per (record, haplotype) task: k-mer seeding (chained hash of the
//                      haplotype 6-mers), sdpkpp chaining (max-Fenwick tree over y, events merged in
//                      order, tuple tie-breaks), traceback to the anchor polyline, band ranges per
//                      column in closed form, and the certificate (below).
//   sw_banded_kernel   the systolic packed-i16 DP of vtx_kernels.hip with per-column row ranges:
//                      cells outside the band hold "-inf" (G), 0 (Q, E, F) — see the header there.
//
// Band in closed form.  Every cell the crate adds (set_boundaries' lazy extensions, add_kmer,
// add_entry for continued k-mers, add_gap) lies on ONE monotone, connected staircase from
// (first - d0) to (last_end + d1); each adds the (2w+1)-square around it and ranges only grow by
// min/max.  With rmin[c] / rmax[c] = first / last anchor row in anchored column c in [cA, cB]:
//     lo[j] = max(0, rmin[max(j - w, cA)] - w),  hi[j] = min(rows, rmax[min(j + w, cB)] + w + 1)
// for j in [cA - w, cB + w], empty elsewhere.  tests/ check this against the oracle's literal
// add_entry loops.
//
// Certificate.  banded <= full always (the band only removes paths).  The anchor staircase itself
// is an in-band path, so its local-alignment score (affine gaps along the vertical / horizontal
// pieces) is a lower bound of the banded score.  If it equals the full score (already computed by
// sw_full_kernel) the banded score IS the full score and the task is done; otherwise the task is
// appended to the hard list and sw_banded_kernel computes it exactly.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vtx_device.h"

#define KMER 6
#define BANDW 20
#define HASH_BITS 9
#define HASH_SIZE (1 << HASH_BITS)

struct band_scratch {      // per-task slices of the workspace (all sized by the launch)
    uint16_t* head;        // HASH_SIZE
    uint16_t* next;        // n
    uint32_t* mt;          // M_cap packed (x << 16 | y)
    int32_t* dps;          // M_cap
    int32_t* dpp;          // M_cap
    int32_t* tree_v;       // n + KMER + 4
    int32_t* tree_i;       // n + KMER + 4
    uint16_t* rmin;        // n + 2
    uint16_t* rmax;        // n + 2
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
__device__ int band_task(const uint8_t* x, int m, const uint8_t* y, int n, band_scratch sc, uint32_t m_cap,
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
    const int tn = n + KMER + 2;
    for (int i = 0; i <= tn; ++i) { sc.tree_v[i] = INT32_MIN; sc.tree_i[i] = -1; }
    int32_t best_v = KMER, best_i = 0;
    uint32_t s_ptr = 0, e_ptr = 0;    // next start / end event (both in match order)
    while (e_ptr < M) {
        // next event: start (xs, ys, s+M) vs end (xe+K, ye+K, e); start ids sort after end ids
        bool take_start = false;
        if (s_ptr < M) {
            const uint32_t ms = sc.mt[s_ptr], me = sc.mt[e_ptr];
            const uint32_t sx = ms >> 16, sy = ms & 0xffff, ex = (me >> 16) + KMER, ey = (me & 0xffff) + KMER;
            take_start = (sx < ex) || (sx == ex && sy < ey);   // equal coordinates: end first
        }
        if (take_start) {
            const uint32_t p = s_ptr++;
            const int32_t px = (int32_t)(sc.mt[p] >> 16), py = (int32_t)(sc.mt[p] & 0xffff);
            int32_t dv = KMER, dp = -1;
            int32_t bv = INT32_MIN, bi = -1;
            for (int i = py + 1; i > 0; i -= i & (-i))
                if (ent_gt(sc.tree_v[i], sc.tree_i[i], bv, bi)) { bv = sc.tree_v[i]; bi = sc.tree_i[i]; }
            if (bi >= 0) {
                const int32_t cand = bv - 5 - (px + py) + KMER;      // stored v = dp + (xe + ye); gap_open -5, extend -1
                if (cand > dv || (cand == dv && bi > dp)) { dv = cand; dp = bi; }
            }
            sc.dps[p] = dv; sc.dpp[p] = dp;
        } else {
            const uint32_t p = e_ptr++;
            const int32_t px = (int32_t)(sc.mt[p] >> 16), py = (int32_t)(sc.mt[p] & 0xffff);
            if (px > 0 && py > 0) {
                // continuation of the match one step up the diagonal: binary search (px-1, py-1)
                const uint32_t key = ((uint32_t)(px - 1) << 16) | (uint32_t)(py - 1);
                int32_t a = 0, b = (int32_t)p - 1, c = -1;
                while (a <= b) {
                    const int32_t mid = (a + b) >> 1;
                    const uint32_t v = sc.mt[mid];
                    if (v == key) { c = mid; break; }
                    if (v < key) a = mid + 1; else b = mid - 1;
                }
                if (c >= 0) {
                    const int32_t cand = sc.dps[c] + 1;
                    if (cand > sc.dps[p] || (cand == sc.dps[p] && c > sc.dpp[p])) { sc.dps[p] = cand; sc.dpp[p] = c; }
                }
            }
            const int32_t v = sc.dps[p] + (px + KMER) + (py + KMER);
            for (int i = py + KMER + 1; i <= tn; i += i & (-i))
                if (ent_gt(v, (int32_t)p, sc.tree_v[i], sc.tree_i[i])) { sc.tree_v[i] = v; sc.tree_i[i] = (int32_t)p; }
            if (ent_gt(sc.dps[p], (int32_t)p, best_v, best_i)) { best_v = sc.dps[p]; best_i = (int32_t)p; }
        }
    }
    // ---- traceback: reverse the prev links in place so the chain can be walked forward ----
    int32_t cur = best_i, nxt = -1;
    while (cur >= 0) { const int32_t pv = sc.dpp[cur]; sc.dpp[cur] = nxt; nxt = cur; cur = pv; }
    const int32_t first = nxt;                       // dpp[] now holds the successor on the chain
    // ---- anchor staircase -> rmin / rmax per anchored column, and the certificate walk ----
    const int fx = (int)(sc.mt[first] >> 16), fy = (int)(sc.mt[first] & 0xffff);
    int d0 = fx < fy ? fx : fy; if (d0 > 2 * KMER) d0 = 2 * KMER;
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
        int steps = KMER;
        if (nx >= 0) {
            const int qx = (int)(sc.mt[nx] >> 16), qy = (int)(sc.mt[nx] & 0xffff);
            if (qx == px + 1 && qy == py + 1) steps = 1;     // next match continues: advance one cell only
        }
        for (int i = 0; i < steps; ++i) STEP_DIAG()
        p = nx;
    }
    int d1 = (m - r) < (n - c) ? (m - r) : (n - c); if (d1 > 2 * KMER) d1 = 2 * KMER;
    for (int i = 0; i < d1; ++i) STEP_DIAG()
#undef STEP_DIAG
#undef STEP_DOWN
#undef STEP_RIGHT
    *cA_out = cA; *cB_out = c;
    *cert_out = w.best;
    return 0;
}

// per-column ranges from the staircase (closed form, see the file header)
__device__ void band_ranges(const band_scratch& sc, int cA, int cB, int m, int n, uint16_t* lo, uint16_t* hi) {
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
// ref_score / alt_score hold the FULL scores on entry (sw_full_kernel); certified tasks keep them
// (banded == full), hard tasks are appended to hard_list with their ranges in band[] and get their
// exact score from sw_banded_kernel.  counters[0] = hard tasks, counters[1] = capacity overflows.
__global__ __launch_bounds__(64) void band_kernel(
    const uint32_t* __restrict__ tasks, uint32_t n_tasks, uint32_t task_base,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus, const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    uint8_t* __restrict__ workspace, uint64_t ws_stride, uint32_t m_cap, uint32_t max_hap,
    const int32_t* __restrict__ ref_score, const int32_t* __restrict__ alt_score,
    uint16_t* __restrict__ band, uint32_t band_stride, uint32_t* __restrict__ hard_list,
    uint32_t* __restrict__ overflow_list, uint32_t* __restrict__ counters) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_tasks) return;
    const uint32_t task = tasks ? tasks[slot] : task_base + slot;
    const uint32_t rid = task >> 1, hap = task & 1;
    const vtx_record rec = records[rid];
    const vtx_locus loc = loci[rec_locus[rid]];
    const uint8_t* x = read_arena + rec.read_off;
    const uint8_t* y = hap_arena + (hap ? loc.alt_off : loc.ref_off);
    const int m = (int)rec.read_len, n = (int)(hap ? loc.alt_len : loc.ref_len);
    if (m == 0 || n == 0) return;                    // score 0 either way
    uint8_t* ws = workspace + (uint64_t)slot * ws_stride;
    band_scratch sc;
    size_t o = 0;
    sc.head = (uint16_t*)(ws + o); o += HASH_SIZE * 2;
    sc.next = (uint16_t*)(ws + o); o += ((size_t)max_hap + 2) * 2;
    sc.rmin = (uint16_t*)(ws + o); o += ((size_t)max_hap + 2) * 2;
    sc.rmax = (uint16_t*)(ws + o); o += ((size_t)max_hap + 2) * 2;
    o = (o + 15) & ~(size_t)15;
    sc.tree_v = (int32_t*)(ws + o); o += ((size_t)max_hap + KMER + 4) * 4;
    sc.tree_i = (int32_t*)(ws + o); o += ((size_t)max_hap + KMER + 4) * 4;
    sc.mt = (uint32_t*)(ws + o); o += (size_t)m_cap * 4;
    sc.dps = (int32_t*)(ws + o); o += (size_t)m_cap * 4;
    sc.dpp = (int32_t*)(ws + o); o += (size_t)m_cap * 4;
    int32_t cert = 0;
    int cA = 0, cB = 0;
    const int rc = band_task(x, m, y, n, sc, m_cap, &cert, &cA, &cB);
    if (rc) { overflow_list[atomicAdd(&counters[1], 1u)] = task; return; }
    const int32_t full = hap ? alt_score[rid] : ref_score[rid];
    if (cert == INT32_MAX || cert == full) return;   // banded == full
    const uint32_t h = atomicAdd(&counters[0], 1u);
    hard_list[h] = task;
    uint16_t* lo = band + (size_t)h * 2 * band_stride;
    band_ranges(sc, cA, cB, m, n, lo, lo + band_stride);
}

extern "C" size_t vtxk_band_ws_stride(uint32_t m_cap, uint32_t max_hap) {
    size_t o = HASH_SIZE * 2 + 3 * ((size_t)max_hap + 2) * 2;
    o = (o + 15) & ~(size_t)15;
    o += 2 * ((size_t)max_hap + KMER + 4) * 4 + 3 * (size_t)m_cap * 4;
    return (o + 63) & ~(size_t)63;
}

extern "C" hipError_t vtxk_launch_band(const uint32_t* tasks, uint32_t n_tasks, uint32_t task_base,
                                       const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                       const uint8_t* read_arena, const uint8_t* hap_arena, uint8_t* workspace,
                                       uint64_t ws_stride, uint32_t m_cap, uint32_t max_hap, int32_t* ref_score,
                                       int32_t* alt_score, uint16_t* band, uint32_t band_stride, uint32_t* hard_list,
                                       uint32_t* overflow_list, uint32_t* counters, hipStream_t s) {
    if (!n_tasks) return hipSuccess;
    hipLaunchKernelGGL(band_kernel, dim3((n_tasks + 63) / 64), dim3(64), 0, s, tasks, n_tasks, task_base, records,
                       rec_locus, loci, read_arena, hap_arena, workspace, ws_stride, m_cap, max_hap, ref_score, alt_score,
                       band, band_stride, hard_list, overflow_list, counters);
    return hipGetLastError();
}
