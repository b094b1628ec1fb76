// vtx_kernels.hip — hand-written gfx950 (CDNA4) kernels of the VarTrix hot path.
//
// What is computed (reference 10XGenomics/vartrix v1.1.22, src/main.rs):
//   sw_full_kernel      : aligner.local(seq, ref_hap).score and .local(seq, alt_hap).score
//                         for every record (src/main.rs:898-901, :926-927), full-matrix
//                         affine local Smith-Waterman, +1/-5, gap -5-L.
//   group/count/collapse/emit kernels : evaluate_scores (:1019-1030), parse_scores
//                         (:1041-1109), convert_to_counts (:1032-1039) and the three matrix
//                         modes (:1111-1164) as a per-(row, cell) histogram + ordered emit.
//
// Machine mapping (MI355X: 256 CUs x 4 SIMD, wave64, no MFMA — this is integer
// DP, VALU-bound; see DESIGN.md for the roofline):
//   * One 16-lane DPP row = one record.  A wave carries 4 records, a 256-thread
//     workgroup 16.  Lane l of a row owns read rows [l*R, (l+1)*R) in VGPRs
//     (R = rows per lane, template parameter chosen per read-length bucket).
//   * Both haplotypes of the record are aligned at once: every DP quantity is a
//     packed pair of 16-bit lanes {REF haplotype, ALT haplotype} in one VGPR,
//     updated with v_pk_* instructions (2 DP cells per VALU op).
//   * Systolic anti-diagonal wavefront: at step t lane l processes haplotype
//     column t-l for all its R rows; the bottom-row (H, F) pair moves to lane
//     l+1 with one DPP row_shr:1 each per step.  No LDS traffic for DP state.
//   * Haplotype columns are staged once per record into LDS as packed
//     {ref byte, alt byte} words with never-matching sentinels before column 0
//     and after the last column; each lane reads its own column with one
//     ds_read_b32 per step (consecutive lanes -> consecutive banks).
//   * Arithmetic: H,E,F >= 0 kept with unsigned saturating subtracts
//     (v_pk_sub_u16 clamp), the diagonal term as a signed v_pk_mad_i16 — 12
//     packed VALU ops per cell pair.  Scores <= min(read, hap) length < 2^15.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vtx_device.h"

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

#define AS_S(x) __builtin_bit_cast(v2s, (x))
#define AS_U(x) __builtin_bit_cast(v2u, (x))
#define AS_I(x) __builtin_bit_cast(uint32_t, (x))

__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) {
    return AS_I(__builtin_elementwise_sub_sat(AS_U(a), AS_U(b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return AS_I(__builtin_elementwise_max(AS_S(a), AS_S(b)));
}
// The two ops below are pinned with inline asm: written as C++ the compiler
// canonicalises min(x,1)*-6+g into per-half compare/select/perm chains (6 VALU
// ops instead of 2).  Pure register ops: no memory, no hazards beyond what the
// assembler / hardware interlocks handle for VALU->VALU.
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return AS_I(AS_S(a) + AS_S(b));
}
// a * b + c per 16-bit lane
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#define PK(x) ((uint32_t)(uint16_t)(x) * 0x00010001u)

// DPP controls (GFX9): row_shr:1 shifts within a 16-lane row, wave_shr:1 across the wave.
#define DPP_ROW_SHR1 0x111
#define DPP_WAVE_SHR1 0x138

template <int CTRL>
__device__ __forceinline__ uint32_t lane_shr1(uint32_t old, uint32_t v) {
    // lanes without a source (lane 0 of the row / wave) keep `old`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, 0xf, false);
}

// ---------------------------------------------------------------------------
// Full-matrix Smith-Waterman, R rows per lane, GL lanes per record (16 or 64).
// work[] lists the record ids of this read-length bucket.
// LDS: per record slot (lcols) words of packed haplotype columns.
// ---------------------------------------------------------------------------
template <int R, int GL>
__global__ __launch_bounds__(256) void sw_full_kernel(
    const uint32_t* __restrict__ work, uint32_t n_work,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus,
    const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score, uint32_t lcols) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int GROUPS_PER_BLOCK = 256 / GL;
    constexpr int DPP = (GL == 16) ? DPP_ROW_SHR1 : DPP_WAVE_SHR1;
    constexpr int PRE = GL;   // sentinel words before column 0

    const int tid = threadIdx.x;
    const int grp = tid / GL;            // record slot in the block
    const int l = tid % GL;              // lane within the record
    const uint32_t widx = blockIdx.x * GROUPS_PER_BLOCK + grp;
    const bool active = widx < n_work;

    uint32_t rid = 0, m = 0, nr = 0, na = 0, roff = 0, ref_off = 0, alt_off = 0;
    if (active) {
        rid = work[widx];
        const vtx_record rec = records[rid];
        const vtx_locus loc = loci[rec_locus[rid]];
        m = rec.read_len; roff = rec.read_off;
        nr = loc.ref_len; na = loc.alt_len; ref_off = loc.ref_off; alt_off = loc.alt_off;
    }
    const uint32_t n = nr > na ? nr : na;
    // wave-uniform step count: every record slot of this wave runs the same loop
    uint32_t nwave = n;
    if (GL == 16) {
        nwave = max(nwave, (uint32_t)__shfl_xor((int)nwave, 16));
        nwave = max(nwave, (uint32_t)__shfl_xor((int)nwave, 32));
    }
    // scalar (SGPR) loop bound: uniform by construction
    const uint32_t steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nwave + GL - 1));

    // ---- stage packed haplotype columns into LDS (sentinels never match) ----
    uint32_t* cols = smem + (size_t)grp * lcols;
    const uint32_t HAP_PAD = 0x0200u, READ_PAD = 0x0100u;
    for (uint32_t idx = l; idx < PRE + steps + 1; idx += GL) {   // +1: prefetch of the last step
        const int j = (int)idx - PRE;
        uint32_t rc = HAP_PAD, ac = HAP_PAD;
        if (j >= 0) {
            if ((uint32_t)j < nr) rc = hap_arena[ref_off + j];
            if ((uint32_t)j < na) ac = hap_arena[alt_off + j];
        }
        cols[idx] = rc | (ac << 16);
    }

    // ---- this lane's read rows, replicated into both halves ----
    uint32_t c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)(l * R + r);
        const uint32_t ch = (i < m) ? (uint32_t)read_arena[roff + i] : READ_PAD;
        c[r] = ch * 0x00010001u;
    }
    __syncthreads();

    // DP state per row: G = H+1, Q = max(H-6, 0), E (all of column j-1).  Two copies
    // (ping-pong) so the 2x-unrolled step loop needs no register-rotation moves.
    uint32_t Ga[R], Qa[R], Ea[R], Gb[R], Qb[R], Eb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { Ga[r] = PK(1); Qa[r] = 0; Ea[r] = 0; }
    uint32_t best = 0;
    // Values received from the lane above.  Lane 0 of the record never receives
    // (row_shr / wave_shr leave lanes without a source untouched), so these
    // registers keep the matrix boundary there: H = 0 (G = 1, Q = 0), F = 0.
    uint32_t gu_a = PK(1), gu_b = PK(1), qu = 0, fu = 0;

    const uint32_t* colp = cols + PRE - l;  // column of step t is colp[t]

    // One systolic step: column `hp`; reads state S (column j-1), writes state D (column j).
    // gprev = G of the row above at column j-1 (diagonal of the lane's first row).
#define SW_STEP(GS, QS, ES, GD, QD, ED, gprev, gcur, hp)                                        \
    {                                                                                           \
        gcur = lane_shr1<DPP>(gcur, GD##_last);                                                 \
        qu = lane_shr1<DPP>(qu, q_last);                                                        \
        fu = lane_shr1<DPP>(fu, f_last);                                                        \
        uint32_t gd = gprev, qa = qu, fa = fu;                                                  \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                         \
            const uint32_t ne = pk_min_u(c[r] ^ hp, one);          /* 0 match, 1 mismatch */    \
            const uint32_t tt = pk_mad(ne, neg6, gd);              /* diag + (1 | -5) */        \
            gd = GS[r];                                                                         \
            const uint32_t e = pk_max(pk_sub_sat(ES[r], PK(1)), QS[r]);   /* max(E-1, H-6) */    \
            const uint32_t f = pk_max(pk_sub_sat(fa, PK(1)), qa);                               \
            const uint32_t h = pk_max(pk_max(tt, e), f);                                        \
            best = pk_max(best, tt);   /* an optimal local alignment ends on a match */         \
            ED[r] = e;                                                                          \
            GD[r] = pk_add(h, PK(1));                                                           \
            QD[r] = pk_sub_sat(h, PK(6));                                                       \
            fa = f; qa = QD[r];                                                                 \
        }                                                                                       \
        f_last = fa;                                                                            \
    }
    // bottom row of this lane for the column just finished (what the lane below receives)
    uint32_t f_last = 0;
#define Ga_last Gb[R - 1]   /* step A (writes Ga) forwards the bottom row written by step B */
#define Gb_last Ga[R - 1]
#define q_last q_bottom
    uint32_t q_bottom = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) { Gb[r] = PK(1); Qb[r] = 0; Eb[r] = 0; }
    const uint32_t one = PK(1), neg6 = PK(-6);
    const uint32_t steps2 = (steps + 1) >> 1;
    for (uint32_t t2 = 0; t2 < steps2; ++t2) {
        const uint32_t hp0 = colp[2 * t2], hp1 = colp[2 * t2 + 1];
        // step A: state b (column j-1) -> state a (column j); lane above finished column j in its step B'
        q_bottom = Qb[R - 1];
        SW_STEP(Gb, Qb, Eb, Ga, Qa, Ea, gu_b, gu_a, hp0)
        q_bottom = Qa[R - 1];
        SW_STEP(Ga, Qa, Ea, Gb, Qb, Eb, gu_a, gu_b, hp1)
    }
#undef SW_STEP
#undef Ga_last
#undef Gb_last
#undef q_last

    // max over the record's lanes
#pragma unroll
    for (int off = 1; off < GL; off <<= 1) best = pk_max(best, (uint32_t)__shfl_xor((int)best, off));
    if (active && l == 0) {
        ref_score[rid] = (int32_t)(int16_t)(best & 0xffffu);
        alt_score[rid] = (int32_t)(int16_t)(best >> 16);
    }
}

#define INST(R, GL) template __global__ void sw_full_kernel<R, GL>(                                     \
    const uint32_t*, uint32_t, const vtx_record*, const uint32_t*, const vtx_locus*, const uint8_t*,     \
    const uint8_t*, int32_t*, int32_t*, uint32_t);
INST(2, 16) INST(4, 16) INST(6, 16) INST(8, 16) INST(10, 16) INST(12, 16) INST(16, 16) INST(8, 64) INST(16, 64)

extern "C" hipError_t vtxk_launch_sw_full(int R, int GL, uint32_t n_work, const uint32_t* work,
                                          const vtx_record* records, const uint32_t* rec_locus,
                                          const vtx_locus* loci, const uint8_t* read_arena,
                                          const uint8_t* hap_arena, int32_t* ref_score, int32_t* alt_score,
                                          uint32_t max_hap_len, hipStream_t stream) {
    if (n_work == 0) return hipSuccess;
    const uint32_t groups = 256 / GL;
    // slot stride = 16 (mod 32) words: the two 16-lane records of a 32-lane LDS group hit disjoint banks
    const uint32_t lcols = (((GL + max_hap_len + GL + 3) + 31u) & ~31u) + 16u;
    const size_t shmem = (size_t)groups * lcols * sizeof(uint32_t);
    const dim3 grid((n_work + groups - 1) / groups), block(256);
#define CASE(r, gl)                                                                                     \
    if (R == r && GL == gl) {                                                                           \
        if (shmem > 48 * 1024) {                                                                        \
            hipError_t e = hipFuncSetAttribute((const void*)sw_full_kernel<r, gl>,                      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
            if (e != hipSuccess) return e;                                                              \
        }                                                                                               \
        hipLaunchKernelGGL((sw_full_kernel<r, gl>), grid, block, shmem, stream, work, n_work, records,  \
                           rec_locus, loci, read_arena, hap_arena, ref_score, alt_score, lcols);        \
        return hipGetLastError();                                                                       \
    }
    CASE(2, 16) CASE(4, 16) CASE(6, 16) CASE(8, 16) CASE(10, 16) CASE(12, 16) CASE(16, 16) CASE(8, 64) CASE(16, 64)
#undef CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// Grouping (depends only on the records, computed once per submit).
// head_cell[r] = 1 iff record r opens a (row, cell) group (itertools group_by
// over the sorted scores, src/main.rs:1044); head_umi[r] = 1 iff it opens a
// (row, cell, umi) sub-group (the per-cell HashMap keyed by UMI, :1047-1057).
// ---------------------------------------------------------------------------
__global__ void group_heads_kernel(const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus,
                                   uint32_t n, uint32_t* __restrict__ head_cell, uint32_t* __restrict__ head_umi) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint32_t hc = 1, hu = 1;
    if (r > 0 && rec_locus[r - 1] == rec_locus[r] && records[r - 1].cell_index == records[r].cell_index) {
        hc = 0;
        hu = records[r - 1].umi_id != records[r].umi_id;
    }
    head_cell[r] = hc;
    head_umi[r] = hu;
}

// After the inclusive scans: gid = scan - 1.  Head records publish their group's (row, col).
__global__ void group_table_kernel(const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus,
                                   const vtx_locus* __restrict__ loci, uint32_t n,
                                   const uint32_t* __restrict__ head_cell, const uint32_t* __restrict__ head_umi,
                                   const uint32_t* __restrict__ cell_scan, const uint32_t* __restrict__ umi_scan,
                                   uint32_t* __restrict__ grp_row, uint32_t* __restrict__ grp_col,
                                   uint32_t* __restrict__ umi_cellgrp) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t cg = cell_scan[r] - 1;
    if (head_cell[r]) { grp_row[cg] = loci[rec_locus[r]].row; grp_col[cg] = records[r].cell_index; }
    if (head_umi[r]) umi_cellgrp[umi_scan[r] - 1] = cg;
}

// evaluate_scores (src/main.rs:1019-1030) + histogram.  cnt layout: 3 counters
// (ref, alt, unk) per group.  Non-UMI mode counts straight into the cell group
// (:1090-1105); UMI mode counts into the UMI sub-group first (:1048-1057).
__global__ void count_calls_kernel(const int32_t* __restrict__ ref_score, const int32_t* __restrict__ alt_score,
                                   uint32_t n, int32_t min_score, const uint32_t* __restrict__ gscan,
                                   uint32_t* __restrict__ cnt) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int32_t rs = ref_score[r], as = alt_score[r];
    if ((rs < min_score) & (as < min_score)) return;           // None
    const uint32_t which = rs > as ? 0u : (as > rs ? 1u : 2u);  // REF / ALT / UNKNOWN
    atomicAdd(&cnt[3u * (gscan[r] - 1u) + which], 1u);
}

// UMI collapse (src/main.rs:1058-1082): per UMI sub-group with at least one call,
// ALT if alt/total >= 0.75, else REF if ref/total >= 0.75, else UNKNOWN.
// 0.75 is exact in binary and |x/t - 3/4| >= 1/(4t), so 4x >= 3t decides identically.
__global__ void umi_collapse_kernel(const uint32_t* __restrict__ umi_cnt, uint32_t n_umi,
                                    const uint32_t* __restrict__ umi_cellgrp, uint32_t* __restrict__ cell_cnt) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_umi) return;
    const uint32_t r = umi_cnt[3u * u], a = umi_cnt[3u * u + 1], k = umi_cnt[3u * u + 2];
    const uint32_t t = r + a + k;
    if (t == 0) return;   // all reads of this UMI were None: no entry (:1050-1052)
    const uint32_t which = (4u * a >= 3u * t) ? 1u : ((4u * r >= 3u * t) ? 0u : 2u);
    atomicAdd(&cell_cnt[3u * umi_cellgrp[u] + which], 1u);
}

// keep flag per cell group: consensus drops groups with no REF and no ALT call
// (src/main.rs:1120-1126); alt_frac / coverage emit every group (:1140-1142, :1160-1161).
__global__ void keep_flags_kernel(const uint32_t* __restrict__ cell_cnt, uint32_t n_grp, int mode,
                                  uint32_t* __restrict__ keep) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_grp) return;
    keep[g] = (mode != VTX_MODE_CONSENSUS) || (cell_cnt[3u * g] > 0) || (cell_cnt[3u * g + 1] > 0);
}

__global__ void emit_coo_kernel(const uint32_t* __restrict__ cell_cnt, uint32_t n_grp, int mode,
                                const uint32_t* __restrict__ keep, const uint32_t* __restrict__ keep_scan,
                                const uint32_t* __restrict__ grp_row, const uint32_t* __restrict__ grp_col,
                                uint32_t* __restrict__ o_row, uint32_t* __restrict__ o_col,
                                uint32_t* __restrict__ o_alt, uint32_t* __restrict__ o_ref,
                                uint32_t* __restrict__ o_unk, double* __restrict__ o_val,
                                double* __restrict__ o_refval) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_grp || !keep[g]) return;
    const uint32_t o = keep_scan[g] - 1u;
    const uint32_t r = cell_cnt[3u * g], a = cell_cnt[3u * g + 1], k = cell_cnt[3u * g + 2];
    double v, rv = 0.0;
    if (mode == VTX_MODE_CONSENSUS) v = (r > 0 && a > 0) ? 3.0 : (a > 0 ? 2.0 : 1.0);
    else if (mode == VTX_MODE_ALT_FRAC) v = (double)a / ((double)r + (double)a + (double)k);   // NaN for 0/0
    else { v = (double)a; rv = (double)r; }
    o_row[o] = grp_row[g]; o_col[o] = grp_col[g];
    o_alt[o] = a; o_ref[o] = r; o_unk[o] = k;
    o_val[o] = v; o_refval[o] = rv;
}

static inline dim3 grid1d(uint32_t n, uint32_t b) { return dim3((n + b - 1) / b); }

extern "C" hipError_t vtxk_group_heads(const vtx_record* records, const uint32_t* rec_locus, uint32_t n,
                                       uint32_t* head_cell, uint32_t* head_umi, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(group_heads_kernel, grid1d(n, 256), dim3(256), 0, s, records, rec_locus, n, head_cell, head_umi);
    return hipGetLastError();
}
extern "C" hipError_t vtxk_group_table(const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                       uint32_t n, const uint32_t* head_cell, const uint32_t* head_umi,
                                       const uint32_t* cell_scan, const uint32_t* umi_scan, uint32_t* grp_row,
                                       uint32_t* grp_col, uint32_t* umi_cellgrp, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(group_table_kernel, grid1d(n, 256), dim3(256), 0, s, records, rec_locus, loci, n, head_cell,
                       head_umi, cell_scan, umi_scan, grp_row, grp_col, umi_cellgrp);
    return hipGetLastError();
}
extern "C" hipError_t vtxk_count_calls(const int32_t* ref_score, const int32_t* alt_score, uint32_t n,
                                       int32_t min_score, const uint32_t* gscan, uint32_t* cnt, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(count_calls_kernel, grid1d(n, 256), dim3(256), 0, s, ref_score, alt_score, n, min_score, gscan, cnt);
    return hipGetLastError();
}
extern "C" hipError_t vtxk_umi_collapse(const uint32_t* umi_cnt, uint32_t n_umi, const uint32_t* umi_cellgrp,
                                        uint32_t* cell_cnt, hipStream_t s) {
    if (!n_umi) return hipSuccess;
    hipLaunchKernelGGL(umi_collapse_kernel, grid1d(n_umi, 256), dim3(256), 0, s, umi_cnt, n_umi, umi_cellgrp, cell_cnt);
    return hipGetLastError();
}
extern "C" hipError_t vtxk_keep_flags(const uint32_t* cell_cnt, uint32_t n_grp, int mode, uint32_t* keep, hipStream_t s) {
    if (!n_grp) return hipSuccess;
    hipLaunchKernelGGL(keep_flags_kernel, grid1d(n_grp, 256), dim3(256), 0, s, cell_cnt, n_grp, mode, keep);
    return hipGetLastError();
}
extern "C" hipError_t vtxk_emit_coo(const uint32_t* cell_cnt, uint32_t n_grp, int mode, const uint32_t* keep,
                                    const uint32_t* keep_scan, const uint32_t* grp_row, const uint32_t* grp_col,
                                    uint32_t* o_row, uint32_t* o_col, uint32_t* o_alt, uint32_t* o_ref,
                                    uint32_t* o_unk, double* o_val, double* o_refval, hipStream_t s) {
    if (!n_grp) return hipSuccess;
    hipLaunchKernelGGL(emit_coo_kernel, grid1d(n_grp, 256), dim3(256), 0, s, cell_cnt, n_grp, mode, keep, keep_scan,
                       grp_row, grp_col, o_row, o_col, o_alt, o_ref, o_unk, o_val, o_refval);
    return hipGetLastError();
}

// Read bases as the BAM holds them (two per byte, high nibble first; SAM spec 4.2.3: "=ACMGRSVTWYHKDBN") -> one byte per base,
// what rec.seq().as_bytes() gives the reference (src/main.rs:896).  vtx_set_read_format(VTX_READS_NIBBLES): the host ships half
// the bytes and this kernel writes the arena every other kernel reads.  16 input bytes -> 32 output bytes per thread.
__global__ __launch_bounds__(256) void unpack_nibbles_kernel(const uint8_t* __restrict__ in, uint64_t n_in, uint8_t* __restrict__ out) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    if (i >= n_in) return;
    const uint64_t lut_lo = 0x565352474d43413dull, lut_hi = 0x4e42444b48595754ull;      // "=ACMGRSV", "TWYHKDBN" (little endian)
    auto dec = [&](uint32_t nib) -> uint32_t { return (uint32_t)(((nib & 8u) ? lut_hi : lut_lo) >> (8u * (nib & 7u))) & 0xffu; };
    if (i + 16 <= n_in) {
        uint4 v;
        __builtin_memcpy(&v, in + i, 16);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t b0 = (w[k] >> (16 * h)) & 0xffu, b1 = (w[k] >> (16 * h + 8)) & 0xffu;
                o[2 * k + h] = dec(b0 >> 4) | (dec(b0 & 15u) << 8) | (dec(b1 >> 4) << 16) | (dec(b1 & 15u) << 24);
            }
        }
        uint4 a = make_uint4(o[0], o[1], o[2], o[3]), b = make_uint4(o[4], o[5], o[6], o[7]);
        __builtin_memcpy(out + 2 * i, &a, 16);
        __builtin_memcpy(out + 2 * i + 16, &b, 16);
    } else {
        for (uint64_t j = i; j < n_in; ++j) { const uint32_t b0 = in[j]; out[2 * j] = (uint8_t)dec(b0 >> 4); out[2 * j + 1] = (uint8_t)dec(b0 & 15u); }
    }
}
extern "C" hipError_t vtxk_unpack_nibbles(const uint8_t* in, uint64_t n_in, uint8_t* out, hipStream_t s) {
    if (!n_in) return hipSuccess;
    const uint64_t threads = (n_in + 15) / 16;
    hipLaunchKernelGGL(unpack_nibbles_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, in, n_in, out);
    return hipGetLastError();
}

// Matrix values of gathered triplets (vtx_gather_coo): the arithmetic of emit_coo_kernel on the three counts.
__global__ void values_from_counts_kernel(const uint32_t* __restrict__ alt, const uint32_t* __restrict__ ref,
                                          const uint32_t* __restrict__ unk, uint32_t n, int mode,
                                          double* __restrict__ o_val, double* __restrict__ o_refval) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t r = ref[g], a = alt[g], k = unk[g];
    double v, rv = 0.0;
    if (mode == VTX_MODE_CONSENSUS) v = (r > 0 && a > 0) ? 3.0 : (a > 0 ? 2.0 : 1.0);
    else if (mode == VTX_MODE_ALT_FRAC) v = (double)a / ((double)r + (double)a + (double)k);   // NaN for 0/0
    else { v = (double)a; rv = (double)r; }
    o_val[g] = v; o_refval[g] = rv;
}
extern "C" hipError_t vtxk_values_from_counts(const uint32_t* alt, const uint32_t* ref, const uint32_t* unk, uint32_t n, int mode,
                                              double* o_val, double* o_refval, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(values_from_counts_kernel, grid1d(n, 256), dim3(256), 0, s, alt, ref, unk, n, mode, o_val, o_refval);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Band-masked Smith-Waterman for the banded aligner flavour (vtx_band.hip).
// Same systolic packed-i16 scheme, but the two 16-bit halves are two INDEPENDENT
// tasks (task = 2 * record + haplotype) taken pairwise from the hard list, each
// with its own read, haplotype and per-column row ranges [lo, hi) (oracle
// coordinates: DP row ii = read index + 1, column jj = hap index + 1, row / column
// 0 = boundary).  Out-of-band cells hold G = H+1 = "-inf" and Q = E = F = 0, so
// nothing flows through them except the free local restart at 0 that every
// in-band cell has anyway; `best` only sees in-band cells.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return AS_I(AS_S(a) - AS_S(b)); }
__device__ __forceinline__ uint32_t pk_sign(uint32_t a) {   // 0xffff per half iff that half is negative
    const v2s sh = {15, 15};
    const v2s r = AS_S(a) >> sh;
    return AS_I(r);
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

#define NEG_G PK(-16000)

// MODE (round 4).  0: the band is per-column row ranges in `band` (band_sweep_kernel / band_expand_kernel wrote them).
// 2: the band is ONE diagonal stretch widened by the (2w + 1)-squares, one word per listed task (packs[s] = vtxf::band_pack:
//    (d + 256) << 16 | ca << 8 | cb): what band_diag_kernel / band_refine_kernel leave when every off-diagonal match is harmless —
//    the chain, hence the staircase, IS that stretch — but the bounds do not meet; the ranges are expanded here, in LDS.
// 1: the same kernel WITHOUT a band — every cell is in band, the mask arithmetic folds away (12 packed ops per cell pair
//    instead of 19) — used as a CHECK, not as a score (experiment hook VTX_BAND_CHECK=1): the listed tasks carry a provisional
//    score (the certificate, a lower bound of the banded score: cert <= banded <= full); where the full-matrix score EQUALS it
//    the provisional score is the banded score and stays; the other tasks (and their packs) are appended to recheck_list /
//    recheck_pack (recheck_count) for the masked DP.
// n_dev != nullptr: the list length lives on the device (min(*n_dev, n_hard) entries; the grid is sized for n_hard).
template <int R, int GL, int MODE>
__global__ __launch_bounds__(256) void sw_banded_kernel(
    const uint32_t* __restrict__ hard, uint32_t n_hard, const uint32_t* __restrict__ n_dev,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus,
    const vtx_locus* __restrict__ loci, const uint8_t* __restrict__ read_arena,
    const uint8_t* __restrict__ hap_arena, const uint16_t* __restrict__ band, uint32_t band_stride,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score, uint32_t lcols,
    uint32_t* __restrict__ recheck_list, uint32_t* __restrict__ recheck_count, uint8_t* __restrict__ stage,
    const uint32_t* __restrict__ packs, uint32_t* __restrict__ recheck_pack) {
    constexpr bool FULL = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    if (n_dev) { const uint32_t nd = *n_dev; n_hard = nd < n_hard ? nd : n_hard; }
    const int GROUPS_PER_BLOCK = (int)blockDim.x / GL;     // 256 threads, or fewer when three LDS arrays per record slot of a wide haplotype do not fit
    constexpr int DPP = (GL == 16) ? DPP_ROW_SHR1 : DPP_WAVE_SHR1;
    // One extra leading column: hap index j = -1 is the oracle's boundary column 0, processed like any
    // other column (never-matching sentinel base, ranges lo[0], hi[0]), so the boundary state is
    // produced by the same masked step.  Lane l processes hap index t - l - 1 at step t.
    constexpr int PRE = GL + 1;
    const int tid = threadIdx.x;
    const int grp = tid / GL;
    const int l = tid % GL;
    const uint32_t pair = blockIdx.x * GROUPS_PER_BLOCK + grp;
    if (2u * (blockIdx.x * (uint32_t)GROUPS_PER_BLOCK) >= n_hard) return;      // (whole workgroup: uniform)

    // the two tasks of this record slot
    uint32_t task[2], m[2] = {0, 0}, n[2] = {0, 0}, roff[2] = {0, 0}, hoff[2] = {0, 0}, pk[2] = {0, 0};
    bool act[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t s = 2 * pair + k;
        act[k] = s < n_hard;
        task[k] = 0;
        if (act[k]) {
            task[k] = hard[s];
            if (MODE != 0 && packs) pk[k] = packs[s];
            const uint32_t rid = task[k] >> 1;
            const vtx_record rec = records[rid];
            const vtx_locus loc = loci[rec_locus[rid]];
            m[k] = rec.read_len; roff[k] = rec.read_off;
            n[k] = (task[k] & 1) ? loc.alt_len : loc.ref_len;
            hoff[k] = (task[k] & 1) ? loc.alt_off : loc.ref_off;
        }
    }
    uint32_t nwave = n[0] > n[1] ? n[0] : n[1];
    if (GL == 16) {
        nwave = max(nwave, (uint32_t)__shfl_xor((int)nwave, 16));
        nwave = max(nwave, (uint32_t)__shfl_xor((int)nwave, 32));
    }
    const uint32_t steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nwave + GL));

    // LDS per slot: columns, lo pairs, hi pairs (index PRE + j <-> hap index j, oracle column j + 1)
    uint32_t* cols = smem + (size_t)grp * 3 * lcols;
    uint32_t* los = cols + lcols;
    uint32_t* his = los + lcols;
    const uint32_t HAP_PAD = 0x0200u, READ_PAD = 0x0100u;
    const uint16_t* bandA = band + (size_t)(2 * pair) * 2 * band_stride;        // lo[0..n], hi at + band_stride
    const uint16_t* bandB = band + (size_t)(2 * pair + 1) * 2 * band_stride;
    for (uint32_t idx = l; idx < PRE + steps + 2; idx += GL) {
        const int j = (int)idx - PRE;
        uint32_t ca = HAP_PAD, cb = HAP_PAD, la = 0x7fff, lb = 0x7fff, ha = 0, hb = 0;
        if (j >= -1) {
            if (act[0] && j < (int)n[0]) { if (j >= 0) ca = hap_arena[hoff[0] + j]; if (MODE == 0) { la = bandA[j + 1]; ha = bandA[band_stride + j + 1]; } }
            if (act[1] && j < (int)n[1]) { if (j >= 0) cb = hap_arena[hoff[1] + j]; if (MODE == 0) { lb = bandB[j + 1]; hb = bandB[band_stride + j + 1]; } }
            if (MODE == 2) {
                // column jj = j + 1 of the DP matrix; anchors (c - d, c), c = cA .. cB; the (2w + 1)-squares (W = 20, src/main.rs:34)
                const int jj = j + 1;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (!act[k] || j >= (int)n[k]) continue;
                    const int d = (int)(pk[k] >> 16) - 256, cA = (int)((pk[k] >> 8) & 0xffu), cB = (int)(pk[k] & 0xffu);
                    uint32_t lv = 0x7fffu, hv = 0;
                    if (jj >= cA - 20 && jj <= cB + 20) {
                        const int c0 = jj - 20 > cA ? jj - 20 : cA, c1 = jj + 20 < cB ? jj + 20 : cB;
                        const int l_ = c0 - d - 20, h_ = c1 - d + 21;
                        lv = (uint32_t)(l_ > 0 ? l_ : 0);
                        hv = (uint32_t)(h_ < (int)m[k] + 1 ? h_ : (int)m[k] + 1);
                    }
                    if (k == 0) { la = lv; ha = hv; } else { lb = lv; hb = hv; }
                }
            }
        }
        cols[idx] = ca | (cb << 16);
        if (!FULL) { los[idx] = la | (lb << 16); his[idx] = ha | (hb << 16); }
    }
    uint32_t c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)(l * R + r);
        const uint32_t ca = (act[0] && i < m[0]) ? (uint32_t)read_arena[roff[0] + i] : READ_PAD;
        const uint32_t cb = (act[1] && i < m[1]) ? (uint32_t)read_arena[roff[1] + i] : READ_PAD;
        c[r] = ca | (cb << 16);
    }
    __syncthreads();

    const uint32_t one = PK(1), neg6 = PK(-6);
    const uint32_t rowbase = PK(l * R + 1);                 // oracle row of this lane's r = 0
    // boundary row 0 in band:  lo == 0 && hi > 0
#define ROW0_MASK(lo2, hi2) (pk_sign(pk_sub(lo2, one)) & pk_sign(pk_sub(0u, hi2)))

    uint32_t Ga[R], Qa[R], Ea[R], Gb[R], Qb[R], Eb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { Ga[r] = NEG_G; Qa[r] = 0; Ea[r] = 0; Gb[r] = NEG_G; Qb[r] = 0; Eb[r] = 0; }   // "column -1": outside
    uint32_t best = 0;
    uint32_t gu_a = NEG_G, gu_b = NEG_G, qu = 0, fu = 0, f_last = 0, q_bottom = 0;
    const uint32_t* colp = cols + PRE - 1 - l;
    const uint32_t* lop = los + PRE - 1 - l;
    const uint32_t* hip = his + PRE - 1 - l;

#define SWB_STEP(GS, QS, ES, GD, QD, ED, gprev, gcur, gsrc, t)                                     \
    {                                                                                              \
        const uint32_t hp = colp[t], lo2 = FULL ? 0u : lop[t], hi2 = FULL ? PK(0x7fff) : hip[t];   \
        gcur = FULL ? one : bfi(ROW0_MASK(lo2, hi2), one, NEG_G);   /* lane 0: boundary row of this column */ \
        gcur = lane_shr1<DPP>(gcur, gsrc);                                                         \
        qu = lane_shr1<DPP>(qu, q_bottom);                                                         \
        fu = lane_shr1<DPP>(fu, f_last);                                                           \
        const uint32_t am = pk_sub(lo2, rowbase), bm = pk_sub(hi2, rowbase);                       \
        uint32_t gd = gprev, qa = qu, fa = fu;                                                     \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                            \
            const uint32_t mk = FULL ? 0xffffffffu : (pk_sign(pk_sub(am, PK(r + 1))) & pk_sign(pk_sub(PK(r), bm))); \
            const uint32_t ne = pk_min_u(c[r] ^ hp, one);                                          \
            const uint32_t tt = pk_mad(ne, neg6, gd) & mk;                                         \
            gd = GS[r];                                                                            \
            const uint32_t e = pk_max(pk_sub_sat(ES[r], PK(1)), QS[r]) & mk;                       \
            const uint32_t f = pk_max(pk_sub_sat(fa, PK(1)), qa) & mk;                             \
            const uint32_t h = pk_max(pk_max(tt, e), f);                                           \
            best = pk_max(best, tt);                                                               \
            ED[r] = e;                                                                             \
            GD[r] = bfi(mk, pk_add(h, PK(1)), NEG_G);                                              \
            QD[r] = pk_sub_sat(h, PK(6)) & mk;                                                     \
            fa = f; qa = QD[r];                                                                    \
        }                                                                                          \
        f_last = fa;                                                                               \
    }
    const uint32_t steps2 = (steps + 1) >> 1;
    for (uint32_t t2 = 0; t2 < steps2; ++t2) {
        q_bottom = Qb[R - 1];
        SWB_STEP(Gb, Qb, Eb, Ga, Qa, Ea, gu_b, gu_a, Gb[R - 1], 2 * t2)
        q_bottom = Qa[R - 1];
        SWB_STEP(Ga, Qa, Ea, Gb, Qb, Eb, gu_a, gu_b, Ga[R - 1], 2 * t2 + 1)
    }
#undef SWB_STEP
#undef ROW0_MASK

#pragma unroll
    for (int off = 1; off < GL; off <<= 1) best = pk_max(best, (uint32_t)__shfl_xor((int)best, off));
    if (l == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!act[k]) continue;
            const int32_t sc = (int32_t)(int16_t)((best >> (16 * k)) & 0xffffu);
            int32_t* dst = ((task[k] & 1) ? alt_score : ref_score) + (task[k] >> 1);
            if (FULL) {
                if (*dst != sc) { const uint32_t pos = atomicAdd(recheck_count, 1u); recheck_list[pos] = task[k]; if (recheck_pack) recheck_pack[pos] = pk[k]; }
                else if (stage) stage[task[k]] = 3;
            } else {
                *dst = sc;
                if (MODE == 2 && stage) stage[task[k]] = 7;
            }
        }
    }
}

static hipError_t launch_sw_pairs(int mode, int R, int GL, uint32_t n_hard, const uint32_t* hard, const uint32_t* n_dev,
                                  const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                  const uint8_t* read_arena, const uint8_t* hap_arena, const uint16_t* band,
                                  uint32_t band_stride, int32_t* ref_score, int32_t* alt_score, uint32_t max_hap_len,
                                  uint32_t* recheck_list, uint32_t* recheck_count, uint8_t* stage, const uint32_t* packs,
                                  uint32_t* recheck_pack, hipStream_t stream) {
    if (n_hard == 0) return hipSuccess;
    const uint32_t pairs = (n_hard + 1) / 2;
    const uint32_t lcols = ((GL + max_hap_len + GL + 8) + 3u) & ~3u;
    // 256 threads per workgroup unless the three LDS arrays per record slot do not fit (haplotypes above ~800 bases: round 3
    // found the launch failing there — no test had a hard task on a wide window): then 128 or 64
    uint32_t threads = 256;
    while (threads > 64 && (size_t)(threads / GL) * 3 * lcols * sizeof(uint32_t) > 150 * 1024) threads >>= 1;
    if (threads < (uint32_t)GL) threads = (uint32_t)GL;
    const uint32_t groups = threads / GL;
    const size_t shmem = (size_t)groups * 3 * lcols * sizeof(uint32_t);      // (the check variant keeps the layout, it only skips two of the arrays)
    if (shmem > 160 * 1024 - 512) return hipErrorInvalidValue;
    const dim3 grid((pairs + groups - 1) / groups), block(threads);
#define CASE_F(r, gl, f)                                                                                  \
    {                                                                                                     \
        if (shmem > 48 * 1024) {                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)sw_banded_kernel<r, gl, f>,                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);   \
            if (e != hipSuccess) return e;                                                                \
        }                                                                                                 \
        hipLaunchKernelGGL((sw_banded_kernel<r, gl, f>), grid, block, shmem, stream, hard, n_hard, n_dev, records, \
                           rec_locus, loci, read_arena, hap_arena, band, band_stride, ref_score, alt_score, lcols, \
                           recheck_list, recheck_count, stage, packs, recheck_pack);                      \
        return hipGetLastError();                                                                         \
    }
#define CASE(r, gl) if (R == r && GL == gl) { if (mode == 1) CASE_F(r, gl, 1) else if (mode == 2) CASE_F(r, gl, 2) else CASE_F(r, gl, 0) }
    CASE(2, 16) CASE(4, 16) CASE(6, 16) CASE(8, 16) CASE(10, 16) CASE(12, 16) CASE(16, 16) CASE(8, 64) CASE(16, 64)
#undef CASE
#undef CASE_F
    return hipErrorInvalidValue;
}

extern "C" hipError_t vtxk_launch_sw_banded(int R, int GL, uint32_t n_hard, const uint32_t* hard,
                                            const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                            const uint8_t* read_arena, const uint8_t* hap_arena, const uint16_t* band,
                                            uint32_t band_stride, int32_t* ref_score, int32_t* alt_score,
                                            uint32_t max_hap_len, hipStream_t stream) {
    return launch_sw_pairs(0, R, GL, n_hard, hard, nullptr, records, rec_locus, loci, read_arena, hap_arena, band, band_stride,
                           ref_score, alt_score, max_hap_len, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
// the masked DP over a list whose length lives on the device (n_dev; n_cap bounds the grid)
extern "C" hipError_t vtxk_launch_sw_banded_dev(int R, int GL, uint32_t n_cap, const uint32_t* hard, const uint32_t* n_dev,
                                                const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                                const uint8_t* read_arena, const uint8_t* hap_arena, const uint16_t* band,
                                                uint32_t band_stride, int32_t* ref_score, int32_t* alt_score,
                                                uint32_t max_hap_len, hipStream_t stream) {
    return launch_sw_pairs(0, R, GL, n_cap, hard, n_dev, records, rec_locus, loci, read_arena, hap_arena, band, band_stride,
                           ref_score, alt_score, max_hap_len, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
// the masked DP over tasks whose band is one diagonal stretch (packs[i] = vtxf::band_pack of list[i]); stage: VTX_STAGE_DIAG_DP
extern "C" hipError_t vtxk_launch_sw_diag_band(int R, int GL, uint32_t n_cap, const uint32_t* list, const uint32_t* packs,
                                               const uint32_t* n_dev, const vtx_record* records, const uint32_t* rec_locus,
                                               const vtx_locus* loci, const uint8_t* read_arena, const uint8_t* hap_arena,
                                               int32_t* ref_score, int32_t* alt_score, uint32_t max_hap_len, uint8_t* stage,
                                               hipStream_t stream) {
    return launch_sw_pairs(2, R, GL, n_cap, list, n_dev, records, rec_locus, loci, read_arena, hap_arena, nullptr, 0,
                           ref_score, alt_score, max_hap_len, nullptr, nullptr, stage, packs, nullptr, stream);
}
// full-matrix CHECK of provisional scores (see the kernel): tasks whose full score differs go to recheck_list / recheck_pack
extern "C" hipError_t vtxk_launch_sw_check(int R, int GL, uint32_t n_cap, const uint32_t* list, const uint32_t* packs, const uint32_t* n_dev,
                                           const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                           const uint8_t* read_arena, const uint8_t* hap_arena, int32_t* ref_score,
                                           int32_t* alt_score, uint32_t max_hap_len, uint32_t* recheck_list, uint32_t* recheck_pack,
                                           uint32_t* recheck_count, uint8_t* stage, hipStream_t stream) {
    return launch_sw_pairs(1, R, GL, n_cap, list, n_dev, records, rec_locus, loci, read_arena, hap_arena, nullptr, 0,
                           ref_score, alt_score, max_hap_len, recheck_list, recheck_count, stage, packs, recheck_pack, stream);
}

__global__ void fill_i32_kernel(int32_t* __restrict__ p, uint32_t n, int32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
extern "C" hipError_t vtxk_fill_i32(int32_t* p, uint32_t n, int32_t v, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fill_i32_kernel, grid1d(n, 256), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}
// both tasks of every listed RECORD
__global__ void mark_stage_records_kernel(const uint32_t* __restrict__ recs, uint32_t n, uint8_t code, uint8_t* __restrict__ stage) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { stage[2 * (size_t)recs[i]] = code; stage[2 * (size_t)recs[i] + 1] = code; }
}
extern "C" hipError_t vtxk_mark_stage_records(const uint32_t* recs, uint32_t n, uint8_t code, uint8_t* stage, hipStream_t s) {
    if (!n || !stage) return hipSuccess;
    hipLaunchKernelGGL(mark_stage_records_kernel, grid1d(n, 256), dim3(256), 0, s, recs, n, code, stage);
    return hipGetLastError();
}
// stage[list[i]] = code for the first min(n, *n_dev) entries (vtx_fetch_stage: which stage decided a task's score)
__global__ void mark_stage_kernel(const uint32_t* __restrict__ list, uint32_t n, const uint32_t* __restrict__ n_dev, uint8_t code,
                                  uint8_t* __restrict__ stage) {
    if (n_dev) { const uint32_t nd = *n_dev; n = nd < n ? nd : n; }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stage[list[i]] = code;
}
extern "C" hipError_t vtxk_mark_stage(const uint32_t* list, uint32_t n, const uint32_t* n_dev, uint8_t code, uint8_t* stage, hipStream_t s) {
    if (!n || !stage) return hipSuccess;
    hipLaunchKernelGGL(mark_stage_kernel, grid1d(n, 256), dim3(256), 0, s, list, n, n_dev, code, stage);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// sw_full_lut_kernel — sw_full_kernel with the substitution term read from LDS.
//
// Per locus (shared by every record of the workgroup that aligns to it) a table
//   lut[idx * 6 + code] = { ref[j] == base(code) ? +1 : -5 , alt[j] == base(code) ? +1 : -5 }   (i16 pair)
// idx = PRE + j, code 0..4 = A C G T N, code 5 = "row beyond the read" (never matches), sentinel
// columns never match.  Each lane keeps one LDS byte address per row (table base + its column
// offset + the row's base code); the per-step column advance is folded into the ds_read
// immediate offset.  The diagonal term becomes t = H_diag + w (one v_pk_add_i16) and the state is H
// itself: 9 packed VALU ops per cell pair instead of 12 (no xor / min / mad / H+1).
// Reads holding a byte outside ACGTN are appended to `redo` and scored by sw_full_kernel.
// The stride of 6 words per column spreads the 16 lanes of a record over distinct even banks.
// ---------------------------------------------------------------------------
#define LUT_CODES 6

__device__ __forceinline__ uint32_t base_code(uint32_t ch) {
    return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : ch == 'N' ? 4u : 6u;   // 6 = not representable
}

template <int R>
__global__ __launch_bounds__(256) void sw_full_lut_kernel(
    const uint32_t* __restrict__ work, uint32_t n_work,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus,
    const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score, uint32_t lcols, uint32_t loci_cap,
    uint32_t* __restrict__ redo, uint32_t* __restrict__ redo_count) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int GL = 16;
    constexpr int GROUPS_PER_BLOCK = 16;
    constexpr int PRE = GL;
    const int tid = threadIdx.x;
    const int grp = tid / GL;
    const int l = tid % GL;
    const uint32_t widx = blockIdx.x * GROUPS_PER_BLOCK + grp;
    const bool active = widx < n_work;

    uint32_t rid = 0, m = 0, nr = 0, na = 0, roff = 0, my_locus = 0;
    if (active) {
        rid = work[widx];
        const vtx_record rec = records[rid];
        my_locus = rec_locus[rid];
        const vtx_locus loc = loci[my_locus];
        m = rec.read_len; roff = rec.read_off;
        nr = loc.ref_len; na = loc.alt_len;
    }
    // locus range of the workgroup (work lists are in record order => loci ascending)
    const uint32_t w_first = blockIdx.x * GROUPS_PER_BLOCK;
    const uint32_t w_last = min(n_work - 1, w_first + GROUPS_PER_BLOCK - 1);
    const uint32_t l_first = rec_locus[work[w_first]], l_last = rec_locus[work[w_last]];
    const uint32_t n_loc = l_last - l_first + 1;          // <= loci_cap (checked by the host when choosing this kernel)
    if (!active) my_locus = l_first;                      // idle slots read a valid table

    const uint32_t n = nr > na ? nr : na;
    uint32_t nwave = n;
    nwave = max(nwave, (uint32_t)__shfl_xor((int)nwave, 16));
    nwave = max(nwave, (uint32_t)__shfl_xor((int)nwave, 32));
    const uint32_t steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nwave + GL - 1));

    // ---- build the tables of loci l_first .. l_last: lcols columns x 6 codes each ----
    const uint32_t tab_words = lcols * LUT_CODES;
    for (uint32_t t = 0; t < n_loc && t < loci_cap; ++t) {
        const vtx_locus loc = loci[l_first + t];
        uint32_t* tab = smem + (size_t)t * tab_words;
        for (uint32_t idx = tid; idx < lcols; idx += 256) {
            const int j = (int)idx - PRE;
            uint32_t rc = 0x200u, ac = 0x200u;            // sentinel: equals no base
            if (j >= 0) {
                if ((uint32_t)j < loc.ref_len) rc = hap_arena[loc.ref_off + j];
                if ((uint32_t)j < loc.alt_len) ac = hap_arena[loc.alt_off + j];
            }
            const uint32_t bases[5] = {'A', 'C', 'G', 'T', 'N'};
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint32_t wr = (rc == bases[k]) ? 0x0001u : 0xfffbu;
                const uint32_t wa = (ac == bases[k]) ? 0x0001u : 0xfffbu;
                tab[idx * LUT_CODES + k] = wr | (wa << 16);
            }
            tab[idx * LUT_CODES + 5] = 0xfffbfffbu;
        }
    }

    // ---- this lane's rows: LDS byte address of (column of step 0, row's base code) ----
    const uint32_t tab_base = (uint32_t)(my_locus - l_first) * tab_words;
    uint32_t addr[R];
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)(l * R + r);
        uint32_t code = 5;
        if (i < m) { code = base_code(read_arena[roff + i]); if (code > 5) { bad = true; code = 5; } }
        addr[r] = (tab_base + (uint32_t)(PRE - l) * LUT_CODES + code) * 4u;
    }
    __syncthreads();
    // any lane of the record saw a byte outside ACGTN: leave the record to the generic kernel
    uint32_t badm = bad ? 1u : 0u;
#pragma unroll
    for (int off = 1; off < GL; off <<= 1) badm |= (uint32_t)__shfl_xor((int)badm, off);

    uint32_t Ha[R], Hb[R], Q[R], E[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { Ha[r] = 0; Hb[r] = 0; Q[r] = 0; E[r] = 0; }
    uint32_t best = 0;
    uint32_t hu_a = 0, hu_b = 0, qu = 0, fu = 0, f_last = 0, q_bottom = 0;
    const char* lds = (const char*)smem;

    // state S = column j-1 (H only; Q and E are updated in place), D = column j
#define LUT_STEP(HS, HD, hprev, hcur, hsrc, OFF)                                                   \
    {                                                                                              \
        hcur = lane_shr1<DPP_ROW_SHR1>(hcur, hsrc);                                                \
        qu = lane_shr1<DPP_ROW_SHR1>(qu, q_bottom);                                                \
        fu = lane_shr1<DPP_ROW_SHR1>(fu, f_last);                                                  \
        uint32_t hd = hprev, qa = qu, fa = fu;                                                     \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                            \
            const uint32_t w = *(const uint32_t*)(lds + addr[r] + (OFF));                          \
            const uint32_t tt = pk_add(hd, w);                                                     \
            hd = HS[r];                                                                            \
            const uint32_t e = pk_max(pk_sub_sat(E[r], PK(1)), Q[r]);                              \
            const uint32_t f = pk_max(pk_sub_sat(fa, PK(1)), qa);                                  \
            const uint32_t h = pk_max(pk_max(tt, e), f);                                           \
            best = pk_max(best, tt);                                                               \
            E[r] = e;                                                                              \
            HD[r] = h;                                                                             \
            Q[r] = pk_sub_sat(h, PK(6));                                                           \
            fa = f; qa = Q[r];                                                                     \
        }                                                                                          \
        f_last = fa; q_bottom = Q[R - 1];                                                          \
    }
    const uint32_t steps4 = (steps + 3) >> 2;
    for (uint32_t t4 = 0; t4 < steps4; ++t4) {
        LUT_STEP(Hb, Ha, hu_b, hu_a, Hb[R - 1], 0 * LUT_CODES * 4)
        LUT_STEP(Ha, Hb, hu_a, hu_b, Ha[R - 1], 1 * LUT_CODES * 4)
        LUT_STEP(Hb, Ha, hu_b, hu_a, Hb[R - 1], 2 * LUT_CODES * 4)
        LUT_STEP(Ha, Hb, hu_a, hu_b, Ha[R - 1], 3 * LUT_CODES * 4)
#pragma unroll
        for (int r = 0; r < R; ++r) addr[r] += 4 * LUT_CODES * 4;
    }
#undef LUT_STEP

#pragma unroll
    for (int off = 1; off < GL; off <<= 1) best = pk_max(best, (uint32_t)__shfl_xor((int)best, off));
    if (active && l == 0) {
        if (badm) {
            redo[atomicAdd(redo_count, 1u)] = rid;
        } else {
            ref_score[rid] = (int32_t)(int16_t)(best & 0xffffu);
            alt_score[rid] = (int32_t)(int16_t)(best >> 16);
        }
    }
}

// ---------------------------------------------------------------------------
// sw_full_duo_kernel — the LUT kernel with the REF == ALT haplotype prefix computed once for TWO reads.
//
// Before the variant column v the two haplotypes of a locus are the same string, so the {REF, ALT} halves
// of the LUT kernel compute the same numbers there.  Here a 16-lane row takes two records A and B and
// runs ONE systolic pipeline through three phases of whole steps:
//     P : steps [0, T1)               halves = {A vs REF, B vs REF}   (two LUT lookups + one v_perm per cell)
//     A : steps [T1, T1 + S)          halves = {A vs REF, A vs ALT}   (as the LUT kernel)
//     B : steps [T1 + S, T1 + 2 S)    halves = {B vs REF, B vs ALT}
// T1 = (common prefix of the wave's haplotype pairs) rounded down to 4.  Every lane changes phase at the same
// step; lane l is then at column T1 - l, which is still inside the common prefix, where {A vs REF, A vs ALT}
// hold equal numbers — so re-packing {A, B} into {A, A} there is exact.  At step T1 the DPP exchange delivers
// the {A, B} values the lane above computed for exactly that column: their B halves, B's H / E of the previous
// column, its diagonal carry and best are set aside (packed with v_perm, one VGPR per row — the registers that
// held B's LUT addresses), the A halves are duplicated.  At step T1 + S the lane restarts B from that state at
// column T1 - l (addresses rebuilt from B's 3-bit row codes).  S = n - T1 + 15 rounded up to 4, so the last
// column is reached by every lane.  Steps per pair: T1 + 2 S = 332 instead of 2 x 216 = 432 for an SNV at
// padding 100; there are no partially-switched steps, the two re-packings cost ~5 R instructions each per pair.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t perm_b32(uint32_t hi_src, uint32_t lo_src, uint32_t sel) {
    return __builtin_amdgcn_perm(hi_src, lo_src, sel);      // selector bytes 0-3: lo_src, 4-7: hi_src
}
#define SEL_LO_LO 0x01000100u      /* {x.lo, x.lo} of lo_src                     */
#define SEL_HI_HI 0x03020302u      /* {x.hi, x.hi} of lo_src                     */
#define SEL_LOA_LOB 0x05040100u    /* {lo_src.lo, hi_src.lo}                     */
#define SEL_HIA_HIB 0x07060302u    /* {lo_src.hi, hi_src.hi}                     */
#define SEL_IDENT 0x03020100u      /* lo_src                                     */

template <int R>
__global__ __launch_bounds__(256) void sw_full_duo_kernel(
    const uint32_t* __restrict__ work, uint32_t n_work,
    const vtx_record* __restrict__ records, const uint32_t* __restrict__ rec_locus,
    const vtx_locus* __restrict__ loci,
    const uint8_t* __restrict__ read_arena, const uint8_t* __restrict__ hap_arena,
    int32_t* __restrict__ ref_score, int32_t* __restrict__ alt_score, uint32_t lcols, uint32_t loci_cap,
    uint32_t* __restrict__ redo, uint32_t* __restrict__ redo_count, uint32_t pcols) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t s_v[8];                           // common REF/ALT prefix length per table
    constexpr int GL = 16;
    constexpr int PAIRS_PER_BLOCK = 16;
    constexpr int PRE = GL;
    const int tid = threadIdx.x;
    const int grp = tid / GL;
    const int l = tid % GL;
    const uint32_t pair = blockIdx.x * PAIRS_PER_BLOCK + grp;
    const bool active = 2 * pair < n_work;
    const bool has_b0 = 2 * pair + 1 < n_work;

    uint32_t rid_a = 0, rid_b = 0, m_a = 0, m_b = 0, roff_a = 0, roff_b = 0, loc_a = 0, loc_b = 0, n = 0;
    uint32_t cross_b = 0xffffffffu;
    const uint32_t w_first = blockIdx.x * 2 * PAIRS_PER_BLOCK;
    const uint32_t w_last = min(n_work - 1, w_first + 2 * PAIRS_PER_BLOCK - 1);
    const uint32_t l_first = rec_locus[work[w_first]], l_last = rec_locus[work[w_last]];
    const uint32_t n_loc = l_last - l_first + 1;          // <= loci_cap (checked when this kernel is chosen)
    if (active) {
        rid_a = work[2 * pair];
        rid_b = has_b0 ? work[2 * pair + 1] : rid_a;
        const vtx_record ra = records[rid_a], rb = records[rid_b];
        loc_a = rec_locus[rid_a]; loc_b = rec_locus[rid_b];
        // pair-table mode (pcols != 0) looks both reads up in ONE table: a pair that straddles two loci scores A
        // alone here and hands B to the redo list (one record per locus boundary)
        if (pcols && loc_b != loc_a) { cross_b = rid_b; rid_b = rid_a; loc_b = loc_a; }
        const vtx_locus la = loci[loc_a], lb = loci[loc_b];
        m_a = ra.read_len; roff_a = ra.read_off; m_b = rb.read_len; roff_b = rb.read_off;
        n = max(max(la.ref_len, la.alt_len), max(lb.ref_len, lb.alt_len));
    } else {
        loc_a = loc_b = l_first;                           // idle rows read a valid table
    }

    // ---- tables of loci l_first .. l_last (as sw_full_lut_kernel) + their common prefix lengths ----
    const uint32_t tab_words = lcols * LUT_CODES;
    if ((uint32_t)tid < n_loc && (uint32_t)tid < 8) s_v[tid] = min(loci[l_first + tid].ref_len, loci[l_first + tid].alt_len);
    __syncthreads();
    for (uint32_t t = 0; t < n_loc && t < loci_cap; ++t) {
        const vtx_locus loc = loci[l_first + t];
        uint32_t* tab = smem + (size_t)t * tab_words;
        for (uint32_t idx = tid; idx < lcols; idx += 256) {
            const int j = (int)idx - PRE;
            uint32_t rc = 0x200u, ac = 0x200u;            // sentinel: equals no base
            if (j >= 0) {
                if ((uint32_t)j < loc.ref_len) rc = hap_arena[loc.ref_off + j];
                if ((uint32_t)j < loc.alt_len) ac = hap_arena[loc.alt_off + j];
                if (rc != ac && (uint32_t)j < min(loc.ref_len, loc.alt_len)) atomicMin(&s_v[t], (uint32_t)j);
            }
            const uint32_t bases[5] = {'A', 'C', 'G', 'T', 'N'};
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint32_t wr = (rc == bases[k]) ? 0x0001u : 0xfffbu;
                const uint32_t wa = (ac == bases[k]) ? 0x0001u : 0xfffbu;
                tab[idx * LUT_CODES + k] = wr | (wa << 16);
            }
            tab[idx * LUT_CODES + 5] = 0xfffbfffbu;
        }
    }

    // ---- pair tables (deep data): pair[t][column][codeA * 6 + codeB] = {m(codeA), m(codeB)} against the common
    //      prefix, so phase P needs ONE lookup per cell and no v_perm.  38 words per column: the 16 lanes of a row
    //      (consecutive columns) fall on distinct even banks, as with the 6-word stride of the single tables. ----
    constexpr uint32_t PSTRIDE = 38;
    const uint32_t pair_base_words = loci_cap * tab_words;
    if (pcols) {
        for (uint32_t t = 0; t < n_loc && t < loci_cap; ++t) {
            const vtx_locus loc = loci[l_first + t];
            uint32_t* ptab = smem + pair_base_words + (size_t)t * pcols * PSTRIDE;
            for (uint32_t item = tid; item < pcols * LUT_CODES; item += 256) {
                const uint32_t idx = item / LUT_CODES, a = item % LUT_CODES;
                const int j = (int)idx - PRE;
                uint32_t rc = 0x200u;
                if (j >= 0 && (uint32_t)j < loc.ref_len) rc = hap_arena[loc.ref_off + j];
                const uint32_t bases[6] = {'A', 'C', 'G', 'T', 'N', 0x300u};
                const uint32_t ma = (rc == bases[a]) ? 0x0001u : 0xfffbu;
#pragma unroll
                for (int b = 0; b < LUT_CODES; ++b)
                    ptab[idx * PSTRIDE + a * LUT_CODES + b] = ma | (((rc == bases[b]) ? 0x0001u : 0xfffbu) << 16);
            }
        }
    }

    // ---- rows of this lane for both reads ----
    const uint32_t lane_base_a = ((uint32_t)(loc_a - l_first) * tab_words + (uint32_t)(PRE - l) * LUT_CODES) * 4u;
    const uint32_t lane_base_b = ((uint32_t)(loc_b - l_first) * tab_words + (uint32_t)(PRE - l) * LUT_CODES) * 4u;
    uint32_t addr[R], X[R];                               // X: B's LUT addresses in phase P, B's saved state in phase A
    uint64_t codes_b = 0;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t i = (uint32_t)(l * R + r);
        uint32_t ca = 5, cb = 5;
        if (i < m_a) { ca = base_code(read_arena[roff_a + i]); if (ca > 5) { bad = true; ca = 5; } }
        if (i < m_b) { cb = base_code(read_arena[roff_b + i]); if (cb > 5) { bad = true; cb = 5; } }
        addr[r] = lane_base_a + ca * 4u;
        X[r] = pcols ? (pair_base_words + (uint32_t)(loc_a - l_first) * pcols * PSTRIDE + (uint32_t)(PRE - l) * PSTRIDE +
                        ca * LUT_CODES + cb) * 4u
                     : lane_base_b + cb * 4u;
        codes_b |= (uint64_t)cb << (3 * r);
    }
    __syncthreads();
    uint32_t badm = bad ? 1u : 0u;
#pragma unroll
    for (int off = 1; off < GL; off <<= 1) badm |= (uint32_t)__shfl_xor((int)badm, off);

    // ---- wave-uniform phase lengths ----
    // All lanes change phase at the same STEP: P for steps [0, T1), A for [T1, T1 + S), B for [T1 + S, T1 + 2S).
    // Lane l is at column t - l, so it leaves phase P at column T1 - l <= T1 <= v: still inside the common prefix,
    // where {A vs REF, A vs ALT} are the same numbers, so re-packing {A, B} -> {A, A} is exact at any such column.
    // Lane l then runs columns [T1 - l, T1 - l + S) for A and again for B; S = n - T1 + 15 rounded up to 4 covers
    // the last column for every lane.
    uint32_t vw = min(s_v[loc_a - l_first], s_v[loc_b - l_first]);
    vw = min(vw, (uint32_t)__shfl_xor((int)vw, 16));
    vw = min(vw, (uint32_t)__shfl_xor((int)vw, 32));
    uint32_t nw = n;
    nw = max(nw, (uint32_t)__shfl_xor((int)nw, 16));
    nw = max(nw, (uint32_t)__shfl_xor((int)nw, 32));
    nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)nw);
    uint32_t T1 = min((uint32_t)__builtin_amdgcn_readfirstlane((int)vw), nw);
    if (pcols) T1 = min(T1, pcols - PRE);              // the pair tables hold PRE + (pcols - PRE) prefix columns
    T1 &= ~3u;
    const uint32_t S = (nw - T1 + (GL - 1) + 3) & ~3u;

    uint32_t Ha[R], Hb[R], Q[R], E[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { Ha[r] = 0; Hb[r] = 0; Q[r] = 0; E[r] = 0; }
    uint32_t best = 0, out_a = 0;
    uint32_t hu_a = 0, hu_b = 0, qu = 0, fu = 0, f_last = 0, q_bottom = 0;
    const char* lds = (const char*)smem;

#define DUO_EXCHANGE(hcur, hsrc)                                                                   \
        hcur = lane_shr1<DPP_ROW_SHR1>(hcur, hsrc);                                                \
        qu = lane_shr1<DPP_ROW_SHR1>(qu, q_bottom);                                                \
        fu = lane_shr1<DPP_ROW_SHR1>(fu, f_last);
    // the DP column update given the substitution words W(r)
#define DUO_COLUMN(HS, HD, hprev, W)                                                               \
    {                                                                                              \
        uint32_t hd = hprev, qa = qu, fa = fu;                                                     \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                            \
            const uint32_t w = W;                                                                  \
            const uint32_t tt = pk_add(hd, w);                                                     \
            hd = HS[r];                                                                            \
            const uint32_t e = pk_max(pk_sub_sat(E[r], PK(1)), Q[r]);                              \
            const uint32_t f = pk_max(pk_sub_sat(fa, PK(1)), qa);                                  \
            const uint32_t h = pk_max(pk_max(tt, e), f);                                           \
            best = pk_max(best, tt);                                                               \
            E[r] = e;                                                                              \
            HD[r] = h;                                                                             \
            Q[r] = pk_sub_sat(h, PK(6));                                                           \
            fa = f; qa = Q[r];                                                                     \
        }                                                                                          \
        f_last = fa; q_bottom = Q[R - 1];                                                          \
    }
#define LDSW(a, OFF) (*(const uint32_t*)(lds + (a) + (OFF)))
#define W_P(OFF) perm_b32(LDSW(X[r], OFF), LDSW(addr[r], OFF), SEL_LOA_LOB)
#define W_S(OFF) LDSW(addr[r], OFF)
#define P_STEP(HS, HD, hprev, hcur, OFF) { DUO_EXCHANGE(hcur, HS[R - 1]) DUO_COLUMN(HS, HD, hprev, W_P(OFF)) }
#define S_STEP(HS, HD, hprev, hcur, OFF) { DUO_EXCHANGE(hcur, HS[R - 1]) DUO_COLUMN(HS, HD, hprev, W_S(OFF)) }
    // a group of four steps starts with "previous column" = Hb, diagonal carry = hu_b, incoming carry = hu_a

    // ---------------- phase P: steps [0, T1), halves {A, B} ----------------
    if (pcols) {
#define W_PP(OFF) LDSW(X[r], OFF)
#define PP_STEP(HS, HD, hprev, hcur, OFF) { DUO_EXCHANGE(hcur, HS[R - 1]) DUO_COLUMN(HS, HD, hprev, W_PP(OFF)) }
        for (uint32_t t4 = 0; t4 < (T1 >> 2); ++t4) {
            PP_STEP(Hb, Ha, hu_b, hu_a, 0 * PSTRIDE * 4)
            PP_STEP(Ha, Hb, hu_a, hu_b, 1 * PSTRIDE * 4)
            PP_STEP(Hb, Ha, hu_b, hu_a, 2 * PSTRIDE * 4)
            PP_STEP(Ha, Hb, hu_a, hu_b, 3 * PSTRIDE * 4)
#pragma unroll
            for (int r = 0; r < R; ++r) { addr[r] += 4 * LUT_CODES * 4; X[r] += 4 * PSTRIDE * 4; }
        }
#undef PP_STEP
#undef W_PP
    } else {
        for (uint32_t t4 = 0; t4 < (T1 >> 2); ++t4) {
            P_STEP(Hb, Ha, hu_b, hu_a, 0 * LUT_CODES * 4)
            P_STEP(Ha, Hb, hu_a, hu_b, 1 * LUT_CODES * 4)
            P_STEP(Hb, Ha, hu_b, hu_a, 2 * LUT_CODES * 4)
            P_STEP(Ha, Hb, hu_a, hu_b, 3 * LUT_CODES * 4)
#pragma unroll
            for (int r = 0; r < R; ++r) { addr[r] += 4 * LUT_CODES * 4; X[r] += 4 * LUT_CODES * 4; }
        }
    }
    // ---------------- phases A and B: single lookup, S steps each ----------------
    uint32_t in_b = 0, dg_b = 0;          // B halves of the carries received at step T1 / of diagonal carry and best
    for (uint32_t phase = 0; phase < 2; ++phase) {
        // first step of the phase: the exchange delivers what the lane above produced in its LAST step of the previous
        // phase.  At P -> A those are {A, B} values of exactly the column this lane is about to process.
        DUO_EXCHANGE(hu_a, Hb[R - 1])
        if (phase == 0) {
            in_b = perm_b32(qu, hu_a, SEL_HIA_HIB);                   // {h_up_B, q_up_B}
            dg_b = perm_b32(best, hu_b, SEL_HIA_HIB);                 // {diag_B, best_B}
            const uint32_t fu_b = fu >> 16;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                X[r] = perm_b32(E[r], Hb[r], SEL_HIA_HIB);            // {H_B, E_B} of the column before the switch
                Hb[r] = perm_b32(0, Hb[r], SEL_LO_LO);
                E[r] = perm_b32(0, E[r], SEL_LO_LO);
                Q[r] = perm_b32(0, Q[r], SEL_LO_LO);
            }
            hu_a = perm_b32(0, hu_a, SEL_LO_LO); qu = perm_b32(0, qu, SEL_LO_LO); fu = perm_b32(0, fu, SEL_LO_LO);
            hu_b = perm_b32(0, hu_b, SEL_LO_LO); best = perm_b32(0, best, SEL_LO_LO);
            out_a = fu_b;                                             // parked here until A is finished
        } else {
            const uint32_t fu_b = out_a;
            out_a = best;                                             // A's {REF, ALT} best of this lane's rows
            // B restarts at the column where this lane left phase P, with the carries the lane above sent for it
            hu_a = perm_b32(0, in_b, SEL_LO_LO); qu = perm_b32(0, in_b, SEL_HI_HI); fu = fu_b * 0x10001u;
            hu_b = perm_b32(0, dg_b, SEL_LO_LO); best = perm_b32(0, dg_b, SEL_HI_HI);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Hb[r] = perm_b32(0, X[r], SEL_LO_LO);
                E[r] = perm_b32(0, X[r], SEL_HI_HI);
                Q[r] = pk_sub_sat(Hb[r], PK(6));
                addr[r] = lane_base_b + T1 * (LUT_CODES * 4) + ((uint32_t)(codes_b >> (3 * r)) & 7u) * 4u;
            }
        }
        DUO_COLUMN(Hb, Ha, hu_b, W_S(0 * LUT_CODES * 4))
        S_STEP(Ha, Hb, hu_a, hu_b, 1 * LUT_CODES * 4)
        S_STEP(Hb, Ha, hu_b, hu_a, 2 * LUT_CODES * 4)
        S_STEP(Ha, Hb, hu_a, hu_b, 3 * LUT_CODES * 4)
#pragma unroll
        for (int r = 0; r < R; ++r) addr[r] += 4 * LUT_CODES * 4;
        for (uint32_t t4 = 1; t4 < (S >> 2); ++t4) {
            S_STEP(Hb, Ha, hu_b, hu_a, 0 * LUT_CODES * 4)
            S_STEP(Ha, Hb, hu_a, hu_b, 1 * LUT_CODES * 4)
            S_STEP(Hb, Ha, hu_b, hu_a, 2 * LUT_CODES * 4)
            S_STEP(Ha, Hb, hu_a, hu_b, 3 * LUT_CODES * 4)
#pragma unroll
            for (int r = 0; r < R; ++r) addr[r] += 4 * LUT_CODES * 4;
        }
    }
#undef P_STEP
#undef S_STEP
#undef W_P
#undef W_S
#undef LDSW
#undef DUO_COLUMN
#undef DUO_EXCHANGE

#pragma unroll
    for (int off = 1; off < GL; off <<= 1) {
        best = pk_max(best, (uint32_t)__shfl_xor((int)best, off));
        out_a = pk_max(out_a, (uint32_t)__shfl_xor((int)out_a, off));
    }
    if (active && l == 0) {
        if (cross_b != 0xffffffffu) redo[atomicAdd(redo_count, 1u)] = cross_b;
        const bool has_b = has_b0 && cross_b == 0xffffffffu;
        if (badm) {
            const uint32_t k = atomicAdd(redo_count, has_b ? 2u : 1u);
            redo[k] = rid_a;
            if (has_b) redo[k + 1] = rid_b;
        } else {
            ref_score[rid_a] = (int32_t)(int16_t)(out_a & 0xffffu);
            alt_score[rid_a] = (int32_t)(int16_t)(out_a >> 16);
            if (has_b) {
                ref_score[rid_b] = (int32_t)(int16_t)(best & 0xffffu);
                alt_score[rid_b] = (int32_t)(int16_t)(best >> 16);
            }
        }
    }
}

extern "C" hipError_t vtxk_launch_sw_full_duo(int R, uint32_t n_work, const uint32_t* work, const vtx_record* records,
                                              const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                              const uint8_t* hap_arena, int32_t* ref_score, int32_t* alt_score,
                                              uint32_t max_hap_len, uint32_t loci_cap, uint32_t* redo, uint32_t* redo_count,
                                              uint32_t pair_cols, hipStream_t stream) {
    if (n_work == 0) return hipSuccess;
    // columns: PRE sentinels + haplotype (at least 16) + lane skew + 4x unroll slack
    const uint32_t lcols = 16 + (max_hap_len > 16 ? max_hap_len : 16) + 16 + 4;
    // pair_cols != 0: per locus also a pair table of pair_cols columns x 38 words (see the kernel)
    const size_t shmem = ((size_t)loci_cap * lcols * LUT_CODES + (size_t)loci_cap * pair_cols * 38) * sizeof(uint32_t);
    if (shmem > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid((n_work + 31) / 32), block(256);
#define CASE(r)                                                                                          \
    if (R == r) {                                                                                        \
        if (shmem > 48 * 1024) {                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)sw_full_duo_kernel<r>,                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);  \
            if (e != hipSuccess) return e;                                                               \
        }                                                                                                \
        hipLaunchKernelGGL((sw_full_duo_kernel<r>), grid, block, shmem, stream, work, n_work, records,   \
                           rec_locus, loci, read_arena, hap_arena, ref_score, alt_score, lcols, loci_cap, redo, redo_count, pair_cols); \
        return hipGetLastError();                                                                        \
    }
    CASE(2) CASE(4) CASE(6) CASE(8) CASE(10) CASE(12) CASE(16)
#undef CASE
    return hipErrorInvalidValue;
}

extern "C" hipError_t vtxk_launch_sw_full_lut(int R, uint32_t n_work, const uint32_t* work, const vtx_record* records,
                                              const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                              const uint8_t* hap_arena, int32_t* ref_score, int32_t* alt_score,
                                              uint32_t max_hap_len, uint32_t loci_cap, uint32_t* redo, uint32_t* redo_count,
                                              hipStream_t stream) {
    if (n_work == 0) return hipSuccess;
    // columns: PRE sentinels + haplotype + (GL - 1 + 3) trailing steps (4x unrolled loop) + 1
    const uint32_t lcols = 16 + max_hap_len + 16 + 4;
    const size_t shmem = (size_t)loci_cap * lcols * LUT_CODES * sizeof(uint32_t);
    const dim3 grid((n_work + 15) / 16), block(256);
#define CASE(r)                                                                                          \
    if (R == r) {                                                                                        \
        if (shmem > 48 * 1024) {                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)sw_full_lut_kernel<r>,                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);  \
            if (e != hipSuccess) return e;                                                               \
        }                                                                                                \
        hipLaunchKernelGGL((sw_full_lut_kernel<r>), grid, block, shmem, stream, work, n_work, records,   \
                           rec_locus, loci, read_arena, hap_arena, ref_score, alt_score, lcols, loci_cap, redo, redo_count); \
        return hipGetLastError();                                                                        \
    }
    CASE(2) CASE(4) CASE(6) CASE(8) CASE(10) CASE(12) CASE(16)
#undef CASE
    return hipErrorInvalidValue;
}
