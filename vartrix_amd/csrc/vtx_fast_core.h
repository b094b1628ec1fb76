// vtx_fast_core.h — per-task logic of band_diag_kernel, the first stage of the banded flavour
// (bio 0.30.0 banded::Aligner::local as restated in oracle/vtx_oracle.c; reference call site src/main.rs:898-901).
//
// Plain C++ without HIP types, so that the SAME source compiles into the device kernel (vtx_band.hip) and into the
// host unit test of the kernel logic (tests/fastcore_host.cpp, checked against the oracle on the CPU).  The host build
// is test infrastructure only; nothing in the product path calls it.
//
// What a task is: one read x (m bases) against one haplotype y (n bases).  band_run_kernel decides a task without a DP
// when  cert == ub  (cert <= banded <= full <= ub; proof in vtx_band.hip / oracle/vtx_certify.c) but builds the general
// machinery — all k-mer matches as diagonal pieces, a closed-form sdpkpp over the pieces, the run bound over all
// ordered pairs — for every task: 28 k lane-instructions per alignment, half of them exec-mask bookkeeping.  Nearly all
// tasks have ONE diagonal that carries the alignment and a handful of isolated spurious 6-mer matches elsewhere
// (150 x 196 / 4^6 = 7 expected).  This file decides exactly those tasks, with the same two bounds, from:
//
//   M      the match mask of the main diagonal d (bit i: x[i] == y[i + d]) — 8-byte compares, no hashing
//   S      the off-diagonal k-mer matches.  A row whose main-diagonal k-mer is intact AND unique in the haplotype
//          cannot have another match (band_run_kernel's continuation shortcut, as a bit mask): only the other rows
//          are probed in the k-mer table (~28 of 145 on the synthetic workloads).
//
//   chain  sdpkpp restricted to the main diagonal is a closed form over its pieces (runs of >= 6 matching bases):
//          inside a piece every k-mer continues the previous one (+1), the first k-mer of a piece takes the best
//          earlier piece end (V = dp + xe + ye, ties to the later piece) if that gives >= 6.  An off-diagonal match s
//          is HARMLESS when no main-diagonal match can take it as its predecessor:  V_s - (x_p + y_p) + 1 < dp(p)  for
//          every main match p that starts at or after the end of s (checked at the first such p of every piece: the left
//          side falls by 2 per row, the right side grows by 1), and dp(s) < the best chain score.  dp(s) itself is exact:
//          its candidates are the main matches and the earlier off-diagonal matches that end before it, plus the
//          continuation of an off-diagonal match one step up its diagonal.  If every s is harmless, every dp and every
//          predecessor of the main matches is what it would be without S, so the chain of the reference — first match,
//          last match — is the chain of the closed form.  Anything else (a tie that would need the match order, an s
//          that is not harmless, too many pieces / matches) leaves the task to band_run_kernel.
//   cert   the chain lies on diagonal d, so the staircase of band_finish is the diagonal itself from (first - d0 - t0)
//          to (last + K + d1 + t1): the certificate is the best local score (match +1, mismatch -5, restart at 0) of M
//          over that window — a scan over the runs of M.
//   ub     the run bound over the GENERIC pieces only: the main-diagonal pieces plus the off-diagonal pieces within
//          T = 20 diagonals of the hull of the generic diagonals (closure).  The other off-diagonal pieces are FAR: every
//          one lies >= T diagonals outside the hull.  With E = the number of far k-mer matches (sum of len - 5 over the
//          far pieces, <= SM = 20):
//            * a chain of far pieces between two generic pieces g1, g2 nets at most E - 5 - 2T - |d1 - d2| (every join
//              costs >= 5 + its diagonal difference, the way out and back is >= 2T + |d1 - d2|), the direct join g1 -> g2
//              costs 5 + |d1 - d2| or J_same <= 10:  replacing the excursion by the direct join never lowers the value
//              (E <= 2T, E + 5 <= 2T);
//            * far pieces before the first / after the last generic piece net at most E - T <= 0: dropping them never lowers it;
//            * a chain of far pieces only is worth at most E + 5.
//          Hence  full <= max(5, E + 5, ub_generic)  with ub_generic the fixpoint of run_ub over the generic pieces.
//          (oracle/vtx_certify.c: vtxo_runs_ub_generic restates this on the CPU; tests check full <= it.)
//
// Capacities: reads up to 192 bases (3 mask words), RM main pieces, SM off-diagonal matches, GM generic off-diagonal
// pieces; tasks beyond them are not wrong, they are band_run_kernel's.
#ifndef VTX_FAST_CORE_H
#define VTX_FAST_CORE_H

#include <stdint.h>

#include "../../include/vtx_band_semantics.h"

#ifdef __HIPCC__
#define VTXF_FN __device__ __forceinline__
#define VTXF_MEM __device__ __forceinline__
#define VTXF_HD static __host__ __device__ inline
#else
#define VTXF_FN static inline
#define VTXF_MEM inline
#define VTXF_HD static inline
#endif

namespace vtxf {

constexpr int K = 6;
constexpr int W = 20;
constexpr int LAZY = VTX_BAND_LAZY_EXT(6);
constexpr int MAX_READ = 192;   // mask capacity
constexpr int RM = 6;           // main-diagonal pieces
constexpr int SM = 20;          // off-diagonal k-mer matches
constexpr int GM = 4;           // off-diagonal pieces admitted to the generic set
constexpr int TFAR = 20;        // far = at least this many diagonals outside the hull of the generic diagonals
constexpr int PM = RM + GM;     // generic pieces
constexpr int LANE_WORDS = SM + 2 * PM;   // per-lane scratch: off-diagonal matches, then (id, len | G << 16) per generic piece
static_assert(2 * TFAR >= SM + 5 && TFAR >= SM, "far-piece lemma: E <= SM must satisfy E + 5 <= 2T and E <= T");
constexpr uint32_t UQ_PAD_WORDS = 6;      // zero words in front of a table's unique-k-mer bit array (a negative diagonal reads them)
constexpr uint32_t CH_END_ = 0xffffu;

// reasons a task is left to band_run_kernel (statistics)
enum Why : uint32_t { W_OK = 0, W_SHAPE = 1, W_NO_DIAG = 2, W_PIECES = 3, W_MATCHES = 4, W_NOT_HARMLESS = 5, W_TIE = 6,
                      W_GENERIC = 7, W_NOT_TIGHT = 8, W_NO_MAIN = 9, W_COUNT = 10 };

// k-mer table of one haplotype in global memory (layout: band_table_stride / build_tables in vtx_band.hip):
// ent[max_hap] {bytes 0-3, bytes 4-5 | next << 16}, head[n_heads] u16, bytes[max_hap + 8], fb[max_hap + 8],
// uq[UQ_PAD_WORDS + max_hap / 32 + 8] (bit y of the array behind the padding: the k-mer starting at y is unique)
struct Tab {
    const uint8_t* gt;      // uniform base of the table buffer
    uint32_t ent, head, bytes, uq;   // byte offsets of this haplotype's arrays
    uint32_t hmask;         // n_heads - 1
};
VTXF_HD uint32_t tab_bytes_off(uint32_t max_hap, uint32_t n_heads) { return max_hap * 8u + n_heads * 2u; }
VTXF_HD uint32_t tab_fb_off(uint32_t max_hap, uint32_t n_heads) { return tab_bytes_off(max_hap, n_heads) + max_hap + 8u; }
VTXF_HD uint32_t tab_uq_off(uint32_t max_hap, uint32_t n_heads) { return (tab_fb_off(max_hap, n_heads) + max_hap + 8u + 3u) & ~3u; }
VTXF_HD uint32_t tab_uq_words(uint32_t max_hap) { return UQ_PAD_WORDS + (max_hap + 31u) / 32u + 8u; }
VTXF_HD uint32_t tab_stride(uint32_t max_hap, uint32_t n_heads) { return (tab_uq_off(max_hap, n_heads) + 4u * tab_uq_words(max_hap) + 15u) & ~15u; }

VTXF_FN uint32_t kw_hash(uint32_t lo, uint32_t hi, uint32_t head_mask) {
    const uint32_t h = (lo ^ (hi << 11) ^ (hi >> 3)) * 0x9E3779B1u;
    return (h >> 18) & head_mask;
}
VTXF_FN uint64_t ld8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
VTXF_FN uint32_t ld4(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
VTXF_FN uint32_t ld2(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
VTXF_FN int imin(int a, int b) { return a < b ? a : b; }
VTXF_FN int imax(int a, int b) { return a > b ? a : b; }
VTXF_FN int iabs(int a) { return a < 0 ? -a : a; }

// ---- 192-bit masks ----
struct M192 { uint64_t w[3]; };
VTXF_FN int ctz64(uint64_t v) { return __builtin_ctzll(v); }
VTXF_FN M192 m_and(M192 a, M192 b) { return M192{{a.w[0] & b.w[0], a.w[1] & b.w[1], a.w[2] & b.w[2]}}; }
VTXF_FN M192 m_andn(M192 a, M192 b) { return M192{{a.w[0] & ~b.w[0], a.w[1] & ~b.w[1], a.w[2] & ~b.w[2]}}; }
template <int S> VTXF_FN M192 m_shr(M192 a) {      // 0 < S < 64
    return M192{{(a.w[0] >> S) | (a.w[1] << (64 - S)), (a.w[1] >> S) | (a.w[2] << (64 - S)), a.w[2] >> S}};
}
// bits [lo, hi) set, 0 <= lo, hi <= 192
VTXF_FN uint64_t ones_below(int n) { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull)); }
VTXF_FN M192 m_range(int lo, int hi) {
    M192 r;
    for (int k = 0; k < 3; ++k) r.w[k] = ones_below(hi - 64 * k) & ~ones_below(lo - 64 * k);
    return r;
}
// first set bit at or after p (192 if none)
VTXF_FN int m_next_set(const M192& a, int p) {
    if (p >= 192) return 192;
    int k = p >> 6;
    uint64_t v = a.w[k] & ~ones_below(p & 63);
    while (v == 0) { if (++k == 3) return 192; v = a.w[k]; }
    return 64 * k + ctz64(v);
}
VTXF_FN int m_next_clear(const M192& a, int p) {
    if (p >= 192) return 192;
    int k = p >> 6;
    uint64_t v = ~a.w[k] & ~ones_below(p & 63);
    while (v == 0) { if (++k == 3) return 192; v = ~a.w[k]; }
    return 64 * k + ctz64(v);
}

// 8 byte-equality flags of two 8-byte words as 8 bits
VTXF_FN uint32_t eq8(uint64_t a, uint64_t b) {
    const uint64_t z = a ^ b, lo7 = 0x7f7f7f7f7f7f7f7full;
    const uint64_t t = ~(((z & lo7) + lo7) | z | lo7);        // 0x80 in every byte of z that is zero
    const uint32_t l = (uint32_t)(t >> 7), h = (uint32_t)(t >> 39);   // flags at bits 0, 8, 16, 24
    const uint32_t cl = (l | (l >> 7) | (l >> 14) | (l >> 21)) & 0xfu;
    const uint32_t ch = (h | (h >> 7) | (h >> 14) | (h >> 21)) & 0xfu;
    return cl | (ch << 4);
}

VTXF_FN int join_same(int D) {                      // run_ub's same-diagonal join (vtx_band.hip: ub_join_same)
    const int c = 6 * ((D + 10) / 6) - D;
    const int g = imax(7, 13 - D);
    return imin(c, g);
}

// per-lane scratch: element i at base[i * stride] (device: LDS, the 64 lanes of a wavefront interleaved; host: stride 1)
struct Lane {
    uint32_t* base; int stride;
    VTXF_MEM uint32_t& at(int i) const { return base[i * stride]; }
};

struct Result { int32_t score; uint32_t why; };

// The match mask of diagonal d: bit i = (x[i] == y[i + d]), i in [max(0, -d), min(m, n - d)).
VTXF_FN M192 diag_mask(const uint8_t* x, int m, const Tab& tb, int n, int d) {
    M192 M{{0, 0, 0}};
    const uint8_t* yb = tb.gt + tb.bytes;
    // only the 8-base words that overlap the haplotype: the 8-byte loads stay within 7 bytes of bytes[0, n)
    const int w0 = d < 0 ? (-d) >> 3 : 0;
    for (int w = w0; 8 * w < m && 8 * w + d < n; ++w) {
        const uint64_t rd = ld8(x + 8 * w);
        const uint64_t hp = ld8(yb + (8 * w + d));     // (bytes outside [0, n) are padding / neighbouring arrays: masked below)
        M.w[w >> 3] |= (uint64_t)eq8(rd, hp) << (8 * (w & 7));
    }
    return m_and(M, m_range(imax(0, -d), imin(m, n - d)));
}

// Decide one task.  Returns {score, W_OK} or {-1, reason}.
VTXF_FN Result fast_task(const uint8_t* x, int m, const Tab& tb, int n, const Lane& ln) {
    if (m < K || n < K || m > MAX_READ) return Result{-1, W_SHAPE};
    const uint8_t* ent = tb.gt + tb.ent;
    const uint8_t* head = tb.gt + tb.head;
    const uint32_t* uq = (const uint32_t*)(tb.gt + tb.uq);

    // ---- 1. the main diagonal: a sampled row whose k-mer has exactly one match, on a unique haplotype k-mer; the
    //         candidate is kept if its mask holds at least one piece and 20 matching bases ----
    int d = 0;
    M192 M{{0, 0, 0}};
    bool have_d = false;
    {
        const int last = m - K;
        const int step = imax(1, last / 5);
        // middle rows first: the ends of a read hang over the padded window more often than its middle
        const int order[6] = {2, 3, 1, 4, 0, 5};
        for (int t = 0; t < 6 && !have_d; ++t) {
            const int row = imin(order[t] * step, last);
            const uint64_t w8 = ld8(x + row);
            const uint32_t lo = (uint32_t)w8, hi = (uint32_t)(w8 >> 32) & 0xffffu;
            uint32_t yc = ld2(head + 2u * kw_hash(lo, hi, tb.hmask));
            int cnt = 0, ycand = 0;
            while (yc != CH_END_) {
                const uint64_t e = ld8(ent + 8u * yc);
                if ((uint32_t)e == lo && ((uint32_t)(e >> 32) & 0xffffu) == hi) { ++cnt; ycand = (int)yc; }
                yc = (uint32_t)(e >> 48);
            }
            if (cnt != 1) continue;
            const uint32_t ub_ = (uint32_t)ycand + 32u * UQ_PAD_WORDS;
            if (!((uq[ub_ >> 5] >> (ub_ & 31u)) & 1u)) continue;
            const int dc = ycand - row;
            const M192 Mc = diag_mask(x, m, tb, n, dc);
            const int pop = __builtin_popcountll(Mc.w[0]) + __builtin_popcountll(Mc.w[1]) + __builtin_popcountll(Mc.w[2]);
            if (pop < 20) continue;
            d = dc; M = Mc; have_d = true;
        }
    }
    if (!have_d) return Result{-1, W_NO_DIAG};

    // ---- 2. main-diagonal pieces: runs of >= K matching bases [pu, pv] ----
    int pu[RM], pv[RM];
    int r = 0;
    {
        int p = 0;
        for (;;) {
            const int u = m_next_set(M, p);
            if (u >= m) break;
            const int e = imin(m_next_clear(M, u), m);        // run [u, e)
            if (e - u >= K) {
                if (r == RM) return Result{-1, W_PIECES};
                pu[r] = u; pv[r] = e - 1; ++r;
            }
            p = e;
        }
    }
    if (r == 0) return Result{-1, W_NO_MAIN};

    // ---- 3. rows that may hold an off-diagonal match: all but those whose main-diagonal k-mer is intact and unique ----
    M192 need;
    {
        const M192 A = m_and(M, m_shr<1>(M));
        const M192 B = m_and(A, m_shr<2>(A));
        const M192 I6 = m_and(B, m_shr<4>(A));              // bit i: bases i .. i + 5 all match
        // unique-k-mer bits of haplotype positions [d, d + 192): the array carries 192 zero bits in front
        const uint32_t bo = (uint32_t)(d + 32 * (int)UQ_PAD_WORDS);
        const uint32_t wi = bo >> 5, sh = bo & 31u;
        uint32_t q[7];
        for (int k = 0; k < 7; ++k) q[k] = uq[wi + k];
        M192 U;
        for (int k = 0; k < 3; ++k) {
            const uint32_t a = (uint32_t)((((uint64_t)q[2 * k + 1] << 32) | q[2 * k]) >> sh);
            const uint32_t b = (uint32_t)((((uint64_t)q[2 * k + 2] << 32) | q[2 * k + 1]) >> sh);
            U.w[k] = ((uint64_t)b << 32) | a;
        }
        need = m_andn(m_range(0, m - K + 1), m_and(I6, U));
    }

    // ---- 4. probe those rows: off-diagonal matches, in (x, y) order ----
    int ns = 0;
    for (int row = m_next_set(need, 0); row < 192; row = m_next_set(need, row + 1)) {
        const uint64_t w8 = ld8(x + row);
        const uint32_t lo = (uint32_t)w8, hi = (uint32_t)(w8 >> 32) & 0xffffu;
        uint32_t yc = ld2(head + 2u * kw_hash(lo, hi, tb.hmask));
        while (yc != CH_END_) {
            const uint64_t e = ld8(ent + 8u * yc);
            if ((uint32_t)e == lo && ((uint32_t)(e >> 32) & 0xffffu) == hi && (int)yc - row != d) {
                if (ns == SM) return Result{-1, W_MATCHES};
                ln.at(ns++) = ((uint32_t)row << 16) | yc;
            }
            yc = (uint32_t)(e >> 48);
        }
    }

    // ---- 5. sdpkpp on the main diagonal (closed form over the pieces) ----
    // piece i: k-mer matches at rows a .. b = pv - 5; dpf = dp of its first match, dpl of its last
    int dpf[RM], pred[RM];
    int best_i = 0;
    for (int i = 0; i < r; ++i) {
        int bc = -1000000, bj = -1;
        for (int j = 0; j < i; ++j) {
            const int dpl_j = dpf[j] + (pv[j] - 5 - pu[j]);
            const int c = dpl_j + 1 - 2 * (pu[i] - (pv[j] + 1));       // gap of pu[i] - (b_j + 6) rows and columns
            if (c >= bc) { bc = c; bj = j; }                          // equal V: the later piece (larger match index)
        }
        if (bj >= 0 && bc >= K) { dpf[i] = bc; pred[i] = bj; } else { dpf[i] = K; pred[i] = -1; }
        const int dpl_i = dpf[i] + (pv[i] - 5 - pu[i]);
        const int dpl_b = dpf[best_i] + (pv[best_i] - 5 - pu[best_i]);
        if (dpl_i >= dpl_b) best_i = i;                               // equal score: the later match
    }
    const int best_dp = dpf[best_i] + (pv[best_i] - 5 - pu[best_i]);
    int root = best_i;
    while (pred[root] >= 0) root = pred[root];

    // ---- 6. every off-diagonal match must be harmless (see the file header) ----
    // V of the best main match visible to a start (px, py): ended at or before it in both coordinates
    int sdp[SM];
    for (int k = 0; k < ns; ++k) {
        const int sx = (int)(ln.at(k) >> 16), sy = (int)(ln.at(k) & 0xffffu);
        // exact dp of this match: candidates = main matches and earlier off-diagonal matches that end before it
        int bv = -1000000;
        {
            const int lim = imin(sx, sy - d) - K;                     // last main row that ends before (sx, sy)
            for (int i = 0; i < r; ++i) {
                const int t = imin(lim - pu[i], pv[i] - 5 - pu[i]);
                if (t < 0) continue;
                const int v = dpf[i] + t + 2 * (pu[i] + t + K) + d;
                bv = imax(bv, v);
            }
        }
        int cont = -1;
        for (int j = 0; j < k; ++j) {
            const int jx = (int)(ln.at(j) >> 16), jy = (int)(ln.at(j) & 0xffffu);
            if (jx + 1 == sx && jy + 1 == sy) cont = j;
            if (jx + K <= sx && jy + K <= sy) {
                bv = imax(bv, sdp[j] + jx + jy + 2 * K);
            }
        }
        int dp = K;
        if (bv > -1000000) {
            const int cand = bv - (sx + sy) + 1;
            if (cand >= K) dp = cand;          // (WHICH predecessor an off-diagonal match takes never matters: it is not on the chain)
        }
        if (cont >= 0 && sdp[cont] + 1 >= dp) dp = sdp[cont] + 1;
        sdp[k] = dp;
        if (dp >= best_dp) return Result{-1, W_NOT_HARMLESS};         // could end the chain
        const int vs = dp + sx + sy + 2 * K;
        // no main match may prefer it: first match of every piece that starts at or after its end
        for (int i = 0; i < r; ++i) {
            const int t0 = imax(0, imax(sx + K - pu[i], sy + K - d - pu[i]));
            if (t0 > pv[i] - 5 - pu[i]) continue;
            const int cand = vs - (2 * (pu[i] + t0) + d) + 1;
            if (cand >= dpf[i] + t0) return Result{-1, W_NOT_HARMLESS};
        }
    }

    // ---- 7. certificate: best local score of M over the in-band stretch of the diagonal ----
    int cert = 0;
    {
        const int fx = pu[root], fy = fx + d;
        const int d0 = imin(imin(fx, fy), LAZY);
        const int t0 = imin(imin(fx - d0, fy - d0), W);
        const int lo = fx - d0 - t0;
        int re = pv[best_i] + 1, ce = re + d;                          // cell after the last k-mer: (b + K, b + K + d)
        const int d1 = imin(imin(m - re, n - ce), LAZY);
        re += d1; ce += d1;
        const int t1 = imin(imin(m - re, n - ce), W);
        const int hi = re + t1;
        int s = 0, p = lo;
        while (p < hi) {
            const int q = imin(m_next_set(M, p), hi);
            s = imax(0, s - 5 * (q - p));
            if (q >= hi) break;
            const int e = imin(m_next_clear(M, q), hi);
            s += e - q;
            cert = imax(cert, s);
            p = e;
        }
    }

    // ---- 8. run bound over the generic pieces; far pieces by the lemma ----
    // generic list: (id = x0 << 16 | y0, bases | G << 16) at ln[SM + 2 j], ln[SM + 2 j + 1]
    int ng = 0;
    for (int i = 0; i < r; ++i) {
        ln.at(SM + 2 * ng) = ((uint32_t)pu[i] << 16) | (uint32_t)(pu[i] + d);
        ln.at(SM + 2 * ng + 1) = (uint32_t)(pv[i] - pu[i] + 1);
        ++ng;
    }
    int far_e = 0;
    {
        // off-diagonal pieces: a match that continues another one belongs to its piece (matches are in (x, y) order, a
        // continuation sits in the next row)
        int hull_lo = d, hull_hi = d;
        uint32_t used = 0;                                            // matches already assigned (heads of generic pieces, or members)
        bool grew = true;
        while (grew) {
            grew = false;
            for (int k = 0; k < ns; ++k) {
                if ((used >> k) & 1u) continue;
                const int sx = (int)(ln.at(k) >> 16), sy = (int)(ln.at(k) & 0xffffu);
                bool is_head = true;
                for (int j = 0; j < k; ++j) if ((int)(ln.at(j) >> 16) + 1 == sx && (int)(ln.at(j) & 0xffffu) + 1 == sy) is_head = false;
                if (!is_head) continue;
                const int ds = sy - sx;
                if (ds > hull_hi + TFAR - 1 || ds < hull_lo - TFAR + 1) continue;     // (still) far
                // piece length in k-mers
                int len = 1;
                uint32_t members = 1u << k;
                for (int j = k + 1; j < ns; ++j)
                    if ((int)(ln.at(j) >> 16) == sx + len && (int)(ln.at(j) & 0xffffu) == sy + len) { ++len; members |= 1u << j; }
                if (ng == PM) return Result{-1, W_GENERIC};
                ln.at(SM + 2 * ng) = ln.at(k);
                ln.at(SM + 2 * ng + 1) = (uint32_t)(len + K - 1);
                ++ng;
                used |= members;
                hull_lo = imin(hull_lo, ds); hull_hi = imax(hull_hi, ds);
                grew = true;
            }
        }
        for (int k = 0; k < ns; ++k) if (!((used >> k) & 1u)) ++far_e;     // E = far k-mer matches = sum over far pieces of bases - 5
    }
    int ub = imax(K - 1, far_e > 0 ? far_e + 5 : 0);
    {
        bool changed = true;
        for (int pass = 0; pass < 6 && changed; ++pass) {
            changed = false;
            for (int p = 0; p < ng; ++p) {
                const uint32_t idp = ln.at(SM + 2 * p), dlp = ln.at(SM + 2 * p + 1);
                const int xp = (int)(idp >> 16), yp = (int)(idp & 0xffffu), lp = (int)(dlp & 0xffffu);
                const int g0 = (int)(dlp >> 16);
                int g = g0;
                for (int q = 0; q < ng; ++q) {
                    if (q == p) continue;
                    const uint32_t idq = ln.at(SM + 2 * q), dlq = ln.at(SM + 2 * q + 1);
                    const int xq = (int)(idq >> 16), yq = (int)(idq & 0xffffu), lq = (int)(dlq & 0xffffu), gq = (int)(dlq >> 16);
                    int s = imax(xq + lq - xp, yq + lq - yp);
                    s = imin(imax(s, 0), lp - 1);
                    const int t = imin(lq - 1, imin(xp - xq, yp - yq) + s - 1);
                    if (t < 0) continue;
                    const int dd = (yp - xp) - (yq - xq);
                    int J = 5 + iabs(dd);
                    if (dd == 0) { const int D = xp + s - xq - t - 1; J = D == 0 ? 0 : join_same(D); }
                    g = imax(g, t + 1 + gq - J - s);
                }
                if (g != g0) { ln.at(SM + 2 * p + 1) = (dlp & 0xffffu) | ((uint32_t)g << 16); changed = true; }
            }
        }
        if (changed) return Result{-1, W_NOT_TIGHT};
        for (int p = 0; p < ng; ++p) {
            const uint32_t dl = ln.at(SM + 2 * p + 1);
            ub = imax(ub, (int)(dl & 0xffffu) + (int)(dl >> 16));
        }
    }
    if (cert != ub) return Result{-1, W_NOT_TIGHT};
    return Result{cert, W_OK};
}

}  // namespace vtxf
#endif
