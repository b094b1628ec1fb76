// vtx_fast_core.h — per-task logic of band_diag_kernel, the first stage of the banded flavour
// (bio 0.30.0 banded::Aligner::local as restated in oracle/vtx_oracle.c; reference call site src/main.rs:898-901).
//
// Plain C++ without HIP types, so that the SAME source compiles into the device kernel (vtx_band.hip) and into the
// host unit test of the kernel logic (tests/fastcore/fastcore_host.cpp, checked against the oracle on the CPU).  The host
// build is test infrastructure only; nothing in the product path calls it.
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
//          inside a piece every k-mer continues the previous one (+1); the first k-mer of a piece takes the best
//          earlier piece end (largest V = dp + xe + ye, ties to the later piece) if that gives >= 6.  An off-diagonal
//          match s is HARMLESS when no main-diagonal match can take it as its predecessor:
//              V_s - (x_p + y_p) + 1 < dp(p)   for every main match p that starts at or after the end of s
//          (checked at the first such p of every piece: the left side falls by 2 per row, the right side grows by 1), and
//          dp(s) < the best chain score.  For dp(s) an UPPER bound is enough (the test is one-sided):
//              dp(s) <= max(6, best main match that ends before s, 1 + max dp bound of the off-diagonal matches before s)
//          — a match reached from another off-diagonal match s' gets dp(s') + 1 - gap <= dp(s') (gap >= 1 unless it
//          continues s' on its diagonal: + 1).  If every s is harmless, every dp and every predecessor of the main
//          matches is what it would be without S, so the reference's chain — first match, last match — is the chain
//          of the closed form.  Anything else leaves the task to band_run_kernel.
//   cert   the chain lies on diagonal d, so the staircase of band_finish is the diagonal itself from (first - d0 - t0)
//          to (last + K + d1 + t1): the certificate is the best local score (match +1, mismatch -5, restart at 0) of M
//          over that window — a scan over the zeros of M.
//   ub     the run bound over the GENERIC pieces only: the main-diagonal pieces plus the off-diagonal pieces of the closure
//          below.  The other off-diagonal pieces are FAR: piece f lies D_f >= 1 diagonals outside the hull of the generic
//          diagonals and holds E_f k-mer matches (bases - 5).  Condition (*): for every far piece j
//              sum of E_f over the far pieces with D_f <= D_j   <=   bound(D_j) = min(D_j, 2 D_j - 6).
//          Take a chain of sub-runs of pieces (the objects of the run bound, oracle/vtx_certify.c) and a maximal run S of
//          consecutive far pieces in it, D = the largest D_f in S, E = sum of E_f over S <= bound(D) by (*):
//            * between two generic pieces g1, g2 (diagonals d1, d2 inside the hull): the k + 1 joins cost >= 5 each plus the
//              way out to distance D and back, >= 2 D + |d1 - d2|, the k pieces bring E + 5 k bases: the excursion nets
//              <= E - 5 - 2 D - |d1 - d2|; the direct join g1 -> g2 (possible: g2 is entered after g1 is left, in both
//              coordinates) costs 5 + |d1 - d2|, or J_same <= 11 when d1 == d2.  E <= 2 D - 6: replacing the excursion by
//              the direct join never lowers the value;
//            * before the first / after the last generic piece: S nets <= E - D <= 0 (k - 1 joins among the far pieces, one
//              join >= 5 + the way to the hull): dropping it never lowers the value;
//            * a chain of far pieces only is worth <= (all far matches) + 5.
//          Hence  full <= max(5, E_far + 5, ub_generic)  with ub_generic the fixpoint of run_ub over the generic pieces.
//          (Round 3 first used the special case "every far piece >= T = max(ns, 5) diagonals out": with the 20-40 chance
//          matches of real sequence T made everything generic.)
//
// Capacities: reads up to 64 * NW bases (NW mask words: 256 since round 6, 192 before), RM main pieces, SM off-diagonal matches, GM generic off-diagonal
// pieces; tasks beyond them are not wrong, they are band_run_kernel's.
//
// Three phases, so that the device can do the middle one cooperatively (a wavefront's probes pooled over its lanes):
//   front()   diagonal, mask, pieces + chain, certificate, the rows to probe
//   probe     every match (row, y) with y - row != d of the rows in Front::need, appended to the lane's list
//   back()    sort the list, harmless test, closure, run bound, verdict
#ifndef VTX_FAST_CORE_H
#define VTX_FAST_CORE_H

#include <stdint.h>

#include "../../include/vtx_band_semantics.h"

#ifdef __HIPCC__
#define VTXF_FN __device__ __forceinline__
#define VTXF_MEM __device__ __forceinline__
#define VTXF_HD static __host__ __device__ inline
#define VTXF_UNROLL _Pragma("unroll")
#else
#define VTXF_UNROLL
#define VTXF_FN static inline
#define VTXF_MEM inline
#define VTXF_HD static inline
#endif

namespace vtxf {

constexpr int K = VTX_REF_K;
constexpr int W = VTX_REF_W;
static_assert(K == 6 && W == 20 && VTX_REF_MATCH == 1 && VTX_REF_MISMATCH == -5 && VTX_REF_GAP_OPEN == -5 && VTX_REF_GAP_EXTEND == -1,
              "the closed forms below (piece dp, join_same, the far-piece lemma, the +1 / -5 scan) are derived for the reference's scoring only");
constexpr int LAZY = VTX_BAND_LAZY_EXT(6);
#ifndef VTXF_NW
#define VTXF_NW 4
#endif
constexpr int NW = VTXF_NW;     // 64-bit words of a diagonal's match mask (3: rounds 3 - 5; the A/B build of tools/gpu_campaign.sh)
static_assert(NW == 3 || NW == 4, "rows, columns and piece ends are bytes: 256 bases at most");
constexpr int MAX_READ = 64 * NW;   // mask capacity
constexpr int RM = 8;           // main-diagonal pieces
#ifndef VTXF_S_WORDS
#define VTXF_S_WORDS 20
#endif
constexpr int S_WORDS = VTXF_S_WORDS;     // LDS words per lane for the off-diagonal k-mer matches: 40 two-byte entries (haplotypes <= 255 bases) or 20 four-byte ones
constexpr int GM = 6;           // off-diagonal pieces admitted to the generic set
constexpr int LANE_WORDS = S_WORDS + RM;   // per-lane scratch: off-diagonal matches, main pieces (+ GM words for back(): generic off-diagonal pieces)
constexpr int DMAX = 120;       // diagonal offsets of generic pieces are stored in a signed byte
constexpr uint32_t UQ_PAD_WORDS = 8;      // zero words in front of a table's unique-k-mer bit array (a negative diagonal reads them: 2 * NW at most)
constexpr uint32_t HEAD_END = 0xffffu;    // empty bucket / end of a chain
constexpr uint32_t HEAD_MULTI = 15u;      // tag of a bucket with more than one entry

// reasons a task is left to band_run_kernel (statistics)
enum Why : uint32_t { W_OK = 0, W_SHAPE = 1, W_NO_DIAG = 2, W_PIECES = 3, W_MATCHES = 4, W_NOT_HARMLESS = 5, W_UNUSED6 = 6,
                      W_GENERIC = 7, W_NOT_TIGHT = 8, W_NO_MAIN = 9, W_COUNT = 10 };

// k-mer table of one haplotype in global memory (layout: band_table_stride / build_tables in vtx_band.hip):
// ent[max_hap] {bytes 0-3, bytes 4-5 | next << 16}, head[n_heads] u16 (position of the first entry | tag << 12: a bucket
// with ONE entry carries 4 hash bits that are not part of the bucket index, a bucket with more HEAD_MULTI),
// bytes[max_hap + 8], fb[max_hap + 8], uq[UQ_PAD_WORDS + max_hap / 32 + 8] (bit y behind the padding: the k-mer
// starting at y is unique in the haplotype), pb[128] (4096 bits: bit kw_code(k-mer) set for every k-mer of the haplotype — a
// probe of a k-mer that is not there ends at ONE word of these 512 bytes instead of a head word of a 2 KB array)
struct Tab {
    const uint8_t* gt;      // uniform base of the table buffer
    uint32_t ent, head, bytes, uq, pb;   // byte offsets of this haplotype's arrays
    uint32_t hmask;         // n_heads - 1
};
VTXF_HD uint32_t tab_bytes_off(uint32_t max_hap, uint32_t n_heads) { return max_hap * 8u + n_heads * 2u; }
VTXF_HD uint32_t tab_fb_off(uint32_t max_hap, uint32_t n_heads) { return tab_bytes_off(max_hap, n_heads) + max_hap + 8u; }
VTXF_HD uint32_t tab_uq_off(uint32_t max_hap, uint32_t n_heads) { return (tab_fb_off(max_hap, n_heads) + max_hap + 8u + 3u) & ~3u; }
VTXF_HD uint32_t tab_uq_words(uint32_t max_hap) { return UQ_PAD_WORDS + (max_hap + 31u) / 32u + 8u; }
VTXF_HD uint32_t tab_pb_off(uint32_t max_hap, uint32_t n_heads) { return tab_uq_off(max_hap, n_heads) + 4u * tab_uq_words(max_hap); }
// tw[TW_BYTES] behind pb[] (round 6): the haplotype's TWIN LIST — byte 0 = its length (TW_NONE: no list — more than TW_MAX pairs, a
// byte >= 0x80 in the haplotype, or a haplotype above 256 bases), from byte 8 on the pairs (y, y'), y != y', of positions that hold
// the same k-mer, in (y, y') order.  A read row whose main-diagonal k-mer is intact has, as off-diagonal matches, exactly the twins
// of its haplotype position: band_diag_kernel takes them from this list (twin_matches) and probes only the rows whose k-mer is NOT
// intact — on the headline workload 5 of the 28 rows a task probed, and 82 % of the bucket walks, were of this kind.
constexpr uint32_t TW_MAX = 60, TW_BYTES = 128, TW_NONE = 0xffu;
VTXF_HD uint32_t tab_tw_off(uint32_t max_hap, uint32_t n_heads) { return tab_pb_off(max_hap, n_heads) + 512u; }
// t3[256] behind tw[] (round 6), 8 bytes each: the presence of THREE consecutive read k-mers in one load.  Eight read bases
// b0 .. b7 hold the k-mers of rows r, r + 1, r + 2, which share the four bases b2 .. b5: entry S = their 8-bit code (kw_code's two
// bits per base) carries three 16-bit sets — bits 0-15: the (b0, b1) in front of S that exist in the haplotype as a k-mer b0 b1 S,
// bits 16-31: the (b1, b6) around S, bits 32-47: the (b6, b7) behind it.  band_diag_kernel's pooled probes test a block of three rows
// with one 8-byte load per haplotype instead of one 4-byte load per row (the rows a task probes come in runs: the bases that hang over
// the haplotype's window, the six rows around an error).  Like pb[], a filter: bytes outside ACGT alias, the walk compares the bytes.
#ifndef VTXF_T3
#define VTXF_T3 1          // (0: tables without t3[], a queue entry per row — the A/B build libvtx_not3.so of tools/gpu_campaign.sh variants)
#endif
constexpr uint32_t T3_BYTES = VTXF_T3 ? 2048 : 0;
VTXF_HD uint32_t tab_t3_off(uint32_t max_hap, uint32_t n_heads) { return tab_tw_off(max_hap, n_heads) + TW_BYTES; }
VTXF_HD uint32_t tab_stride(uint32_t max_hap, uint32_t n_heads) { return (tab_t3_off(max_hap, n_heads) + T3_BYTES + 15u) & ~15u; }
// (a table in LDS — band_run_kernel without a global table buffer — ends behind pb[]: only band_diag_kernel reads tw[] and t3[])
VTXF_HD uint32_t tab_stride_lds(uint32_t max_hap, uint32_t n_heads) { return (tab_tw_off(max_hap, n_heads) + 15u) & ~15u; }
// where k-mer code c (kw_code) sits in t3: word index (entry S, 2 words each: sets 0 | 1 << 16, set 2) and bit, for its three roles
VTXF_HD uint32_t t3_word_a(uint32_t c) { return 2u * (c >> 4); }
VTXF_HD uint32_t t3_bit_a(uint32_t c) { return c & 15u; }
VTXF_HD uint32_t t3_word_b(uint32_t c) { return 2u * ((c >> 2) & 0xffu); }
VTXF_HD uint32_t t3_bit_b(uint32_t c) { return 16u + ((c & 3u) | ((c >> 10) << 2)); }
VTXF_HD uint32_t t3_word_c(uint32_t c) { return 2u * (c & 0xffu) + 1u; }
VTXF_HD uint32_t t3_bit_c(uint32_t c) { return c >> 8; }
// 16-bit code of eight bases (two bits per byte, like kw_code): base i at bits 2 i
VTXF_HD uint32_t kw_code8(uint64_t w8) {
    uint32_t a = ((uint32_t)w8 >> 1) & 0x03030303u, b = ((uint32_t)(w8 >> 32) >> 1) & 0x03030303u;
    a |= a >> 6; a = (a | (a >> 12)) & 0xffu;
    b |= b >> 6; b = (b | (b >> 12)) & 0xffu;
    return a | (b << 8);
}
// 12-bit code of a k-mer, two bits per byte ((b >> 1) & 3: A 0, C 1, T 2, G 3; any other byte lands on one of them — equal
// bytes always give equal codes, which is all a presence filter needs)
VTXF_HD uint32_t kw_code(uint32_t lo, uint32_t hi) {
    uint32_t a = (lo >> 1) & 0x03030303u, b = (hi >> 1) & 0x0303u;
    a |= a >> 6; a = (a | (a >> 12)) & 0xffu;          // bytes 0-3 -> bits 0-7
    b = (b | (b >> 6)) & 0xfu;                         // bytes 4-5 -> bits 0-3
    return a | (b << 8);
}

// hash of a k-mer (bytes 0-3 in lo, bytes 4-5 in hi): bucket = bits 18.. of the product, tag = its top 4 bits
VTXF_HD uint32_t kw_mix(uint32_t lo, uint32_t hi) { return (lo ^ (hi << 11) ^ (hi >> 3)) * 0x9E3779B1u; }
VTXF_HD uint32_t kw_bucket(uint32_t h, uint32_t head_mask) { return (h >> 18) & head_mask; }
VTXF_HD uint32_t kw_tag(uint32_t h) { const uint32_t t = h >> 28; return t == HEAD_MULTI ? HEAD_MULTI - 1u : t; }

VTXF_FN uint64_t ld8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
VTXF_FN uint32_t ld2(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
VTXF_FN int imin(int a, int b) { return a < b ? a : b; }
VTXF_FN int imax(int a, int b) { return a > b ? a : b; }
VTXF_FN int iabs(int a) { return a < 0 ? -a : a; }

// ---- masks of MAX_READ bits: NW 64-bit words, always indexed with compile-time constants (the loops below unroll) ----
struct M192 { uint64_t w[NW]; };           // (the name is round 3's: 192 bits then)
VTXF_FN int ctz64(uint64_t v) { return __builtin_ctzll(v); }
VTXF_FN M192 m_zero() { M192 r; VTXF_UNROLL for (int k = 0; k < NW; ++k) r.w[k] = 0; return r; }
VTXF_FN M192 m_and(M192 a, M192 b) { M192 r; VTXF_UNROLL for (int k = 0; k < NW; ++k) r.w[k] = a.w[k] & b.w[k]; return r; }
VTXF_FN M192 m_andn(M192 a, M192 b) { M192 r; VTXF_UNROLL for (int k = 0; k < NW; ++k) r.w[k] = a.w[k] & ~b.w[k]; return r; }
VTXF_FN bool m_any(M192 a) { uint64_t v = 0; VTXF_UNROLL for (int k = 0; k < NW; ++k) v |= a.w[k]; return v != 0; }
VTXF_FN int m_pop(M192 a) { int c = 0; VTXF_UNROLL for (int k = 0; k < NW; ++k) c += __builtin_popcountll(a.w[k]); return c; }
template <int S> VTXF_FN M192 m_shr(M192 a) {      // 0 < S < 64
    M192 r;
    VTXF_UNROLL
    for (int k = 0; k < NW; ++k) r.w[k] = (a.w[k] >> S) | (k + 1 < NW ? a.w[k + 1 < NW ? k + 1 : k] << (64 - S) : 0ull);
    return r;
}
template <int S> VTXF_FN M192 m_shl(M192 a) {      // 0 < S < 64; bits shifted beyond the mask are lost
    M192 r;
    VTXF_UNROLL
    for (int k = 0; k < NW; ++k) r.w[k] = (a.w[k] << S) | (k > 0 ? a.w[k > 0 ? k - 1 : 0] >> (64 - S) : 0ull);
    return r;
}
VTXF_FN uint64_t ones_below(int n) { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull)); }
// (A: the words that can hold a one — band_diag_kernel is built for reads up to 192 bases as well, A = 3: everything derived from a
//  mask whose last word is the constant 0 costs nothing there, and the headline workload keeps round 5's instruction count)
template <int A = NW> VTXF_FN M192 m_range(int lo, int hi) {             // bits [lo, hi)
    M192 r;
    VTXF_UNROLL
    for (int k = 0; k < NW; ++k) r.w[k] = k < A ? ones_below(hi - 64 * k) & ~ones_below(lo - 64 * k) : 0ull;
    return r;
}
// lowest set bit: its index (MAX_READ: none), cleared in a
VTXF_FN int m_pop_lowest(M192& a) {
    uint64_t cur = a.w[NW - 1];
    int idx = NW - 1;
    VTXF_UNROLL
    for (int k = NW - 2; k >= 0; --k) if (a.w[k] != 0) { cur = a.w[k]; idx = k; }
    if (cur == 0) return MAX_READ;
    const int r = 64 * idx + ctz64(cur);
    const uint64_t nxt = cur & (cur - 1);
    VTXF_UNROLL
    for (int k = 0; k < NW; ++k) a.w[k] = idx == k ? nxt : a.w[k];
    return r;
}
// the set bits of a mask in ascending order, one per call: the words shift down as they run empty (NW - 1 times in a mask's life), so a
// call is one count-trailing-zeros and one clear on ONE word — m_pop_lowest selects among the words on every call
struct MIter {
    uint64_t cur, nx[NW - 1];
    int st;                                  // bits still to deliver | base << 16 (one register: band_diag_kernel has none to spare)
    VTXF_MEM int left() const { return st & 0xffff; }
};
VTXF_FN MIter m_iter(M192 a) {
    MIter it;
    it.cur = a.w[0];
    VTXF_UNROLL
    for (int k = 1; k < NW; ++k) it.nx[k - 1] = a.w[k];
    it.st = m_pop(a);
    return it;
}
VTXF_FN int m_next(MIter& it) {              // precondition: it.left() > 0
    while (it.cur == 0) {
        it.cur = it.nx[0];
        VTXF_UNROLL
        for (int k = 0; k + 2 < NW; ++k) it.nx[k] = it.nx[k + 1];
        it.nx[NW - 2] = 0; it.st += 64 << 16;
    }
    const int r = (it.st >> 16) + ctz64(it.cur);
    it.cur &= it.cur - 1;
    --it.st;
    return r;
}
// f(position) for every set bit, ascending
template <class F> VTXF_FN void m_for_each(M192 a, F f) {
    VTXF_UNROLL
    for (int k = 0; k < NW; ++k)
        for (uint64_t v = a.w[k]; v; v &= v - 1) f(64 * k + ctz64(v));
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

// Same-diagonal joins of the run bound (proof: oracle/vtx_certify.c).  D >= 1 bases between two runs on one diagonal:
//   join_free(D)     6 ceil((D + 5) / 6) - D: the gap-free stretch with as few mismatches as runs of <= 5 allow (a caller that
//                    does not look at the bases; vtx_band.hip: ub_join_same)
//   join_same(D, e)  min(6 e - D, J_gap(D)): the gap-free stretch costs exactly 6 e - D when e of the D bases mismatch (an e
//                    below the true count only loosens the bound), a stretch with gaps at least J_gap(D) = {7, 9, 11, 10, 9, 8}
//                    [D mod 6] (J_gap(1) = 12 > 6 e - D = 5)
VTXF_FN int div6(int D) { return (D * 10923) >> 16; }                // D / 6 for 0 <= D < 30000 ((D * 43) >> 8, first version, is off from D = 131: found by tests/test_fastcore.py::test_join_closed_forms_of_the_kernel_header)
VTXF_FN int join_free(int D) { return 6 * div6(D + 10) - D; }
VTXF_FN int join_same(int D, int e) {
    const int jg = (int)((0x89ab97u >> (4 * (D - 6 * div6(D)))) & 15u);
    return imin(6 * e - D, jg);
}

// ---- refinement of a same-diagonal join whose gap-free cost 6 e - D exceeds J_gap(D) (three or more errors within a few
//      bases): J_gap prices a HYPOTHETICAL stretch over perfectly matching neighbour diagonals; the real ones are looked at.
//      A stretch either stays within CORR diagonals of the runs' diagonal — then its cost is at least the exact optimum of an
//      affine DP without a floor over that corridor, started up to mu bases before the first run's end and ended up to mu bases
//      behind the second run's start at 1 per base given up (a stretch that gives up more costs >= 7 + mu + 1 >= 6 e - D for
//      mu = 6 e - D - 8) — or its gaps total >= CORR + 1 per direction: J_gap restricted to G >= 3 (join_gap3; brute force and
//      proof as for J_gap: oracle/vtx_certify.c, tests/test_certify.py).  The join costs the smaller of the two. ----
constexpr int CORR = 2;
VTXF_FN int join_gap3(int D) {                    // J_gap for G >= 3, D >= 3 (a lower bound of it above D = 22: 11 or 12)
    if (D > 22) return 11;
    const uint64_t lo = 0x0122001232012345ull;    // D = 3 .. 18: value - 11, a nibble each
    const uint32_t hi = 0x1200u;                  // D = 19 .. 22
    return 11 + (int)(D <= 18 ? (lo >> (4 * (D - 3))) & 15u : (hi >> (4 * (D - 19))) & 15u);
}
// join_gap3 assumes what J_gap assumes: between the two runs every exact-match run on ANY diagonal has <= 5 bases (a longer one would
// be a piece).  A task with main pieces only still has its FAR off-diagonal pieces (far_e k-mer matches in all), and an excursion may
// run over them: a far piece of E matches is a run of 5 + E bases.  Shortening every such run to 5 bases turns the excursion into one
// of the model over D - t bases (t <= far_e bases removed, gaps untouched) that scores t less, so the excursion costs at least
// min over t <= far_e of join_gap3(D - t) - t; and by the far-piece lemma (header; condition (*) holds for these tasks) any join through
// far pieces costs >= 11.  Round 6: found by the first full audit of the 8 % workload — the plain join_gap3 priced a join at 13 that a
// 7-base run four diagonals out made for 12 (ub 36 < full 37; the banded score happened to be 36 too).
VTXF_FN int join_gap3_far(int D, int far_e) {
    int j = join_gap3(D);
    if (far_e > 0) {
        VTXF_UNROLL
        for (int t = 1; t <= 5; ++t)                   // (join_gap3 <= 16: beyond t = 5 the floor of 11 is reached anyway)
            if (t <= far_e && D - t >= 3) j = imin(j, join_gap3(D - t) - t);
        j = imax(j, 11);
    }
    return j;
}
// x, yb: the read and the haplotype bytes; (xb, xb + d): the first run's last base; cells are prefix cells (i, j) = i read
// bases and j haplotype bases consumed, diagonal index k = j - i - d + CORR
VTXF_FN int corridor_cost(const uint8_t* x, int m, const uint8_t* yb, int n, int xb, int d, int D, int mu_a, int mu_b) {
    constexpr int NEG = -100000, W = 2 * CORR + 1;
    const int r0 = xb + 1 - mu_a, r1 = xb + D + 2 + mu_b;
    int Hp[W], Fp[W];
    VTXF_UNROLL
    for (int k = 0; k < W; ++k) { Hp[k] = NEG; Fp[k] = NEG; }
    Hp[CORR] = -mu_a;
    {
        int E = NEG;
        VTXF_UNROLL
        for (int k = CORR + 1; k < W; ++k) { E = imax(E - 1, Hp[k - 1] - 6); Hp[k] = E; }
    }
    for (int i = r0 + 1; i <= r1; ++i) {
        const uint32_t xc = (i >= 1 && i <= m) ? x[i - 1] : 0x100u;
        int Hc[W], Fc[W];
        int E = NEG;
        VTXF_UNROLL
        for (int k = 0; k < W; ++k) {
            const int j = i + d + (k - CORR);
            int h = NEG, f = NEG;
            if (j >= 0 && j <= n && i <= m) {
                if (j >= 1 && Hp[k] > NEG / 2) h = Hp[k] + ((uint32_t)yb[j - 1] == xc ? 1 : -5);
                if (k + 1 < W) { f = imax(Fp[k + 1] - 1, Hp[k + 1] - 6); if (f < NEG / 2) f = NEG; }
                if (k >= 1) { E = imax(E - 1, Hc[k - 1] - 6); if (E < NEG / 2) E = NEG; } else E = NEG;
                h = imax(h, imax(f, E));
            } else E = NEG;
            Hc[k] = h; Fc[k] = f;
        }
        VTXF_UNROLL
        for (int k = 0; k < W; ++k) { Hp[k] = Hc[k]; Fp[k] = Fc[k]; }
    }
    return Hp[CORR] <= NEG / 2 ? (1 << 20) : mu_b + 1 - Hp[CORR];
}

// per-lane scratch (device: LDS, the 64 lanes of a wavefront interleaved; host: stride 1)
//   s(k), k < SMAX          off-diagonal matches x << XS | y, ST = uint16_t (XS = 8, 40 entries: every haplotype of the batch has
//                           <= 255 bases — padding 100 with indels up to 54) or uint32_t (XS = 16, 20 entries)
//   at(i), i < RM           main pieces: first base | last base << 8 | dp of the first k-mer << 16 | G << 24
// and a second, GM-word scratch for back() only (the device lends it the probe queue's LDS):
//   at(i), i < GM           generic off-diagonal pieces: x | (delta + 128) << 8 | bases << 16 | G << 24
struct Lane {
    uint32_t* base; int stride;
    VTXF_MEM uint32_t& at(int i) const { return base[i * stride]; }
};
// (UB4: back_rest runs the main-only run bound on piece words held in registers — not in band_diag_kernel's four-word build, which
//  has no register to spare: a spilled dword costs every launch its scratch set-up)
template <class ST, int SW = S_WORDS, bool UB4_ = true> struct LaneS {
    typedef ST SType;
    static constexpr bool UB4 = UB4_;
    uint32_t* base; int stride;
    ST* sb; int sstride;
    static constexpr int XS = sizeof(ST) == 2 ? 8 : 16;
    static constexpr int SMAX = SW * 4 / (int)sizeof(ST);
    static constexpr uint32_t YM = (1u << XS) - 1u, ONE = (1u << XS) | 1u;
    static constexpr bool TIGHT = false;
    VTXF_MEM uint32_t& at(int i) const { return base[i * stride]; }
    VTXF_MEM ST& s(int k) const { return sb[k * sstride]; }
};
// The lane of the SECOND stage (band_diag2_kernel, round 5): two-byte entries, S2_WORDS words of them, and a byte per entry for the
// dp bound of the harmless test (u(k)): with it the bound of an off-diagonal match comes from the matches that can really precede it
// (back_harmless), not from every earlier one.
#ifndef VTXF_S2_WORDS
#define VTXF_S2_WORDS 32
#endif
constexpr int S2_WORDS = VTXF_S2_WORDS;   // 64 entries: what back_rest's closure holds (a bit set, byte counters).  (With 120 the list also
                                           // settled the harmless test of tasks with 65 .. 120 matches; band_stream_kernel does that for any number,
                                           // and the smaller list lets ten wavefronts per CU overlap their loads instead of six: -4 ms of 178.)
template <int W> struct LaneS2T {
    typedef uint16_t SType;
    uint32_t* base; int stride;
    uint16_t* sb; int sstride;
    uint8_t* ub; int ustride;
    static constexpr int XS = 8;
    static constexpr int SMAX = W * 2;
    static constexpr uint32_t YM = 0xffu, ONE = 0x101u;
    static constexpr bool TIGHT = true;
    static constexpr bool UB4 = true;
    VTXF_MEM uint32_t& at(int i) const { return base[i * stride]; }
    VTXF_MEM uint16_t& s(int k) const { return sb[k * sstride]; }
    VTXF_MEM uint8_t& u(int k) const { return ub[k * ustride]; }
};
typedef LaneS2T<S2_WORDS> LaneS2;
#ifndef VTXF_WIN_WORDS
#define VTXF_WIN_WORDS 32
#endif
constexpr int WIN_WORDS = VTXF_WIN_WORDS;  // band_stream_kernel's window: 64 entries
typedef LaneS2T<WIN_WORDS> LaneW;

struct Front {
    uint32_t why;           // W_OK: go on with the probes
    int d, r;               // main diagonal, main pieces
    int best_dp, cert;
    int ca, cb;             // the chain's staircase: the cells (c - d, c), c = ca .. cb (lazy extensions included) — with every off-diagonal
                            // match harmless this IS the reference's staircase, so the band is its (2w + 1)-squares (band_pack)
    uint32_t zc;            // nibble i: mismatching bases between main pieces i - 1 and i (capped at 15; nibble 0 unused)
    M192 need;              // rows to probe
};

// The band of a task whose chain lies on ONE diagonal, as one word: (d + 256) << 16 | ca << 8 | cb (haplotypes up to 255 bases).
// Column j of the DP matrix is in band for j in [ca - W, cb + W]: rows [max(0, max(j - W, ca) - d - W), min(m + 1, min(j + W, cb) - d + W + 1))
// (vtx_band.hip's closed form with rmin[c] = rmax[c] = c - d).  sw_banded_kernel<.., 2> expands it.
// (with haplotypes above 255 bases the word still carries d — band_refine_kernel reads it there — and ca / cb are not used)
VTXF_FN uint32_t band_pack(const Front& fr) { return ((uint32_t)(fr.d + 256) << 16) | (((uint32_t)fr.ca & 0xffu) << 8) | ((uint32_t)fr.cb & 0xffu); }

// The read's first 192 bases as 8-byte words in registers (RW words): loaded once per task — by the device with 16-byte
// loads split between the two haplotype lanes of a record (read_words_pair in vtx_band.hip), by the host plainly.  (The bases of a
// fourth mask word — reads above 192 bases — are loaded where they are compared: eight more words in registers would not fit
// band_diag_kernel's 128 without spills.)
constexpr int RW = 192 / 8;
struct ReadWords { uint64_t w[RW]; };
VTXF_FN ReadWords read_words(const uint8_t* x, int m) {
    ReadWords r;
    for (int k = 0; k < RW; ++k) r.w[k] = 8 * k < m ? ld8(x + 8 * k) : 0ull;       // (a read's bytes are followed by >= 8 readable ones)
    return r;
}

// The match mask of diagonal d: bit i = (x[i] == y[i + d]), i in [max(0, -d), min(m, n - d)).
// Eight 8-base words per mask word, their haplotype bytes loaded together (a loop of load - wait - compare steps is a chain of
// 19 L2 round trips per task) and SIXTEEN bytes per load: every lane has its own address, so what such a load costs the address
// unit does not depend on its width — four loads per mask word instead of eight.  Words that do not overlap the haplotype are
// masked, not skipped: their bytes lie inside the same table (the head words in front of bytes[], the flag bytes behind it),
// readable and ignored.
struct W16 { uint64_t a, b; };
VTXF_FN W16 ld16(const uint8_t* p) { W16 v; __builtin_memcpy(&v, p, 16); return v; }
template <int C> VTXF_FN uint64_t diag_mask_word(const ReadWords& rw, const uint8_t* x, const uint8_t* yb, int d, int wa, int wb) {
    W16 h[4];
VTXF_UNROLL
    for (int k = 0; k < 4; ++k) h[k] = ld16(yb + (8 * (8 * C + 2 * k) + d));
    uint64_t out = 0;
VTXF_UNROLL
    for (int k = 0; k < 8; ++k) {
        const int w = 8 * C + k;
        const uint64_t xw = w < RW ? rw.w[w < RW ? w : 0] : ld8(x + 8 * (w >= wa && w < wb ? w : 0));        // (compile-time choice: C is)
        const uint64_t e = (uint64_t)eq8(xw, (k & 1) ? h[k >> 1].b : h[k >> 1].a) << (8 * k);
        out |= (w >= wa && w < wb) ? e : 0ull;
    }
    return out;
}
template <int A = NW> VTXF_FN M192 diag_mask(const ReadWords& rw, const uint8_t* x, int m, const Tab& tb, int n, int d) {
    const uint8_t* yb = tb.gt + tb.bytes;
    // only the 8-base words that overlap the haplotype: the 8-byte loads stay within 7 bytes of bytes[0, n)
    const int wa = d < 0 ? (-d) >> 3 : 0;
    const int wb = imin((m + 7) >> 3, (n - d + 7) >> 3);
    if (wa >= wb) return m_zero();
    M192 M;
    M.w[0] = diag_mask_word<0>(rw, x, yb, d, wa, wb);
    M.w[1] = wb > 8 ? diag_mask_word<1>(rw, x, yb, d, wa, wb) : 0ull;
    M.w[2] = wb > 16 ? diag_mask_word<2>(rw, x, yb, d, wa, wb) : 0ull;
    if constexpr (NW > 3) M.w[NW - 1] = (A > 3 && wb > 24) ? diag_mask_word<NW - 1>(rw, x, yb, d, wa, wb) : 0ull;
    return m_and(M, m_range<A>(imax(0, -d), imin(m, n - d)));
}

// One bucket lookup for the k-mer in w8's low 6 bytes: f(y) for every position of the haplotype that holds it.
// head word `raw` already loaded (so that callers can issue the loads of several rows together).
template <class F> VTXF_FN void walk_bucket(const Tab& tb, uint64_t w8, uint32_t h, uint32_t raw, F f) {
    if (raw == HEAD_END) return;
    const uint32_t tag = raw >> 12;
    if (tag != HEAD_MULTI && tag != kw_tag(h)) return;     // the bucket's only k-mer is another one
    const uint32_t lo = (uint32_t)w8, hi = (uint32_t)(w8 >> 32) & 0xffffu;
    const uint8_t* ent = tb.gt + tb.ent;
    uint32_t yc = raw & 0xfffu;
    do {
        const uint64_t e = ld8(ent + 8u * yc);
        if ((uint32_t)e == lo && ((uint32_t)(e >> 32) & 0xffffu) == hi) f(yc);
        yc = (uint32_t)(e >> 48);
    } while (yc != HEAD_END);
}

// ---- phase 1 ----
constexpr int NO_DIAG = -100000;
// row of the t-th sample (t = 0 .. 5) of the main-diagonal search: middle rows first — the ends of a read hang over the padded
// window more often than its middle
// (t = 6 .. 11, round 4: six more rows between the first six — a clean read whose six samples all fell into the overhang, onto an
// error or onto a repeated k-mer was 60 % of what band_diag_kernel left on the headline workload)
constexpr int N_SAMPLES = 12;
VTXF_FN int sample_row(int t, int m) {
    const int last = m - K;
    if (t < 6) return imin((int)((0x504132u >> (4 * t)) & 0xfu) * imax(1, last / 5), last);
    return imin((int)((0x619375u >> (4 * (t - 6))) & 0xfu) * last / 10 + (t == 11 ? 3 : 0), last);      // tenths 5, 7, 3, 9, 1, and 6 (+ 3 rows)
}
// candidate diagonal from one row: its k-mer's bucket holds ONE k-mer with its four tag bits — most likely the k-mer itself, and then
// the haplotype holds it exactly once.  (Until round 6 the bucket's entry was loaded and compared: one more dependent load per
// round of the search.  A candidate is only ever a candidate — verify_diag looks at eight bases of its diagonal, the mask must hold
// twenty matching ones — so a bucket whose k-mer merely shares the tag costs a wasted check, nothing else.)
VTXF_FN int cand_diag(const uint8_t* x, int row, const Tab& tb) {
    const uint64_t w8 = ld8(x + row);
    const uint32_t hh = kw_mix((uint32_t)w8, (uint32_t)(w8 >> 32) & 0xffffu);
    const uint32_t raw = ld2(tb.gt + tb.head + 2u * kw_bucket(hh, tb.hmask));
    // (Measured and not kept: the first entry of a bucket with SEVERAL k-mers as a candidate too, checked on all eight bases — fewer second
    //  rounds of the search, headline 13.61 -> 13.55 ms, but in repeats it names diagonals a unit off: real sequence 152 -> 158 ms, the
    //  config-5 shape 32.0 -> 33.3.)
    if (raw == HEAD_END || (raw >> 12) != kw_tag(hh)) return NO_DIAG;
    return (int)(raw & 0xfffu) - row;
}
// cheap check of a candidate diagonal before its whole mask is computed: 8 bases in the middle of the overlap, at least 6 of
// them equal (a chance k-mer match elsewhere in the haplotype passes with probability ~1e-3)
VTXF_FN bool verify_diag(const uint8_t* x, int m, const Tab& tb, int n, int dc) {
    const int vlo = imax(0, -dc), vhi = imin(m, n - dc);
    if (vhi - vlo < 20) return false;                     // (the mask must hold >= 20 matching bases anyway)
    const int p = vlo + ((vhi - vlo - 8) >> 1);
    return __builtin_popcount(eq8(ld8(x + p), ld8(tb.gt + tb.bytes + (p + dc)))) >= 6;
}
template <class LN, int A = NW> VTXF_FN Front front_rest(const uint8_t* x, int m, const Tab& tb, int n, const LN& ln, int d, M192 M, bool tw = false);

// one lane on its own: the six sample rows in turn; a candidate is kept if its mask has at least 20 matching bases
template <class LN> VTXF_FN Front front(const uint8_t* x, int m, const Tab& tb, int n, const LN& ln, bool tw = false) {
    Front fr;
    fr.why = W_OK; fr.d = 0; fr.r = 0; fr.best_dp = 0; fr.cert = 0; fr.ca = fr.cb = 0; fr.zc = 0; fr.need = m_zero();
    if (m < K || n < K || m > MAX_READ) { fr.why = W_SHAPE; return fr; }
    int prev = NO_DIAG;
    const ReadWords rw = read_words(x, m);
    for (int t = 0; t < N_SAMPLES; ++t) {
        const int dc = cand_diag(x, sample_row(t, m), tb);
        if (dc == NO_DIAG || dc == prev) continue;
        prev = dc;
        if (!verify_diag(x, m, tb, n, dc)) continue;
        const M192 Mc = diag_mask(rw, x, m, tb, n, dc);
        if (m_pop(Mc) >= 20) return front_rest(x, m, tb, n, ln, dc, Mc, tw);
    }
    fr.why = W_NO_DIAG;
    return fr;
}

// everything of phase 1 behind the choice of the diagonal d (M = diag_mask(d))
// tw: the caller takes the off-diagonal matches of the rows with an intact main-diagonal k-mer from the haplotype's twin list
// (twin_matches): fr.need = the rows whose k-mer is not intact, whether the haplotype's k-mer there is unique or not
template <class LN, int A> VTXF_FN Front front_rest(const uint8_t* x, int m, const Tab& tb, int n, const LN& ln, int d, M192 M, bool tw) {
    (void)x;
    Front fr;
    fr.why = W_OK; fr.d = 0; fr.r = 0; fr.best_dp = 0; fr.cert = 0; fr.ca = fr.cb = 0; fr.zc = 0; fr.need = m_zero();
    fr.d = d;

    // ---- main pieces (runs of >= K matching bases, found between the zeros of M) and sdpkpp on the diagonal ----
    // piece: k-mer matches at rows a = first base .. b = last base - 5; dpf = dp of its first match, dpl of its last.
    // bestV = max over earlier pieces of dpl + 2 (b + K) (their V without the constant d): the first match of a piece takes
    // it as predecessor if that gives >= K (ties: the later piece = the larger match index)
    int r = 0;
    int bestV = -1000000, bestRoot = 0;
    int best_dp = -1, best_root = 0, best_last = 0;
    bool too_many = false;
    uint32_t zc = 0;
    {
        // (bases outside [vlo, vhi) face no haplotype base: zeros of M, but nothing to iterate over)
        const int vlo = imax(0, -d), vhi = imin(m, n - d);
        int prev = vlo - 1;
        int nz = 0, nz_piece = 0;                           // zeros of M seen so far / when the last piece was taken
        auto piece = [&](int u, int v) {                    // bases [u, v]
            if (v - u + 1 < K) return;
            if (r == RM) { too_many = true; return; }
            zc |= (uint32_t)imin(nz - nz_piece, 15) << (4 * r);
            nz_piece = nz;
            const int a = u, b = v - 5;
            const int c = bestV - 2 * a + 1;
            int dpf = K, root = a;
            if (c >= K) { dpf = c; root = bestRoot; }
            const int dpl = dpf + (b - a);
            const int Vl = dpl + 2 * (b + K);
            if (Vl >= bestV) { bestV = Vl; bestRoot = root; }
            if (dpl >= best_dp) { best_dp = dpl; best_root = root; best_last = v; }
            ln.at(r) = (uint32_t)u | ((uint32_t)v << 8) | ((uint32_t)dpf << 16);
            ++r;
        };
        m_for_each(m_andn(m_range<A>(vlo, vhi), M), [&](int z) { piece(prev + 1, z - 1); prev = z; ++nz; });
        piece(prev + 1, vhi - 1);
    }
    if (too_many) { fr.why = W_PIECES; return fr; }
    if (r == 0) { fr.why = W_NO_MAIN; return fr; }
    fr.r = r; fr.best_dp = best_dp; fr.zc = zc;

    // ---- certificate: best local score of M over the in-band stretch of the diagonal ----
    {
        const int fx = best_root, fy = fx + d;
        const int d0 = imin(imin(fx, fy), LAZY);
        const int t0 = imin(imin(fx - d0, fy - d0), W);
        const int lo = fx - d0 - t0;
        int re = best_last + 1, ce = re + d;                           // cell after the last k-mer: (b + K, b + K + d)
        const int d1 = imin(imin(m - re, n - ce), LAZY);
        re += d1; ce += d1;
        fr.ca = fy - d0; fr.cb = ce;                                   // first / last anchor column of the staircase
        const int t1 = imin(imin(m - re, n - ce), W);
        const int hi = re + t1;
        int s = 0, best = 0, prev = lo - 1;
        m_for_each(m_andn(m_range<A>(lo, hi), M), [&](int z) {
            s += z - prev - 1;
            best = imax(best, s);
            s = imax(0, s - 5);
            prev = z;
        });
        s += hi - prev - 1;
        fr.cert = imax(best, s);
    }

    // ---- rows that may hold an off-diagonal match: all but those whose main-diagonal k-mer is intact and unique ----
    {
        const M192 P2 = m_and(M, m_shr<1>(M));
        const M192 P4 = m_and(P2, m_shr<2>(P2));
        const M192 I6 = m_and(P4, m_shr<4>(P2));            // bit i: bases i .. i + 5 all match
        // unique-k-mer bits of haplotype positions [d, d + MAX_READ): the array carries 32 * UQ_PAD_WORDS zero bits in front
        static_assert(32 * (int)UQ_PAD_WORDS >= MAX_READ, "d >= -(m - K)");
        const uint32_t* uq = (const uint32_t*)(tb.gt + tb.uq);
        const uint32_t bo = (uint32_t)(d + 32 * (int)UQ_PAD_WORDS);
        const uint32_t wi = bo >> 5, sh = bo & 31u;
        uint32_t q[2 * NW + 1];
        VTXF_UNROLL
        for (int k = 0; k < 2 * NW + 1; ++k) q[k] = (k < 2 * A + 1 && !tw) ? uq[wi + k] : 0u;
        M192 U;
        VTXF_UNROLL
        for (int k = 0; k < NW; ++k) {
            const uint32_t a = (uint32_t)((((uint64_t)q[2 * k + 1] << 32) | q[2 * k]) >> sh);
            const uint32_t b = (uint32_t)((((uint64_t)q[2 * k + 2 < 2 * NW + 1 ? 2 * k + 2 : 0] << 32) | q[2 * k + 1]) >> sh);
            U.w[k] = k < A ? (tw ? ~0ull : ((uint64_t)b << 32) | a) : 0ull;
        }
        fr.need = m_andn(m_range<A>(0, m - K + 1), m_and(I6, U));
    }
    return fr;
}

// ---- the twin list (Tab: tw[]): the off-diagonal matches of the rows whose main-diagonal k-mer is intact — a row i of [0, m - K]
//      that front_rest(.., tw = true) did NOT put into fr.need — are (i, y') for the twins y' of haplotype position i + d.  Appended
//      to ln[0 ..) in (x, y) order (the list's own); returns their number (entries beyond the lane's capacity are counted, not stored).
//      Precondition: the list exists (tw[0] != TW_NONE). ----
VTXF_FN bool tab_has_twins(const Tab& tb) { return tb.gt[tb.pb + 512u] != TW_NONE; }
// (the list's first 32 bytes — its length and twelve pairs: band_diag_kernel asks for them before front_rest, they arrive under it)
struct TwinHead { W16 h0, h1; };
VTXF_FN TwinHead twin_head(const Tab& tb) {
    TwinHead t;
    const uint8_t* tw = tb.gt + tb.pb + 512u;
    __builtin_memcpy(&t.h0, tw, 16);
    __builtin_memcpy(&t.h1, tw + 16, 16);
    return t;
}
template <class LN> VTXF_FN int twin_matches(const Tab& tb, const Front& fr, int m, const LN& ln, const TwinHead& th) {
    const uint8_t* tw = tb.gt + tb.pb + 512u;
    int ns = 0;
    auto four = [&](uint64_t w, int i0, int cnt) {                            // pairs i0 .. i0 + 3 of the list
VTXF_UNROLL
        for (int j = 0; j < 4; ++j) {
            const int y = (int)((w >> (16 * j)) & 0xffu), y2 = (int)((w >> (16 * j + 8)) & 0xffu);
            const int row = y - fr.d;
            if (i0 + j >= cnt || row < 0 || row > m - K) continue;
            const uint64_t nw = row < 64 ? fr.need.w[0] : (row < 128 ? fr.need.w[1] : (NW > 3 && row >= 192 ? fr.need.w[NW - 1] : fr.need.w[2]));
            if ((nw >> (row & 63)) & 1ull) continue;                         // the row's k-mer is not intact: it is probed
            if (ns < LN::SMAX) ln.s(ns) = (typename LN::SType)(((uint32_t)row << LN::XS) | (uint32_t)y2);
            ++ns;
        }
    };
    // (the usual haplotype has about ten pairs)
    const int cnt = (int)(th.h0.a & 0xffu);
    four(th.h0.b, 0, cnt);
    four(th.h1.a, 4, cnt);
    four(th.h1.b, 8, cnt);
    for (int i0 = 12; i0 < cnt; i0 += 4) four(ld8(tw + 8 + 2 * i0), i0, cnt);
    return ns;
}
template <class LN> VTXF_FN int twin_matches(const Tab& tb, const Front& fr, int m, const LN& ln) { return twin_matches(tb, fr, m, ln, twin_head(tb)); }

// ---- phase 2, one lane on its own: the rows of fr.need four at a time (their loads go out together); a row whose k-mer is not
//      in the haplotype's presence bitmap is done after one word.  Returns the number of off-diagonal matches appended to
//      ln[0 ..), or SM + 1 when there are more than SM. ----
template <class LN> VTXF_FN int probe_rows(const uint8_t* x, const Tab& tb, const Front& fr, const LN& ln, int ns0 = 0) {
    constexpr int SM = LN::SMAX;
    const uint8_t* head = tb.gt + tb.head;
    const uint32_t* pb = (const uint32_t*)(tb.gt + tb.pb);
    M192 need = fr.need;
    int ns = ns0;                                  // (matches the lane holds already: twin_matches)
    if (ns > SM) return SM + 1;
    while (m_any(need)) {
        int row[4];
        uint64_t w8[4];
        uint32_t code[4], bits[4];
        for (int t = 0; t < 4; ++t) row[t] = m_pop_lowest(need);
        for (int t = 0; t < 4; ++t) w8[t] = ld8(x + (row[t] < MAX_READ ? row[t] : 0));
        for (int t = 0; t < 4; ++t) {
            code[t] = kw_code((uint32_t)w8[t], (uint32_t)(w8[t] >> 32) & 0xffffu);
            bits[t] = pb[code[t] >> 5];
        }
        for (int t = 0; t < 4; ++t) {
            if (row[t] >= MAX_READ || !((bits[t] >> (code[t] & 31u)) & 1u)) continue;       // not a k-mer of this haplotype
            const uint32_t hh = kw_mix((uint32_t)w8[t], (uint32_t)(w8[t] >> 32) & 0xffffu);
            walk_bucket(tb, w8[t], hh, ld2(head + 2u * kw_bucket(hh, tb.hmask)), [&](uint32_t yc) {
                if ((int)yc - row[t] == fr.d) return;
                if (ns < SM) ln.s(ns) = (typename LN::SType)(((uint32_t)row[t] << LN::XS) | yc);
                ++ns;
            });
        }
        if (ns > SM) return SM + 1;
    }
    return ns;
}

// ---- phase 3 ----
// Returns the score (>= 0) or -1 with *why set.
// rf != nullptr: same-diagonal joins between main pieces are refined by the corridor DP (band_refine_kernel; the host test)
struct Refine { const uint8_t* x; const uint8_t* yb; int m, n; };

// The run bound over main pieces only (pl.at(i), i < r: first base | last base << 8 | . | G << 24, in base order; zc: mismatching
// bases between consecutive pieces, a nibble each): all on one diagonal — a predecessor always lies before its successor, so one
// pass over the ordered pairs q < p settles every G (no back edges), and every join is a same-diagonal join.
template <class PL> VTXF_FN int main_pieces_ub(const PL& pl, int r, uint32_t zc, int d, const Refine* rf, int far_e) {
    int ub = 0;
    for (int p = 0; p < r; ++p) {
        const uint32_t wp = pl.at(p);
        const int xp = (int)(wp & 0xffu), lp = (int)((wp >> 8) & 0xffu) - xp + 1;
        int g = 0, e = 0;
        for (int q = p - 1; q >= 0; --q) {
            const uint32_t wq = pl.at(q);
            const int xq = (int)(wq & 0xffu), lq = (int)((wq >> 8) & 0xffu) - xq + 1, gq = (int)(wq >> 24);
            // entry at the first base of p (s = 0), q used whole (t = lq - 1): D bases between them, e of them mismatches
            const int D = xp - (xq + lq);
            e += (int)((zc >> (4 * (q + 1))) & 15u);
            int J = D == 0 ? 0 : join_same(D, e);
            if (rf && D > 0 && J < 6 * e - D && e < 15) {
                // (e is exact below the nibbles' cap; the whole of q may be given up to its first base, p to its last)
                const int mu = imax(0, 6 * e - D - 8);
                const int inside = corridor_cost(rf->x, rf->m, rf->yb, rf->n, xq + lq - 1, d, D, imin(mu, lq - 1), imin(mu, lp - 1));
                J = imin(6 * e - D, imin(inside, join_gap3_far(D, far_e)));
            }
            g = imax(g, lq + gq - J);
        }
        pl.at(p) = (wp & 0x00ffffffu) | ((uint32_t)g << 24);
        ub = imax(ub, lp + g);
    }
    return ub;
}
// the same for up to four pieces held in registers, without the refinement (band_diag_kernel's usual task: no LDS round trips)
VTXF_FN int main_pieces_ub4(uint32_t (&pw4)[4], int r, uint32_t zc) {
    int ub = 0;
    VTXF_UNROLL
    for (int p = 0; p < 4; ++p) {
        if (p >= r) continue;
        const uint32_t wp = pw4[p];
        const int xp = (int)(wp & 0xffu), lp = (int)((wp >> 8) & 0xffu) - xp + 1;
        int g = 0, e = 0;
        VTXF_UNROLL
        for (int q = p - 1; q >= 0; --q) {
            const uint32_t wq = pw4[q];
            const int xq = (int)(wq & 0xffu), lq = (int)((wq >> 8) & 0xffu) - xq + 1, gq = (int)(wq >> 24);
            const int D = xp - (xq + lq);
            e += (int)((zc >> (4 * (q + 1))) & 15u);
            const int J = D == 0 ? 0 : join_same(D, e);
            g = imax(g, lq + gq - J);
        }
        pw4[p] = (wp & 0x00ffffffu) | ((uint32_t)g << 24);
        ub = imax(ub, lp + g);
    }
    return ub;
}
// aux (optional): when the verdict is W_NOT_TIGHT with main pieces only, the number of far matches (the refinement needs nothing
// else of the off-diagonal matches); 0xffffffff otherwise
// ---- the harmless test in two parts, so that the device can pool the first over a wavefront (band_diag_kernel) ----
// harmless_item: one off-diagonal match (sx, sy) of a task against the task's main pieces pl.at(i), i < r.  Returns A | T << 8:
//   A  the dp bound the main matches that end before it give (bv - q + 1; 0: none),
//   T  the largest dp + 1 that keeps it harmless: it may neither end the chain (dp < best_dp) nor be preferred by a main match
//      (dp + (sx + K) + (sy + K) - (x_p + y_p) + 1 < dp(p) at the first match p of every piece that starts after it ends).
// Both clamped to a byte: dp bounds never exceed best_dp <= 187, and A = 255 fails every T.
// harmless_step: the matches of a task in (x, y) order: dp = max(K, 1 + the largest bound so far, A); harmless iff dp < T.
template <class PL> VTXF_FN uint32_t harmless_item(const PL& pl, int r, int d, int best_dp, int sx, int sy) {
    const int q = sx + sy - d;
    const int lim = imin(sx, sy - d) - K;                          // last main row that ends before (sx, sy)
    int bv = -1000000, minH = 1000000;
    for (int i = 0; i < r; ++i) {
        const uint32_t pw = pl.at(i);
        const int pu = (int)(pw & 0xffu), pv = (int)((pw >> 8) & 0xffu), dpf = (int)((pw >> 16) & 0xffu);
        const int lm1 = pv - 5 - pu;
        const int t = imin(lim - pu, lm1);
        if (t >= 0) bv = imax(bv, dpf + t + 2 * (pu + t + K));               // V of the piece's last visible match (without d)
        const int t0 = imax(0, imax(sx + K - pu, sy + K - d - pu));          // first match of the piece that starts after s ends
        if (t0 <= lm1) minH = imin(minH, dpf + 3 * t0 + 2 * pu);
    }
    const int A = bv > -1000000 ? imin(imax(bv - q + 1, 0), 255) : 0;
    const int T = imin(imax(imin(best_dp, minH - q - 2 * K - 1), 0), 255);
    return (uint32_t)A | ((uint32_t)T << 8);
}
// the same against up to four main pieces held in registers (a piece word of 0 is no piece: it ends before it starts) — the usual task
// has one to three, and the per-match loop over the lane's LDS words was a chain of dependent round trips
VTXF_FN uint32_t harmless_item4(const uint32_t (&pw4)[4], int d, int best_dp, int sx, int sy) {
    const int q = sx + sy - d;
    const int lim = imin(sx, sy - d) - K;
    int bv = -1000000, minH = 1000000;
    VTXF_UNROLL
    for (int i = 0; i < 4; ++i) {
        const uint32_t pw = pw4[i];
        const int pu = (int)(pw & 0xffu), pv = (int)((pw >> 8) & 0xffu), dpf = (int)((pw >> 16) & 0xffu);
        const int lm1 = pv - 5 - pu;
        const int t = imin(lim - pu, lm1);
        if (t >= 0) bv = imax(bv, dpf + t + 2 * (pu + t + K));
        const int t0 = imax(0, imax(sx + K - pu, sy + K - d - pu));
        if (t0 <= lm1) minH = imin(minH, dpf + 3 * t0 + 2 * pu);
    }
    const int A = bv > -1000000 ? imin(imax(bv - q + 1, 0), 255) : 0;
    const int T = imin(imax(imin(best_dp, minH - q - 2 * K - 1), 0), 255);
    return (uint32_t)A | ((uint32_t)T << 8);
}
VTXF_FN bool harmless_step(uint32_t at, int& runmax) {
    const int dp = imax(imax(K, runmax + 1), (int)(at & 0xffu));
    runmax = imax(runmax, dp);
    return dp < (int)(at >> 8);
}
// (x, y) order of the off-diagonal matches (a lane probing its own rows produces it; pooled probes arrive in any order)
// (first: the leading entries that are in order already — the twin list's, which twin_matches appends in (x, y) order)
template <class LN> VTXF_FN void back_sort(int ns, const LN& ln, int first = 0) {
    for (int k = imax(first, 1); k < ns; ++k) {
        const uint32_t v = ln.s(k);
        int j = k - 1;
        while (j >= 0 && ln.s(j) > v) { ln.s(j + 1) = ln.s(j); --j; }
        ln.s(j + 1) = (typename LN::SType)v;
    }
}
// The dp bound of an off-diagonal match s (harmless_step): max(6, A from the main matches that end before it, 1 + the bound of a
// match it can be reached from).  First stage: "a match it can be reached from" = every earlier match of the (x, y) order — one
// running maximum, no storage, and a bound that grows by one per match: 80 chance matches of a repeat-rich window make the last
// ones look dangerous whatever they are.  Second stage (LN::TIGHT, a byte per match): only the matches that have ENDED when s
// starts (x' + K <= x; sdpkpp jumps to a match from matches that end at or before its start — the y condition is dropped: a
// superset) and its diagonal continuation partner (x - 1, y - 1).  A jump adds at most 1 (gap 0), the continuation exactly 1.
template <class LN> VTXF_FN bool back_harmless(const Front& fr, int ns, const LN& ln) {
    if constexpr (LN::TIGHT) {
        int ended = 0, j = 0;                              // max bound over the matches with x' + K <= x of the current one
        for (int k = 0; k < ns; ++k) {
            const uint32_t w = ln.s(k);
            const int sx = (int)(w >> LN::XS), sy = (int)(w & LN::YM);
            while (j < k && (int)((uint32_t)ln.s(j) >> LN::XS) + K <= sx) { ended = imax(ended, (int)ln.u(j)); ++j; }
            const uint32_t at = harmless_item(ln, fr.r, fr.d, fr.best_dp, sx, sy);
            int dp = imax(K, (int)(at & 0xffu));
            if (ended) dp = imax(dp, ended + 1);
            for (int i = k - 1; i >= 0; --i) {             // the partner (sx - 1, sy - 1): among the matches of the previous row
                const uint32_t wi = ln.s(i);
                if ((int)(wi >> LN::XS) < sx - 1) break;
                if (wi + LN::ONE == w) { dp = imax(dp, (int)ln.u(i) + 1); break; }
            }
            if (!(dp < (int)(at >> 8))) return false;
            ln.u(k) = (uint8_t)imin(dp, 255);
        }
        return true;
    } else {
        int runmax = 0;
        if (fr.r <= 4) {
            uint32_t pw4[4];
            VTXF_UNROLL
            for (int i = 0; i < 4; ++i) pw4[i] = i < fr.r ? (ln.at(i) & 0x00ffffffu) : 0u;
            // four matches per trip: their words are loaded together, the items are independent, only the steps are a chain
            for (int k0 = 0; k0 < ns; k0 += 4) {
                uint32_t w4[4], at[4];
                VTXF_UNROLL
                for (int j = 0; j < 4; ++j) w4[j] = ln.s(imin(k0 + j, ns - 1));
                VTXF_UNROLL
                for (int j = 0; j < 4; ++j) at[j] = harmless_item4(pw4, fr.d, fr.best_dp, (int)(w4[j] >> LN::XS), (int)(w4[j] & LN::YM));
                VTXF_UNROLL
                for (int j = 0; j < 4; ++j)
                    if (k0 + j < ns && !harmless_step(at[j], runmax)) return false;
            }
            return true;
        }
        for (int k = 0; k < ns; ++k) {
            const uint32_t w = ln.s(k);
            if (!harmless_step(harmless_item(ln, fr.r, fr.d, fr.best_dp, (int)(w >> LN::XS), (int)(w & LN::YM)), runmax)) return false;
        }
        return true;
    }
}
template <class LN> VTXF_FN int32_t back_rest(const Front& fr, int ns, const LN& ln, const Lane& gl, uint32_t* why, int ablate,
                                              const Refine* rf, uint32_t* aux);
template <class LN> VTXF_FN int32_t back(const Front& fr, int ns, const LN& ln, const Lane& gl, uint32_t* why, int ablate = 0,
                                         const Refine* rf = nullptr, uint32_t* aux = nullptr) {
    if (aux) *aux = 0xffffffffu;
    if (ns > LN::SMAX) { *why = W_MATCHES; return -1; }
    back_sort(ns, ln);
    if (!back_harmless(fr, ns, ln)) { *why = W_NOT_HARMLESS; return -1; }
    return back_rest(fr, ns, ln, gl, why, ablate, rf, aux);
}
// closure, run bound, verdict (the matches sorted and found harmless)
template <class LN> VTXF_FN int32_t back_rest(const Front& fr, int ns, const LN& ln, const Lane& gl, uint32_t* why, int ablate,
                                              const Refine* rf, uint32_t* aux) {
    if (aux) *aux = 0xffffffffu;
    constexpr int XS = LN::XS;
    constexpr uint32_t YM = LN::YM, ONE = LN::ONE;
    const int d = fr.d, r = fr.r;
    int far_e = 0;
    const int nc = ns;
    if (ablate == 5) { *why = W_NOT_TIGHT; return -1; }                   // (profiling aid) sort + harmless tests only
    // ---- generic set: the main pieces plus the off-diagonal pieces that may not be left out as FAR (header: condition (*)).
    //      Far matches are binned by their distance D from the hull of the generic diagonals; the cumulative count up to a
    //      bin's upper edge must stay within bound(lower edge), bound(t) = min(t, 2t - 6) — conservative for (*), which asks
    //      cnt(D' <= D) <= bound(D) at every far match.  On a violation the nearest far piece joins the generic set (the hull
    //      grows, every distance is taken again). ----
    int ng = 0;
    {
        int hull_lo = 0, hull_hi = 0;
        uint64_t used = 0;
        for (;;) {
            // cumulative counters, one byte each: D <= 5, 7, 11, 15, 23, 31, 39 (a match at D >= 40 > ns can never be in the way)
            uint64_t cum = 0;
            int near_k = -1, near_d = 1 << 20;
            for (int k0 = 0; k0 < nc; k0 += 4) {               // (four words per trip, loaded together)
                uint32_t w4[4];
                VTXF_UNROLL
                for (int j = 0; j < 4; ++j) w4[j] = ln.s(imin(k0 + j, nc - 1));
                VTXF_UNROLL
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + j;
                    if (k >= nc || ((used >> k) & 1ull)) continue;
                    const uint32_t w = w4[j];
                    const int delta = (int)(w & YM) - (int)(w >> XS) - d;
                    const int D = delta > hull_hi ? delta - hull_hi : (delta < hull_lo ? hull_lo - delta : 0);
                    if (D < near_d) { near_d = D; near_k = k; }
                    if (D < 40) {
                        const int bin = (D >= 6) + (D >= 8) + (D >= 12) + (D >= 16) + (D >= 24) + (D >= 32);
                        cum += 0x0001010101010101ull << (8 * bin);
                    }
                }
            }
            if (near_k < 0) break;                                        // nothing is far
            bool ok = near_d >= 4;                                         // bound(t) <= 0 for t <= 3
            if (ok) {
                const uint64_t lim = 0x002018100c080602ull;                // bound(4, 6, 8, 12, 16, 24, 32), one byte each
                // every byte of cum <= its byte of lim: byte-wise (0x80 + lim) - cum keeps bit 7 (no borrow crosses a byte: cum <= SM < 128)
                ok = (((lim | 0x8080808080808080ull) - cum) & 0x0080808080808080ull) == 0x0080808080808080ull;
            }
            if (ok) break;
            // the nearest far match: its piece (head = the match no other match continues into; members = its continuations)
            int k = near_k;
            for (int j = k - 1; j >= 0; --j) {
                const uint32_t wj = ln.s(j);
                if ((int)(wj >> XS) + 1 < (int)((uint32_t)ln.s(k) >> XS)) break;
                if (wj + ONE == (uint32_t)ln.s(k)) { k = j; }
            }
            const uint32_t w = ln.s(k);
            const int sx = (int)(w >> XS), sy = (int)(w & YM);
            const int delta = sy - sx - d;
            int len = 1;
            uint64_t members = 1ull << k;
            for (int j = k + 1; j < nc; ++j) {
                const uint32_t wj = ln.s(j);
                if ((int)(wj >> XS) > sx + len) break;
                if (wj == w + (uint32_t)len * ONE) { ++len; members |= 1ull << j; }
            }
            if (ng == GM || iabs(delta) > DMAX) { *why = W_GENERIC; return -1; }
            gl.at(ng) = (uint32_t)sx | ((uint32_t)(delta + 128) << 8) | ((uint32_t)(len + K - 1) << 16);
            ++ng;
            used |= members;
            hull_lo = imin(hull_lo, delta); hull_hi = imax(hull_hi, delta);
        }
        // what is left is far.  E = their k-mer matches
        far_e = nc - __builtin_popcountll(used);
    }
    if (ablate == 6) { *why = W_NOT_TIGHT; return -1 - far_e - ng; }      // (profiling aid) ... + closure
    // ---- run bound over the generic pieces: main pieces at ln[SM, SM + r), off-diagonal ones at gl[0, ng) ----
    int ub = imax(K - 1, far_e > 0 ? far_e + 5 : 0);
    {
        const int n_all = r + ng;
        auto word = [&](int i) -> uint32_t& { return i < r ? ln.at(i) : gl.at(i - r); };
        auto decode = [&](int i, uint32_t w, int& xp, int& yp, int& lp) {
            if (i < r) { xp = (int)(w & 0xffu); yp = xp + d; lp = (int)((w >> 8) & 0xffu) - xp + 1; }
            else { xp = (int)(w & 0xffu); yp = xp + d + (int)((w >> 8) & 0xffu) - 128; lp = (int)((w >> 16) & 0xffu); }
        };
        // G lives in the top byte of every piece word
        for (int i = 0; i < n_all; ++i) word(i) &= 0x00ffffffu;
        bool changed = n_all > 1;
        if (ng == 0) {
            if (LN::UB4 && r <= 4 && !rf) {
                uint32_t pw4[4];
                VTXF_UNROLL
                for (int i = 0; i < 4; ++i) pw4[i] = i < r ? ln.at(i) : 0u;
                ub = imax(ub, main_pieces_ub4(pw4, r, fr.zc));
                VTXF_UNROLL
                for (int i = 0; i < 4; ++i) if (i < r) ln.at(i) = pw4[i];
            } else {
                ub = imax(ub, main_pieces_ub(ln, r, fr.zc, d, rf, far_e));
            }
            changed = false;
        }
        uint64_t zpre = 0;                                  // byte i: mismatching bases between main pieces 0 and i
        if (changed) {
            int acc = 0;
            for (int i = 1; i < r; ++i) { acc += (int)((fr.zc >> (4 * i)) & 15u); zpre |= (uint64_t)acc << (8 * i); }
        }
        for (int pass = 0; pass < 6 && changed; ++pass) {
            changed = false;
            for (int p = 0; p < n_all; ++p) {
                const uint32_t wp = word(p);
                int xp, yp, lp;
                decode(p, wp, xp, yp, lp);
                const int g0 = (int)(wp >> 24);
                int g = g0;
                for (int q0 = 0; q0 < n_all; q0 += 4) {          // (four piece words per trip, loaded together: an LDS round trip per PAIR was most of this loop)
                  uint32_t wq4[4];
                  VTXF_UNROLL
                  for (int j = 0; j < 4; ++j) wq4[j] = word(imin(q0 + j, n_all - 1));
                  VTXF_UNROLL
                  for (int j = 0; j < 4; ++j) {
                    const int q = q0 + j;
                    if (q >= n_all || q == p) continue;
                    const uint32_t wq = wq4[j];
                    int xq, yq, lq;
                    decode(q, wq, xq, yq, lq);
                    const int gq = (int)(wq >> 24);
                    int s = imax(xq + lq - xp, yq + lq - yp);
                    s = imin(imax(s, 0), lp - 1);
                    const int t = imin(lq - 1, imin(xp - xq, yp - yq) + s - 1);
                    if (t < 0) continue;
                    const int dd = (yp - xp) - (yq - xq);
                    int J = 5 + iabs(dd);
                    if (dd == 0) {
                        const int D = xp + s - xq - t - 1;
                        // two main pieces (q before p, whole: s = 0, t = lq - 1): the mismatches between them are counted
                        J = D == 0 ? 0 : (p < r && q < r ? join_same(D, (int)((zpre >> (8 * p)) & 0xffu) - (int)((zpre >> (8 * q)) & 0xffu)) : join_free(D));
                    }
                    g = imax(g, t + 1 + gq - J - s);
                  }
                }
                if (g != g0) { word(p) = (wp & 0x00ffffffu) | ((uint32_t)g << 24); changed = true; }
            }
        }
        if (changed) { *why = W_NOT_TIGHT; return -1; }
        for (int p = 0; p < n_all; ++p) {
            const uint32_t wp = word(p);
            int xp, yp, lp;
            decode(p, wp, xp, yp, lp);
            ub = imax(ub, lp + (int)(wp >> 24));
        }
    }
    if (fr.cert != ub) { *why = W_NOT_TIGHT; if (aux && ng == 0) *aux = (uint32_t)far_e; return -1; }
    *why = W_OK;
    return fr.cert;
}

// ======== the corridor certificate (round 6; band_corridor_kernel; proof below, brute-force check in tests/test_fastcore.py) ========
// For a task whose off-diagonal matches are all harmless the reference's chain is the main diagonal's stretch and its band B the
// (2w + 1)-squares along it (band_pack: rows [ca - d - w, cb - d + w], columns [ca - w, cb + w]).  The run bound prices every join by
// a closed form; on noisy reads (clusters of errors, a few chance matches far out) it stops meeting the certificate for a fifth of
// the tasks although the banded score IS the certificate's.  This bound looks at the cells themselves, but only at those that matter:
//
//   K = the cells of B within CC diagonals of d (the corridor).  An exact affine-gap DP over K (one lane, 2 CC + 1 values per row)
//   gives the best alignment that stays in K — a LOWER bound of the banded score.  It becomes an UPPER bound with one more kind of
//   edge, the excursion: whatever a path of B does outside K is replaced by "leave K, come back later".
//
//   Claim.  Let P be any local alignment path inside B.  Cut it at its H-cells (cells where it is not inside a gap).  A maximal
//   stretch of H-cells outside K (a) is entered and left by gaps (a diagonal step keeps the diagonal) that cost at least
//   5 + (CC + 1 - |delta|) when they start / end at a K-cell of diagonal offset delta — the gap has to reach offset CC + 1 —, and
//   (b) scores at most 5 + E - g, E = the k-mer matches it covers, g = its gaps: its g + 1 gap-free pieces are runs of matching bases
//   separated by mismatches, a run of 5 + e bases covers e k-mer matches, so a piece of t runs scores <= 5 t + e - 5 (t - 1) = 5 + e,
//   and every gap between two pieces costs >= 6.  (c) The k-mer matches it covers are OFF-DIAGONAL matches more than CC diagonals out
//   (the list band_diag_kernel holds, complete for these tasks), each read row belongs to at most one of its diagonal steps, and all
//   of them start at rows between the row it left K and the row it comes back: E <= the number of rows in that range with such a
//   match (far bit).  Hence, with X_i = the best score a path can have while it is outside K having consumed i read bases:
//       X_i = max(5, X_(i-1) + far(i - 1), max over the K-cells (i, delta) of H(i, delta) - exit(delta) + 5)
//       H(i, delta) = max(0, diagonal, gap from the left, gap from above — all inside K —, X_(i-1) - exit(delta))
//   bounds every path of B from above (X_(i-1), not X_i, for the re-entry: a way out and back within one row is a single gap inside K,
//   which the DP has; a path that STARTS outside K is the 5 in X's maximum, one that ENDS outside is X itself):
//   banded <= U = max(max H, max X).  Rows of B are intervals in every column and vice versa, so a gap between two K-cells runs
//   through K-cells only: the DP misses no path of B that keeps its H-cells in K.
//
//   Exactness.  Every value carries one more bit, "no excursion edge on the way here" (value = 2 score + clean: adding even numbers
//   keeps it, a maximum prefers the clean one of two equal scores, everything that comes through X is not clean).  If U is reached by
//   a clean value, a path inside K — inside B — scores U: banded >= U, so banded = U.  (Every cell of an optimal clean path holds
//   exactly that path's partial score — a larger value there would continue to more than U — and so its clean bit survives.)
//   Otherwise the task keeps its band and takes the masked DP.  Measured on the CPU (tests/test_fastcore.py): it decides 99.7 % of the
//   tasks the run bound leaves at 8 % substitution errors, all of them at 3 %.
constexpr int CC = 8;                      // half-width of the corridor (5: decides 95 % where 8 decides 99.7 %)
constexpr int CW = 2 * CC + 1;
static_assert(CC <= W, "the corridor lies inside the band's squares");
// rows that hold an off-diagonal k-mer match more than CC diagonals from the main one (ln.s(0 .. ns): the task's complete list)
template <class LN> VTXF_FN M192 far_rows(int d, int ns, const LN& ln) {
    M192 f = m_zero();
    for (int k = 0; k < ns; ++k) {
        const uint32_t w = ln.s(k);
        const int sx = (int)(w >> LN::XS), sy = (int)(w & LN::YM);
        if (iabs(sy - sx - d) <= CC) continue;
        const uint64_t bit = 1ull << (sx & 63);
        VTXF_UNROLL
        for (int q = 0; q < NW; ++q) f.w[q] |= (sx >> 6) == q ? bit : 0ull;
    }
    return f;
}
// The banded score of the task (>= 0) when the bound above is exact, -1 otherwise.  x / yb: read and haplotype bytes (8 readable bytes
// behind each), d / ca / cb: band_pack's fields, far: far_rows().
VTXF_FN int corridor_bound(const uint8_t* x, int m, const uint8_t* yb, int n, int d, int ca, int cb, const M192& far) {
    constexpr int NEGV = -(1 << 20);       // (a value no real one reaches: 300 steps of -12 do not bring it near an overflow)
    const int r0 = imax(0, ca - d - W), r1 = imin(m, cb - d + W);
    const int j0 = imax(0, ca - W), j1 = imin(n, cb + W);
    if (r0 > r1 || j0 > j1) return -1;
    int Hp[CW], Fp[CW];
    VTXF_UNROLL
    for (int k = 0; k < CW; ++k) { Hp[k] = NEGV; Fp[k] = NEGV; }
    // the haplotype bytes of the row's cells: cell k of prefix row i compares x[i - 1] with y[i + d - CC + k - 1]; 24 bytes in three
    // words, byte 0 = cell 0, shifted by one byte per row (the new byte: cell CW - 1 of the next row)
    uint64_t yw[3] = {0, 0, 0};
    {
        const int p0 = r0 + d - CC - 1;
        for (int k = 0; k < CW; ++k) {
            const int p = p0 + k;
            const uint64_t b = (p >= 0 && p < n) ? yb[p] : 0xffull;
            yw[k >> 3] |= b << (8 * (k & 7));
        }
    }
    int X = 5, best = 1;
    uint64_t xw = 0;
    for (int i = r0; i <= r1; ++i) {
        // ---- the row's constants: the read base, the far bit of the row before, which cells exist ----
        const int ri = i - 1;                                             // read index of this row's base
        if (ri >= 0 && (i == r0 || (ri & 7) == 0)) xw = ld8(x + (ri & ~7));
        const uint32_t xc = ri >= 0 ? (uint32_t)(xw >> (8 * (ri & 7))) & 0xffu : 0x100u;
        uint32_t fbit = 0;
        if (ri >= 0) {
            uint64_t fwd = far.w[0];
            VTXF_UNROLL
            for (int q = 1; q < NW; ++q) fwd = (ri >> 6) == q ? far.w[q] : fwd;
            fbit = (uint32_t)(fwd >> (ri & 63)) & 1u;
        }
        const int X2 = 2 * X;                                             // (what an excursion is worth when this row is entered)
        int Xn = imax(5, X + (int)fbit);
        const int jb = i + d - CC;                                        // column of cell 0
        const int klo = imax(0, j0 - jb), khi = imin(CW - 1, j1 - jb);
        const uint32_t vm = klo > khi ? 0u : ((2u << khi) - 1u) & ~((1u << klo) - 1u);
        int Hc[CW], Fc[CW];
        int E = NEGV, rowx = NEGV;
        VTXF_UNROLL
        for (int k = 0; k < CW; ++k) {
            constexpr int dummy = 0; (void)dummy;
            const int exitc = 5 + CC + 1 - (k < CC ? CC - k : k - CC);    // the gap to the nearest cell outside the corridor
            const uint32_t yk = (uint32_t)(yw[k >> 3] >> (8 * (k & 7))) & 0xffu;
            int h = imax(1, X2 - 2 * exitc);                              // a fresh start (clean) or the way back from an excursion (not clean)
            h = imax(h, Hp[k] + (yk == xc ? 2 : -10));
            int f = NEGV;
            if (k + 1 < CW) { f = imax(Fp[k + 1 < CW ? k + 1 : k] - 2, Hp[k + 1 < CW ? k + 1 : k] - 12); h = imax(h, f); }
            if (k >= 1) { E = imax(E - 2, Hc[k >= 1 ? k - 1 : 0] - 12); h = imax(h, E); }
            const bool valid = (vm >> k) & 1u;
            h = valid ? h : NEGV; f = valid ? f : NEGV; E = valid ? E : NEGV;
            Hc[k] = h; Fc[k] = f;
            best = imax(best, h);
            rowx = imax(rowx, h - 2 * exitc);
        }
        Xn = imax(Xn, (rowx >> 1) + 5);
        VTXF_UNROLL
        for (int k = 0; k < CW; ++k) { Hp[k] = Hc[k]; Fp[k] = Fc[k]; }
        X = Xn;
        // ---- the window moves on by one column ----
        const int pn = i + 1 + d - CC - 1 + (CW - 1);                     // haplotype byte of cell CW - 1 of the next row
        const uint64_t nb = (pn >= 0 && pn < n) ? yb[pn] : 0xffull;
        yw[0] = (yw[0] >> 8) | (yw[1] << 56);
        yw[1] = (yw[1] >> 8) | (yw[2] << 56);
        yw[2] = (yw[2] >> 8) | (nb << (8 * ((CW - 1) & 7)));
    }
    if (2 * X >= best || !(best & 1)) return -1;                         // (an excursion that never comes back, or one on the way to the maximum)
    return best >> 1;
}

// A read that matches the haplotype base for base on its diagonal needs nothing else: cert == m (every base of the read a one of the
// mask, in band).  The FULL score is at most m (a local alignment scores at most +1 per read base), so full = m; and the banded score
// is m whatever the off-diagonal matches are: sdpkpp's dp of a match never exceeds x + K (dp = K at a chain's first match, + 1 per row
// at best: a jump over dx + dy > 0 bases loses dx + dy), so the best chain has dp = m, and only a chain of m - K + 1 continuing
// matches from row 0 on ONE diagonal reaches it — this diagonal or, by the tie rule, another one that is just as perfect; its
// staircase spans the whole read, the band holds that diagonal from end to end, and the DP scores m on it.  No probes, no bounds.
VTXF_FN bool whole_read(const Front& fr, int m) { return fr.why == W_OK && fr.cert == m; }

struct Result { int32_t score; uint32_t why; };
// all three phases on one lane (host test; device variant without pooled probes)
template <class LN> VTXF_FN Result fast_task(const uint8_t* x, int m, const Tab& tb, int n, const LN& ln, const Lane& gl, bool refine) {
    const Front fr = front(x, m, tb, n, ln);
    if (fr.why != W_OK) return Result{-1, fr.why};
    if (whole_read(fr, m)) return Result{m, W_OK};
    const int ns = probe_rows(x, tb, fr, ln);
    uint32_t why = W_OK;
    const Refine rf{x, tb.gt + tb.bytes, m, n};
    const int32_t sc = back(fr, ns, ln, gl, &why, 0, refine ? &rf : nullptr);
    return Result{sc, why};
}

// The harmless test WITHOUT the list (band_stream_kernel: tasks with more off-diagonal matches than the second stage's list holds).
// The probes deliver the matches row by row, and the bound of a match needs only (a) the maximum over the matches that have ended
// (x' + K <= x): a running maximum, fed from the maxima of the last rows (a byte per row, eight rows in one 64-bit word) as they fall
// K behind, and (b) its partner in the previous row: the (y, bound) pairs of TWO rows are all that is kept (ln.s: y | bound << 8,
// two halves of SMAX / 2 entries that swap roles).  Same bounds, same verdict as back_harmless with LN::TIGHT on a list that holds
// everything (tests/test_fastcore.py checks the two against each other).  Returns 1: every match harmless; 0: one is not;
// -1: a row with more than `cap` matches.
template <class LN> VTXF_FN int probe_harmless_stream(const uint8_t* x, const Tab& tb, const Front& fr, const LN& ln, int cap = LN::SMAX / 2) {
    constexpr int HALF = LN::SMAX / 2;
    const uint8_t* head = tb.gt + tb.head;
    const uint32_t* pb = (const uint32_t*)(tb.gt + tb.pb);
    M192 need = fr.need;
    uint64_t rowmax = 0;                  // byte r & 7: the largest bound among the matches of row r, for the rows that have not ended yet
    int folded = -1;                      // rows <= folded are in `ended`
    int ended = 0, verdict = 1;
    int cur = 0, cn = 0, pn = 0, crow = -2;     // the current row's entries: ln.s(cur * HALF + i), i < cn; the previous row's: the other half, pn
    while (m_any(need) && verdict == 1) {
        int row[4];
        uint64_t w8[4];
        uint32_t code[4], bits[4], raw[4];
        for (int t = 0; t < 4; ++t) row[t] = m_pop_lowest(need);
        for (int t = 0; t < 4; ++t) w8[t] = ld8(x + (row[t] < MAX_READ ? row[t] : 0));
        for (int t = 0; t < 4; ++t) {
            code[t] = kw_code((uint32_t)w8[t], (uint32_t)(w8[t] >> 32) & 0xffffu);
            bits[t] = pb[code[t] >> 5];
            raw[t] = ld2(head + 2u * kw_bucket(kw_mix((uint32_t)w8[t], (uint32_t)(w8[t] >> 32) & 0xffffu), tb.hmask));
        }
        for (int t = 0; t < 4 && verdict == 1; ++t) {
            if (row[t] >= MAX_READ || !((bits[t] >> (code[t] & 31u)) & 1u)) continue;       // not a k-mer of this haplotype
            const int sx = row[t];
            for (int r = imax(folded + 1, sx - K - 7); r <= sx - K; ++r) {             // rows that have ended by this one
                const int sh = 8 * (r & 7);
                ended = imax(ended, (int)((rowmax >> sh) & 0xffu));
                rowmax &= ~(0xffull << sh);
            }
            folded = imax(folded, sx - K);
            if (sx == crow + 1) { cur ^= 1; pn = cn; } else pn = 0;                    // (a row without matches leaves cn = 0)
            cn = 0; crow = sx;
            const int cb = cur * HALF, pbase = (cur ^ 1) * HALF;
            int rmax = 0;
            const uint32_t hh = kw_mix((uint32_t)w8[t], (uint32_t)(w8[t] >> 32) & 0xffffu);
            walk_bucket(tb, w8[t], hh, raw[t], [&](uint32_t yc) {
                if ((int)yc - sx == fr.d || verdict != 1) return;
                const uint32_t at = harmless_item(ln, fr.r, fr.d, fr.best_dp, sx, (int)yc);
                int dp = imax(K, (int)(at & 0xffu));
                if (ended) dp = imax(dp, ended + 1);
                for (int i = 0; i < pn; ++i) {                                         // the partner (sx - 1, yc - 1)
                    const uint32_t wi = ln.s(pbase + i);
                    if ((wi & 0xffu) + 1u == yc) { dp = imax(dp, (int)(wi >> 8) + 1); break; }
                }
                if (!(dp < (int)(at >> 8))) { verdict = 0; return; }
                if (cn == cap) { verdict = -1; return; }
                dp = imin(dp, 255);
                ln.s(cb + cn) = (uint16_t)(yc | ((uint32_t)dp << 8));
                ++cn;
                rmax = imax(rmax, dp);
            });
            rowmax |= (uint64_t)rmax << (8 * (sx & 7));
        }
    }
    return verdict;
}

// The SECOND stage on one lane (band_diag2_kernel; host test): a task the first stage left because its off-diagonal matches did not
// fit the lane's list (or did not pass its coarse harmless test).  Verdict: T2_SCORE — cert == ub, the score; T2_TIGHT — every
// off-diagonal match is harmless, so the reference's chain lies on the main diagonal and the band is band_pack(fr)'s one diagonal
// stretch (masked DP, no sweep); T2_SWEEP — neither (no diagonal, a match that may matter); T2_STREAM — more matches than the list
// holds: fast_task2_stream (band_stream_kernel) runs the harmless test alone over a two-row window.
enum T2Verdict : uint32_t { T2_SCORE = 0, T2_TIGHT = 1, T2_SWEEP = 2, T2_STREAM = 3 };
struct Result2 { uint32_t verdict; int32_t score; uint32_t pack; uint32_t why; };
VTXF_FN Result2 fast_task2_list(const uint8_t* x, int m, const Tab& tb, int n, const LaneS2& ln, const Lane& gl) {
    const Front fr = front(x, m, tb, n, ln);
    if (fr.why != W_OK) return Result2{T2_SWEEP, -1, 0u, fr.why};
    if (whole_read(fr, m)) return Result2{T2_SCORE, m, 0u, W_OK};                 // (band_diag_kernel decides these itself: a list of band_run_kernel's overflows may hold one)
    const int ns = probe_rows(x, tb, fr, ln);
    if (ns > LaneS2::SMAX) return Result2{T2_STREAM, -1, (uint32_t)fr.d, W_MATCHES};          // (pack: the diagonal, for fast_task2_stream)
    back_sort(ns, ln);
    if (!back_harmless(fr, ns, ln)) return Result2{T2_SWEEP, -1, 0u, W_NOT_HARMLESS};
    uint32_t why = W_GENERIC;
    int32_t sc = -1;
    if (ns <= 64) sc = back_rest(fr, ns, ln, gl, &why, 0, nullptr, nullptr);      // (the closure's bit set holds 64 matches)
    if (sc >= 0) return Result2{T2_SCORE, sc, 0u, W_OK};
    return Result2{T2_TIGHT, fr.cert, band_pack(fr), why};
}
// ... and what follows for T2_STREAM: front again on the diagonal d the list stage found (the same pieces and certificate, without
// the search for the diagonal), then the probes with the harmless test on the fly.  T2_TIGHT or T2_SWEEP (a match that may matter,
// or a row with more matches than half the window).
template <class LN> VTXF_FN Result2 fast_task2_stream(const uint8_t* x, int m, const Tab& tb, int n, const LN& wl, int d) {
    if (m < K || n < K || m > MAX_READ || d < -(m - K) || d > n - K) return Result2{T2_SWEEP, -1, 0u, W_SHAPE};   // (not what the list stage hands over)
    const Front fr = front_rest(x, m, tb, n, wl, d, diag_mask(read_words(x, m), x, m, tb, n, d));
    if (fr.why != W_OK) return Result2{T2_SWEEP, -1, 0u, fr.why};                 // (the list stage's front accepted this task)
    const int v = probe_harmless_stream(x, tb, fr, wl);
    if (v == 1) return Result2{T2_TIGHT, fr.cert, band_pack(fr), W_MATCHES};
    return Result2{T2_SWEEP, -1, 0u, v == 0 ? W_NOT_HARMLESS : W_MATCHES};
}
// both, one after the other (host tests; the device runs them as two kernels)
VTXF_FN Result2 fast_task2(const uint8_t* x, int m, const Tab& tb, int n, const LaneS2& ln, const Lane& gl) {
    const Result2 r = fast_task2_list(x, m, tb, n, ln, gl);
    if (r.verdict != T2_STREAM) return r;
    return fast_task2_stream(x, m, tb, n, LaneW{ln.base, ln.stride, ln.sb, ln.sstride, ln.ub, ln.ustride}, (int)r.pack);
}

}  // namespace vtxf
#endif
