// vtx_band_trim.h — the run bound of a one-diagonal task restricted to its BAND: an upper bound of the BANDED score, not of the full one.
//
// STATUS: included by vtx_band.hip; band_refine_kernel runs it behind its `ub == cert` test (round 6; stage VTX_STAGE_BAND_CERT).
// Host harness and tests: tests/fastcore, tests/test_fastcore.py::test_band_trimmed_bound_*, tools/band_trim_stress.py
// (profiles/r05_band_trim_cpu.txt: 2.0 M tasks against the oracle).
//
// Why.  On noisy reads (3 % / 8 % substitution errors) the tasks that leave band_diag_kernel / band_refine_kernel WITH a certificate
// have cert == banded for 95 % — and banded < full for 27 % / 51 %: sdpkpp drops short end pieces (a jump over D bases costs 2 D), the
// band ends 20 columns behind the chain, and the pieces beyond it count for the full-matrix score only.  The run bound (ub >= full)
// can never meet the certificate there, so the task takes the masked DP (10 ns; 151 of the 326 ms a step takes at 8 % errors).
// Measured on the host build of the kernel logic with the corridor refinement on (9 144 tasks each): of the tasks still undecided
// 71 % (8 % errors) have banded < full, and the refined bound EQUALS the full score for most of them — it is tight, against the
// wrong score.
//
// What.  With every off-diagonal match harmless the reference's band is band_pack(fr): DP columns ca - W .. cb + W around ONE
// diagonal.  A path inside the band visits in-band cells only, so the run bound holds for the banded score with the exact-match
// runs of the in-band cells: on the main diagonal the read bases i with ca - W <= i + 1 + d <= cb + W (DP cell (i + 1, i + 1 + d)),
// i.e. the main pieces TRIMMED to rows [ca - W - 1 - d, cb + W - 1 - d].  A trimmed piece is a shorter run (counting it as a piece
// whatever its length only enlarges the bound); the join costs are lower bounds of what it costs to get from one run to the
// next through ANY cells, so they stay lower bounds when cells are taken away; far matches: a chain of far pieces is worth
// <= E + 5 as before.  Hence  banded <= max(K - 1, E_far + 5, main_pieces_ub over the trimmed pieces),  and equality with the
// certificate (a path inside the band: cert <= banded) decides the task.
// Only tasks with main pieces alone (no generic off-diagonal piece: back_rest's aux) take this bound — the tasks
// band_refine_kernel sees.  Their score is then NOT the full-matrix score: a stage of its own in the audit (banded != full allowed).
//
// Measured (host, refinement on): decides 58 % (8 % errors) and 53 % (3 %) of what the refinement leaves, every decided score the
// oracle's banded score (tests).  Projected on the device: the one-diagonal DP of 8 % errors 151 -> ~65 ms per step.
//
// On the device: band_refine_kernel holds everything in its record — band_pack (d, ca, cb), the certificate, the far matches, the
// pieces and the mismatch nibbles — and calls main_pieces_ub_band for batches whose haplotypes have <= 255 bases (ca / cb are bytes
// of the pack).  tests/audit_util.py: "banded != full => a DP stage or this one".
#ifndef VTX_BAND_TRIM_H
#define VTX_BAND_TRIM_H
#include "vtx_fast_core.h"

namespace vtxf {

// rows of the read whose main-diagonal cell lies inside the band of a one-diagonal task (band_pack(fr))
VTXF_FN void band_rows(const Front& fr, int m, int& lo, int& hi) {
    lo = imax(0, fr.ca - W - 1 - fr.d);
    hi = imin(m - 1, fr.cb + W - 1 - fr.d);
}

// main_pieces_ub (vtx_fast_core.h) over the pieces trimmed to rows [lo, hi]: pieces outside do not exist, a piece cut at its low
// end has no predecessor (everything before it is out of band), a piece cut at its high end is nobody's predecessor.
template <class PL> VTXF_FN int main_pieces_ub_band(const PL& pl, int r, uint32_t zc, int d, const Refine* rf, int lo, int hi, int far_e) {
    int ub = 0;
    for (int p = 0; p < r; ++p) {
        const uint32_t wp = pl.at(p);
        const int xp0 = (int)(wp & 0xffu), xl0 = (int)((wp >> 8) & 0xffu);
        const int xp = imax(xp0, lo), xl = imin(xl0, hi);
        if (xl < xp) { pl.at(p) = wp & 0x00ffffffu; continue; }       // out of band
        const int lp = xl - xp + 1;
        int g = 0, e = 0;
        if (xp == xp0) {
            for (int q = p - 1; q >= 0; --q) {
                const uint32_t wq = pl.at(q);
                const int xq0 = (int)(wq & 0xffu), ql0 = (int)((wq >> 8) & 0xffu), gq = (int)(wq >> 24);
                if (ql0 < lo) break;                                   // q and everything before it: out of band
                const int xq = imax(xq0, lo), lq = ql0 - xq + 1;       // (q ends below p, which starts inside the band: ql0 <= hi)
                const int D = xp - (ql0 + 1);
                e += (int)((zc >> (4 * (q + 1))) & 15u);
                int J = D == 0 ? 0 : join_same(D, e);
                if (rf && D > 0 && J < 6 * e - D && e < 15) {
                    const int mu = imax(0, 6 * e - D - 8);
                    const int inside = corridor_cost(rf->x, rf->m, rf->yb, rf->n, ql0, d, D, imin(mu, lq - 1), imin(mu, lp - 1));
                    J = imin(6 * e - D, imin(inside, join_gap3_far(D, far_e)));
                }
                g = imax(g, lq + gq - J);
            }
        }
        pl.at(p) = (wp & 0x00ffffffu) | ((uint32_t)g << 24);
        ub = imax(ub, lp + g);
    }
    return ub;
}

// Both bounds in ONE pass over the ordered pairs (band_refine_kernel, round 6): the refined bound of the full score (main_pieces_ub) and
// the bound of the banded score (main_pieces_ub_band) price the same joins — the corridor DP of a join, which is where the time goes,
// is run once and only repeated for the (at most two) pieces the band actually cuts, where the bases a run may give up differ.
// G of the full bound in the top byte of a piece word as before, G of the band bound in bits 16 - 23 (the refine record has no use for
// the dp byte kept there).  Returns the full bound, *ub_band the other.
template <class PL> VTXF_FN int main_pieces_ub_both(const PL& pl, int r, uint32_t zc, int d, const Refine* rf, int far_e, int lo, int hi, int* ub_band) {
    int ub = 0, ubb = 0;
    for (int p = 0; p < r; ++p) {
        const uint32_t wp = pl.at(p);
        const int xp0 = (int)(wp & 0xffu), xl0 = (int)((wp >> 8) & 0xffu), lp0 = xl0 - xp0 + 1;
        const int xpb = imax(xp0, lo), xlb = imin(xl0, hi), lpb = xlb - xpb + 1;
        const bool in_band = xlb >= xpb, cut_lo = xpb != xp0;
        int g = 0, gb = 0, e = 0;
        bool band_open = in_band && !cut_lo;                             // (a piece cut at its low end has no in-band predecessor)
        for (int q = p - 1; q >= 0; --q) {
            const uint32_t wq = pl.at(q);
            const int xq0 = (int)(wq & 0xffu), ql0 = (int)((wq >> 8) & 0xffu), lq0 = ql0 - xq0 + 1;
            const int gq = (int)(wq >> 24), gqb = (int)((wq >> 16) & 0xffu);
            const int D = xp0 - (ql0 + 1);
            e += (int)((zc >> (4 * (q + 1))) & 15u);
            const int Js = D == 0 ? 0 : join_same(D, e);
            const bool refine = rf && D > 0 && Js < 6 * e - D && e < 15;
            const int mu = imax(0, 6 * e - D - 8);
            int J = Js;
            if (refine) {
                const int inside = corridor_cost(rf->x, rf->m, rf->yb, rf->n, ql0, d, D, imin(mu, lq0 - 1), imin(mu, lp0 - 1));
                J = imin(6 * e - D, imin(inside, join_gap3_far(D, far_e)));
            }
            g = imax(g, lq0 + gq - J);
            if (band_open) {
                if (ql0 < lo) band_open = false;                         // q and everything before it: out of band
                else {
                    const int lqb = ql0 - imax(xq0, lo) + 1;
                    int Jb = J;
                    if (refine && (imin(mu, lqb - 1) != imin(mu, lq0 - 1) || imin(mu, lpb - 1) != imin(mu, lp0 - 1))) {
                        const int inside = corridor_cost(rf->x, rf->m, rf->yb, rf->n, ql0, d, D, imin(mu, lqb - 1), imin(mu, lpb - 1));
                        Jb = imin(6 * e - D, imin(inside, join_gap3_far(D, far_e)));
                    }
                    gb = imax(gb, lqb + gqb - Jb);
                }
            }
        }
        pl.at(p) = (wp & 0x0000ffffu) | ((uint32_t)(in_band ? gb : 0) << 16) | ((uint32_t)g << 24);
        ub = imax(ub, lp0 + g);
        if (in_band) ubb = imax(ubb, lpb + gb);
    }
    *ub_band = ubb;
    return ub;
}

// After back_rest left a task W_NOT_TIGHT with main pieces only (aux = its far k-mer matches): the bound of the BANDED score.
// Returns the score (= the certificate) or -1.
template <class LN> VTXF_FN int32_t band_trim_verdict(const Front& fr, int m, const LN& ln, uint32_t far_e, const Refine* rf) {
    int lo, hi;
    band_rows(fr, m, lo, hi);
    int ub = imax(K - 1, far_e > 0 ? (int)far_e + 5 : 0);
    ub = imax(ub, main_pieces_ub_band(ln, fr.r, fr.zc, fr.d, rf, lo, hi, (int)far_e));
    return ub == fr.cert ? fr.cert : -1;
}

}  // namespace vtxf
#endif
