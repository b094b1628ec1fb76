/*
 * vtx_certify.c — CPU ORACLE, TEST INFRASTRUCTURE ONLY (see vtx_oracle.h).
 *
 * CPU restatement of the two bounds the device uses to decide the banded score of an
 * alignment WITHOUT running a DP (vartrix_amd/csrc/vtx_band.hip, "certificate"):
 *
 *   lower bound  vtxo_chain_cert : local-alignment score of the anchor staircase of the
 *                sdpkpp chain (lazy extensions included).  The staircase lies inside the
 *                band bio's banded aligner builds (oracle/vtx_oracle.c:vtxo_band_create),
 *                so cert <= banded <= full.
 *   upper bound  vtxo_runs_ub / vtxo_runs_ub_exact : bound on the FULL-matrix local score
 *                from the maximal exact-match runs of length >= k ("pieces" — exactly the
 *                diagonal runs of k-mer matches of find_kmer_matches).
 *
 * If cert == ub then cert == banded == full and the task needs no DP.
 *
 * Proof of the upper bound (scoring +1 / -5, gap of length L costs 5 + L; src/main.rs:35-38):
 *   Write an alignment as maximal runs of match columns separated by EVENTS (a mismatch
 *   column: cost 5; a gap of length L: cost 5 + L).  Call a run LONG if it has >= k = 6
 *   columns; a long run is a sub-run of a piece.  Between two consecutive long runs there
 *   are e >= 1 events and e - 1 short runs (<= 5 columns each, possibly empty), so that
 *   stretch nets at most 5 (e - 1) - 5 e - sum(L) = -5 - sum(L), and sum(L) >= |d' - d|
 *   (difference of the two diagonals).  Before the first long run there are as many short
 *   runs as events (net <= 0), likewise after the last one; an alignment with no long run
 *   scores <= 5.  Hence
 *       full <= max(5, max over chains of sub-runs of pieces of
 *                       sum(len) - sum over joins of J),          J >= 5 + |d' - d|.
 *   Same diagonal, D >= 1 positions between the two runs.  Without gaps the stretch IS the diagonal: it costs exactly
 *   6 e - D, e = the mismatching ones among the D positions (the bases are at hand), and e >= ceil((D + 5) / 6) in any
 *   case (short runs <= 5): J_free(D) = 6 ceil((D + 5) / 6) - D.  With gaps: g >= 2 gap events whose insertions and
 *   deletions both total G >= ceil(g / 2), so D - G diagonal columns, mm of them mismatches; the g + mm events separate
 *   at most g + mm - 1 short runs, D - G - mm <= 5 (g + mm - 1); cost = 5 g + 2 G + 5 mm - (D - G - mm).  Minimised over
 *   (g, G, mm) (tests/test_certify.py does it by brute force):
 *       J_gap(1) = 12,   J_gap(D) = {7, 9, 11, 10, 9, 8}[D mod 6]  (>= J_free(D), and J_gap(D + 1) >= J_gap(D) - 1),
 *       J_same(D, e) = min(6 e - D, J_gap(D));      J_same(D) = J_free(D) when the bases are not looked at.
 *   (Round 2 used max(7, 13 - D) for the gapped path: it forgot that 5 (g - 1) matches cannot span a long stretch, and
 *   took e at its minimum: at 3 % substitution errors that made 10.5 % of the tasks hard, this form 6.0 %.)
 *   Refinement (vtxo_set_corridor(2); on the device: band_refine_kernel / vtx_fast_core.h corridor_cost).  Where 6 e - D >
 *   J_gap(D) — three or more errors within a few bases — J_gap is the price of a stretch over PERFECTLY matching neighbour
 *   diagonals.  A stretch either stays within kc diagonals of the runs' diagonal: then it costs at least the optimum of an affine
 *   DP without a floor over that corridor, with the real bases, from a base of the first run to a base of the second, 1 per base
 *   of a run given up (up to mu = 6 e - D - 8 bases each side: a stretch that gives up more costs >= 7 + mu + 1 >= 6 e - D) —
 *   or it reaches a diagonal >= kc + 1 away and comes back: gaps of total length G >= kc + 1 per direction, priced by J_gap
 *   restricted to G >= kc + 1 (same brute force; for kc = 2: 16, 15, 14, 13, 12, 11, 13, 14, ... >= 11).  The join costs the
 *   smaller of the two.  Both forms keep J(D + 1) >= J(D) - 1, so the piece-level bound may still join at the latest exit and the
 *   earliest entry.  At 3 % substitution errors: 6.0 -> 2.4 % of the tasks with cert != ub.
 *   Any designation of further (short) runs as chain members keeps the inequality, so the
 *   maximisation may range over ALL sub-runs of pieces (length >= 1).
 */
#include "vtx_oracle.h"
#include "../include/vtx_band_semantics.h"

#include <stdlib.h>
#include <string.h>

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* ---- lower bound: the staircase walk of band_task / band_finish (vtx_band.hip) ---- */
typedef struct { int s, gap, best, dir; } walk_t;
static void walk_diag(walk_t* w, int match) {
    int v = (w->s > w->gap ? w->s : w->gap) + (match ? 1 : -5);
    w->s = v > 0 ? v : 0; w->gap = -100000; w->dir = 0;
    if (w->s > w->best) w->best = w->s;
}
static void walk_gap(walk_t* w, int dir) {
    int open = w->s - 6;
    int ext = (w->dir == dir) ? w->gap - 1 : -100000;
    w->gap = open > ext ? open : ext; w->dir = dir;
    w->s = w->gap > 0 ? w->gap : 0;
}

/* Returns the certificate, or -1 when there is no k-mer match (Band::full_matrix). */
int32_t vtxo_chain_cert(const uint8_t* x, int m, const uint8_t* y, int n, int k) {
    const int band_w = 20;   /* W of src/main.rs:34 */
    uint32_t* mt = NULL;
    int64_t M = vtxo_find_kmer_matches(x, m, y, n, k, &mt);
    if (M == 0) { free(mt); return -1; }
    int64_t* path = (int64_t*)malloc(sizeof(int64_t) * (size_t)M);
    int64_t L = vtxo_sdpkpp(mt, M, k, 1, -5, -1, path, NULL);
    const int lazy = VTX_BAND_LAZY_EXT(k);
    const int fx = (int)mt[2 * path[0]], fy = (int)mt[2 * path[0] + 1];
    int d0 = imin(imin(fx, fy), lazy);
    int r = fx - d0, c = fy - d0;
    walk_t w = {0, -100000, 0, 0};
    /* the (2w+1)-square around the first anchor is in band: the diagonal may be walked from up to w cells before it */
    {
        const int t0 = imin(imin(r, c), band_w);
        int rr = r - t0, cc = c - t0;
        for (int i = 0; i < t0; ++i) { ++rr; ++cc; walk_diag(&w, x[rr - 1] == y[cc - 1]); }
    }
    for (int i = 0; i < d0; ++i) { ++r; ++c; walk_diag(&w, x[r - 1] == y[c - 1]); }
    for (int64_t t = 0; t < L; ++t) {
        const int px = (int)mt[2 * path[t]], py = (int)mt[2 * path[t] + 1];
        int dr = px - r, dc = py - c;
        int dg = imin(dr, dc);
        for (int i = 0; i < dg; ++i) { ++r; ++c; walk_diag(&w, x[r - 1] == y[c - 1]); }
        dr = px - r; dc = py - c;
        for (int i = 0; i < dr; ++i) { ++r; walk_gap(&w, 1); }
        for (int i = 0; i < dc; ++i) { ++c; walk_gap(&w, 2); }
        int steps = VTX_BAND_KMER_LAST_ANCHOR(k);
        if (t + 1 < L) {
            const int qx = (int)mt[2 * path[t + 1]], qy = (int)mt[2 * path[t + 1] + 1];
            if (qx == px + 1 && qy == py + 1) steps = 1;
        }
        for (int i = 0; i < steps; ++i) { ++r; ++c; walk_diag(&w, x[r - 1] == y[c - 1]); }
    }
    int d1 = imin(imin(m - r, n - c), lazy);
    for (int i = 0; i < d1; ++i) { ++r; ++c; walk_diag(&w, x[r - 1] == y[c - 1]); }
    /* ... and up to w cells past the last anchor */
    {
        const int t1 = imin(imin(m - r, n - c), band_w);
        for (int i = 0; i < t1; ++i) { ++r; ++c; walk_diag(&w, x[r - 1] == y[c - 1]); }
    }
    free(path); free(mt);
    return w.best;
}

/* ---- pieces: maximal diagonal runs of matching bases of length >= k ---- */
typedef struct { int xs, ys, len; } piece_t;

static int64_t find_pieces(const uint8_t* x, int m, const uint8_t* y, int n, int k, piece_t** out) {
    uint32_t* mt = NULL;
    int64_t M = vtxo_find_kmer_matches(x, m, y, n, k, &mt);
    piece_t* ps = (piece_t*)malloc(sizeof(piece_t) * (size_t)(M > 0 ? M : 1));
    int64_t np = 0;
    /* a k-mer match (i, j) opens a piece iff (i-1, j-1) is not a k-mer match, i.e. i == 0 or j == 0 or
       x[i-1] != y[j-1] (then the run of matching bases starts at i) */
    for (int64_t t = 0; t < M; ++t) {
        const int i = (int)mt[2 * t], j = (int)mt[2 * t + 1];
        if (i > 0 && j > 0 && x[i - 1] == y[j - 1]) continue;
        int len = k;
        while (i + len < m && j + len < n && x[i + len] == y[j + len]) ++len;
        ps[np].xs = i; ps[np].ys = j; ps[np].len = len; ++np;
    }
    free(mt);
    *out = ps;
    return np;
}

int vtxo_join_same(int D);
int vtxo_join_gap(int D);
static int g_join_exact = 1;     /* 0: the closed form J_free(D) only (what a caller without the bases uses) */
void vtxo_set_join_exact(int on) { g_join_exact = on; }
/* J_gap restricted to stretches whose gaps total G >= 3 / 4 / 5 per direction (they reach a diagonal >= G away and come back):
   brute force over (g, G, mm) as for J_gap, tests/test_certify.py */
static int join_gap_min_g(int D, int g0) {
    int best = 1 << 20;
    for (int g = 2; g <= 2 * D + 4; ++g)
        for (int G = (g + 1) / 2 > g0 ? (g + 1) / 2 : g0; G <= D; ++G)
            for (int mm = 0; mm <= D - G; ++mm) {
                const int matches = D - G - mm;
                if (matches <= 5 * (g + mm - 1)) { const int c = 5 * g + 2 * G + 5 * mm - matches; if (c < best) best = c; break; }
            }
    return best;
}
/* Exact cost of the cheapest stretch between base (xb, yb) of one run and the base D + 1 further on the same diagonal that
   stays within kc diagonals of it, leaving the first run up to mu_a bases early / entering the second up to mu_b bases late at
   1 per base (affine DP without a floor over the corridor; the runs' own bases are part of the corridor).                    */
static int corridor_cost(const uint8_t* x, int m, const uint8_t* y, int n, int xb, int yb, int D, int kc, int mu_a, int mu_b) {
    const int NEG = -100000;
    const int d = yb - xb;
    const int r0 = xb + 1 - mu_a, r1 = xb + D + 1 + mu_b + 1;       /* prefix cells (i, j): rows r0 .. r1 */
    const int W = 2 * kc + 1;
    int Hp[16], Fp[16], Hc[16], Fc[16];
    for (int k = 0; k < W; ++k) { Hp[k] = NEG; Fp[k] = NEG; }
    Hp[kc] = -mu_a;                                                     /* cell (r0, r0 + d) */
    /* row r0: horizontal gaps from the seed */
    { int E = NEG; for (int k = kc + 1; k < W; ++k) { E = (E - 1 > Hp[k - 1] - 6) ? E - 1 : Hp[k - 1] - 6; Hp[k] = E; } }
    for (int i = r0 + 1; i <= r1; ++i) {
        int E = NEG;
        for (int k = 0; k < W; ++k) {
            const int j = i + d + (k - kc);
            int h = NEG, f = NEG;
            if (j >= 0 && j <= n && i >= 0 && i <= m) {
                /* diagonal: from (i - 1, j - 1): same k */
                if (i >= 1 && j >= 1 && Hp[k] > NEG / 2) h = Hp[k] + (x[i - 1] == y[j - 1] ? 1 : -5);
                /* vertical (consumes x): from (i - 1, j): diagonal index k + 1 in the previous row */
                if (k + 1 < W) { const int a = Fp[k + 1] - 1, b = Hp[k + 1] - 6; f = a > b ? a : b; if (f < NEG / 2) f = NEG; }
                /* horizontal (consumes y): from (i, j - 1): k - 1 in this row */
                if (k >= 1) { const int a = E - 1, b = Hc[k - 1] - 6; E = a > b ? a : b; if (E < NEG / 2) E = NEG; } else E = NEG;
                if (f > h) h = f;
                if (E > h) h = E;
            } else E = NEG;
            Hc[k] = h; Fc[k] = f;
        }
        for (int k = 0; k < W; ++k) { Hp[k] = Hc[k]; Fp[k] = Fc[k]; }
    }
    /* cell (r1, r1 + d): the second run's bases 0 .. mu_b consumed */
    return Hp[kc] <= NEG / 2 ? (1 << 20) : -(Hp[kc] - (mu_b + 1));
}
static int g_corridor = 0;       /* experiment: diagonals each side of the corridor (0: off) */
void vtxo_set_corridor(int kc) { g_corridor = kc; }
/* avail_a: bases of the first run in front of (xb, yb) that the chain may give up (leave early); avail_b: likewise behind the
   second run's entry base */
static int join_same_e(const uint8_t* x, int m, const uint8_t* y, int n, int xb, int yb, int D, int avail_a, int avail_b) {
    /* bases (xb+1 .. xb+D) on the diagonal of (xb, yb) */
    if (!g_join_exact) return vtxo_join_same(D);
    int e = 0;
    for (int i = 1; i <= D; ++i) e += x[xb + i] != y[yb + i];
    const int free_cost = 6 * e - D, jg = vtxo_join_gap(D);
    if (free_cost <= jg || !g_corridor) return imin(free_cost, jg);
    /* the gap-free stretch costs more than a hypothetical one with gaps: price the real ones.  Inside the corridor exactly
       (a stretch that gives up more than mu bases of a run costs >= 7 + mu + 1 >= free_cost); outside it the gaps total >= kc + 1 */
    const int mu = imax(0, free_cost - 8);
    const int inside = corridor_cost(x, m, y, n, xb, yb, D, g_corridor, imin(mu, avail_a), imin(mu, avail_b));
    return imin(inside, join_gap_min_g(D, g_corridor + 1));
}
int vtxo_join_gap(int D) {
    static const int jg[6] = {7, 9, 11, 10, 9, 8};
    return D == 1 ? 12 : jg[D % 6];
}
int vtxo_join_same(int D) {
    return 6 * ((D + 10) / 6) - D;             /* 6 ceil((D + 5) / 6) - D  (<= J_gap(D)) */
}

/* Exact form of the bound: DP over every matched base of every piece (O(N^2)).               */
int32_t vtxo_runs_ub_exact(const uint8_t* x, int m, const uint8_t* y, int n, int k) {
    piece_t* ps = NULL;
    int64_t np = find_pieces(x, m, y, n, k, &ps);
    int total = 0;
    for (int64_t p = 0; p < np; ++p) total += ps[p].len;
    int* nx = (int*)malloc(sizeof(int) * (size_t)(total + 1) * 4);
    int* ny = nx + (total + 1); int* nd = ny + (total + 1); int* ub = nd + (total + 1);
    int N = 0;
    /* nodes in order of x (then anything): bucket by x */
    for (int xx = 0; xx < m; ++xx)
        for (int64_t p = 0; p < np; ++p)
            if (xx >= ps[p].xs && xx < ps[p].xs + ps[p].len) {
                nx[N] = xx; ny[N] = ps[p].ys + (xx - ps[p].xs); nd[N] = ny[N] - xx; ++N;
            }
    int best = 5;
    for (int a = 0; a < N; ++a) {
        int v = 0;
        for (int b = 0; b < a; ++b) {
            if (nx[b] >= nx[a] || ny[b] >= ny[a]) continue;
            int cand;
            if (nd[b] == nd[a]) {
                const int D = nx[a] - nx[b] - 1;
                cand = D == 0 ? ub[b] : ub[b] - join_same_e(x, m, y, n, nx[b], ny[b], D, 0, 0);
            } else {
                cand = ub[b] - 5 - abs(nd[a] - nd[b]);
            }
            if (cand > v) v = cand;
        }
        ub[a] = v + 1;
        if (ub[a] > best) best = ub[a];
    }
    free(nx); free(ps);
    return best;
}

/* Piece-level form (what the device computes): one number G per piece = best value a chain can bring
   into the piece minus the entry offset; a predecessor is used up to the last base that precedes the
   entry point in both coordinates; fixpoint over ordered pairs.  >= the exact form (it ignores that a
   piece must be entered before it is left).  *passes_out = passes until stable.                       */
int32_t vtxo_runs_ub(const uint8_t* x, int m, const uint8_t* y, int n, int k, int* passes_out, int* npieces_out) {
    piece_t* ps = NULL;
    int64_t np = find_pieces(x, m, y, n, k, &ps);
    int* G = (int*)calloc((size_t)(np > 0 ? np : 1), sizeof(int));
    int passes = 0, changed = 1;
    while (changed && passes < 64) {
        changed = 0; ++passes;
        for (int64_t p = 0; p < np; ++p) {
            for (int64_t q = 0; q < np; ++q) {
                if (q == p) continue;
                const int xeq = ps[q].xs + ps[q].len, yeq = ps[q].ys + ps[q].len;
                int s = imax(xeq - ps[p].xs, yeq - ps[p].ys);
                s = imax(0, imin(s, ps[p].len - 1));
                int t = imin(ps[q].len - 1, imin(ps[p].xs + s - 1 - ps[q].xs, ps[p].ys + s - 1 - ps[q].ys));
                if (t < 0) continue;
                const int dq = ps[q].ys - ps[q].xs, dp = ps[p].ys - ps[p].xs;
                int J;
                if (dq == dp) {
                    const int D = (ps[p].xs + s) - (ps[q].xs + t) - 1;
                    J = D == 0 ? 0 : join_same_e(x, m, y, n, ps[q].xs + t, ps[q].ys + t, D, t, ps[p].len - 1 - s);
                } else {
                    J = 5 + abs(dp - dq);
                }
                const int cand = t + 1 + G[q] - J - s;
                if (cand > G[p]) { G[p] = cand; changed = 1; }
            }
        }
    }
    int best = 5;
    for (int64_t p = 0; p < np; ++p) best = imax(best, ps[p].len + G[p]);
    if (passes_out) *passes_out = passes;
    if (npieces_out) *npieces_out = (int)np;
    free(G); free(ps);
    return best;
}

/* Per task (2 * record + hap) of a packed batch: full, banded, cert, ub_exact (optional), ub, passes, pieces. */
int vtxo_batch_certify(const vtx_batch* b, const vtx_config* cfg, int32_t* full, int32_t* banded, int32_t* cert,
                       int32_t* ub_exact, int32_t* ub, int32_t* passes, int32_t* npieces, int threads) {
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
#endif
    for (int64_t l = 0; l < (int64_t)b->n_loci; ++l) {
        const vtx_locus* L = &b->loci[l];
        for (uint32_t r = L->rec_begin; r < L->rec_begin + L->rec_count; ++r) {
            const vtx_record* R = &b->records[r];
            const uint8_t* seq = b->read_arena + R->read_off;
            const int m = (int)R->read_len;
            for (int h = 0; h < 2; ++h) {
                const uint8_t* hp = b->hap_arena + (h ? L->alt_off : L->ref_off);
                const int n = (int)(h ? L->alt_len : L->ref_len);
                const size_t t = 2 * (size_t)r + (size_t)h;
                if (full) full[t] = vtxo_sw_full(seq, m, hp, n, cfg->match_score, cfg->mismatch_score, cfg->gap_open, cfg->gap_extend);
                if (banded) banded[t] = vtxo_sw_banded(seq, m, hp, n, cfg->match_score, cfg->mismatch_score, cfg->gap_open,
                                                       cfg->gap_extend, cfg->kmer_k, cfg->band_w);
                if (cert) cert[t] = vtxo_chain_cert(seq, m, hp, n, cfg->kmer_k);
                if (ub_exact) ub_exact[t] = vtxo_runs_ub_exact(seq, m, hp, n, cfg->kmer_k);
                int ps = 0, np = 0;
                if (ub) ub[t] = vtxo_runs_ub(seq, m, hp, n, cfg->kmer_k, &ps, &np);
                if (passes) passes[t] = ps;
                if (npieces) npieces[t] = np;
            }
        }
    }
    return 0;
}
