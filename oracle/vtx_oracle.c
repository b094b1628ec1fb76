/*
 * vtx_oracle.c — CPU ORACLE, TEST INFRASTRUCTURE ONLY (see vtx_oracle.h).
 *
 * Plain-C restatement of the reference's hot path.  Every function cites the
 * reference lines (src/main.rs of 10XGenomics/vartrix v1.1.22) or the module of
 * crate bio 0.30.0 whose published algorithm it restates.  Written for
 * clarity first; it is also the "port" CPU baseline bench.py times.
 */
#include "vtx_oracle.h"
#include "../include/vtx_band_semantics.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* Per-thread scratch slabs (grown on demand, never shrunk): the per-read aligner of the reference allocates
 * its DP vectors once per Aligner (banded.rs: `with_capacity`) — one malloc per call here would measure the
 * allocator, not the algorithm, when this file is timed as the CPU baseline.                                */
#define VTXO_SLABS 8
static _Thread_local void* t_slab[VTXO_SLABS];
static _Thread_local size_t t_slab_cap[VTXO_SLABS];
static void* slab(int k, size_t bytes) {
    if (t_slab_cap[k] < bytes) {
        free(t_slab[k]);
        t_slab_cap[k] = bytes + bytes / 2 + 64;
        t_slab[k] = malloc(t_slab_cap[k]);
        if (!t_slab[k]) { fprintf(stderr, "vtx_oracle: out of memory (%zu bytes of scratch)\n", t_slab_cap[k]); abort(); }
    }
    return t_slab[k];
}

/* ------------------------------------------------------------------------- */
/* Affine local alignment, full matrix.                                       */
/* bio::alignment::pairwise::Aligner::custom with all four clip penalties 0    */
/* (= `local`): S = max(diag + score(x_i, y_j), I, D, 0); a gap of length L     */
/* costs gap_open + L * gap_extend; score = max over all cells.               */
/* Column-major like the crate (outer loop over y).                           */
/* ------------------------------------------------------------------------- */
int32_t vtxo_sw_full(const uint8_t* x, int m, const uint8_t* y, int n,
                     int match, int mismatch, int gap_open, int gap_extend) {
    if (m <= 0 || n <= 0) return 0;
    int32_t* S = (int32_t*)slab(0, sizeof(int32_t) * (size_t)(m + 1) * 2);
    int32_t* D = S + (m + 1); /* D[i] = gap-in-x state of column j-1 for row i */
    int32_t best = 0;
    for (int i = 0; i <= m; ++i) { S[i] = 0; D[i] = VTXO_MIN_SCORE; }
    for (int j = 1; j <= n; ++j) {
        const uint8_t q = y[j - 1];
        int32_t diag = 0;           /* S[i-1][j-1] */
        int32_t up_s = 0;           /* S[i-1][j]   */
        int32_t up_i = VTXO_MIN_SCORE; /* I[i-1][j] */
        for (int i = 1; i <= m; ++i) {
            const int32_t left_s = S[i];   /* S[i][j-1] */
            /* D[i][j] = max(D[i][j-1] + e, S[i][j-1] + o + e) */
            int32_t d = imax(D[i] + gap_extend, left_s + gap_open + gap_extend);
            /* I[i][j] = max(I[i-1][j] + e, S[i-1][j] + o + e) */
            int32_t ii = imax(up_i + gap_extend, up_s + gap_open + gap_extend);
            int32_t s = diag + (x[i - 1] == q ? match : mismatch);
            s = imax(s, imax(d, ii));
            s = imax(s, 0);
            diag = left_s;
            S[i] = s; D[i] = d;
            up_s = s; up_i = ii;
            if (s > best) best = s;
        }
    }
    return best;
}

/* ------------------------------------------------------------------------- */
/* bio::alignment::sparse::find_kmer_matches                                  */
/* The crate hashes the k-mers of the shorter sequence and scans the other;    */
/* the result — every (i, j) with x[i..i+k] == y[j..j+k], sorted — does not     */
/* depend on which side is hashed.  Here: chained hash of y's k-mers.          */
/* ------------------------------------------------------------------------- */
static uint32_t kmer_hash(const uint8_t* s, int k) {
    uint32_t h = 2166136261u;
    for (int i = 0; i < k; ++i) { h ^= s[i]; h *= 16777619u; }
    return h;
}

int64_t vtxo_find_kmer_matches(const uint8_t* x, int m, const uint8_t* y, int n,
                               int k, uint32_t** out) {
    *out = NULL;
    if (k <= 0 || m < k || n < k) return 0;
    const int ny = n - k + 1, nx = m - k + 1;
    int tbits = 4;
    while ((1 << tbits) < 2 * ny) ++tbits;
    const uint32_t tmask = (1u << tbits) - 1u;
    int32_t* head = (int32_t*)slab(1, sizeof(int32_t) * (((size_t)1 << tbits) + (size_t)ny));
    int32_t* next = head + ((size_t)1 << tbits);
    for (uint32_t i = 0; i <= tmask; ++i) head[i] = -1;
    /* insert in descending j so each chain lists j ascending */
    for (int j = ny - 1; j >= 0; --j) {
        uint32_t h = kmer_hash(y + j, k) & tmask;
        next[j] = head[h]; head[h] = j;
    }
    int64_t cap = 256, cnt = 0;
    uint32_t* mt = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)cap);
    for (int i = 0; i < nx; ++i) {
        uint32_t h = kmer_hash(x + i, k) & tmask;
        for (int j = head[h]; j >= 0; j = next[j]) {
            if (memcmp(x + i, y + j, (size_t)k) == 0) {
                if (cnt == cap) { cap *= 2; mt = (uint32_t*)realloc(mt, sizeof(uint32_t) * 2 * (size_t)cap); }
                mt[2 * cnt] = (uint32_t)i; mt[2 * cnt + 1] = (uint32_t)j; ++cnt;
            }
        }
    }
    *out = mt;
    return cnt; /* already sorted by (i, j): i ascending outer, j ascending inner */
}

/* ------------------------------------------------------------------------- */
/* bio::alignment::sparse::sdpkpp — sparse DP over k-mer matches ("LCSk++      */
/* with gap penalties").  Published algorithm: every match p = (x, y) has a     */
/* start event (x, y) and an end event (x+k, y+k); events are processed in      */
/* lexicographic order.  At p's start, dp[p] = k*match, or the best earlier      */
/* chain q with end(q) <= start(p) component-wise, paying                      */
/* gap_open + d*gap_extend for the gap of d = dx + dy positions, plus k*match;  */
/* the predecessor is found with a max-Fenwick tree over y holding               */
/* position-offset scores.  At p's end, p may instead continue the match one     */
/* step up its diagonal (LCSk++ rule): dp = dp[(x-1, y-1)] + match.  Ties are    */
/* resolved by tuple comparison (score, match index).  Result: the highest        */
/* scoring chain as match indices.                                             */
/* ------------------------------------------------------------------------- */
/* Test hooks (tests/test_band_variants.py, tools/band_semantics_table.py): the recollected details of the un-vendored crate,
 * switchable one at a time, so that the exposure of every one of them can be measured.  Process globals: single-threaded use.
 *   VTXO_VAR_LAZY_EXT     Band::set_boundaries' lazy extension (default VTX_BAND_LAZY_EXT(k); any value >= 0)
 *   VTXO_VAR_LAST_ANCHOR  add_kmer's last anchor offset (default VTX_BAND_KMER_LAST_ANCHOR(k) = k; alternative k - 1)
 *   VTXO_VAR_NO_SEED      1: no k-mer match = whole matrix (default); 0: empty band (score 0)
 *   VTXO_VAR_TIE          sdpkpp ties: 1 = the larger match index wins (default, tuple comparison); 0 = the smaller
 * (The `x > 0 && y > 0` guard of the LCSk++ continuation is not a variant: without it the lookup of (x - 1, y - 1) wraps and
 * finds nothing, the same result.)                                                                                          */
static int g_var[4] = {-1, -1, -1, -1};
void vtxo_set_variant(int which, int value) { if (which >= 0 && which < 4) g_var[which] = value; }
static inline int var_tie_larger(void) { return g_var[3] != 0; }

typedef struct { int64_t v; int64_t idx; } bit_ent;
static inline int ent_gt(bit_ent a, bit_ent b) { return a.v > b.v || (a.v == b.v && (var_tie_larger() ? a.idx > b.idx : (b.idx < 0 || (a.idx >= 0 && a.idx < b.idx)))); }

typedef struct { uint32_t x, y, id; } sdp_event;
static int ev_cmp(const void* a, const void* b) {
    const sdp_event* p = (const sdp_event*)a; const sdp_event* q = (const sdp_event*)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    if (p->id != q->id) return p->id < q->id ? -1 : 1;
    return 0;
}

static int64_t match_find(const uint32_t* mt, int64_t n, uint32_t x, uint32_t y) {
    int64_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        uint32_t mx = mt[2 * mid], my = mt[2 * mid + 1];
        if (mx == x && my == y) return mid;
        if (mx < x || (mx == x && my < y)) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

int64_t vtxo_sdpkpp(const uint32_t* mt, int64_t M, int k, int match_score,
                    int gap_open, int gap_extend, int64_t* path_out, int64_t* score_out) {
    if (score_out) *score_out = 0;
    if (M <= 0) return 0;
    sdp_event* ev = (sdp_event*)slab(2, sizeof(sdp_event) * 2 * (size_t)M);
    uint32_t nmax = 0;
    for (int64_t p = 0; p < M; ++p) {
        uint32_t x = mt[2 * p], y = mt[2 * p + 1];
        ev[2 * p] = (sdp_event){x, y, (uint32_t)(p + M)};
        ev[2 * p + 1] = (sdp_event){x + (uint32_t)k, y + (uint32_t)k, (uint32_t)p};
        if (x + (uint32_t)k > nmax) nmax = x + (uint32_t)k;
        if (y + (uint32_t)k > nmax) nmax = y + (uint32_t)k;
    }
    qsort(ev, 2 * (size_t)M, sizeof(sdp_event), ev_cmp);
    /* max-Fenwick tree over column index 0..nmax (1-based internally) */
    const int64_t tn = (int64_t)nmax + 2;
    bit_ent* tree = (bit_ent*)slab(3, sizeof(bit_ent) * (size_t)(tn + 1));
    for (int64_t i = 0; i <= tn; ++i) tree[i] = (bit_ent){INT64_MIN, -1};
    int64_t* dps = (int64_t*)slab(4, sizeof(int64_t) * 2 * (size_t)M);
    int64_t* dpp = dps + M;
    const int64_t kscore = (int64_t)k * match_score;
    bit_ent best = {kscore, 0};
    for (int64_t e = 0; e < 2 * M; ++e) {
        const int is_start = ev[e].id >= (uint32_t)M;
        const int64_t p = is_start ? (int64_t)ev[e].id - M : (int64_t)ev[e].id;
        if (is_start) {
            dps[p] = kscore; dpp[p] = -1;
            /* prefix max over end columns <= y */
            bit_ent bq = {INT64_MIN, -1};
            for (int64_t i = (int64_t)ev[e].y + 1; i > 0; i -= i & (-i))
                if (ent_gt(tree[i], bq)) bq = tree[i];
            if (bq.idx >= 0) {
                /* stored value = dp[q] - gap_extend*(xe+ye); gap d = (x+y)-(xe+ye) */
                int64_t cand = bq.v + (int64_t)gap_open
                             + (int64_t)gap_extend * ((int64_t)ev[e].x + (int64_t)ev[e].y) + kscore;
                if (cand > dps[p] || (cand == dps[p] && (var_tie_larger() ? bq.idx > dpp[p] : dpp[p] < 0))) { dps[p] = cand; dpp[p] = bq.idx; }
            }
        } else {
            /* does this k-mer continue the diagonal of the match one step back? */
            const uint32_t x = ev[e].x - (uint32_t)k, y = ev[e].y - (uint32_t)k;
            if (x > 0 && y > 0) {   /* (the guard only avoids an unsigned wrap: without it (x - 1, y - 1) would never be found either) */
                int64_t c = match_find(mt, M, x - 1, y - 1);
                if (c >= 0) {
                    int64_t cand = dps[c] + match_score;
                    if (cand > dps[p] || (cand == dps[p] && (var_tie_larger() ? c > dpp[p] : (dpp[p] < 0 || c < dpp[p])))) { dps[p] = cand; dpp[p] = c; }
                }
            }
            bit_ent me = {dps[p] - (int64_t)gap_extend * ((int64_t)ev[e].x + (int64_t)ev[e].y), p};
            for (int64_t i = (int64_t)ev[e].y + 1; i <= tn; i += i & (-i))
                if (ent_gt(me, tree[i])) tree[i] = me;
            bit_ent cur = {dps[p], p};
            if (ent_gt(cur, best)) best = cur;
        }
    }
    /* traceback */
    int64_t len = 0;
    for (int64_t p = best.idx; p >= 0; p = dpp[p]) ++len;
    int64_t w = len;
    for (int64_t p = best.idx; p >= 0; p = dpp[p]) path_out[--w] = p;
    if (score_out) *score_out = best.v;
    return len;
}

/* ------------------------------------------------------------------------- */
/* bio::alignment::pairwise::banded::Band                                      */
/* rows = m+1, cols = n+1, one row range per column; ranges only grow (min of   */
/* starts, max of ends).  Cells are added as (2w+1)-squares around anchor        */
/* cells: add_entry = one anchor; add_kmer = the anchors of a k-mer's diagonal;   */
/* add_gap = diagonal run of min(dr, dc) then the straight remainder.           */
/* set_boundaries (free clipping on both sequences, i.e. local mode): the band   */
/* is extended lazily, diagonally by 2k cells past the first / last chained       */
/* k-mer (clipped at the matrix edge).  No matches => full matrix.              */
/* ------------------------------------------------------------------------- */
typedef struct { int rows, cols, w; int32_t* lo; int32_t* hi; } band_t;

static void band_add_entry(band_t* b, int r, int c) {
    const int istart = imax(r - b->w, 0), iend = imin(r + b->w + 1, b->rows);
    const int j0 = imax(c - b->w, 0), j1 = imin(c + b->w + 1, b->cols);
    for (int j = j0; j < j1; ++j) {
        if (istart < b->lo[j]) b->lo[j] = istart;
        if (iend > b->hi[j]) b->hi[j] = iend;
    }
}
static void band_add_kmer(band_t* b, int r, int c, int k) {
    const int last = g_var[1] >= 0 ? g_var[1] : VTX_BAND_KMER_LAST_ANCHOR(k);
    for (int d = 0; d <= last; ++d) band_add_entry(b, r + d, c + d);
}

/* Test hook: override of VTX_BAND_LAZY_EXT for the sensitivity tests (tests/test_band_variants.py); < 0 = the constant. */
#define g_lazy_override (g_var[0])
void vtxo_set_lazy_extension(int ext) { g_var[0] = ext; }
static void band_add_gap(band_t* b, int r0, int c0, int r1, int c1) {
    const int dr = r1 - r0, dc = c1 - c0;
    const int diag = imin(dr, dc);
    for (int d = 0; d <= diag; ++d) band_add_entry(b, r0 + d, c0 + d);
    if (dr > dc) { for (int r = r0 + diag; r <= r1; ++r) band_add_entry(b, r, c0 + diag); }
    else { for (int c = c0 + diag; c <= c1; ++c) band_add_entry(b, r0 + diag, c); }
}

int64_t vtxo_band_create(const uint8_t* x, int m, const uint8_t* y, int n,
                         int k, int w, int32_t* lo, int32_t* hi) {
    band_t b = {m + 1, n + 1, w, lo, hi};
    for (int j = 0; j <= n; ++j) { lo[j] = m + 1; hi[j] = 0; }
    uint32_t* mt = NULL;
    int64_t M = vtxo_find_kmer_matches(x, m, y, n, k, &mt);
    if (M == 0) {
        /* VTX_BAND_NO_SEED_FULL_MATRIX (g_var[2] == 0: the alternative — an empty band) */
        if (g_var[2] >= 0 ? g_var[2] : VTX_BAND_NO_SEED_FULL_MATRIX) for (int j = 0; j <= n; ++j) { lo[j] = 0; hi[j] = m + 1; }
    } else {
        int64_t* path = (int64_t*)slab(5, sizeof(int64_t) * (size_t)M);
        int64_t L = vtxo_sdpkpp(mt, M, k, 1 /* match_fn.score(b'A', b'A') */, -5, -1, path, NULL);
        /* NOTE: the aligner passes its own scoring's gap penalties; the
         * reference constructs it with (-5, -1) (src/main.rs:899).            */
        const int lazy = g_lazy_override >= 0 ? g_lazy_override : VTX_BAND_LAZY_EXT(k);
        const int fx = (int)mt[2 * path[0]], fy = (int)mt[2 * path[0] + 1];
        const int lx = (int)mt[2 * path[L - 1]] + k, ly = (int)mt[2 * path[L - 1] + 1] + k;
        int d = imin(imin(fx, fy), lazy);
        band_add_gap(&b, fx - d, fy - d, fx, fy);
        d = imin(imin(m - lx, n - ly), lazy);
        band_add_gap(&b, lx, ly, lx + d, ly + d);
        int px = -1, py = -1;
        for (int64_t t = 0; t < L; ++t) {
            const int cx = (int)mt[2 * path[t]], cy = (int)mt[2 * path[t] + 1];
            if (t > 0 && cx == px + 1 && cy == py + 1) {
                const int last = g_var[1] >= 0 ? g_var[1] : VTX_BAND_KMER_LAST_ANCHOR(k);
                band_add_entry(&b, cx + last, cy + last);
            } else {
                if (t > 0) band_add_gap(&b, px + k, py + k, cx, cy);
                band_add_kmer(&b, cx, cy, k);
            }
            px = cx; py = cy;
        }
    }
    free(mt);
    int64_t cells = 0;
    for (int j = 0; j <= n; ++j) if (hi[j] > lo[j]) cells += hi[j] - lo[j];
    return cells;
}

/* ------------------------------------------------------------------------- */
/* banded::Aligner::compute_alignment in local mode: the recurrences of         */
/* vtxo_sw_full restricted to the band; every cell outside the band holds       */
/* MIN_SCORE in S, I and D; every in-band cell may start at 0 (prefix clips are   */
/* free) and the score is the maximum over in-band cells (suffix clips free).     */
/* Row 0 / column 0 cells are 0 when in band, MIN_SCORE otherwise.              */
/* ------------------------------------------------------------------------- */
int32_t vtxo_sw_ranges(const uint8_t* x, int m, const uint8_t* y, int n,
                       int match, int mismatch, int gap_open, int gap_extend,
                       const int32_t* lo, const int32_t* hi) {
    const size_t R = (size_t)m + 1;
    int32_t* buf = (int32_t*)slab(6, sizeof(int32_t) * R * 4);
    int32_t *Sp = buf, *Dp = buf + R, *Sc = buf + 2 * R, *Dc = buf + 3 * R;
    int32_t best = 0;
    for (size_t i = 0; i < 4 * R; ++i) buf[i] = VTXO_MIN_SCORE;
    /* column 0: in-band cells are 0 (y prefix clipped) */
    for (int i = lo[0]; i < hi[0]; ++i) Sp[i] = 0;
    /* Sc / Dc still hold column j-2 when column j is written: only its band range needs resetting to
     * MIN_SCORE (everything else in the buffer already is), not the whole column.                     */
    int plo = 0, phi = 0;                       /* band range last written into Sc / Dc */
    int qlo = lo[0], qhi = hi[0];               /* ... into Sp / Dp */
    for (int j = 1; j <= n; ++j) {
        for (int i = plo; i < phi; ++i) { Sc[i] = VTXO_MIN_SCORE; Dc[i] = VTXO_MIN_SCORE; }
        const uint8_t q = y[j - 1];
        int32_t up_i = VTXO_MIN_SCORE;
        for (int i = lo[j]; i < hi[j]; ++i) {
            if (i == 0) { Sc[0] = 0; up_i = VTXO_MIN_SCORE; continue; }
            int32_t d = imax(Dp[i] + gap_extend, Sp[i] + gap_open + gap_extend);
            int32_t ii = imax(up_i + gap_extend, Sc[i - 1] + gap_open + gap_extend);
            int32_t s = Sp[i - 1] + (x[i - 1] == q ? match : mismatch);
            s = imax(s, imax(d, ii));
            s = imax(s, 0);
            Sc[i] = s; Dc[i] = d; up_i = ii;
            if (s > best) best = s;
        }
        plo = qlo; phi = qhi;
        qlo = lo[j]; qhi = hi[j] > lo[j] ? hi[j] : lo[j];
        int32_t* t;
        t = Sp; Sp = Sc; Sc = t;
        t = Dp; Dp = Dc; Dc = t;
    }
    return best;
}

int32_t vtxo_sw_banded(const uint8_t* x, int m, const uint8_t* y, int n,
                       int match, int mismatch, int gap_open, int gap_extend, int k, int w) {
    if (m <= 0 || n <= 0) return 0;
    int32_t* lo = (int32_t*)slab(7, sizeof(int32_t) * 2 * ((size_t)n + 1));
    int32_t* hi = lo + (n + 1);
    int64_t cells = vtxo_band_create(x, m, y, n, k, w, lo, hi);
    int32_t s;
    if (cells > VTXO_MAX_CELLS) s = VTXO_MIN_SCORE; /* banded.rs: empty alignment, score MIN_SCORE */
    else s = vtxo_sw_ranges(x, m, y, n, match, mismatch, gap_open, gap_extend, lo, hi);
    return s;
}

/* evaluate_scores, src/main.rs:1019-1030 */
int vtxo_evaluate_scores(int32_t ref_score, int32_t alt_score, int32_t min_score) {
    if ((ref_score < min_score) & (alt_score < min_score)) return VTXO_CALL_NONE;
    if (ref_score > alt_score) return VTXO_CALL_REF;
    if (alt_score > ref_score) return VTXO_CALL_ALT;
    return VTXO_CALL_UNKNOWN;
}

/* ------------------------------------------------------------------------- */
/* evaluate_chunk over a packed batch.  Loci are cut into contiguous chunks of    */
/* max(n_loci / threads, 1) (src/main.rs:250-254) and the chunks are mapped over   */
/* a pool of `threads` workers (src/main.rs:279-291); inside a chunk loci and       */
/* reads are sequential (src/main.rs:602-605, :829).  Per read: both haplotypes    */
/* (src/main.rs:900-901).                                                        */
/* ------------------------------------------------------------------------- */
static int32_t align_one(const vtx_config* c, const uint8_t* x, int m, const uint8_t* y, int n) {
    if (c->aligner == VTX_ALIGNER_FULL)
        return vtxo_sw_full(x, m, y, n, c->match_score, c->mismatch_score, c->gap_open, c->gap_extend);
    return vtxo_sw_banded(x, m, y, n, c->match_score, c->mismatch_score, c->gap_open, c->gap_extend,
                          c->kmer_k, c->band_w);
}

int vtxo_batch_scores(const vtx_batch* b, const vtx_config* cfg,
                      int32_t* ref_score, int32_t* alt_score, int threads) {
    if (threads < 1) threads = 1;
    const int64_t nl = b->n_loci;
    const int64_t chunk = nl / threads > 1 ? nl / threads : 1;
    const int64_t nchunks = (nl + chunk - 1) / chunk;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int64_t l0 = ch * chunk, l1 = (l0 + chunk < nl) ? l0 + chunk : nl;
        for (int64_t l = l0; l < l1; ++l) {
            const vtx_locus* L = &b->loci[l];
            const uint8_t* rh = b->hap_arena + L->ref_off;
            const uint8_t* ah = b->hap_arena + L->alt_off;
            for (uint32_t r = L->rec_begin; r < L->rec_begin + L->rec_count; ++r) {
                const vtx_record* R = &b->records[r];
                const uint8_t* seq = b->read_arena + R->read_off;
                ref_score[r] = align_one(cfg, seq, (int)R->read_len, rh, (int)L->ref_len);
                alt_score[r] = align_one(cfg, seq, (int)R->read_len, ah, (int)L->alt_len);
            }
        }
    }
    return 0;
}

uint64_t vtxo_batch_cells(const vtx_batch* b, const vtx_config* cfg, int threads) {
    uint64_t total = 0;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads) reduction(+ : total)
#endif
    for (int64_t l = 0; l < (int64_t)b->n_loci; ++l) {
        const vtx_locus* L = &b->loci[l];
        for (uint32_t r = L->rec_begin; r < L->rec_begin + L->rec_count; ++r) {
            const vtx_record* R = &b->records[r];
            if (cfg->aligner == VTX_ALIGNER_FULL) {
                total += (uint64_t)R->read_len * (L->ref_len + L->alt_len);
            } else {
                const uint8_t* seq = b->read_arena + R->read_off;
                for (int h = 0; h < 2; ++h) {
                    const uint8_t* hp = b->hap_arena + (h ? L->alt_off : L->ref_off);
                    const int n = (int)(h ? L->alt_len : L->ref_len);
                    int32_t* lo = (int32_t*)malloc(sizeof(int32_t) * 2 * ((size_t)n + 1));
                    int32_t* hi = lo + (n + 1);
                    vtxo_band_create(seq, (int)R->read_len, hp, n, cfg->kmer_k, cfg->band_w, lo, hi);
                    for (int j = 1; j <= n; ++j) {
                        int a = lo[j] < 1 ? 1 : lo[j];
                        if (hi[j] > a) total += (uint64_t)(hi[j] - a);
                    }
                    free(lo);
                }
            }
        }
    }
    return total;
}

/* ------------------------------------------------------------------------- */
/* Merge loop src/main.rs:320-348 = parse_scores (src/main.rs:1041-1109) then    */
/* consensus_scoring / alt_frac / coverage (src/main.rs:1111-1164).              */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t umi; int call; } umi_call;
static int umi_cmp(const void* a, const void* b) {
    const umi_call* p = (const umi_call*)a; const umi_call* q = (const umi_call*)b;
    return p->umi < q->umi ? -1 : (p->umi > q->umi ? 1 : 0);
}

int64_t vtxo_batch_reduce(const vtx_batch* b, const vtx_config* cfg,
                          const int32_t* ref_score, const int32_t* alt_score,
                          uint32_t* row, uint32_t* col, uint32_t* alt, uint32_t* ref,
                          uint32_t* unk, double* value, double* ref_value) {
    int64_t nnz = 0;
    umi_call* tmp = NULL; size_t tmp_cap = 0;
    for (uint32_t l = 0; l < b->n_loci; ++l) {
        const vtx_locus* L = &b->loci[l];
        uint32_t r = L->rec_begin;
        const uint32_t rend = L->rec_begin + L->rec_count;
        while (r < rend) {
            /* group_by(cell_index) over the sorted scores, src/main.rs:1044 */
            const uint32_t cell = b->records[r].cell_index;
            uint32_t g1 = r;
            while (g1 < rend && b->records[g1].cell_index == cell) ++g1;
            uint32_t rc = 0, ac = 0, uc = 0;
            if (!cfg->use_umi) {
                /* src/main.rs:1090-1105 */
                for (uint32_t i = r; i < g1; ++i) {
                    int c = vtxo_evaluate_scores(ref_score[i], alt_score[i], cfg->min_score);
                    if (c == VTXO_CALL_REF) ++rc; else if (c == VTXO_CALL_ALT) ++ac;
                    else if (c == VTXO_CALL_UNKNOWN) ++uc;
                }
            } else {
                /* src/main.rs:1045-1088: per UMI, collapse calls with the 0.75 rule */
                size_t nc = 0;
                if ((size_t)(g1 - r) > tmp_cap) { tmp_cap = (size_t)(g1 - r) * 2; tmp = (umi_call*)realloc(tmp, tmp_cap * sizeof(umi_call)); }
                for (uint32_t i = r; i < g1; ++i) {
                    int c = vtxo_evaluate_scores(ref_score[i], alt_score[i], cfg->min_score);
                    if (c == VTXO_CALL_NONE) continue;   /* :1050-1052 */
                    tmp[nc].umi = b->records[i].umi_id; tmp[nc].call = c; ++nc;
                }
                qsort(tmp, nc, sizeof(umi_call), umi_cmp);
                size_t i = 0;
                while (i < nc) {
                    size_t j = i; double r_ = 0, a_ = 0, u_ = 0;
                    while (j < nc && tmp[j].umi == tmp[i].umi) {
                        if (tmp[j].call == VTXO_CALL_REF) r_ += 1; else if (tmp[j].call == VTXO_CALL_ALT) a_ += 1; else u_ += 1;
                        ++j;
                    }
                    const double ref_frac = r_ / (a_ + r_ + u_);   /* :1070-1073 */
                    const double alt_frac = a_ / (a_ + r_ + u_);
                    if ((ref_frac < 0.75) & (alt_frac < 0.75)) ++uc;       /* :1074-1075 */
                    else if (alt_frac >= 0.75) ++ac;                       /* :1076-1077 */
                    else ++rc;                                             /* :1078-1080 */
                    i = j;
                }
            }
            int emit = 1; double v = 0.0, rv = 0.0;
            if (cfg->scoring_mode == VTX_MODE_CONSENSUS) {          /* :1120-1126 */
                if (rc > 0 && ac > 0) v = 3.0; else if (ac > 0) v = 2.0; else if (rc > 0) v = 1.0; else emit = 0;
            } else if (cfg->scoring_mode == VTX_MODE_ALT_FRAC) {    /* :1140-1142 */
                v = (double)ac / ((double)rc + (double)ac + (double)uc);
            } else {                                               /* :1160-1161 */
                v = (double)ac; rv = (double)rc;
            }
            if (emit) {
                row[nnz] = L->row; col[nnz] = cell; alt[nnz] = ac; ref[nnz] = rc; unk[nnz] = uc;
                value[nnz] = v; ref_value[nnz] = rv; ++nnz;
            }
            r = g1;
        }
    }
    free(tmp);
    return nnz;
}

/* ------------------------------------------------------------------------- */
/* rust-htslib 0.36 CigarStringView::read_pos as used at src/main.rs:796        */
/* (include_softclips = false, include_dels = true).  Walks the ops from the     */
/* alignment start; returns Some(query pos) when ref_pos lies in M/=/X (or in D   */
/* with include_dels, or S with include_softclips), None otherwise.  Errors:     */
/* hard clip that is not at the ends, and (per the crate's documented checks)     */
/* a leading D/N before any query-consuming op.                                */
/* ------------------------------------------------------------------------- */
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

int vtxo_cigar_read_pos(const uint32_t* cigar, int n_ops, int64_t pos, int64_t ref_pos,
                        int include_softclips, int include_dels, int64_t* qpos_out) {
    int64_t rpos = pos;  /* reference position at the start of the current op */
    int64_t qpos = 0;    /* query position at the start of the current op */
    int j = 0;
    /* phase 1: find the first op that refers to query position 0 */
    for (int i = 0; i < n_ops; ++i) {
        const int op = (int)(cigar[i] & 0xf); const int64_t l = cigar[i] >> 4;
        if (op == OP_M || op == OP_X || op == OP_EQ || op == OP_I) { j = i; break; }
        if (op == OP_S) {
            j = i;
            if (include_softclips) rpos = rpos >= l ? rpos - l : 0;   /* POS excludes the clip */
            break;
        }
        if (op == OP_D || op == OP_N) return -1;  /* D / N before any op describing read sequence */
        if (op == OP_H && i > 0 && i < n_ops - 1) return -1;  /* H between operations */
        if ((op == OP_P || op == OP_H) && i == n_ops - 1) return 0;  /* only pads / hard clips */
        /* leading H / P: consume nothing */
    }
    /* phase 2: walk */
    while (rpos <= ref_pos && j < n_ops) {
        const int op = (int)(cigar[j] & 0xf); const int64_t l = cigar[j] >> 4;
        const int contains = (rpos <= ref_pos) && (rpos + l > ref_pos);
        switch (op) {
        case OP_M: case OP_X: case OP_EQ:
            if (contains) { if (qpos_out) *qpos_out = qpos + (ref_pos - rpos); return 1; }
            rpos += l; qpos += l; ++j; break;
        case OP_S:
            if (include_softclips && contains) { if (qpos_out) *qpos_out = qpos + (ref_pos - rpos); return 1; }
            qpos += l; ++j; if (include_softclips) rpos += l; break;
        case OP_D:
            if (include_dels && contains) { if (qpos_out) *qpos_out = qpos; return 1; }
            rpos += l; ++j; break;
        case OP_N: rpos += l; ++j; break;
        case OP_I: qpos += l; ++j; break;
        case OP_P: ++j; break;
        case OP_H:
            if (j < n_ops - 1) return -1;   /* hard clip in between operations */
            return 0;
        default: return -1;
        }
    }
    return 0;
}

/* useful_alignment, src/main.rs:790-806: any i in start..=end (inclusive) */
int vtxo_useful_alignment(const uint32_t* cigar, int n_ops, int64_t pos,
                          int64_t locus_start, int64_t locus_end) {
    for (int64_t i = locus_start; i <= locus_end; ++i) {
        int r = vtxo_cigar_read_pos(cigar, n_ops, pos, i, 0, 1, NULL);
        if (r == 1) return 1;
        if (r < 0) return 0;   /* :799-802 invalid CIGAR => skip read */
    }
    return 0;
}

/* construct_haplotypes + read_locus, src/main.rs:936-994.  Flanks and the REF    */
/* window are upper-cased (:952); ALT allele bytes are copied verbatim (:979).     */
static uint8_t up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

void vtxo_construct_haplotypes(const uint8_t* contig, int64_t contig_len,
                               int64_t start, int64_t end,
                               const uint8_t* alt, int64_t alt_len, int64_t padding,
                               uint8_t* ref_out, int64_t* ref_out_len,
                               uint8_t* alt_out, int64_t* alt_out_len) {
    /* alt_hap: get_range(start.saturating_sub(padding), start) ++ alt ++
     *          get_range(end, min(end + padding, chrom_len))   (:977-981)  */
    int64_t n = 0;
    int64_t ls = start >= padding ? start - padding : 0;
    int64_t le = start < contig_len ? start : contig_len;     /* read_locus: min(end, chrom_len) :945 */
    for (int64_t i = ls; i < le; ++i) alt_out[n++] = up(contig[i]);
    for (int64_t i = 0; i < alt_len; ++i) alt_out[n++] = alt[i];
    int64_t re = end + padding < contig_len ? end + padding : contig_len;
    for (int64_t i = end; i < re; ++i) alt_out[n++] = up(contig[i]);
    *alt_out_len = n;
    /* ref_hap: read_locus(locus, padding, padding): max(0, start - pad) as i32
     * (:944) .. min(end + pad, chrom_len) (:945)                            */
    int64_t rs = (int32_t)start - (int32_t)padding; if (rs < 0) rs = 0;
    n = 0;
    for (int64_t i = rs; i < re; ++i) ref_out[n++] = up(contig[i]);
    *ref_out_len = n;
}

/* ------------------------------------------------------------------------- */
/* Rust `{}` for f64 (what sprs' write_matrix_market prints, src/main.rs:381):    */
/* shortest digit string that round-trips, positional notation, no exponent, no    */
/* trailing ".0"; NaN -> "NaN", infinities -> "inf" / "-inf".                    */
/* ------------------------------------------------------------------------- */
int vtxo_format_f64(double v, char* buf) {
    if (isnan(v)) return sprintf(buf, "NaN");
    if (isinf(v)) return sprintf(buf, v < 0 ? "-inf" : "inf");
    if (v == 0.0) return sprintf(buf, signbit(v) ? "-0" : "0");
    char tmp[40];
    int prec;
    for (prec = 1; prec <= 17; ++prec) {
        snprintf(tmp, sizeof tmp, "%.*e", prec - 1, v);
        if (strtod(tmp, NULL) == v) break;
    }
    /* tmp = d.ddddde[+-]XX */
    char digits[24]; int nd = 0; int neg = 0;
    const char* p = tmp;
    if (*p == '-') { neg = 1; ++p; }
    for (; *p && *p != 'e'; ++p) if (*p >= '0' && *p <= '9') digits[nd++] = *p;
    int ex = atoi(p + 1);
    while (nd > 1 && digits[nd - 1] == '0') --nd;
    char* o = buf;
    if (neg) *o++ = '-';
    if (ex >= 0) {
        for (int i = 0; i <= ex; ++i) *o++ = i < nd ? digits[i] : '0';
        if (nd > ex + 1) { *o++ = '.'; for (int i = ex + 1; i < nd; ++i) *o++ = digits[i]; }
    } else {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -ex - 1; ++i) *o++ = '0';
        for (int i = 0; i < nd; ++i) *o++ = digits[i];
    }
    *o = 0;
    return (int)(o - buf);
}
