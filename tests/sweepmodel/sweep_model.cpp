// sweep_model.cpp — scalar CPU model of band_sweep_kernel's per-task algorithm (vartrix_amd/csrc/vtx_sweep.hip).
// TEST INFRASTRUCTURE ONLY (tests/test_sweep_model.py): it restates, step by step and with the same packed words, what
// the eight lanes of a task do on the device, so that the ALGORITHM (row sweep instead of the sorted event list,
// dp finalised at the start event, the section log instead of per-match predecessor links, the closed-form band) is
// checked against the oracle's literal restatement of bio 0.30.0 (oracle/vtx_oracle.c: vtxo_find_kmer_matches,
// vtxo_sdpkpp, vtxo_band_create; reference call site src/main.rs:898-901) on the CPU, without a GPU.
//
// The algorithm (K = 6, W = 20; include/vtx_band_semantics.h):
//   rows     match mask of read row x over the haplotype columns: M6(x) = AND_t Eq[x[x + t]] >> t  (bit y: the 6-mers at
//            x and y are equal).  Bytes outside ACGTN: the task is declined (the general kernel takes it).
//   sweep    x ascending; END events of row x (the matches that started at row x - 6) enter a max-Fenwick tree over
//            their end column with the word  V << 16 | xq << 8 | yq  (V = dp + xe + ye: the tuple order of sdpkpp's
//            tree, ties to the larger match index = the lexicographically larger start); then the START events of
//            row x: dp = max(6, prefix-max(y).V - (x + y) + 1, dp(x - 1, y - 1) + 1) — the continuation wins ties, a
//            jump needs >= 6 (oracle: cand > dp || cand == dp && larger index; every jump source has a smaller index
//            than the continuation partner).  dp is FINAL at the start event: the continuation partner's dp was.
//   log      every match that does NOT continue its diagonal opens a section: (x, y, source or none).
//   chain    from the best (dp, x, y): the section is the log entry of this diagonal with the largest x' <= x; go on
//            from its source.
//   band     anchors of the staircase (lazy extension, sections, gaps) -> rmin / rmax per column -> lo / hi
//            (closed form, vtx_band.hip's header).
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#include "../../include/vtx_band_semantics.h"

namespace {
constexpr int K = VTX_REF_K, W = VTX_REF_W;
constexpr int MAXLEN = 255;

inline int code_of(uint8_t b) {
    switch (b) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'N': return 4; default: return -1; }
}

struct Mask256 { uint32_t w[8]; };
inline Mask256 shr(const Mask256& a, int s) {          // bit j of the result = bit j + s of a
    Mask256 r;
    for (int i = 0; i < 8; ++i) {
        const uint32_t lo = a.w[i] >> s;
        const uint32_t hi = (i + 1 < 8 && s) ? a.w[i + 1] << (32 - s) : 0u;
        r.w[i] = lo | hi;
    }
    return r;
}
inline Mask256 band(const Mask256& a, const Mask256& b) { Mask256 r; for (int i = 0; i < 8; ++i) r.w[i] = a.w[i] & b.w[i]; return r; }
}  // namespace

extern "C" {

// status: 0 band written; 1 declined (bytes outside ACGTN / lengths above 255); 2 log capacity exceeded; 3 too many sections.
// lo / hi: n + 1 entries (oracle format: empty columns lo = m + 1, hi = 0).  stats[0] = log entries, [1] = sections, [2] = matches.
int vtxs_band(const uint8_t* x, int m, const uint8_t* y, int n, int log_cap, int sec_cap, int32_t* lo, int32_t* hi, int32_t* stats) {
    if (stats) stats[0] = stats[1] = stats[2] = 0;
    if (m > MAXLEN || n > MAXLEN) return 1;
    std::vector<int> cx(m), cy(n);
    for (int i = 0; i < m; ++i) if ((cx[i] = code_of(x[i])) < 0) return 1;
    for (int j = 0; j < n; ++j) if ((cy[j] = code_of(y[j])) < 0) return 1;
    Mask256 eq[5];
    memset(eq, 0, sizeof eq);
    for (int j = 0; j < n; ++j) eq[cy[j]].w[j >> 5] |= 1u << (j & 31);
    const int rows_m = m - K + 1;                         // rows with a 6-mer
    std::vector<Mask256> m6(std::max(rows_m, 0));
    long total = 0;
    for (int r = 0; r < rows_m; ++r) {
        Mask256 a = eq[cx[r]];
        for (int t = 1; t < K; ++t) a = band(a, shr(eq[cx[r + t]], t));
        m6[r] = a;
        for (int i = 0; i < 8; ++i) total += __builtin_popcount(a.w[i]);
    }
    if (stats) stats[2] = (int32_t)total;
    for (int j = 0; j <= n; ++j) { lo[j] = m + 1; hi[j] = 0; }
    if (total == 0) {
        if (VTX_BAND_NO_SEED_FULL_MATRIX) for (int j = 0; j <= n; ++j) { lo[j] = 0; hi[j] = m + 1; }
        return 0;
    }
    // ---- sweep ----
    uint8_t ring[8][256];
    memset(ring, 0, sizeof ring);
    uint32_t tree[257];
    memset(tree, 0, sizeof tree);
    std::vector<uint32_t> log;                            // x << 24 | y << 16 | source (xq << 8 | yq, 0xffff: none)
    uint32_t best = 0;
    for (int r = 0; r <= m; ++r) {
        if (r >= K && r - K < rows_m) {                   // END events of row r
            const int xs = r - K;
            for (int yy = 0; yy < n; ++yy) {
                if (!((m6[xs].w[yy >> 5] >> (yy & 31)) & 1u)) continue;
                const uint32_t dp = ring[xs & 7][yy];
                const uint32_t V = dp + (uint32_t)r + (uint32_t)(yy + K);
                const uint32_t val = (V << 16) | ((uint32_t)xs << 8) | (uint32_t)yy;
                for (int i = yy + K + 1; i <= 256; i += i & (-i)) tree[i] = std::max(tree[i], val);
                best = std::max(best, (dp << 16) | ((uint32_t)xs << 8) | (uint32_t)yy);
            }
        }
        if (r < rows_m) {                                 // START events of row r
            memset(ring[r & 7], 0, 256);
            for (int yy = 0; yy < n; ++yy) {
                if (!((m6[r].w[yy >> 5] >> (yy & 31)) & 1u)) continue;
                uint32_t q = 0;
                for (int i = yy + 1; i > 0; i -= i & (-i)) q = std::max(q, tree[i]);
                int dv = K;
                uint32_t src = 0xffffu;
                bool cont = false;
                if (q) {
                    const int cand = (int)(q >> 16) - (r + yy) + 1;
                    if (cand >= K) { dv = cand; src = q & 0xffffu; }
                }
                if (r > 0 && yy > 0) {
                    const int dpc = ring[(r - 1) & 7][yy - 1];
                    if (dpc && dpc + 1 >= dv) { dv = dpc + 1; cont = true; }
                }
                ring[r & 7][yy] = (uint8_t)dv;
                if (!cont) {
                    if ((int)log.size() >= log_cap) return 2;
                    log.push_back(((uint32_t)r << 24) | ((uint32_t)yy << 16) | src);
                }
            }
        }
    }
    if (stats) stats[0] = (int32_t)log.size();
    // ---- chain: sections, last first ----
    struct Sec { int x0, y0, len; };
    std::vector<Sec> secs;
    int cxr = (int)((best >> 8) & 0xffu), cyr = (int)(best & 0xffu);
    for (;;) {
        const int d = cyr - cxr;
        int found = -1, fx = -1;
        for (size_t e = 0; e < log.size(); ++e) {
            const int ex = (int)(log[e] >> 24), ey = (int)((log[e] >> 16) & 0xffu);
            if (ey - ex == d && ex <= cxr && ex > fx) { fx = ex; found = (int)e; }
        }
        if (found < 0) return 4;                          // cannot happen: every piece start is logged
        const int ex = (int)(log[found] >> 24), ey = (int)((log[found] >> 16) & 0xffu);
        if ((int)secs.size() >= sec_cap) return 3;
        secs.push_back(Sec{ex, ey, cxr - ex + 1});
        const uint32_t src = log[found] & 0xffffu;
        if (src == 0xffffu) break;
        cxr = (int)(src >> 8); cyr = (int)(src & 0xffu);
    }
    std::reverse(secs.begin(), secs.end());
    if (stats) stats[1] = (int32_t)secs.size();
    // ---- anchors -> rmin / rmax ----
    std::vector<int> rmin(n + 2, 1 << 20), rmax(n + 2, -1);
    auto anchor = [&](int r, int c) { rmin[c] = std::min(rmin[c], r); rmax[c] = std::max(rmax[c], r); };
    const int lazy = VTX_BAND_LAZY_EXT(K), last = VTX_BAND_KMER_LAST_ANCHOR(K);
    const int fx0 = secs.front().x0, fy0 = secs.front().y0;
    const int d0 = std::min(std::min(fx0, fy0), lazy);
    for (int t = 0; t <= d0; ++t) anchor(fx0 - d0 + t, fy0 - d0 + t);
    const int cA = fy0 - d0;
    int pr = -1, pc = -1;                                  // end anchor of the previous section = (last match + K): add_gap's origin
    for (size_t s = 0; s < secs.size(); ++s) {
        const Sec& S = secs[s];
        if (s > 0) {
            const int dr = S.x0 - pr, dc = S.y0 - pc, dg = std::min(dr, dc);
            for (int t = 0; t <= dg; ++t) anchor(pr + t, pc + t);
            if (dr > dc) { for (int r = pr + dg; r <= S.x0; ++r) anchor(r, pc + dg); }
            else { for (int c = pc + dg; c <= S.y0; ++c) anchor(pr + dg, c); }
        }
        const int span = S.len - 1 + last;                // first match: anchors 0 .. last; every continued match adds (+ last)
        for (int t = 0; t <= span; ++t) anchor(S.x0 + t, S.y0 + t);
        pr = S.x0 + S.len - 1 + K; pc = S.y0 + S.len - 1 + K;
    }
    const int lx = pr, ly = pc;
    const int d1 = std::min(std::min(m - lx, n - ly), lazy);
    for (int t = 0; t <= d1; ++t) anchor(lx + t, ly + t);
    const int cB = ly + d1;
    // (with LAST_ANCHOR = k - 1 the cell (lx, ly) itself is only reached by add_gap's origin: it is anchored above either way)
    for (int j = 0; j <= n; ++j) {
        if (j < cA - W || j > cB + W) continue;
        const int c0 = std::max(j - W, cA), c1 = std::min(j + W, cB);
        lo[j] = std::max(rmin[c0] - W, 0);
        hi[j] = std::min(rmax[c1] + W + 1, m + 1);
    }
    return 0;
}


// ---- round 5: the sweep WITHOUT a dp ring (vtx_sweep.hip, band_sweep_kernel) --------------------------------------------------
// dp along a run of continuing matches grows by exactly 1 per row, so dp(x, y) = x + 6 - o with ONE deficit o per SECTION (a run
// that starts at a match which does not continue its diagonal, or at one where a jump beats the continuation).  The kernel keeps
// per DIAGONAL d = y - x the latest section only: (o, x0 = its first row), 16 bits — no per-row / per-column dp bytes.  A
// continuing match costs nothing at its START unless a jump could beat it; the END event of the match that started at row xs
// reads dp = xs + 6 - o from its diagonal's entry.  The one case the latest section does not cover: a jump beats the continuation
// at row r while matches of rows r - 5 .. r - 1 of the OLD section have not ended yet.  Their (end row, column, dp) go to a
// stash — eight buckets by end row, STASH entries each — and enter the tree when their row comes; the END loop skips a match
// whose row lies before its diagonal's x0.  status 5: a stash bucket is full (the general kernel takes the task); status 2: one of
// the eight lanes (32 columns each) opened more than log_cap / 8 sections.
// stats: [0] log entries, [1] sections, [2] matches, [3] jump-beats-continuation events, [4] fullest stash bucket.
// vtxs_band2_lanes (same algorithm, 12 stats): additionally what the kernel's EIGHT LANES would do row by row — [5] START trips (sum
// over rows of the largest per-lane count of matches the lane has to look at: new sections + the continuing matches of a lane whose
// bound pmax.V - 2x - 5 > G fails), [6] END trips (largest per-lane count of ending matches), [7] prefix-maximum queries, [8] rows
// with any START trip, [9] continuing matches looked at, [10] of those: in a lane with NO new section (pure filter failures).
static int g_lane_stats = 0;
static int g_row_start[272], g_row_end[272];             // per row: the trips of the last vtxs_band2_lanes call (START, END)
int vtxs_band2(const uint8_t* x, int m, const uint8_t* y, int n, int log_cap, int sec_cap, int stash_cap, int32_t* lo, int32_t* hi,
               int32_t* stats) {
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = stats[4] = 0;
    if (m > MAXLEN || n > MAXLEN) return 1;
    std::vector<int> cx(m), cy(n);
    for (int i = 0; i < m; ++i) if ((cx[i] = code_of(x[i])) < 0) return 1;
    for (int j = 0; j < n; ++j) if ((cy[j] = code_of(y[j])) < 0) return 1;
    Mask256 eq[5];
    memset(eq, 0, sizeof eq);
    for (int j = 0; j < n; ++j) eq[cy[j]].w[j >> 5] |= 1u << (j & 31);
    const int rows_m = m - K + 1;
    std::vector<Mask256> m6(std::max(rows_m, 0));
    long total = 0;
    for (int r = 0; r < rows_m; ++r) {
        Mask256 a = eq[cx[r]];
        for (int t = 1; t < K; ++t) a = band(a, shr(eq[cx[r + t]], t));
        m6[r] = a;
        for (int i = 0; i < 8; ++i) total += __builtin_popcount(a.w[i]);
    }
    if (stats) stats[2] = (int32_t)total;
    for (int j = 0; j <= n; ++j) { lo[j] = m + 1; hi[j] = 0; }
    if (total == 0) {
        if (VTX_BAND_NO_SEED_FULL_MATRIX) for (int j = 0; j <= n; ++j) { lo[j] = 0; hi[j] = m + 1; }
        return 0;
    }
    auto bit = [&](int r, int yy) { return r >= 0 && r < rows_m && yy >= 0 && yy < n && ((m6[r].w[yy >> 5] >> (yy & 31)) & 1u); };
    const bool lane_stats = g_lane_stats && stats;
    if (lane_stats) for (int i = 5; i < 12; ++i) stats[i] = 0;
    uint32_t pmax_l[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // largest V inserted at an end column below 32 (l + 1)
    int G_l[8];
    for (int l = 0; l < 8; ++l) G_l[l] = 0x3fffffff;
    struct Ent { uint8_t o, x0; };
    Ent off[512];
    memset(off, 0, sizeof off);
    uint32_t tree[257];
    memset(tree, 0, sizeof tree);
    struct St { int y, dp; };
    std::vector<St> stash[8];
    std::vector<uint32_t> log;
    int lane_log[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // the kernel's lane l logs the sections that open in columns 32 l .. 32 l + 31: log_cap / 8 each
    uint32_t best = 0;
    auto end_event = [&](int xs, int yy, uint32_t dp) {
        const uint32_t V = dp + (uint32_t)(xs + K) + (uint32_t)(yy + K);
        const uint32_t val = (V << 16) | ((uint32_t)xs << 8) | (uint32_t)yy;
        for (int i = yy + K + 1; i <= 256; i += i & (-i)) tree[i] = std::max(tree[i], val);
        best = std::max(best, (dp << 16) | ((uint32_t)xs << 8) | (uint32_t)yy);
        for (int l = std::min((yy + K) >> 5, 7); l < 8; ++l) pmax_l[l] = std::max(pmax_l[l], val);
    };
    for (int r = 0; r <= m; ++r) {
        const int xs = r - K;
        if (lane_stats && xs >= 0 && xs < rows_m) {
            int mx = 0;
            for (int l = 0; l < 8; ++l) mx = std::max(mx, __builtin_popcount(m6[xs].w[l]));
            stats[6] += mx;
            g_row_end[r] = mx;
        }
        if (xs >= 0 && xs < rows_m) {
            for (int yy = 0; yy < n; ++yy) {
                if (!bit(xs, yy)) continue;
                const Ent e = off[yy - xs + 256];
                if (xs < (int)e.x0) continue;                  // a match of the diagonal's previous section: its END is in the stash
                end_event(xs, yy, (uint32_t)(r - (int)e.o));   // dp = xs + 6 - o
            }
        }
        for (const St& t : stash[r & 7]) end_event(xs, t.y, (uint32_t)t.dp);
        stash[r & 7].clear();
        if (lane_stats && r < rows_m) {
            // what the eight lanes do (vtx_sweep.hip): per lane the new sections, plus ALL its continuing matches when the lane-level
            // bound fails; G = the bound carried from the previous row
            int trips = 0, Gn[8];
            for (int l = 0; l < 8; ++l) {
                int Gc = 0x3fffffff, n_new = 0, n_cont = 0, gmin_all = 0x3fffffff, nq = 0;
                const int thr = (int)(pmax_l[l] >> 16) - 2 * r - 5;
                for (int b = 0; b < 32; ++b) {
                    const int yy = 32 * l + b;
                    if (!bit(r, yy)) continue;
                    if (bit(r - 1, yy - 1)) {
                        ++n_cont;
                        Gc = std::min(Gc, (b ? G_l[l] : (l ? G_l[l - 1] : 0x3fffffff)) + 1);
                    } else ++n_new;
                }
                const bool chk = n_cont && pmax_l[l] && thr > Gc;
                for (int b = 0; b < 32; ++b) {
                    const int yy = 32 * l + b;
                    if (!bit(r, yy)) continue;
                    if (bit(r - 1, yy - 1)) {
                        const int gy = yy - (int)off[yy - r + 256].o;
                        if (chk && thr > gy) ++nq;
                        gmin_all = std::min(gmin_all, gy);       // (a jump that wins raises it: ignored here, the bound only gets looser)
                    } else if (pmax_l[l] && (int)(pmax_l[l] >> 16) - (r + yy) + 1 >= K) ++nq;
                }
                const int todo = n_new + (chk ? n_cont : 0);
                trips = std::max(trips, todo);
                stats[7] += nq;
                if (chk) { stats[9] += n_cont; if (!n_new) stats[10] += n_cont; }
                Gn[l] = chk ? gmin_all : Gc;                      // new sections' own values join below (after their dv is known): approximated by the continuing ones
                if (!(n_new + n_cont)) Gn[l] = 0x3fffffff;
            }
            for (int l = 0; l < 8; ++l) G_l[l] = Gn[l];
            stats[5] += trips;
            g_row_start[r] = trips;
            if (trips) ++stats[8];
        }
        if (r < rows_m) {
            for (int yy = 0; yy < n; ++yy) {
                if (!bit(r, yy)) continue;
                uint32_t q = 0;
                for (int i = yy + 1; i > 0; i -= i & (-i)) q = std::max(q, tree[i]);
                const int cand = q ? (int)(q >> 16) - (r + yy) + 1 : 0;
                const int d = yy - r + 256;
                if (!bit(r - 1, yy - 1)) {                     // opens its diagonal's run: a section
                    int dv = K;
                    uint32_t src = 0xffffu;
                    if (q && cand >= K) { dv = cand; src = q & 0xffffu; }
                    off[d] = Ent{(uint8_t)(r + K - dv), (uint8_t)r};
                    if (lane_stats) G_l[yy >> 5] = std::min(G_l[yy >> 5], yy - (r + K - dv));
                    if (++lane_log[yy >> 5] > log_cap / 8) return 2;
                    log.push_back(((uint32_t)r << 24) | ((uint32_t)yy << 16) | src);
                } else {
                    const Ent e = off[d];
                    const int dpc = (r - 1) + K - (int)e.o;
                    if (q && cand > dpc + 1) {                 // a jump beats the continuation (ties: the continuation)
                        if (stats) ++stats[3];
                        for (int xp = std::max((int)e.x0, r - (K - 1)); xp <= r - 1; ++xp) {   // matches of the old section still to end
                            std::vector<St>& b = stash[(xp + K) & 7];
                            if ((int)b.size() >= stash_cap) return 5;
                            b.push_back(St{yy - (r - xp), xp + K - (int)e.o});
                            if (stats) stats[4] = std::max(stats[4], (int32_t)b.size());
                        }
                        off[d] = Ent{(uint8_t)(r + K - cand), (uint8_t)r};
                        if (++lane_log[yy >> 5] > log_cap / 8) return 2;
                        log.push_back(((uint32_t)r << 24) | ((uint32_t)yy << 16) | (q & 0xffffu));
                    }
                }
            }
        }
    }
    if (stats) stats[0] = (int32_t)log.size();
    struct Sec { int x0, y0, len; };
    std::vector<Sec> secs;
    int cxr = (int)((best >> 8) & 0xffu), cyr = (int)(best & 0xffu);
    for (;;) {
        const int d = cyr - cxr;
        int found = -1, fx = -1;
        for (size_t e = 0; e < log.size(); ++e) {
            const int ex = (int)(log[e] >> 24), ey = (int)((log[e] >> 16) & 0xffu);
            if (ey - ex == d && ex <= cxr && ex > fx) { fx = ex; found = (int)e; }
        }
        if (found < 0) return 4;
        const int ex = (int)(log[found] >> 24), ey = (int)((log[found] >> 16) & 0xffu);
        if ((int)secs.size() >= sec_cap) return 3;
        secs.push_back(Sec{ex, ey, cxr - ex + 1});
        const uint32_t src = log[found] & 0xffffu;
        if (src == 0xffffu) break;
        cxr = (int)(src >> 8); cyr = (int)(src & 0xffu);
    }
    std::reverse(secs.begin(), secs.end());
    if (stats) stats[1] = (int32_t)secs.size();
    std::vector<int> rmin(n + 2, 1 << 20), rmax(n + 2, -1);
    auto anchor = [&](int r, int c) { rmin[c] = std::min(rmin[c], r); rmax[c] = std::max(rmax[c], r); };
    const int lazy = VTX_BAND_LAZY_EXT(K), last = VTX_BAND_KMER_LAST_ANCHOR(K);
    const int fx0 = secs.front().x0, fy0 = secs.front().y0;
    const int d0 = std::min(std::min(fx0, fy0), lazy);
    for (int t = 0; t <= d0; ++t) anchor(fx0 - d0 + t, fy0 - d0 + t);
    const int cA = fy0 - d0;
    int pr = -1, pc = -1;
    for (size_t s2 = 0; s2 < secs.size(); ++s2) {
        const Sec& S = secs[s2];
        if (s2 > 0) {
            const int dr = S.x0 - pr, dc = S.y0 - pc, dg = std::min(dr, dc);
            for (int t = 0; t <= dg; ++t) anchor(pr + t, pc + t);
            if (dr > dc) { for (int r = pr + dg; r <= S.x0; ++r) anchor(r, pc + dg); }
            else { for (int c = pc + dg; c <= S.y0; ++c) anchor(pr + dg, c); }
        }
        const int span = S.len - 1 + last;
        for (int t = 0; t <= span; ++t) anchor(S.x0 + t, S.y0 + t);
        pr = S.x0 + S.len - 1 + K; pc = S.y0 + S.len - 1 + K;
    }
    const int lx = pr, ly = pc;
    const int d1 = std::min(std::min(m - lx, n - ly), lazy);
    for (int t = 0; t <= d1; ++t) anchor(lx + t, ly + t);
    const int cB = ly + d1;
    for (int j = 0; j <= n; ++j) {
        if (j < cA - W || j > cB + W) continue;
        const int c0 = std::max(j - W, cA), c1 = std::min(j + W, cB);
        lo[j] = std::max(rmin[c0] - W, 0);
        hi[j] = std::min(rmax[c1] + W + 1, m + 1);
    }
    return 0;
}

int vtxs_band2_lanes(const uint8_t* x, int m, const uint8_t* y, int n, int log_cap, int sec_cap, int stash_cap, int32_t* lo, int32_t* hi,
                      int32_t* stats12) {
    g_lane_stats = 1;
    memset(g_row_start, 0, sizeof g_row_start); memset(g_row_end, 0, sizeof g_row_end);
    const int rc = vtxs_band2(x, m, y, n, log_cap, sec_cap, stash_cap, lo, hi, stats12);
    g_lane_stats = 0;
    return rc;
}

void vtxs_last_rows(int32_t* start272, int32_t* end272) { memcpy(start272, g_row_start, sizeof g_row_start); memcpy(end272, g_row_end, sizeof g_row_end); }

}  // extern "C"
