// fastcore_host.cpp — HOST build of the per-task logic of band_diag_kernel (vartrix_amd/csrc/vtx_fast_core.h), for the
// CPU unit tests of that logic against the oracle (tests/test_fastcore.py).  TEST INFRASTRUCTURE ONLY: the product
// (libvtx.so) never links or loads this; it exists because the container that runs `pytest -m "not gpu"` has no GPU and
// the kernel's decisions (which tasks it may score without a DP, and with what) must be checked on millions of cases.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/vtx.h"
#include "../../vartrix_amd/csrc/vtx_fast_core.h"
#include "../../vartrix_amd/csrc/vtx_band_trim.h"

namespace {
// serial restatement of build_tables (vtx_band.hip): same layout, same chain order (ascending y), same flags
void build_table(uint8_t* tb, const uint8_t* hy, uint32_t hn, uint32_t max_hap, uint32_t n_heads) {
    using namespace vtxf;
    memset(tb, 0, tab_stride(max_hap, n_heads));
    uint64_t* ent = (uint64_t*)tb;
    uint16_t* head = (uint16_t*)(tb + (size_t)max_hap * 8);
    uint8_t* bytes = tb + tab_bytes_off(max_hap, n_heads);
    uint8_t* fb = tb + tab_fb_off(max_hap, n_heads);
    uint32_t* uq = (uint32_t*)(tb + tab_uq_off(max_hap, n_heads)) + UQ_PAD_WORDS;
    uint32_t* pb = (uint32_t*)(tb + tab_pb_off(max_hap, n_heads));
    for (uint32_t i = 0; i < n_heads; ++i) head[i] = 0xffff;
    for (uint32_t y = 0; y < hn; ++y) { bytes[y] = hy[y]; fb[y] = hy[y] & 0x7f; }
    if (hn < 6) return;
    auto lo_of = [&](uint32_t y) { return (uint32_t)hy[y] | ((uint32_t)hy[y + 1] << 8) | ((uint32_t)hy[y + 2] << 16) | ((uint32_t)hy[y + 3] << 24); };
    auto hi_of = [&](uint32_t y) { return (uint32_t)hy[y + 4] | ((uint32_t)hy[y + 5] << 8); };
    std::vector<uint32_t> count(n_heads, 0), tag(n_heads, 0);
    for (int y = (int)hn - 6; y >= 0; --y) {
        const uint32_t hh = kw_mix(lo_of((uint32_t)y), hi_of((uint32_t)y));
        const uint32_t h = kw_bucket(hh, n_heads - 1);
        ent[y] = (uint64_t)lo_of((uint32_t)y) | ((uint64_t)hi_of((uint32_t)y) << 32) | ((uint64_t)head[h] << 48);
        head[h] = (uint16_t)y;
        ++count[h]; tag[h] = kw_tag(hh);
        const uint32_t code = kw_code(lo_of((uint32_t)y), hi_of((uint32_t)y));
        pb[code >> 5] |= 1u << (code & 31);
    }
    // tags: a bucket with one entry carries 4 hash bits of its k-mer, a bucket with more HEAD_MULTI
    for (uint32_t h = 0; h < n_heads; ++h)
        if (count[h]) head[h] = (uint16_t)(head[h] | ((count[h] == 1 ? tag[h] : HEAD_MULTI) << 12));
    for (uint32_t y = 0; y + 6 <= hn; ++y) {
        uint32_t same = 0;
        for (uint32_t z = 0; z + 6 <= hn; ++z) same += memcmp(hy + y, hy + z, 6) == 0;
        if (same == 1) { uq[y >> 5] |= 1u << (y & 31); fb[y + 5] |= 0x80; }
    }
    // the twin list (vtx_fast_core.h: tab_tw_off): pairs (y, y') of positions with the same k-mer, in (y, y') order
    uint8_t* tw = tb + tab_tw_off(max_hap, n_heads);
    bool hib = false;
    for (uint32_t y = 0; y < hn; ++y) hib |= (hy[y] & 0x80) != 0;
    uint32_t cnt = 0;
    bool ok = !hib && hn <= 256;
    for (uint32_t y = 0; ok && y + 6 <= hn; ++y)
        for (uint32_t z = 0; ok && z + 6 <= hn; ++z)
            if (z != y && memcmp(hy + y, hy + z, 6) == 0) {
                if (cnt == TW_MAX) { ok = false; break; }
                tw[8 + 2 * cnt] = (uint8_t)y; tw[9 + 2 * cnt] = (uint8_t)z; ++cnt;
            }
    if (!ok) memset(tw, 0, TW_BYTES);
    tw[0] = ok ? (uint8_t)cnt : (uint8_t)TW_NONE;
}
}  // namespace

extern "C" {
// the closed forms and the corridor DP of the run bound, for unit tests against brute force (tests/test_fastcore.py)
int vtxt_join_free(int D) { return vtxf::join_free(D); }
int vtxt_join_same(int D, int e) { return vtxf::join_same(D, e); }
int vtxt_join_gap3(int D) { return vtxf::join_gap3(D); }
int vtxt_corridor_cost(const uint8_t* x, int m, const uint8_t* y, int n, int xb, int d, int D, int mu_a, int mu_b) {
    return vtxf::corridor_cost(x, m, y, n, xb, d, D, mu_a, mu_b);
}
// diag_mask (the match mask of one diagonal) and the pieces / mismatch nibbles front_rest() derives from it, for a single read and
// haplotype: out[0..2] (and out[14]) = mask words, out[3] = number of main pieces, out[4] = zc, out[5] = certificate
int vtxt_front_of_diagonal(const uint8_t* x, int m, const uint8_t* y, int n, int d, uint64_t* out) {
    using namespace vtxf;
    const uint32_t max_hap = (uint32_t)std::max(n, 8), n_heads = 1024;
    std::vector<uint8_t> gt(tab_stride(max_hap, n_heads) + 64);
    build_table(gt.data(), y, (uint32_t)n, max_hap, n_heads);
    Tab tb;
    tb.gt = gt.data(); tb.ent = 0; tb.head = max_hap * 8; tb.bytes = tab_bytes_off(max_hap, n_heads);
    tb.uq = tab_uq_off(max_hap, n_heads); tb.pb = tab_pb_off(max_hap, n_heads); tb.hmask = n_heads - 1;
    std::vector<uint8_t> xb((size_t)m + 16, 0);
    memcpy(xb.data(), x, (size_t)m);
    const ReadWords rw = read_words(xb.data(), m);
    const M192 M = diag_mask(rw, xb.data(), m, tb, n, d);
    out[0] = M.w[0]; out[1] = M.w[1]; out[2] = M.w[2]; out[14] = NW > 3 ? M.w[NW - 1] : 0;     // (out[6 .. 13]: the pieces)
    uint32_t lane[LANE_WORDS];
    const LaneS<uint32_t> ln{lane + S_WORDS, 1, lane, 1};
    const Front fr = front_rest(xb.data(), m, tb, n, ln, d, M);
    out[3] = (uint64_t)fr.r | ((uint64_t)fr.why << 32); out[4] = fr.zc; out[5] = (uint64_t)(int64_t)fr.cert;
    for (int i = 0; i < RM; ++i) out[6 + i] = i < fr.r ? lane[S_WORDS + i] : 0;
    return 0;
}
// The probe phase for a single read and haplotype on diagonal d: need_out[0..3] = the rows front_rest() wants probed, then the
// off-diagonal matches probe_rows() finds (x << 16 | y, up to cap); returns their number (-1: front_rest declined)
int vtxt_probe_of_diagonal(const uint8_t* x, int m, const uint8_t* y, int n, int d, uint64_t* need_out, uint32_t* s_out, int cap) {
    using namespace vtxf;
    const uint32_t max_hap = (uint32_t)std::max(n, 8), n_heads = 1024;
    std::vector<uint8_t> gt(tab_stride(max_hap, n_heads) + 64);
    build_table(gt.data(), y, (uint32_t)n, max_hap, n_heads);
    Tab tb;
    tb.gt = gt.data(); tb.ent = 0; tb.head = max_hap * 8; tb.bytes = tab_bytes_off(max_hap, n_heads);
    tb.uq = tab_uq_off(max_hap, n_heads); tb.pb = tab_pb_off(max_hap, n_heads); tb.hmask = n_heads - 1;
    std::vector<uint8_t> xb((size_t)m + 16, 0);
    memcpy(xb.data(), x, (size_t)m);
    const ReadWords rw = read_words(xb.data(), m);
    const M192 M = diag_mask(rw, xb.data(), m, tb, n, d);
    uint32_t lane[LANE_WORDS];
    const LaneS<uint32_t> ln{lane + S_WORDS, 1, lane, 1};
    const Front fr = front_rest(xb.data(), m, tb, n, ln, d, M);
    if (fr.why != W_OK) return -1;
    for (int k = 0; k < 4; ++k) need_out[k] = k < NW ? fr.need.w[k < NW ? k : 0] : 0;
    // probe_rows() stops counting above the lane's capacity: probe the rows in slices so that every match is seen
    int total = 0;
    M192 need = fr.need;
    while (m_any(need)) {
        Front one = fr;
        one.need = m_zero();
        const int row = m_pop_lowest(need);
        one.need.w[row >> 6] = 1ull << (row & 63);
        const int ns = probe_rows(xb.data(), tb, one, ln);
        if (ns > LaneS<uint32_t>::SMAX) return -2;                    // more matches in ONE row than a lane holds
        for (int k = 0; k < ns && total < cap; ++k) s_out[total++] = lane[k];
    }
    return total;
}
// The off-diagonal matches of a read against a haplotype, found the two ways band_diag_kernel knows: every row that is not (intact and
// unique) probed (s_a), or the twin list + probes of the rows that are not intact (s_b); both sorted.  Returns na | nb << 16
// (0xffffffff: front declined; 0xfffffffe: the haplotype has no twin list; an entry count of 0xffff: more than cap)
uint32_t vtxt_twin_vs_probe(const uint8_t* x, int m, const uint8_t* y, int n, uint32_t* s_a, uint32_t* s_b, int cap) {
    using namespace vtxf;
    const uint32_t max_hap = (uint32_t)std::max(n, 8), n_heads = 1024;
    std::vector<uint8_t> gt(tab_stride(max_hap, n_heads) + 64);
    build_table(gt.data(), y, (uint32_t)n, max_hap, n_heads);
    Tab tb;
    tb.gt = gt.data(); tb.ent = 0; tb.head = max_hap * 8; tb.bytes = tab_bytes_off(max_hap, n_heads);
    tb.uq = tab_uq_off(max_hap, n_heads); tb.pb = tab_pb_off(max_hap, n_heads); tb.hmask = n_heads - 1;
    if (!tab_has_twins(tb)) return 0xfffffffeu;
    std::vector<uint8_t> xb((size_t)m + 16, 0);
    memcpy(xb.data(), x, (size_t)m);
    uint32_t la[LANE_WORDS], lb[LANE_WORDS];
    const LaneS<uint16_t> lna{la + S_WORDS, 1, (uint16_t*)la, 1}, lnb{lb + S_WORDS, 1, (uint16_t*)lb, 1};
    const Front fa = front(xb.data(), m, tb, n, lna, false), fb = front(xb.data(), m, tb, n, lnb, true);
    if (fa.why != W_OK || fb.why != W_OK) return 0xffffffffu;
    if (fa.d != fb.d || fa.cert != fb.cert || fa.r != fb.r) return 0xfffffffdu;
    int na = probe_rows(xb.data(), tb, fa, lna);
    int nb = probe_rows(xb.data(), tb, fb, lnb, twin_matches(tb, fb, m, lnb));
    if (na <= LaneS<uint16_t>::SMAX) back_sort(na, lna);
    if (nb <= LaneS<uint16_t>::SMAX) back_sort(nb, lnb);
    for (int k = 0; k < std::min(std::min(na, cap), (int)LaneS<uint16_t>::SMAX); ++k) s_a[k] = lna.s(k);
    for (int k = 0; k < std::min(std::min(nb, cap), (int)LaneS<uint16_t>::SMAX); ++k) s_b[k] = lnb.s(k);
    return (uint32_t)(na > LaneS<uint16_t>::SMAX ? 0xffff : na) | ((uint32_t)(nb > LaneS<uint16_t>::SMAX ? 0xffff : nb) << 16);
}
static uint32_t g_last_pack = 0;
// The harmless verdict for a single read and haplotype: 1 = every off-diagonal match is harmless (then the reference's chain lies on
// the main diagonal *d_out), 0 = not, -1 = the logic declined before that (no diagonal, capacities)
int vtxt_harmless(const uint8_t* x, int m, const uint8_t* y, int n, int* d_out, int* cert_out) {
    using namespace vtxf;
    const uint32_t max_hap = (uint32_t)std::max(n, 8), n_heads = 1024;
    std::vector<uint8_t> gt(tab_stride(max_hap, n_heads) + 64);
    build_table(gt.data(), y, (uint32_t)n, max_hap, n_heads);
    Tab tb;
    tb.gt = gt.data(); tb.ent = 0; tb.head = max_hap * 8; tb.bytes = tab_bytes_off(max_hap, n_heads);
    tb.uq = tab_uq_off(max_hap, n_heads); tb.pb = tab_pb_off(max_hap, n_heads); tb.hmask = n_heads - 1;
    std::vector<uint8_t> xb((size_t)m + 16, 0);
    memcpy(xb.data(), x, (size_t)m);
    uint32_t lane[LANE_WORDS];
    const LaneS<uint32_t> ln{lane + S_WORDS, 1, lane, 1};
    const Front fr = front(xb.data(), m, tb, n, ln);
    if (fr.why != W_OK) return -1;
    const int ns = probe_rows(xb.data(), tb, fr, ln);
    if (ns > LaneS<uint32_t>::SMAX) return -1;
    back_sort(ns, ln);
    *d_out = fr.d;
    *cert_out = fr.cert;
    g_last_pack = band_pack(fr);
    return back_harmless(fr, ns, ln) ? 1 : 0;
}
// vtxf::band_pack of the last vtxt_harmless call: (d + 256) << 16 | ca << 8 | cb — the one-diagonal band sw_banded_kernel<.., 2> expands
uint32_t vtxt_last_band_pack(void) { return g_last_pack; }
// Per task t = 2 * record + hap of a packed batch: score[t] (-1: left to band_run_kernel) and why[t].
// n_heads bit 31: use the four-byte match entries (20 per task) even when every haplotype has <= 255 bases — the variant the
// device takes for longer haplotypes.
int vtxt_fastcore_batch(const vtx_batch* b, uint32_t n_heads, int32_t* score, uint32_t* why) {
    const bool force_wide = (n_heads >> 31) != 0;
    const bool refine = ((n_heads >> 30) & 1u) != 0;          // bit 30: the corridor refinement of band_refine_kernel
    const bool corridor = ((n_heads >> 29) & 1u) != 0;        // bit 29: the corridor certificate of band_corridor_kernel (round 6) behind the run bound
    const bool twins = ((n_heads >> 28) & 1u) != 0;           // bit 28 (with bit 29): the twin list instead of probes of the rows with an intact k-mer, as band_diag_kernel does
    n_heads &= 0x0fffffffu;
    using namespace vtxf;
    uint32_t max_hap = 8;
    for (uint32_t l = 0; l < b->n_loci; ++l) max_hap = std::max(max_hap, std::max(b->loci[l].ref_len, b->loci[l].alt_len));
    const uint32_t stride = tab_stride(max_hap, n_heads);
    std::vector<uint8_t> gt((size_t)2 * stride + 64);
    std::vector<uint8_t> readbuf;
    uint32_t lane[LANE_WORDS], generic[GM];
    const bool narrow = max_hap <= 255 && !force_wide;          // (vtxk_launch_band_diag makes the same choice)
    for (uint32_t l = 0; l < b->n_loci; ++l) {
        const vtx_locus& L = b->loci[l];
        build_table(gt.data(), b->hap_arena + L.ref_off, L.ref_len, max_hap, n_heads);
        build_table(gt.data() + stride, b->hap_arena + L.alt_off, L.alt_len, max_hap, n_heads);
        for (uint32_t r = L.rec_begin; r < L.rec_begin + L.rec_count; ++r) {
            const vtx_record& R = b->records[r];
            readbuf.assign(R.read_len + 16, 0);
            memcpy(readbuf.data(), b->read_arena + R.read_off, R.read_len);
            for (int h = 0; h < 2; ++h) {
                Tab tb;
                tb.gt = gt.data();
                tb.ent = (uint32_t)h * stride;
                tb.head = tb.ent + max_hap * 8;
                tb.bytes = tb.ent + tab_bytes_off(max_hap, n_heads);
                tb.uq = tb.ent + tab_uq_off(max_hap, n_heads);
                tb.pb = tb.ent + tab_pb_off(max_hap, n_heads);
                tb.hmask = n_heads - 1;
                Result res;
                if (narrow && corridor) {
                    // what band_diag_kernel + band_corridor_kernel do with a task: the three phases, and for a task that leaves them with a
                    // certificate and harmless matches only (the kernel's tight list) the corridor bound over its one-diagonal band
                    const LaneS<uint16_t> ln{lane + S_WORDS, 1, (uint16_t*)lane, 1};
                    const int m = (int)R.read_len, n = (int)(h ? L.alt_len : L.ref_len);
                    const bool tw = twins && tab_has_twins(tb);
                    const Front fr = front(readbuf.data(), m, tb, n, ln, tw);
                    res = Result{-1, fr.why};
                    if (fr.why == W_OK && whole_read(fr, m)) res = Result{m, W_OK};
                    else if (fr.why == W_OK) {
                        const int ns = probe_rows(readbuf.data(), tb, fr, ln, tw ? twin_matches(tb, fr, m, ln) : 0);
                        uint32_t wy = W_OK;
                        const int32_t sc = back(fr, ns, ln, Lane{generic, 1}, &wy, 0, nullptr);
                        res = Result{sc, wy};
                        if (sc < 0 && (wy == W_NOT_TIGHT || wy == W_GENERIC)) {
                            const M192 far = far_rows(fr.d, ns, ln);
                            const int cs = corridor_bound(readbuf.data(), m, tb.gt + tb.bytes, n, fr.d, fr.ca, fr.cb, far);
                            if (cs >= 0) res = Result{cs, W_OK};
                        }
                    }
                } else if (narrow) {
                    const LaneS<uint16_t> ln{lane + S_WORDS, 1, (uint16_t*)lane, 1};
                    res = fast_task(readbuf.data(), (int)R.read_len, tb, (int)(h ? L.alt_len : L.ref_len), ln, Lane{generic, 1}, refine);
                } else {
                    const LaneS<uint32_t> ln{lane + S_WORDS, 1, lane, 1};
                    res = fast_task(readbuf.data(), (int)R.read_len, tb, (int)(h ? L.alt_len : L.ref_len), ln, Lane{generic, 1}, refine);
                }
                score[2 * (size_t)r + h] = res.score;
                why[2 * (size_t)r + h] = res.why;
            }
        }
    }
    return 0;
}

// ---- the SECOND stage (vtxf::fast_task2: band_diag2_kernel's per-task logic) ----
// one read against one haplotype: returns the verdict (vtxf::T2Verdict), *score (T2_SCORE: the score; T2_TIGHT: the certificate),
// *pack (T2_TIGHT: vtxf::band_pack) and *why
int vtxt_task2(const uint8_t* x, int m, const uint8_t* y, int n, int32_t* score, uint32_t* pack, uint32_t* why) {
    using namespace vtxf;
    const uint32_t max_hap = (uint32_t)std::max(n, 8), n_heads = 1024;
    std::vector<uint8_t> gt(tab_stride(max_hap, n_heads) + 64);
    build_table(gt.data(), y, (uint32_t)n, max_hap, n_heads);
    Tab tb;
    tb.gt = gt.data(); tb.ent = 0; tb.head = max_hap * 8; tb.bytes = tab_bytes_off(max_hap, n_heads);
    tb.uq = tab_uq_off(max_hap, n_heads); tb.pb = tab_pb_off(max_hap, n_heads); tb.hmask = n_heads - 1;
    std::vector<uint8_t> xb((size_t)m + 16, 0);
    memcpy(xb.data(), x, (size_t)m);
    uint32_t lane[S2_WORDS + RM], generic[GM];
    uint8_t ub[LaneS2::SMAX];
    const LaneS2 ln{lane + S2_WORDS, 1, (uint16_t*)lane, 1, ub, 1};
    const Result2 r = n <= 255 ? fast_task2(xb.data(), m, tb, n, ln, Lane{generic, 1}) : Result2{T2_SWEEP, -1, 0u, W_SHAPE};
    *score = r.score; *pack = r.pack; *why = r.why;
    return (int)r.verdict;
}
// every task of a packed batch: verdict[t], score[t], pack[t] (t = 2 * record + hap); haplotypes above 255 bases: T2_SWEEP
int vtxt_fastcore2_batch(const vtx_batch* b, uint32_t n_heads, uint8_t* verdict, int32_t* score, uint32_t* pack, uint32_t* why) {
    using namespace vtxf;
    uint32_t max_hap = 8;
    for (uint32_t l = 0; l < b->n_loci; ++l) max_hap = std::max(max_hap, std::max(b->loci[l].ref_len, b->loci[l].alt_len));
    const uint32_t stride = tab_stride(max_hap, n_heads);
    std::vector<uint8_t> gt((size_t)2 * stride + 64);
    std::vector<uint8_t> readbuf;
    uint32_t lane[S2_WORDS + RM], generic[GM];
    uint8_t ub[LaneS2::SMAX];
    for (uint32_t l = 0; l < b->n_loci; ++l) {
        const vtx_locus& L = b->loci[l];
        build_table(gt.data(), b->hap_arena + L.ref_off, L.ref_len, max_hap, n_heads);
        build_table(gt.data() + stride, b->hap_arena + L.alt_off, L.alt_len, max_hap, n_heads);
        for (uint32_t r = L.rec_begin; r < L.rec_begin + L.rec_count; ++r) {
            const vtx_record& R = b->records[r];
            readbuf.assign(R.read_len + 16, 0);
            memcpy(readbuf.data(), b->read_arena + R.read_off, R.read_len);
            for (int h = 0; h < 2; ++h) {
                Tab tb;
                tb.gt = gt.data();
                tb.ent = (uint32_t)h * stride;
                tb.head = tb.ent + max_hap * 8;
                tb.bytes = tb.ent + tab_bytes_off(max_hap, n_heads);
                tb.uq = tb.ent + tab_uq_off(max_hap, n_heads);
                tb.pb = tb.ent + tab_pb_off(max_hap, n_heads);
                tb.hmask = n_heads - 1;
                const LaneS2 ln{lane + S2_WORDS, 1, (uint16_t*)lane, 1, ub, 1};
                const int n = (int)(h ? L.alt_len : L.ref_len);
                const Result2 res = max_hap <= 255 ? fast_task2(readbuf.data(), (int)R.read_len, tb, n, ln, Lane{generic, 1})
                                                   : Result2{T2_SWEEP, -1, 0u, W_SHAPE};
                const size_t t = 2 * (size_t)r + h;
                verdict[t] = (uint8_t)res.verdict; score[t] = res.score; pack[t] = res.pack; why[t] = res.why;
            }
        }
    }
    return 0;
}

// the streaming harmless test against the list version on one read / haplotype (both with the second stage's bounds): returns
// list_verdict | stream_verdict << 8 | 0x10000 when the list held every match (0xffffffff: front declined); verdicts: 1 harmless, 0 not, 2: -1
uint32_t vtxt_harmless_stream_vs_list(const uint8_t* x, int m, const uint8_t* y, int n, int win) {
    using namespace vtxf;
    const uint32_t max_hap = (uint32_t)std::max(n, 8), n_heads = 1024;
    std::vector<uint8_t> gt(tab_stride(max_hap, n_heads) + 64);
    build_table(gt.data(), y, (uint32_t)n, max_hap, n_heads);
    Tab tb;
    tb.gt = gt.data(); tb.ent = 0; tb.head = max_hap * 8; tb.bytes = tab_bytes_off(max_hap, n_heads);
    tb.uq = tab_uq_off(max_hap, n_heads); tb.pb = tab_pb_off(max_hap, n_heads); tb.hmask = n_heads - 1;
    std::vector<uint8_t> xb((size_t)m + 16, 0);
    memcpy(xb.data(), x, (size_t)m);
    uint32_t lane[S2_WORDS + RM];
    uint8_t ub[LaneS2::SMAX];
    const LaneS2 ln{lane + S2_WORDS, 1, (uint16_t*)lane, 1, ub, 1};
    const Front fr = front(xb.data(), m, tb, n, ln);
    if (fr.why != W_OK) return 0xffffffffu;
    const int ns = probe_rows(xb.data(), tb, fr, ln);
    uint32_t lv = 3;
    if (ns <= LaneS2::SMAX) { back_sort(ns, ln); lv = back_harmless(fr, ns, ln) ? 1u : 0u; }
    const int sv = probe_harmless_stream(xb.data(), tb, fr, ln, win > 0 ? win : LaneS2::SMAX / 2);
    return lv | ((uint32_t)(sv < 0 ? 2 : sv) << 8) | (ns <= LaneS2::SMAX ? 0x10000u : 0u);
}

// vtxt_fastcore_batch with the band-trimmed bound behind it (vtx_band_trim.h; refinement on): score[t] (-1: undecided), why[t], and
// trimmed[t] = 1 where the trimmed bound decided the task (its score is the BANDED score, which may be below the full-matrix one)
int vtxt_fastcore_trim_batch(const vtx_batch* b, uint32_t n_heads, int32_t* score, uint32_t* why, uint8_t* trimmed) {
    using namespace vtxf;
    uint32_t max_hap = 8;
    for (uint32_t l = 0; l < b->n_loci; ++l) max_hap = std::max(max_hap, std::max(b->loci[l].ref_len, b->loci[l].alt_len));
    if (max_hap > 255) return -1;
    const uint32_t stride = tab_stride(max_hap, n_heads);
    std::vector<uint8_t> gt((size_t)2 * stride + 64);
    std::vector<uint8_t> readbuf;
    uint32_t lane[LANE_WORDS], generic[GM];
    for (uint32_t l = 0; l < b->n_loci; ++l) {
        const vtx_locus& L = b->loci[l];
        build_table(gt.data(), b->hap_arena + L.ref_off, L.ref_len, max_hap, n_heads);
        build_table(gt.data() + stride, b->hap_arena + L.alt_off, L.alt_len, max_hap, n_heads);
        for (uint32_t r = L.rec_begin; r < L.rec_begin + L.rec_count; ++r) {
            const vtx_record& R = b->records[r];
            readbuf.assign(R.read_len + 16, 0);
            memcpy(readbuf.data(), b->read_arena + R.read_off, R.read_len);
            for (int h = 0; h < 2; ++h) {
                Tab tb;
                tb.gt = gt.data();
                tb.ent = (uint32_t)h * stride;
                tb.head = tb.ent + max_hap * 8;
                tb.bytes = tb.ent + tab_bytes_off(max_hap, n_heads);
                tb.uq = tb.ent + tab_uq_off(max_hap, n_heads);
                tb.pb = tb.ent + tab_pb_off(max_hap, n_heads);
                tb.hmask = n_heads - 1;
                const size_t t = 2 * (size_t)r + h;
                const int m = (int)R.read_len, n = (int)(h ? L.alt_len : L.ref_len);
                const LaneS<uint16_t> ln{lane + S_WORDS, 1, (uint16_t*)lane, 1};
                const uint8_t* x = readbuf.data();
                trimmed[t] = 0;
                const Front fr = front(x, m, tb, n, ln);
                if (fr.why != W_OK) { score[t] = -1; why[t] = fr.why; continue; }
                if (whole_read(fr, m)) { score[t] = m; why[t] = W_OK; continue; }
                const int ns = probe_rows(x, tb, fr, ln);
                uint32_t w = W_OK, aux = 0xffffffffu;
                const Refine rf{x, tb.gt + tb.bytes, m, n};
                int32_t sc = back(fr, ns, ln, Lane{generic, 1}, &w, 0, &rf, &aux);
                if (sc < 0 && w == W_NOT_TIGHT && aux != 0xffffffffu) {
                    // the kernel's fused pass (main_pieces_ub_both) must give what the two separate passes give
                    uint32_t copy[LANE_WORDS];
                    memcpy(copy, lane, sizeof copy);
                    const LaneS<uint16_t> ln2{copy + S_WORDS, 1, (uint16_t*)copy, 1};
                    int lo, hi, ubb2 = 0;
                    band_rows(fr, m, lo, hi);
                    const int ubf2 = main_pieces_ub_both(ln2, fr.r, fr.zc, fr.d, &rf, (int)aux, lo, hi, &ubb2);
                    const int ubf1 = main_pieces_ub(ln, fr.r, fr.zc, fr.d, &rf, (int)aux);
                    const int ubb1 = main_pieces_ub_band(ln, fr.r, fr.zc, fr.d, &rf, lo, hi, (int)aux);
                    if (ubf1 != ubf2 || ubb1 != ubb2) return -2;
                    sc = band_trim_verdict(fr, m, ln, aux, &rf);
                    if (sc >= 0) { w = W_OK; trimmed[t] = 1; }
                }
                score[t] = sc; why[t] = w;
            }
        }
    }
    return 0;
}
}
