#include "../tests/fastcore/fastcore_host.cpp"
#include <stdio.h>
extern "C" int probe_stats(const vtx_batch* b, uint64_t* out) {
    using namespace vtxf;
    const uint32_t n_heads = 1024;
    uint32_t max_hap = 8;
    for (uint32_t l = 0; l < b->n_loci; ++l) max_hap = std::max(max_hap, std::max(b->loci[l].ref_len, b->loci[l].alt_len));
    const uint32_t stride = tab_stride(max_hap, n_heads);
    std::vector<uint8_t> gt((size_t)2 * stride + 64);
    std::vector<uint8_t> readbuf;
    uint32_t lane[2][LANE_WORDS];
    // out: 0 tasks, 1 whole, 2 live(non-whole ok front), 3 sum need rows, 4 sum pair-union rows, 5 presence hits (own rows), 6 presence hits (pair-union entries, both maps),
    // 7 sum ns, 8 triples (own), 9 triples(pair union), 10 hits excluding main-diag self hit, 11 sum r (pieces), 12 overhang rows, 13 nonunique intact rows, 14 rest rows
    // 20.. histogram of ns (0..40)
    for (uint32_t l = 0; l < b->n_loci; ++l) {
        const vtx_locus& L = b->loci[l];
        build_table(gt.data(), b->hap_arena + L.ref_off, L.ref_len, max_hap, n_heads);
        build_table(gt.data() + stride, b->hap_arena + L.alt_off, L.alt_len, max_hap, n_heads);
        for (uint32_t r = L.rec_begin; r < L.rec_begin + L.rec_count; ++r) {
            const vtx_record& R = b->records[r];
            readbuf.assign(R.read_len + 16, 0);
            memcpy(readbuf.data(), b->read_arena + R.read_off, R.read_len);
            M192 need[2]; bool live[2]; Tab tbs[2]; int ds[2];
            const int m = (int)R.read_len;
            for (int h = 0; h < 2; ++h) {
                Tab tb; tb.gt = gt.data(); tb.ent = (uint32_t)h * stride; tb.head = tb.ent + max_hap * 8;
                tb.bytes = tb.ent + tab_bytes_off(max_hap, n_heads); tb.uq = tb.ent + tab_uq_off(max_hap, n_heads);
                tb.pb = tb.ent + tab_pb_off(max_hap, n_heads); tb.hmask = n_heads - 1;
                tbs[h] = tb;
                const LaneS<uint16_t> ln{lane[h] + S_WORDS, 1, (uint16_t*)lane[h], 1};
                const int n = (int)(h ? L.alt_len : L.ref_len);
                const Front fr = front(readbuf.data(), m, tb, n, ln);
                out[0]++;
                live[h] = false; need[h] = m_zero(); ds[h] = fr.d;
                if (fr.why == W_OK && whole_read(fr, m)) { out[1]++; continue; }
                if (fr.why != W_OK) continue;
                out[2]++; live[h] = true; need[h] = fr.need; out[3] += m_pop(fr.need); out[11] += fr.r;
                // classify rows
                for (int row = 0; row + 6 <= m; ++row) if ((fr.need.w[row >> 6] >> (row & 63)) & 1) {
                    if (row + fr.d < 0 || row + fr.d + 6 > n) out[12]++;
                    else if (memcmp(readbuf.data() + row, b->hap_arena + (h ? L.alt_off : L.ref_off) + row + fr.d, 6) == 0) out[13]++;
                    else out[14]++;
                }
                // presence hits on own rows
                const uint32_t* pb = (const uint32_t*)(tb.gt + tb.pb);
                int trip = 0, last = -10;
                for (int row = 0; row + 6 <= m; ++row) if ((fr.need.w[row >> 6] >> (row & 63)) & 1) {
                    const uint64_t w8 = ld8(readbuf.data() + row);
                    const uint32_t code = kw_code((uint32_t)w8, (uint32_t)(w8 >> 32) & 0xffffu);
                    if ((pb[code >> 5] >> (code & 31)) & 1) out[5]++;
                    if (row > last + 2) { trip++; last = row; }
                }
                out[8] += trip;
                const int ns = probe_rows(readbuf.data(), tb, fr, ln);
                out[7] += ns; out[20 + std::min(ns, 41)]++;
            }
            M192 un; for (int k = 0; k < NW; ++k) un.w[k] = need[0].w[k] | need[1].w[k];
            out[4] += m_pop(un);
            int trip = 0, last = -10;
            for (int row = 0; row + 6 <= m; ++row) if ((un.w[row >> 6] >> (row & 63)) & 1) {
                const uint64_t w8 = ld8(readbuf.data() + row);
                const uint32_t code = kw_code((uint32_t)w8, (uint32_t)(w8 >> 32) & 0xffffu);
                for (int h = 0; h < 2; ++h) if (live[h]) { const uint32_t* pb = (const uint32_t*)(tbs[h].gt + tbs[h].pb); if ((pb[code >> 5] >> (code & 31)) & 1) out[6]++; }
                if (row > last + 2) { trip++; last = row; }
            }
            out[9] += trip;
        }
    }
    return 0;
}
