// vtx_inflate_core.h — raw DEFLATE (RFC 1951) of one BGZF block by ONE LANE: the per-block logic of bgzf_inflate_kernel (vtx_ingest.hip).
//
// What it replaces: rust-htslib -> htslib bgzf_read -> zlib inflate behind `bam.fetch(..)` / `bam.records()` (src/main.rs:822-830) — in
// the packer of round 1-5 (host/vtx_host.cpp) sixteen CPU threads inflating windows of blocks, 0.4 s of a 2.4 s run at config-3 scale.
// BGZF blocks are independent (<= 64 KiB of output each, ISIZE known), a 0.9 GB BAM has 45 000 of them: one lane per block, every block
// of the file in flight at once.  Lanes of a wavefront run in lock step, so the decoder is a flat state machine — one symbol OR one
// 8-byte slice of a match copy OR one slice of a stored block per trip — instead of nested loops (a lane copying a 258-byte match would
// hold 63 others at the loop's exit).
//
// Huffman decoding without lookup tables (there is no room for 64 lanes' tables in LDS): the canonical code of a table lives in 15
// registers, word j = (left-justified first code of length j + 1) << 16 | (index of that length's first symbol) << 4 | (j + 1).  The
// words ascend, so "the last word <= the next 15 input bits (bit-reversed, left-justified)" is a chain of 14 compare + select pairs;
// the symbol is one byte read (plus a bit for the ninth bit of a literal / length symbol) of the lane's symbol list sorted by
// (length, symbol) — 288 + 32 entries in LDS, 444 bytes per lane with the counters.
// The code lengths of a dynamic block are decoded TWICE (count per length, then place the symbols) instead of being stored.
//
// Strictness: what zlib would reject is rejected (over-subscribed or incomplete codes except the one-code distance table of RFC 1951
// 3.2.7, invalid symbols, distances beyond the output so far, a stream that does not end exactly at ISIZE, input that runs out); a
// rejected block makes the whole ingest fall back to the host packer, where zlib stays the authority (vtx_host.cpp: inflate_block).
// Nothing is written outside [out, out + out_len): the neighbouring block belongs to another lane.
//
// Compiles for the host too (tests/inflatecore/: every accepted stream equals zlib's output byte for byte; CPU suite, no GPU needed).
#ifndef VTX_INFLATE_CORE_H
#define VTX_INFLATE_CORE_H

#include <stdint.h>

#ifdef __HIPCC__
#define VTXI_FN __device__ __forceinline__
#define VTXI_MEM __device__ __forceinline__
#define VTXI_UNROLL _Pragma("unroll")
#else
#define VTXI_FN static inline
#define VTXI_MEM inline
#define VTXI_UNROLL
#endif

namespace vtxi {

// Per-lane scratch at a stride (LDS: element i of lane l at [i * 64 + l]): the symbol lists as BYTES — a literal / length symbol's
// ninth bit in a bit array beside them — and 16-bit counters.  444 bytes per lane (round 6, first cut: 744 as 16-bit words): 28.4 KB per
// wavefront, FIVE wavefronts per CU instead of three — the kernel is a latency chain per block, the lanes in flight are its throughput.
struct Scratch {
    uint8_t* b;        // [0, 288) literal / length symbols sorted by (code length, symbol), low byte; [288, 320) distance symbols; [320, 340) code-length symbols
    uint32_t* hi;      // 9 words: bit i = bit 8 of literal / length symbol i
    uint16_t* c;       // 32: per-length counters / cursors while a table is built (16 literal / length or code-length, 16 distance)
    int stride;
    VTXI_MEM uint32_t ll(uint32_t i) const { return (uint32_t)b[i * stride] | (((hi[(i >> 5) * stride] >> (i & 31u)) & 1u) << 8); }
    VTXI_MEM void set_ll(uint32_t i, uint32_t v) const { b[i * stride] = (uint8_t)v; if (v >> 8) hi[(i >> 5) * stride] |= 1u << (i & 31u); }
    VTXI_MEM void clear_hi() const { for (int w = 0; w < 9; ++w) hi[w * stride] = 0; }
    VTXI_MEM uint8_t& d(uint32_t i) const { return b[(288 + i) * stride]; }
    VTXI_MEM uint8_t& cl(uint32_t i) const { return b[(320 + i) * stride]; }
    VTXI_MEM uint16_t& at(int i) const { return c[i * stride]; }      // counters: CUR + l, CUR_D + l
};
constexpr int CUR = 0;         // 16: per-length counters / cursors (code-length code, literal / length table)
constexpr int CUR_D = 16;      // 16: the same for the distance table (both tables of a dynamic block are counted in one pass)
constexpr int BYTES = 344, HI_WORDS = 9, CNT_WORDS = 32;           // per lane: 344 + 36 + 64 = 444 bytes

enum Status : uint32_t { ST_OK = 0, ST_BAD_TYPE = 1, ST_BAD_STORED = 2, ST_BAD_CODE = 3, ST_BAD_SYMBOL = 4, ST_BAD_DIST = 5,
                         ST_OVERRUN = 6, ST_SHORT = 7, ST_INPUT = 8 };

VTXI_FN uint32_t ld4(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
VTXI_FN uint64_t ld8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
VTXI_FN void st8(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
VTXI_FN void st4(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
VTXI_FN void st2(uint8_t* p, uint32_t v) { const uint16_t w = (uint16_t)v; __builtin_memcpy(p, &w, 2); }
VTXI_FN uint32_t rev15(uint32_t v) {      // the low 15 bits, first bit of the stream on top
#ifdef __HIPCC__
    return __brev(v) >> 17;
#else
    uint32_t r = 0;
    for (int i = 0; i < 15; ++i) r |= ((v >> i) & 1u) << (14 - i);
    return r;
#endif
}

// A canonical Huffman code in registers.  p[j] ascending (see the header); n: symbols with a code.
struct Code {
    uint32_t p[15];
    uint32_t n;
};

// next symbol: its index in the table's sorted list (>= c.n: no such code) and its length
VTXI_FN void decode(const Code& c, uint32_t bits, uint32_t& idx, uint32_t& len) {
    const uint32_t rev = rev15(bits);
    const uint32_t key = (rev << 16) | 0xffffu;
    uint32_t sel = c.p[0];
    VTXI_UNROLL
    for (int j = 1; j < 15; ++j) sel = key >= c.p[j] ? c.p[j] : sel;
    len = sel & 15u;
    idx = ((sel >> 4) & 0xfffu) + ((rev - (sel >> 16)) >> (15u - len));
}

// The code of the lengths counted in sc.at(cur + 1 .. cur + 15); leaves the cursors (index of each length's first symbol) there.
// Returns 0: complete code; 1: incomplete (the caller decides: only a distance table with one code may be); 2: over-subscribed.
VTXI_FN int make_code(Code& c, const Scratch& sc, int cur, int max_len) {
    uint32_t first = 0, offs = 0;
    int left = 1;
    bool over = false;
    VTXI_UNROLL
    for (int l = 1; l <= 15; ++l) {
        const uint32_t cnt = l <= max_len ? sc.at(cur + l) : 0u;
        c.p[l - 1] = ((first << (15 - l)) << 16) | (offs << 4) | (uint32_t)l;
        if (l <= max_len) sc.at(cur + l) = (uint16_t)offs;
        left = left * 2 - (int)cnt;
        over |= left < 0;
        if (left < 0) left = 0;
        first = (first + cnt) << 1;
        offs += cnt;
    }
    c.n = offs;
    return over ? 2 : (left != 0 ? 1 : 0);
}

struct Bits {
    const uint8_t* in;       // compressed bytes of the block (readable for 8 bytes past in_len: the buffer is padded)
    uint32_t ip;             // next byte to load
    uint32_t cnt;            // valid bits in buf
    uint64_t buf;
    VTXI_MEM void refill() {  // >= 33 valid bits afterwards (the input may be exhausted: the bits beyond are whatever follows, and
        if (cnt <= 32) {     // `consumed() > in_len * 8` is checked where a block ends)
            buf |= (uint64_t)ld4(in + ip) << cnt;
            ip += 4; cnt += 32;
        }
    }
    VTXI_MEM uint32_t peek15() const { return (uint32_t)buf & 0x7fffu; }
    VTXI_MEM uint32_t take(uint32_t n) { const uint32_t v = (uint32_t)buf & ((1u << n) - 1u); buf >>= n; cnt -= n; return v; }
    VTXI_MEM void drop(uint32_t n) { buf >>= n; cnt -= n; }
    VTXI_MEM uint64_t consumed() const { return (uint64_t)ip * 8 - cnt; }
};

constexpr uint32_t S_HEADER = 0, S_SYM = 1, S_COPY = 2, S_STORED = 3, S_DONE = 4;

// length / distance symbols: base | extra bits << 16 (computed, no tables: lengths 257..285, distances 0..29)
VTXI_FN uint32_t len_base_extra(uint32_t s) {            // s = symbol - 257, 0..28
    if (s < 8) return 3 + s;
    if (s == 28) return 258;
    const uint32_t e = (s - 4) >> 2;                     // 1..5
    return (3 + ((4 + (s & 3)) << e)) | (e << 16);
}
VTXI_FN uint32_t dist_base_extra(uint32_t s) {           // 0..29
    if (s < 4) return 1 + s;
    const uint32_t e = (s - 2) >> 1;                     // 1..13
    return (1 + ((2 + (s & 1)) << e)) | (e << 16);
}

// Inflates in[0 .. in_len) into out[0 .. out_len); returns a Status.  sc: scratch private to this lane.
// trips (optional statistics): state-machine trips taken.
VTXI_FN uint32_t inflate_block(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, const Scratch& sc, uint32_t* trips) {
    Bits b{in, 0u, 0u, 0ull};
    Code ll, dc;
    for (int j = 0; j < 15; ++j) { ll.p[j] = 0; dc.p[j] = 0; }
    ll.n = dc.n = 0;
    uint32_t op = 0;                 // bytes written
    uint32_t st = S_HEADER, err = ST_OK;
    bool last = false;
    uint32_t rem = 0, src = 0;       // S_COPY: bytes left, source position in out; S_STORED: bytes left (source: b.ip)
    uint32_t dist = 0;
    uint32_t n_trips = 0;
    while (st != S_DONE) {
        ++n_trips;
        if (st == S_COPY) {
            // out[op .. op + n) = out[src .. src + n): up to 8 bytes a trip, never more than the distance (the source of the bytes
            // after that is what this trip writes)
            const uint32_t n = rem < 8u ? (dist < rem ? dist : rem) : (dist < 8u ? dist : 8u);
            const uint64_t w = ld8(out + src);     // (may read up to 7 bytes this lane has not written yet: they are not used)
            uint8_t* d = out + op;
            if (n == 8) st8(d, w);
            else {
                uint64_t v = w;
                if (n & 4) { st4(d, (uint32_t)v); d += 4; v >>= 32; }
                if (n & 2) { st2(d, (uint32_t)v); d += 2; v >>= 16; }
                if (n & 1) *d = (uint8_t)v;
            }
            op += n; src += n; rem -= n;
            if (rem == 0) st = S_SYM;
        } else if (st == S_SYM) {
            b.refill();
            uint32_t idx, len;
            decode(ll, b.peek15(), idx, len);
            if (idx >= ll.n) { err = ST_BAD_CODE; st = S_DONE; }
            else {
                b.drop(len);
                const uint32_t sym = sc.ll(idx);
                if (sym < 256) {
                    if (op >= out_len) { err = ST_OVERRUN; st = S_DONE; }
                    else out[op++] = (uint8_t)sym;
                } else if (sym == 256) {
                    if (b.consumed() > (uint64_t)in_len * 8) { err = ST_INPUT; st = S_DONE; }
                    else st = last ? S_DONE : S_HEADER;
                } else if (sym > 285) { err = ST_BAD_SYMBOL; st = S_DONE; }
                else {
                    const uint32_t lb = len_base_extra(sym - 257);
                    const uint32_t mlen = (lb & 0xffffu) + b.take(lb >> 16);
                    b.refill();
                    uint32_t didx, dlen;
                    decode(dc, b.peek15(), didx, dlen);
                    if (didx >= dc.n) { err = ST_BAD_CODE; st = S_DONE; }
                    else {
                        b.drop(dlen);
                        const uint32_t dsym = sc.d(didx);
                        if (dsym > 29) { err = ST_BAD_SYMBOL; st = S_DONE; }
                        else {
                            const uint32_t db = dist_base_extra(dsym);
                            dist = (db & 0xffffu) + b.take(db >> 16);
                            if (dist > op) { err = ST_BAD_DIST; st = S_DONE; }
                            else if (mlen > out_len - op) { err = ST_OVERRUN; st = S_DONE; }
                            else { rem = mlen; src = op - dist; st = S_COPY; }
                        }
                    }
                }
            }
        } else if (st == S_STORED) {
            // byte-aligned input: 8 bytes a trip
            const uint32_t n = rem < 8u ? rem : 8u;
            const uint64_t w = ld8(in + b.ip);
            uint8_t* d = out + op;
            if (n == 8) st8(d, w);
            else {
                uint64_t v = w;
                if (n & 4) { st4(d, (uint32_t)v); d += 4; v >>= 32; }
                if (n & 2) { st2(d, (uint32_t)v); d += 2; v >>= 16; }
                if (n & 1) *d = (uint8_t)v;
            }
            op += n; b.ip += n; rem -= n;
            if (rem == 0) st = last ? S_DONE : S_HEADER;
        } else {   // S_HEADER
            b.refill();
            last = b.take(1) != 0;
            const uint32_t type = b.take(2);
            if (type == 0) {
                b.drop(b.cnt & 7u);                      // to the byte boundary; the whole bytes still buffered go back to the input
                b.ip -= b.cnt >> 3;
                b.cnt = 0; b.buf = 0;
                if ((uint64_t)b.ip + 4 > in_len) { err = ST_INPUT; st = S_DONE; }
                else {
                    const uint32_t w = ld4(in + b.ip);
                    b.ip += 4;
                    const uint32_t slen = w & 0xffffu;
                    if ((slen ^ (w >> 16)) != 0xffffu) { err = ST_BAD_STORED; st = S_DONE; }
                    else if ((uint64_t)b.ip + slen > in_len) { err = ST_INPUT; st = S_DONE; }
                    else if (slen > out_len - op) { err = ST_OVERRUN; st = S_DONE; }
                    else { rem = slen; st = slen ? S_STORED : (last ? S_DONE : S_HEADER); }
                }
            } else if (type == 3) { err = ST_BAD_TYPE; st = S_DONE; }
            else if (type == 1) {
                // fixed code (RFC 1951 3.2.6): 7 bits 256..279, 8 bits 0..143 and 280..287, 9 bits 144..255; 32 distance codes of 5 bits
                for (int l = 1; l <= 15; ++l) sc.at(CUR + l) = 0;
                sc.at(CUR + 7) = 24; sc.at(CUR + 8) = 152; sc.at(CUR + 9) = 112;
                (void)make_code(ll, sc, CUR, 15);
                sc.clear_hi();
                for (uint32_t i = 0; i < 24; ++i) sc.set_ll(i, 256 + i);
                for (uint32_t i = 0; i < 144; ++i) sc.set_ll(24 + i, i);
                for (uint32_t i = 0; i < 8; ++i) sc.set_ll(168 + i, 280 + i);
                for (uint32_t i = 0; i < 112; ++i) sc.set_ll(176 + i, 144 + i);
                for (int l = 1; l <= 15; ++l) sc.at(CUR + l) = 0;
                sc.at(CUR + 5) = 32;
                (void)make_code(dc, sc, CUR, 15);
                for (uint32_t i = 0; i < 32; ++i) sc.d(i) = (uint8_t)i;
                st = S_SYM;
            } else {
                b.refill();
                const uint32_t nll = b.take(5) + 257, nd = b.take(5) + 1, ncl = b.take(4) + 4;
                if (nll > 286 || nd > 30) { err = ST_BAD_CODE; st = S_DONE; }
                else {
                    // the code-length code: 3 bits per symbol in the RFC's order; lengths packed 3 bits per symbol in one word
                    uint64_t cl = 0;
                    for (uint32_t i = 0; i < ncl; ++i) {
                        b.refill();
                        const uint32_t v = b.take(3);
                        // order: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (5 bits each, packed in two words)
                        const uint32_t s = i < 12 ? (uint32_t)((0x022caa324e804a30ull >> (5 * i)) & 31u)           // 16 17 18 0 8 7 9 6 10 5 11 4
                                                  : (uint32_t)((0x00000003c2e1346cull >> (5 * (i - 12))) & 31u);   // 12 3 13 2 14 1 15
                        cl |= (uint64_t)v << (3 * s);
                    }
                    for (int l = 0; l <= 7; ++l) sc.at(CUR + l) = 0;
                    for (int s = 0; s < 19; ++s) { const int l = (int)((cl >> (3 * s)) & 7u); if (l) sc.at(CUR + l) = (uint16_t)(sc.at(CUR + l) + 1); }
                    Code cc;
                    const int vc = make_code(cc, sc, CUR, 7);
                    if (vc != 0) { err = ST_BAD_CODE; st = S_DONE; }
                    else {
                        for (int s = 0; s < 19; ++s) {
                            const int l = (int)((cl >> (3 * s)) & 7u);
                            if (l) { const int k = sc.at(CUR + l); sc.at(CUR + l) = (uint16_t)(k + 1); sc.cl((uint32_t)k) = (uint8_t)s; }
                        }
                        // pass 0: count the lengths of both tables; pass 1 (the same bits again): place the symbols at their cursors
                        const Bits mark = b;
                        const uint32_t total = nll + nd;
                        bool bad = false, has_eob = false;
                        uint32_t d_used = 0, d_one_len = 0;
                        for (int l = 0; l <= 15; ++l) { sc.at(CUR + l) = 0; sc.at(CUR_D + l) = 0; }
                        for (int pass = 0; pass < 2 && !bad; ++pass) {
                            b = mark;
                            if (pass == 1) sc.clear_hi();
                            uint32_t k = 0, pv = 0;
                            while (k < total) {
                                b.refill();
                                uint32_t ci, cn;
                                decode(cc, b.peek15(), ci, cn);
                                if (ci >= cc.n) { bad = true; break; }
                                b.drop(cn);
                                const uint32_t cs = sc.cl(ci);
                                uint32_t rep = 1, val = cs;
                                if (cs == 16) { if (k == 0) { bad = true; break; } val = pv; rep = 3 + b.take(2); }
                                else if (cs == 17) { val = 0; rep = 3 + b.take(3); }
                                else if (cs == 18) { val = 0; rep = 11 + b.take(7); }
                                if (k + rep > total) { bad = true; break; }
                                if (val) {
                                    for (uint32_t t = 0; t < rep; ++t) {
                                        const uint32_t sy = k + t;
                                        const int cur = (sy < nll ? CUR : CUR_D) + (int)val;
                                        const int slot = sc.at(cur);
                                        sc.at(cur) = (uint16_t)(slot + 1);
                                        if (pass == 0) {
                                            if (sy == 256) has_eob = true;
                                            if (sy >= nll) { ++d_used; d_one_len = val; }
                                        } else if (sy < nll) sc.set_ll((uint32_t)slot, sy);
                                        else sc.d((uint32_t)slot) = (uint8_t)(sy - nll);
                                    }
                                }
                                k += rep;
                                pv = val;
                            }
                            if (bad) break;
                            if (pass == 0) {
                                const int v0 = make_code(ll, sc, CUR, 15);
                                const int v1 = make_code(dc, sc, CUR_D, 15);
                                if (v0 != 0 || !has_eob) bad = true;
                                else if (v1 == 2) bad = true;
                                else if (v1 == 1 && !(d_used == 0 || (d_used == 1 && d_one_len == 1))) bad = true;   // RFC 1951 3.2.7
                            }
                        }
                        if (bad) { err = ST_BAD_CODE; st = S_DONE; }
                        else st = S_SYM;
                    }
                }
            }
        }
    }
    if (trips) *trips = n_trips;
    if (err == ST_OK && op != out_len) err = ST_SHORT;
    if (err == ST_OK && b.consumed() > (uint64_t)in_len * 8) err = ST_INPUT;
    return err;
}

}  // namespace vtxi
#endif
