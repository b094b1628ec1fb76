// vtx_inflate.h — one-shot raw DEFLATE (RFC 1951) decoder for BGZF blocks: the whole compressed block and the whole output
// buffer are in memory and the output size is known (BGZF's ISIZE), which is everything a streaming inflater spends its
// generality on.  The reference gets its bytes through rust-htslib -> htslib's bgzf_read -> zlib (src/main.rs:822-830 is the
// loop that consumes them); here the packer's workers inflate the blocks of a window in parallel (vtx_host.cpp: refill) and
// zlib's inflate() was a third to a half of the ingest.  Written from the RFC:
//   * a 64-bit bit buffer refilled by one unaligned load;
//   * one table lookup per symbol: 11 index bits for literal / length codes (longer codes through a subtable), 8 for distances;
//   * a fast loop while at least 8 input bytes and 320 output bytes remain (literals stored directly, matches copied eight bytes
//     at a time), an exact loop for the tail — nothing is ever written outside [out, out + out_len): the neighbouring block is
//     being written by another thread.
// Anything unusual — an over-subscribed or incomplete code, a distance beyond the output so far, a block that does not end
// exactly at out_len, input that runs out — returns false and the caller hands the block to zlib, which stays the authority on
// malformed input (tests/test_host.py: every accepted block equals zlib's output byte for byte).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace vtxinf {

constexpr int LL_BITS = 11, D_BITS = 8;
constexpr uint32_t F_LIT = 0x8000u, F_EOB = 0x4000u, F_LINK = 0x2000u, F_BAD = 0x1000u;
// entry: bits 0-7 code length (link: LL_BITS / D_BITS), bits 8-11 extra bits (link: subtable index bits), flags, bits 16-31
// literal / base value / subtable offset

struct Tables {
    uint32_t ll[(1 << LL_BITS) + 1024];      // 2048 primary + subtables (at most 286 long codes, <= 16 entries each... bounded below)
    uint32_t d[(1 << D_BITS) + 512];
};

static inline uint32_t rev_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

// canonical Huffman table from code lengths; value(sym) gives bits 8-31 of the symbol's entry (flags, extra bits, base).
// complete = the code must be complete (Kraft sum exactly 1); a single-code distance set is the caller's special case.
template <class V>
static bool build(const uint8_t* lens, int n, int tbits, uint32_t* tab, int tab_cap, V value) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    if (count[0] == n) return false;
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left = left * 2 - count[l]; if (left < 0) return false; }
    if (left != 0) return false;                                   // incomplete: zlib decides
    uint32_t next[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
    next[0] = 0;
    const int prim = 1 << tbits;
    for (int i = 0; i < prim; ++i) tab[i] = F_BAD;
    // subtable sizes: the longest code under each primary prefix
    uint8_t sub_bits[1 << LL_BITS];
    memset(sub_bits, 0, (size_t)prim);
    uint32_t codes[320];
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = rev_bits(next[l]++, l);
        codes[s] = c;
        if (l > tbits) {
            const uint32_t p = c & (uint32_t)(prim - 1);
            if (l - tbits > sub_bits[p]) sub_bits[p] = (uint8_t)(l - tbits);
        }
    }
    int used = prim;
    for (int p = 0; p < prim; ++p) {
        if (!sub_bits[p]) continue;
        const int sz = 1 << sub_bits[p];
        if (used + sz > tab_cap) return false;
        tab[p] = (uint32_t)tbits | ((uint32_t)sub_bits[p] << 8) | F_LINK | ((uint32_t)used << 16);
        for (int i = 0; i < sz; ++i) tab[used + i] = F_BAD;
        used += sz;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = codes[s];
        const uint32_t v = value(s);
        if (l <= tbits) {
            const uint32_t e = (uint32_t)l | v;
            for (uint32_t i = c; i < (uint32_t)prim; i += 1u << l) tab[i] = e;
        } else {
            const uint32_t p = c & (uint32_t)(prim - 1);
            const uint32_t link = tab[p];
            const int sb = (int)((link >> 8) & 0xfu);
            const uint32_t base = link >> 16;
            const uint32_t e = (uint32_t)(l - tbits) | v;
            for (uint32_t i = c >> tbits; i < (1u << sb); i += 1u << (l - tbits)) tab[base + i] = e;
        }
    }
    return true;
}

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t ll_value(int s) {
    if (s < 256) return F_LIT | ((uint32_t)s << 16);
    if (s == 256) return F_EOB;
    if (s > 285) return F_BAD;
    return ((uint32_t)kLenExtra[s - 257] << 8) | ((uint32_t)kLenBase[s - 257] << 16);
}
static inline uint32_t d_value(int s) {
    if (s > 29) return F_BAD;
    return ((uint32_t)kDistExtra[s] << 8) | ((uint32_t)kDistBase[s] << 16);
}

struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t buf = 0;
    int cnt = 0;
    bool over = false;                    // more bits were asked for than the input holds
    inline void refill() {
        if (end - p >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(int n) { if (n > cnt) { over = true; n = cnt; } buf >>= n; cnt -= n; }
    inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

static inline bool inflate_raw(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, Tables& T) {
    Bits b;
    b.p = in; b.end = in + in_len;
    uint8_t* o = out;
    uint8_t* const oend = out + out_len;
    bool last = false;
    static const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    while (!last) {
        b.refill();
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (b.over) return false;
        if (type == 0) {
            b.drop(b.cnt & 7);                                     // to the byte boundary
            b.refill();
            if (b.cnt < 32) return false;
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ nlen) != 0xffffu) return false;
            // the bytes still in the bit buffer come first
            uint32_t left = len;
            if ((size_t)(oend - o) < left) return false;
            while (left && b.cnt >= 8) { *o++ = (uint8_t)b.take(8); --left; }
            if (left) {
                if (b.cnt != 0 || (size_t)(b.end - b.p) < left) return false;
                memcpy(o, b.p, left);
                b.p += left; o += left;
                b.buf = 0;                                         // (bits a refill left above cnt belonged to the bytes just skipped)
            }
            continue;
        }
        if (type == 3) return false;
        uint8_t lens[320];
        int nll, nd;
        if (type == 1) {
            nll = 288; nd = 30;
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 30; ++i) lens[288 + i] = 5;
            // (the fixed distance code has 32 five-bit codes, two of them unused: fill the table by hand below)
        } else {
            b.refill();
            nll = (int)b.take(5) + 257; nd = (int)b.take(5) + 1;
            const int ncl = (int)b.take(4) + 4;
            if (nll > 286 || nd > 30) return false;
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncl; ++i) { if (b.cnt < 3) b.refill(); cl[kOrder[i]] = (uint8_t)b.take(3); }
            if (b.over) return false;
            uint32_t pre[128 + 64];
            if (!build(cl, 19, 7, pre, 128 + 64, [](int s) { return (uint32_t)s << 16; })) return false;
            int i = 0;
            while (i < nll + nd) {
                b.refill();
                const uint32_t e = pre[b.peek(7)];
                if (e & (F_BAD | F_LINK)) return false;
                b.drop((int)(e & 0xffu));
                const int sym = (int)(e >> 16);
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                int rep; uint8_t v = 0;
                if (sym == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + (int)b.take(2); }
                else if (sym == 17) rep = 3 + (int)b.take(3);
                else rep = 11 + (int)b.take(7);
                if (i + rep > nll + nd) return false;
                while (rep--) lens[i++] = v;
            }
            if (b.over || lens[256] == 0) return false;
        }
        if (!build(lens, nll, LL_BITS, T.ll, (int)(sizeof T.ll / sizeof T.ll[0]), ll_value)) return false;
        if (type == 1) {
            for (uint32_t c = 0; c < 32; ++c) {
                const uint32_t idx = rev_bits(c, 5);
                const uint32_t e = 5u | d_value((int)c);
                for (uint32_t i2 = idx; i2 < (1u << D_BITS); i2 += 32) T.d[i2] = e;
            }
        } else {
            const uint8_t* dl = lens + nll;
            int used = 0, one = -1;
            for (int i = 0; i < nd; ++i) if (dl[i]) { ++used; one = i; }
            if (used == 0) {
                for (int i = 0; i < (1 << D_BITS); ++i) T.d[i] = F_BAD;          // literals only: any distance is an error
            } else if (used == 1 && dl[one] == 1) {
                // one distance code of one bit (RFC 1951 3.2.7): the other bit pattern is invalid
                for (int i = 0; i < (1 << D_BITS); ++i) T.d[i] = (i & 1) ? F_BAD : (1u | d_value(one));
            } else if (!build(dl, nd, D_BITS, T.d, (int)(sizeof T.d / sizeof T.d[0]), d_value)) return false;
        }
        // ---- symbols ----
        for (;;) {
            const bool fast = (b.end - b.p) >= 16 && (oend - o) >= 320;
            b.refill();
            uint32_t e = T.ll[b.peek(LL_BITS)];
            if (e & F_LINK) {
                b.drop(LL_BITS);
                e = T.ll[(e >> 16) + b.peek((int)((e >> 8) & 0xfu))];
            }
            if (e & F_BAD) return false;
            b.drop((int)(e & 0xffu));
            if (e & F_LIT) {
                if (o >= oend) return false;
                *o++ = (uint8_t)(e >> 16);
                if (fast) {
                    // up to two more literals on the bits already loaded (>= 56 - 15 left)
                    uint32_t e2 = T.ll[b.peek(LL_BITS)];
                    if ((e2 & (F_LIT | F_LINK)) == F_LIT) {
                        b.drop((int)(e2 & 0xffu));
                        *o++ = (uint8_t)(e2 >> 16);
                        e2 = T.ll[b.peek(LL_BITS)];
                        if ((e2 & (F_LIT | F_LINK)) == F_LIT) { b.drop((int)(e2 & 0xffu)); *o++ = (uint8_t)(e2 >> 16); }
                    }
                }
                continue;
            }
            if (e & F_EOB) break;
            // length
            const int xb = (int)((e >> 8) & 0xfu);
            uint32_t len = (e >> 16) + b.take(xb);
            if (b.cnt < 32) b.refill();
            uint32_t de = T.d[b.peek(D_BITS)];
            if (de & F_LINK) {
                b.drop(D_BITS);
                de = T.d[(de >> 16) + b.peek((int)((de >> 8) & 0xfu))];
            }
            if (de & F_BAD) return false;
            b.drop((int)(de & 0xffu));
            const uint32_t dist = (de >> 16) + b.take((int)((de >> 8) & 0xfu));
            if (b.over) return false;
            if (dist > (size_t)(o - out) || len > (size_t)(oend - o)) return false;
            const uint8_t* s = o - dist;
            if (fast && dist >= 8) {
                // (len <= 258 and 320 bytes are free: the last 8-byte store stays inside the block)
                uint8_t* t = o;
                uint8_t* const te = o + len;
                do { uint64_t w; memcpy(&w, s, 8); memcpy(t, &w, 8); s += 8; t += 8; } while (t < te);
                o = te;
            } else {
                for (uint32_t i = 0; i < len; ++i) o[i] = s[i];
                o += len;
            }
        }
        if (b.over) return false;
    }
    return o == oend && !b.over;
}

}  // namespace vtxinf
