// vtx_host.cpp — host-side ingest + filter + pack (libvtxhost.so), see include/vtx_host.h.
//
// MI355X-first structure rather than a transliteration of the reference's
// per-locus loop: the reference does one htslib index seek + BGZF inflate per
// locus (src/main.rs:822) and re-opens the FASTA per locus (:661).  Here the BAM
// is inflated block-parallel and swept ONCE in file order; every record is
// joined against the per-contig sorted locus intervals, so each locus receives
// its reads in exactly the order `bam.fetch(...).records()` yields them.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/vtx_host.h"
#include "vtx_inflate.h"

// Test hooks (window / batch sizes that force the many-window and many-batch paths, the zlib-only inflater, the buffer pool's
// thresholds) exist only in libvtxhost_dev.so (-DVTX_DEVTOOLS; tests/ load it).  The production library reads one environment
// variable, VTXH_PROFILE (phase timings on stderr; changes no result).
#ifdef VTX_DEVTOOLS
#define VTXH_DEV_ENV(name) getenv(name)
#else
#define VTXH_DEV_ENV(name) ((const char*)nullptr)
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

bool ends_with(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// ---- whole-file readers -------------------------------------------------------
bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0);
    out.resize((size_t)n);
    if (n) f.read(&out[0], n);
    return (bool)f || n == 0;
}

bool read_gz(const std::string& path, std::string& out) {   // MultiGzDecoder / plain-text transparent
    gzFile g = gzopen(path.c_str(), "rb");
    if (!g) return false;
    char buf[1 << 16];
    int n;
    out.clear();
    while ((n = gzread(g, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
    gzclose(g);
    return n == 0;
}

// read-only mapping of a (possibly huge) input file: BAM and FASTA are never copied into host memory
struct MappedFile {
    const unsigned char* p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const unsigned char*)m;
        madvise(m, n, MADV_SEQUENTIAL);
        return true;
    }
    ~MappedFile() {
        if (p) munmap((void*)p, n);
        if (fd >= 0) ::close(fd);
    }
    size_t size() const { return n; }
    const unsigned char* data() const { return p; }
};

// BufRead::lines(): split on '\n', drop one trailing '\r'; no empty last line after a final '\n'
std::vector<std::string> split_lines(const std::string& data) {
    std::vector<std::string> lines;
    size_t i = 0;
    while (i < data.size()) {
        size_t j = data.find('\n', i);
        if (j == std::string::npos) j = data.size();
        size_t e = j;
        if (e > i && data[e - 1] == '\r') --e;
        lines.emplace_back(data, i, e - i);
        i = j + 1;
    }
    return lines;
}

// ---- FASTA + .fai (rust-bio fasta::IndexedReader) ------------------------------
struct FaiEntry { std::string name; uint64_t len, offset, linebases, linewidth; };
struct Fasta {
    MappedFile data;                  // mapped, random access through the .fai offsets
    std::vector<FaiEntry> seqs;
    std::unordered_map<std::string, size_t> by_name;
    // fetch [start, end) of contig, upper-cased (read_locus :947-952)
    void fetch_upper(const FaiEntry& e, uint64_t start, uint64_t end, std::string& out) const {
        out.clear();
        if (end <= start) return;
        out.resize((size_t)(end - start));
        char* dst = &out[0];
        uint64_t p = start;
        while (p < end) {                                   // one FASTA line at a time (no division per base)
            const uint64_t col = p % e.linebases;
            const uint64_t n = std::min<uint64_t>(end - p, e.linebases - col);
            const uint64_t off = e.offset + (p / e.linebases) * e.linewidth + col;
            for (uint64_t k = 0; k < n; ++k) {
                unsigned char c = off + k < data.size() ? data.data()[off + k] : 'N';
                if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
                dst[p - start + k] = (char)c;
            }
            p += n;
        }
    }
};

// ---- VCF (text, optionally gz) -------------------------------------------------
struct VcfRec { std::string chrom; int64_t pos; std::vector<std::string> alleles; };

// BCF2 (the binary VCF htslib reads behind bcf::Reader, src/main.rs:220): "BCF\2\2", l_text, the VCF header text, then records
// {l_shared, l_indiv, CHROM i32, POS i32 (0-based), rlen, QUAL, n_info | n_allele << 16, n_fmt | n_sample << 24, ID, alleles ...}
// with typed values (descriptor byte: size << 4 | type; size 15: the real size follows as a typed integer; type 7: characters).
// Only what the reference uses is read: rid -> contig name (header dictionary: IDX= if present, else order of appearance), pos,
// the alleles (record.alleles(), :229-233).  `data` is the inflated file.
bool parse_bcf(const std::string& data, std::vector<VcfRec>& out, std::string& msg) {
    const unsigned char* p = (const unsigned char*)data.data();
    const size_t n = data.size();
    auto u32at = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) | ((uint32_t)p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); };
    if (n < 9 || p[4] != 2) { msg = "unsupported BCF version"; return false; }
    const uint32_t l_text = u32at(5);
    if ((size_t)9 + l_text > n) { msg = "truncated BCF header"; return false; }
    // contig dictionary
    std::vector<std::string> contigs;
    {
        const std::string text(data, 9, l_text);
        size_t seen = 0;
        for (auto& line : split_lines(text)) {
            if (line.compare(0, 10, "##contig=<") != 0) continue;
            std::string id;
            long idx = -1;
            size_t i = 10;
            while (i < line.size() && line[i] != '>') {
                size_t e = line.find_first_of(",>", i), q = line.find('=', i);
                if (q != std::string::npos && q < e) {
                    size_t vb = q + 1;
                    if (vb < line.size() && line[vb] == '"') { const size_t qe = line.find('"', vb + 1); if (qe != std::string::npos) e = line.find_first_of(",>", qe); }
                    if (e == std::string::npos) e = line.size();
                    const std::string key(line, i, q - i), val(line, vb, e - vb);
                    if (key == "ID") id = val;
                    else if (key == "IDX") idx = atol(val.c_str());
                }
                if (e == std::string::npos) break;
                i = e + 1;
            }
            const size_t at = idx >= 0 ? (size_t)idx : seen;
            if (contigs.size() <= at) contigs.resize(at + 1);
            contigs[at] = id;
            ++seen;
        }
    }
    size_t o = 9 + (size_t)l_text;
    // typed string at o -> [ptr, len); advances o.  false: malformed
    auto typed_len = [&](size_t& q, uint32_t& len, uint32_t& type) -> bool {
        if (q >= n) return false;
        const uint32_t d = p[q++];
        type = d & 15u; len = d >> 4;
        if (len == 15) {                                       // the length is a typed integer of its own
            if (q >= n) return false;
            const uint32_t d2 = p[q++], t2 = d2 & 15u;
            if ((d2 >> 4) != 1) return false;
            if (t2 == 1) { if (q + 1 > n) return false; len = p[q]; q += 1; }
            else if (t2 == 2) { if (q + 2 > n) return false; len = (uint32_t)p[q] | ((uint32_t)p[q + 1] << 8); q += 2; }
            else if (t2 == 3) { if (q + 4 > n) return false; len = u32at(q); q += 4; }
            else return false;
        }
        return true;
    };
    while (o + 8 <= n) {
        const uint32_t l_shared = u32at(o), l_indiv = u32at(o + 4);
        const size_t body = o + 8;
        if (body + (size_t)l_shared + l_indiv > n || l_shared < 24) { msg = "truncated BCF record"; return false; }
        const int32_t rid = (int32_t)u32at(body), pos = (int32_t)u32at(body + 4);
        const uint32_t n_allele = u32at(body + 16) >> 16;
        if (rid < 0 || (size_t)rid >= contigs.size() || contigs[(size_t)rid].empty()) { msg = "BCF record on a contig the header does not define"; return false; }
        size_t q = body + 24;
        const size_t end = body + l_shared;
        uint32_t len, type;
        if (!typed_len(q, len, type) || (len && type != 7) || q + len > end) { msg = "malformed BCF record (ID)"; return false; }
        q += len;
        VcfRec r;
        r.chrom = contigs[(size_t)rid];
        r.pos = pos;
        for (uint32_t k = 0; k < n_allele; ++k) {
            if (!typed_len(q, len, type) || (len && type != 7) || q + len > end) { msg = "malformed BCF record (alleles)"; return false; }
            r.alleles.emplace_back((const char*)p + q, len);
            q += len;
        }
        if (r.alleles.empty()) { msg = "BCF record without a REF allele"; return false; }
        out.push_back(std::move(r));
        o = body + (size_t)l_shared + l_indiv;
    }
    if (o != n) { msg = "trailing bytes behind the last BCF record"; return false; }
    return true;
}

// ---- BAM ------------------------------------------------------------------------
const char kNt16[] = "=ACMGRSVTWYHKDBN";
struct Nt16Pairs {                    // byte of two 4-bit codes -> its two letters
    uint16_t v[256];
    Nt16Pairs() { for (int b = 0; b < 256; ++b) { const char two[2] = {kNt16[b >> 4], kNt16[b & 15]}; memcpy(&v[b], two, 2); } }
    const uint16_t& operator[](size_t i) const { return v[i]; }
};
const Nt16Pairs kNt16Pair;
enum { FLAG_UNMAP = 0x4, FLAG_SECONDARY = 0x100, FLAG_DUP = 0x400, FLAG_SUPP = 0x800 };

struct BgzfBlock { size_t coff; uint32_t clen; uint32_t isize; size_t start; };   // start: file offset of the block header

// growable byte buffer that never zero-fills (the inflaters overwrite every byte they are given)
// n elements of uninitialised storage (std::vector::resize would write every byte once on one thread before the
// workers fill it: the pages are touched by the threads that write them instead)
template <class T>
struct RawArr {
    T* p = nullptr;
    size_t n = 0;
    ~RawArr() { free(p); }
    bool alloc(size_t count) { free(p); p = (T*)malloc(std::max<size_t>(count, 1) * sizeof(T)); n = p ? count : 0; return p != nullptr; }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    size_t size() const { return n; }
};

// Buffers of a megabyte and more are kept when released and handed out again (a few, process-wide): their pages have been touched,
// and touched pages are what the packer's time is made of — the pack of the NEXT range of a streamed run
// (vtxh_pack_files_range) then writes into memory that costs nothing to write to.
struct BufPool {
    std::mutex m;
    struct Item { unsigned char* p; size_t cap; };
    std::vector<Item> items;
    static constexpr size_t kKeep = 12;
    size_t held = 0;                                          // bytes kept; bounded (2 GiB: two ranges of a streamed run; vtxh_trim() returns them)
    static size_t min_bytes() { static const size_t v = VTXH_DEV_ENV("VTXH_POOL_MIN") ? (size_t)strtoull(VTXH_DEV_ENV("VTXH_POOL_MIN"), nullptr, 10) : ((size_t)1 << 20); return v; }   // (tests: 64, so that small packs reuse each other's memory)
    static size_t limit() { static const size_t v = VTXH_DEV_ENV("VTXH_POOL_BYTES") ? (size_t)strtoull(VTXH_DEV_ENV("VTXH_POOL_BYTES"), nullptr, 10) : ((size_t)2 << 30); return v; }
    void trim() {
        std::lock_guard<std::mutex> g(m);
        for (auto& it : items) free(it.p);
        items.clear(); held = 0;
    }
    unsigned char* take(size_t want, size_t* cap) {
        std::lock_guard<std::mutex> g(m);
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < items.size(); ++i)
            if (items[i].cap >= want && (best == SIZE_MAX || items[i].cap < items[best].cap)) best = i;
        if (best == SIZE_MAX) return nullptr;
        unsigned char* p = items[best].p; *cap = items[best].cap;
        held -= items[best].cap;
        items.erase(items.begin() + (long)best);
        return p;
    }
    void give(unsigned char* p, size_t cap) {
        if (!p) return;
        if (cap < min_bytes() || VTXH_DEV_ENV("VTXH_NO_BUFFER_POOL")) { free(p); return; }
        std::lock_guard<std::mutex> g(m);
        if (cap > limit()) { free(p); return; }
        while (!items.empty() && (items.size() >= kKeep || held + cap > limit())) {      // make room: the smallest go first
            size_t small = 0;
            for (size_t i = 1; i < items.size(); ++i) if (items[i].cap < items[small].cap) small = i;
            if (items.size() >= kKeep && items[small].cap >= cap) { free(p); return; }
            held -= items[small].cap;
            free(items[small].p); items.erase(items.begin() + (long)small);
        }
        items.push_back(Item{p, cap});
        held += cap;
    }
};
static BufPool g_buf_pool;

struct ByteBuf {
    unsigned char* p = nullptr;
    size_t len = 0, cap = 0;
    ~ByteBuf() { g_buf_pool.give(p, cap); }
    unsigned char* data() { return p; }
    const unsigned char* data() const { return p; }
    size_t size() const { return len; }
    void drop_prefix(size_t n) { if (n) { memmove(p, p + n, len - n); len -= n; } }
    void release() { g_buf_pool.give(p, cap); p = nullptr; len = cap = 0; }
    void swap_with(ByteBuf& o) { std::swap(p, o.p); std::swap(len, o.len); std::swap(cap, o.cap); }
    unsigned char* grow(size_t add) {      // returns the start of the new bytes; nullptr when out of memory
        if (len + add > cap || !p) {
            size_t want = std::max<size_t>(std::max(len + add, cap + cap / 2), 64);
            if (want >= BufPool::min_bytes()) {
                size_t pcap = 0;
                unsigned char* q = g_buf_pool.take(want, &pcap);                // a kept buffer with the geometric headroom (a buffer that
                if (!q) q = g_buf_pool.take(len + add, &pcap);                  // grows step by step is not re-copied at every step), else one
                if (q) {                                                        // that holds what is needed now
                    if (len) memcpy(q, p, len);
                    g_buf_pool.give(p, cap);
                    p = q; cap = pcap;
                    unsigned char* r = p + len;
                    len += add;
                    return r;
                }
            }
            unsigned char* q = (unsigned char*)realloc(p, want);
            if (!q) return nullptr;
            p = q; cap = want;
        }
        unsigned char* r = p + len;
        len += add;
        return r;
    }
};

bool index_bgzf(const MappedFile& file, std::vector<BgzfBlock>& blocks) {
    size_t o = 0;
    while (o + 18 <= file.size()) {
        const unsigned char* h = (const unsigned char*)file.data() + o;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
        uint32_t xlen = h[10] | (h[11] << 8);
        if (o + 12 + xlen > file.size()) return false;           // truncated extra field: never read past the mapping
        uint32_t bsize = 0;
        bool found = false;
        size_t x = 12;
        while (x + 4 <= 12 + xlen) {
            uint32_t slen = h[x + 2] | (h[x + 3] << 8);
            if (x + 4 + slen > 12 + xlen) return false;          // subfield runs past the extra field
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) { bsize = (h[x + 4] | (h[x + 5] << 8)) + 1u; found = true; }
            x += 4 + slen;
        }
        if (!found || o + bsize > file.size() || bsize < 12 + xlen + 8) return false;
        const unsigned char* t = h + bsize - 4;
        uint32_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        blocks.push_back(BgzfBlock{o + 12 + xlen, bsize - 12 - xlen - 8, isize, o});
        o += bsize;
    }
    return o == file.size();
}

// the packer's own decoder first (vtx_inflate.h: one-shot, table driven, ~2.5 x zlib on BAM data); whatever it does not accept
// — malformed or merely unusual — goes to zlib, which decides.  VTXH_ZLIB_INFLATE=1: zlib only (A/B timing, tests).
bool inflate_block(const MappedFile& file, const BgzfBlock& b, unsigned char* dst) {
    if (b.isize == 0) return true;
    const bool zlib_only = VTXH_DEV_ENV("VTXH_ZLIB_INFLATE") != nullptr;      // (per block: a test switches it between two packs of one process)
    if (!zlib_only) {
        vtxinf::Tables T;
        if (vtxinf::inflate_raw((const uint8_t*)file.data() + b.coff, b.clen, dst, b.isize, T)) return true;
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = (Bytef*)(file.data() + b.coff);
    zs.avail_in = b.clen;
    zs.next_out = dst;
    zs.avail_out = b.isize;
    int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0;
}

// Linear index of a .bai (SAM spec 5.2): per reference, for every 16 kb window the smallest virtual file offset of an
// alignment overlapping the window.  Only the linear index is used (bins are skipped).  Returns false when the file is
// absent, malformed or empty (the packer then sweeps the whole BAM, as it does for .csi-only inputs).
bool read_bai_linear(const std::string& path, size_t n_ref_expected, std::vector<std::vector<uint64_t>>& lin) {
    std::string d;
    if (!read_file(path, d) || d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return false;
    const unsigned char* p = (const unsigned char*)d.data();
    size_t o = 4;
    auto u32 = [&](uint32_t& v) { if (o + 4 > d.size()) return false; v = p[o] | (p[o + 1] << 8) | (p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); o += 4; return true; };
    uint32_t n_ref;
    if (!u32(n_ref) || n_ref != n_ref_expected) return false;
    lin.assign(n_ref, {});
    bool any = false;
    for (uint32_t r = 0; r < n_ref; ++r) {
        uint32_t n_bin, n_intv;
        if (!u32(n_bin)) return false;
        for (uint32_t b = 0; b < n_bin; ++b) {
            uint32_t bin, n_chunk;
            if (!u32(bin) || !u32(n_chunk) || o + 16ull * n_chunk > d.size()) return false;
            o += 16ull * n_chunk;
        }
        if (!u32(n_intv) || o + 8ull * n_intv > d.size()) return false;
        lin[r].resize(n_intv);
        for (uint32_t i = 0; i < n_intv; ++i) {
            uint64_t v = 0;
            for (int k = 7; k >= 0; --k) v = (v << 8) | p[o + (size_t)k];
            lin[r][i] = v; o += 8;
            any |= v != 0;
        }
    }
    return any;
}

// The same table from a .csi (CSI v1, htslib: BGZF-compressed; bins of min_shift + 3 * level bits, `depth` levels): no linear index,
// but every bin carries loffset — the smallest virtual offset of a record overlapping the bin.  The window table is rebuilt from the
// LEAF bins (one per 2^min_shift bases: exactly a linear-index entry); a window whose leaf bin is absent takes its nearest present
// ancestor's loffset (records that span a leaf boundary are filed further up; an ancestor's loffset is at or before anything that
// overlaps the window: conservative), else 0 = "not recorded" like an empty .bai slot.  *shift = min_shift (the window size).
bool read_csi_linear(const std::string& path, size_t n_ref_expected, std::vector<std::vector<uint64_t>>& lin, int* shift) {
    std::string d;
    if (!read_gz(path, d) || d.size() < 16 || memcmp(d.data(), "CSI\1", 4) != 0) return false;
    const unsigned char* p = (const unsigned char*)d.data();
    size_t o = 4;
    auto u32 = [&](uint32_t& v) { if (o + 4 > d.size()) return false; v = p[o] | (p[o + 1] << 8) | (p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); o += 4; return true; };
    auto u64 = [&](uint64_t& v) { if (o + 8 > d.size()) return false; v = 0; for (int k = 7; k >= 0; --k) v = (v << 8) | p[o + (size_t)k]; o += 8; return true; };
    uint32_t min_shift, depth, l_aux, n_ref;
    if (!u32(min_shift) || !u32(depth) || !u32(l_aux) || o + l_aux > d.size()) return false;
    o += l_aux;
    if (!u32(n_ref) || n_ref != n_ref_expected || min_shift < 8 || min_shift > 29 || depth > 9) return false;
    auto first_of = [](uint32_t level) -> uint64_t { return (((uint64_t)1 << (3 * level)) - 1) / 7; };
    const uint64_t leaf0 = first_of(depth), n_leaf = (uint64_t)1 << (3 * depth);
    lin.assign(n_ref, {});
    bool any = false;
    for (uint32_t r = 0; r < n_ref; ++r) {
        uint32_t n_bin;
        if (!u32(n_bin)) return false;
        std::unordered_map<uint32_t, uint64_t> loff;
        uint64_t max_leaf = 0;
        bool has = false;
        for (uint32_t b = 0; b < n_bin; ++b) {
            uint32_t bin, n_chunk;
            uint64_t lo;
            if (!u32(bin) || !u64(lo) || !u32(n_chunk) || o + 16ull * n_chunk > d.size()) return false;
            uint64_t last_end_window = 0;
            if (bin < leaf0 + n_leaf) {                         // (the pseudo-bin of the metadata lies beyond the last level)
                loff[bin] = lo;
                // the last window this bin can touch: its level's span
                uint32_t level = depth;
                while (level > 0 && bin < first_of(level)) --level;
                const uint64_t idx = bin - first_of(level), span = (uint64_t)1 << (3 * (depth - level));
                last_end_window = idx * span + span - 1;
                if (n_chunk) { has = true; max_leaf = std::max(max_leaf, last_end_window); }
            }
            o += 16ull * n_chunk;
        }
        if (!has) continue;
        lin[r].assign((size_t)std::min<uint64_t>(max_leaf + 1, n_leaf), 0);
        for (size_t w = 0; w < lin[r].size(); ++w) {
            uint64_t bin = leaf0 + w;
            for (uint32_t level = depth;; --level) {
                auto it = loff.find((uint32_t)bin);
                if (it != loff.end() && it->second) { lin[r][w] = it->second; break; }
                if (level == 0) break;
                bin = first_of(level - 1) + ((bin - first_of(level)) >> 3);      // the parent
            }
            any |= lin[r][w] != 0;
        }
    }
    *shift = (int)min_shift;
    return any;
}

inline uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rdi32(const unsigned char* p) { return (int32_t)rd32(p); }

// rec.aux(tag) matched against Aux::String (src/main.rs:742-748, :753-755): type 'Z' only.
// Returns 1 and [val, val+len) if the tag exists and is a Z string, 0 otherwise.
int aux_string(const unsigned char* aux, size_t n, const char* tag, const unsigned char** val, size_t* len) {
    size_t o = 0;
    while (o + 3 <= n) {
        const unsigned char* t = aux + o;
        char ty = (char)aux[o + 2];
        o += 3;
        size_t size;
        bool is_z = false;
        switch (ty) {
        case 'A': case 'c': case 'C': size = 1; break;
        case 's': case 'S': size = 2; break;
        case 'i': case 'I': case 'f': size = 4; break;
        case 'd': size = 8; break;                              // htslib's bam_aux_get skips a double as 8 bytes
        case 'Z': case 'H': {
            size_t e = o;
            while (e < n && aux[e]) ++e;
            size = e - o + 1;
            is_z = ty == 'Z';
            break;
        }
        case 'B': {
            if (o + 5 > n) return 0;
            char sub = (char)aux[o];
            uint32_t cnt = rd32(aux + o + 1);
            size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            size = 5 + (size_t)cnt * es;
            break;
        }
        default: return 0;
        }
        if (t[0] == (unsigned char)tag[0] && t[1] == (unsigned char)tag[1]) {
            if (!is_z) return 0;
            *val = aux + o; *len = size - 1;
            return 1;
        }
        o += size;
    }
    return 0;
}

// rust-htslib 0.36 CigarStringView::read_pos(ref_pos, include_softclips=false, include_dels=true)
// as called from useful_alignment (src/main.rs:796).  1 = Some, 0 = None, -1 = Err.
int cigar_read_pos(const unsigned char* cig, uint32_t n_ops, int64_t pos, int64_t ref_pos) {
    int64_t rpos = pos;
    uint32_t j = 0;
    for (uint32_t i = 0; i < n_ops; ++i) {
        uint32_t c = rd32(cig + 4 * i);
        uint32_t op = c & 15;
        if (op == 0 || op == 7 || op == 8 || op == 1) { j = i; break; }
        if (op == 4) { j = i; break; }
        if (op == 2 || op == 3) return -1;
        if (op == 5 && i > 0 && i + 1 < n_ops) return -1;
        if ((op == 6 || op == 5) && i + 1 == n_ops) return 0;
    }
    while (rpos <= ref_pos && j < n_ops) {
        uint32_t c = rd32(cig + 4 * j);
        uint32_t op = c & 15;
        int64_t l = c >> 4;
        bool contains = rpos <= ref_pos && rpos + l > ref_pos;
        switch (op) {
        case 0: case 7: case 8: if (contains) return 1; rpos += l; ++j; break;
        case 4: ++j; break;
        case 2: if (contains) return 1; rpos += l; ++j; break;
        case 3: rpos += l; ++j; break;
        case 1: case 6: ++j; break;
        case 5: if (j + 1 < n_ops) return -1; return 0;
        default: return -1;
        }
    }
    return 0;
}

// useful_alignment, src/main.rs:790-806 (probes start..=end, inclusive)
bool useful_alignment(const unsigned char* cig, uint32_t n_ops, int64_t pos, int64_t start, int64_t end) {
    for (int64_t i = start; i <= end; ++i) {
        int r = cigar_read_pos(cig, n_ops, pos, i);
        if (r == 1) return true;
        if (r < 0) return false;    // invalid CIGAR: read skipped (:799-802)
    }
    return false;
}

// Minimal persistent worker pool: run(fn) executes fn(t) for t in [0, n) — t = 0 on the caller — and waits.
// The sweep calls run() three or four times per window of blocks, a few milliseconds apart; the workers sleep on a condition
// variable in between (spinning before the sleep was measured in round 4 and is gone: on the GPU boxes' CPU quota it made the ingest
// 20 % slower, profiles/r04_e2e_cli_spin.log).
class Pool {
  public:
    explicit Pool(int n) : n_(n < 1 ? 1 : n) {
        for (int t = 1; t < n_; ++t) th_.emplace_back([this, t] { loop(t); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; gen_.fetch_add(1); }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return n_; }
    void run(const std::function<void(size_t)>& fn) {
        { std::lock_guard<std::mutex> g(m_); fn_ = &fn; pending_.store(n_ - 1); gen_.fetch_add(1); }
        cv_.notify_all();
        fn(0);
        if (pending_.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> g(m_);
            done_.wait(g, [this] { return pending_.load() == 0; });
        }
        fn_ = nullptr;
    }

  private:
    void loop(int t) {
        uint64_t seen = 0;
        while (true) {
            const std::function<void(size_t)>* fn;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_.load() != seen; });
                seen = gen_.load();
                if (stop_) return;
                fn = fn_;
            }
            (*fn)((size_t)t);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> g(m_); done_.notify_one(); }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    std::atomic<int> pending_{0};
    std::atomic<uint64_t> gen_{0};
    bool stop_ = false;
};

// phase timer: VTXH_PROFILE=1 prints cumulative seconds per phase of vtxh_pack_files to stderr
struct Phases {
    bool on = getenv("VTXH_PROFILE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    std::vector<std::pair<std::string, double>> acc;
    void mark(const char* name) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        const double d = std::chrono::duration<double>(now - t).count();
        t = now;
        for (auto& a : acc) if (a.first == name) { a.second += d; return; }
        acc.emplace_back(name, d);
    }
    ~Phases() { if (on) for (auto& a : acc) fprintf(stderr, "[vtxh] %-18s %.3f s\n", a.first.c_str(), a.second); }
};

struct LocusBuild {
    uint32_t row;
    int64_t start, end;
    uint64_t ref_off = 0, alt_off = 0;          // into the pack's haplotype arena
    uint32_t ref_len = 0, alt_len = 0;
    struct Rec { uint32_t cell, umi; uint64_t read_off; uint32_t read_len; };
};

struct Interval { int64_t start, end; uint32_t locus; };

}  // namespace

struct vtxh_pack {
    uint64_t blocks_inflated = 0, blocks_total = 0, index_jumps = 0;      // ingest statistics (vtxh_get_ingest_stats)
    std::vector<vtx_locus> loci;
    RawArr<vtx_record> records;
    ByteBuf hap_arena;             // REF then ALT haplotype of every locus, in locus order (written by the workers: no zero-fill, no serial append)
    ByteBuf read_arena;            // grown without zero-fill, filled by the sweep's workers
    int read_format = VTX_READS_BYTES;   // VTX_READS_NIBBLES: two bases per byte, offsets / sizes below still count bases
    vtxh_metrics metrics{};
    uint32_t n_variants = 0;
    std::vector<std::string> barcodes, variant_names;
    // raw mode (vtxh_pack_files_raw)
    RawArr<vtx_raw_record> raw_records;
    ByteBuf tag_arena;
    std::string bc_bytes;
    std::vector<uint64_t> bc_offsets;
    // Batches: consecutive loci whose reads (and tags) span less than 4 GiB of the arenas, so that the 32-bit offsets
    // of vtx.h hold relative to the batch's window.  Loci / records of all batches sit back to back in the arrays above.
    struct Batch { uint32_t l0, l1; uint64_t rec0, rec1, rbase, rbytes, tbase, tbytes; };
    std::vector<Batch> batches;
    // plan of a device-side ingest (vtxh_plan_ingest): the file stays mapped, the index of its blocks / record starts / loci below
    bool is_plan = false, planned = false;
    std::string plan_reason;
    std::unique_ptr<MappedFile> bam_map;
    std::vector<vtx_bgzf_block> pl_blocks;
    std::vector<uint64_t> pl_seeds;
    uint64_t pl_end = 0;
    std::vector<vtx_bam_interval> pl_iv;
    std::vector<uint32_t> pl_tid_begin;
    std::vector<int32_t> pl_span;
    uint32_t pl_mapq = 0;
    int32_t pl_primary = 0, pl_nodup = 0;
    char pl_tag[2] = {'C', 'B'};
};

extern "C" {

const char* vtxh_last_error(void) { return g_err.c_str(); }

int vtxh_format_f64(double v, char* buf) {
    // Rust `{}` for f64: shortest round-trip digits, positional, no exponent, no trailing ".0"
    if (std::isnan(v)) { strcpy(buf, "NaN"); return 3; }
    if (std::isinf(v)) { strcpy(buf, v < 0 ? "-inf" : "inf"); return (int)strlen(buf); }
    if (v == 0) { strcpy(buf, std::signbit(v) ? "-0" : "0"); return (int)strlen(buf); }
    char tmp[400];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::fixed);
    size_t n = (size_t)(r.ptr - tmp);
    if (n > 31) {   // very large / tiny magnitudes do not occur for counts and fractions; keep the contract
        memcpy(buf, tmp, 31); buf[31] = 0; return 31;
    }
    memcpy(buf, tmp, n);
    buf[n] = 0;
    return (int)n;
}

int vtxh_write_mtx(const char* path, uint32_t n_rows, uint32_t n_cols, uint64_t nnz, const uint32_t* row,
                   const uint32_t* col, const double* value) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(VTX_E_INVAL, "cannot open %s for writing", path);
    char head[160];
    const int hl = snprintf(head, sizeof head, "%%%%MatrixMarket matrix coordinate real general\n%% written by sprs\n%u %u %llu\n", n_rows,
                            n_cols, (unsigned long long)nnz);
    auto write_at = [&](const char* p, size_t n, uint64_t off) -> bool {
        while (n) {
            const ssize_t w = pwrite(fd, p, n, (off_t)off);
            if (w <= 0) return false;
            p += w; n -= (size_t)w; off += (uint64_t)w;
        }
        return true;
    };
    bool ok = write_at(head, (size_t)hl, 0);
    uint64_t file_off = (uint64_t)hl;
    // lines are formatted by several threads into private buffers (rounds of bounded size); every thread then writes its
    // buffer at its own offset (the copy into the page cache is most of the time of a 300 MB file)
    const uint64_t kRound = 8u << 20;
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t T = nnz < (1u << 16) ? 1 : std::min<size_t>(hw ? hw : 1, 16);
    std::vector<std::string> parts(T);
    std::vector<uint64_t> offs(T);
    std::vector<uint8_t> wok(T, 1);
    for (uint64_t base = 0; base < nnz && ok; base += kRound) {
        const uint64_t n = std::min<uint64_t>(kRound, nnz - base);
        auto fmt = [&](size_t t) {
            std::string& out = parts[t];
            out.clear();
            const uint64_t k0 = base + n * t / T, k1 = base + n * (t + 1) / T;
            out.reserve((size_t)(k1 - k0) * 16);
            char line[96], num[40];
            for (uint64_t k = k0; k < k1; ++k) {
                char* p = line;
                p = std::to_chars(p, p + 12, row[k] + 1u).ptr; *p++ = ' ';
                p = std::to_chars(p, p + 12, col[k] + 1u).ptr; *p++ = ' ';
                const int m = vtxh_format_f64(value[k], num);
                memcpy(p, num, (size_t)m); p += m; *p++ = '\n';
                out.append(line, (size_t)(p - line));
            }
        };
        auto put = [&](size_t t) { wok[t] = write_at(parts[t].data(), parts[t].size(), offs[t]) ? 1 : 0; };
        {
            std::vector<std::thread> th;
            for (size_t t = 1; t < T; ++t) th.emplace_back(fmt, t);
            fmt(0);
            for (auto& t : th) t.join();
        }
        for (size_t t = 0; t < T; ++t) { offs[t] = file_off; file_off += parts[t].size(); }
        {
            std::vector<std::thread> th;
            for (size_t t = 1; t < T; ++t) th.emplace_back(put, t);
            put(0);
            for (auto& t : th) t.join();
        }
        for (size_t t = 0; t < T; ++t) ok = ok && wok[t];
    }
    ok = (close(fd) == 0) && ok;
    return ok ? VTX_OK : fail(VTX_E_INVAL, "error writing %s", path);
}

void vtxh_free(vtxh_pack* p) { delete p; }
void vtxh_trim(void) { g_buf_pool.trim(); }
uint32_t vtxh_num_batches(const vtxh_pack* p) { return (uint32_t)p->batches.size(); }
void vtxh_get_batch_at(const vtxh_pack* p, uint32_t i, vtx_batch* out) {
    memset(out, 0, sizeof *out);
    if (i >= p->batches.size()) return;
    const vtxh_pack::Batch& b = p->batches[i];
    out->loci = p->loci.data() + b.l0; out->n_loci = b.l1 - b.l0;
    out->records = p->records.data() + b.rec0; out->n_records = (uint32_t)(b.rec1 - b.rec0);
    out->hap_arena = (const uint8_t*)p->hap_arena.data(); out->hap_bytes = p->hap_arena.size();
    if (p->read_format == VTX_READS_NIBBLES) { out->read_arena = p->read_arena.data() + b.rbase / 2; out->read_bytes = (b.rbytes + 1) & ~(uint64_t)1; }   // (rbase is even: every read starts at an even base)
    else { out->read_arena = p->read_arena.data() + b.rbase; out->read_bytes = b.rbytes; }
}
void vtxh_get_batch(const vtxh_pack* p, vtx_batch* out) { vtxh_get_batch_at(p, 0, out); }
void vtxh_get_metrics(const vtxh_pack* p, vtxh_metrics* out) { *out = p->metrics; }
void vtxh_get_ingest_stats(const vtxh_pack* p, uint64_t out[3]) { out[0] = p->blocks_inflated; out[1] = p->blocks_total; out[2] = p->index_jumps; }
uint32_t vtxh_num_variants(const vtxh_pack* p) { return p->n_variants; }
uint32_t vtxh_num_barcodes(const vtxh_pack* p) { return (uint32_t)p->barcodes.size(); }
const char* vtxh_variant_name(const vtxh_pack* p, uint32_t i) { return i < p->variant_names.size() ? p->variant_names[i].c_str() : ""; }
const char* vtxh_barcode(const vtxh_pack* p, uint32_t j) { return j < p->barcodes.size() ? p->barcodes[j].c_str() : ""; }

void vtxh_get_raw_batch_at(const vtxh_pack* p, uint32_t i, vtx_raw_batch* out) {
    memset(out, 0, sizeof *out);
    if (i >= p->batches.size()) return;
    const vtxh_pack::Batch& b = p->batches[i];
    out->loci = p->loci.data() + b.l0; out->n_loci = b.l1 - b.l0;
    out->records = p->raw_records.data() + b.rec0; out->n_records = (uint32_t)(b.rec1 - b.rec0);
    out->hap_arena = (const uint8_t*)p->hap_arena.data(); out->hap_bytes = p->hap_arena.size();
    if (p->read_format == VTX_READS_NIBBLES) { out->read_arena = p->read_arena.data() + b.rbase / 2; out->read_bytes = (b.rbytes + 1) & ~(uint64_t)1; }   // (rbase is even: every read starts at an even base)
    else { out->read_arena = p->read_arena.data() + b.rbase; out->read_bytes = b.rbytes; }
    out->tag_arena = (const uint8_t*)p->tag_arena.data() + b.tbase; out->tag_bytes = b.tbytes;
}
void vtxh_get_raw_batch(const vtxh_pack* p, vtx_raw_batch* out) { vtxh_get_raw_batch_at(p, 0, out); }
void vtxh_get_barcode_table(const vtxh_pack* p, const uint8_t** bytes, const uint64_t** offsets, uint32_t* n) {
    *bytes = (const uint8_t*)p->bc_bytes.data(); *offsets = p->bc_offsets.data(); *n = (uint32_t)p->barcodes.size();
}

static int pack_impl(const vtxh_args* a, bool raw, uint32_t row_begin, uint32_t row_end, vtxh_pack** out, bool plan = false);
int vtxh_pack_files(const vtxh_args* a, vtxh_pack** out) { return pack_impl(a, false, 0, 0xffffffffu, out); }
int vtxh_pack_files_raw(const vtxh_args* a, vtxh_pack** out) { return pack_impl(a, true, 0, 0xffffffffu, out); }
int vtxh_pack_files_range(const vtxh_args* a, int raw, uint32_t row_begin, uint32_t row_end, vtxh_pack** out) {
    if (row_begin > row_end) return fail(VTX_E_INVAL, "vtxh_pack_files_range: row_begin > row_end");
    return pack_impl(a, raw != 0, row_begin, row_end, out);
}
int vtxh_plan_ingest(const vtxh_args* a, uint32_t row_begin, uint32_t row_end, vtxh_pack** out) {
    if (row_begin > row_end) return fail(VTX_E_INVAL, "vtxh_plan_ingest: row_begin > row_end");
    return pack_impl(a, true, row_begin, row_end, out, true);
}
int vtxh_get_ingest(const vtxh_pack* p, vtx_bam_ingest* out) {
    if (!p || !out) return fail(VTX_E_INVAL, "vtxh_get_ingest: null argument");
    memset(out, 0, sizeof *out);
    if (!p->is_plan) return fail(VTX_E_STATE, "vtxh_get_ingest: not a plan (vtxh_plan_ingest)");
    if (!p->planned) return fail(VTX_E_UNSUPPORTED, "no device-side ingest for this input: %s", p->plan_reason.c_str());
    out->file = p->bam_map->data(); out->file_bytes = p->bam_map->size();
    out->blocks = p->pl_blocks.data(); out->n_blocks = (uint32_t)p->pl_blocks.size();
    out->n_ref = (uint32_t)p->pl_span.size();
    out->seeds = p->pl_seeds.data(); out->n_seeds = (uint32_t)p->pl_seeds.size();
    out->n_intervals = (uint32_t)p->pl_iv.size();
    out->end_upos = p->pl_end;
    out->intervals = p->pl_iv.data(); out->tid_begin = p->pl_tid_begin.data(); out->tid_max_span = p->pl_span.data();
    out->loci = p->loci.data(); out->n_loci = (uint32_t)p->loci.size();
    out->min_mapq = p->pl_mapq; out->primary_only = p->pl_primary; out->no_duplicates = p->pl_nodup;
    out->hap_arena = (const uint8_t*)p->hap_arena.data(); out->hap_bytes = p->hap_arena.size();
    out->bam_tag[0] = p->pl_tag[0]; out->bam_tag[1] = p->pl_tag[1];
    return VTX_OK;
}
int vtxh_is_plan(const vtxh_pack* p) { return p && p->is_plan ? 1 : 0; }

// VCF records [row_begin, row_end) only: the other records keep their matrix rows (n_variants, names) but get no haplotypes, no
// loci and no reads, and are not counted in the metrics — the packs of consecutive ranges add up to the pack of the whole file.
static int pack_impl(const vtxh_args* a, bool raw, uint32_t row_begin, uint32_t row_end, vtxh_pack** out, bool plan) {
    if (!a || !out || !a->vcf || !a->bam || !a->fasta || !a->cell_barcodes) return fail(VTX_E_INVAL, "vtxh_pack_files: null argument");
    *out = nullptr;
    const std::string bam_tag = a->bam_tag ? a->bam_tag : "CB";
    if (bam_tag.size() != 2) return fail(VTX_E_INVAL, "--bam-tag must be two characters");
    bool valid[256] = {false};
    for (const char* c = a->valid_chars ? a->valid_chars : "ATGCatgc"; *c; ++c) valid[(unsigned char)*c] = true;
    const int threads = a->threads > 0 ? a->threads : 1;
    std::unique_ptr<vtxh_pack> P(new vtxh_pack());
    Phases ph;
    // the BAM is mapped and its BGZF headers are walked (one page of the file touched per block: 130 000 page faults for a 0.9 GB BAM)
    // on a thread of its own while this one reads the barcodes, the VCF and the FASTA index; joined where the BAM header is parsed
    std::unique_ptr<MappedFile> bam_holder(new MappedFile());      // (a plan keeps the mapping: vtx_submit_bam reads the file's bytes)
    std::vector<BgzfBlock> blocks;
    int bam_state = 0;                                             // 1 indexed, -1 cannot open, -2 not BGZF
    std::thread bam_thread([&] {
        if (!bam_holder->open(a->bam)) { bam_state = -1; return; }
        bam_state = index_bgzf(*bam_holder, blocks) ? 1 : -2;
    });
    struct BamJoin { std::thread& t; ~BamJoin() { if (t.joinable()) t.join(); } } bam_join{bam_thread};

    // ---- load_barcodes (:697-718): first-occurrence index, whole line is the key ----
    std::unordered_map<std::string, uint32_t> bc_index;
    {
        std::string data;
        std::string path = a->cell_barcodes;
        bool ok = ends_with(path, ".gz") ? read_gz(path, data) : read_file(path, data);   // open_with_gz :727
        if (!ok) return fail(VTX_E_INVAL, "error open barcodes file: \"%s\"", path.c_str());
        for (auto& line : split_lines(data))
            if (bc_index.emplace(line, (uint32_t)P->barcodes.size()).second) P->barcodes.push_back(line);
        if (P->barcodes.empty()) return fail(VTX_E_INVAL, "Loaded 0 barcodes. Is your barcode file gzipped or empty?");
        if (raw) {
            P->bc_offsets.push_back(0);
            for (auto& b : P->barcodes) { P->bc_bytes += b; P->bc_offsets.push_back(P->bc_bytes.size()); }
        }
    }
    // the per-read lookup (cooked packs): open addressing over the barcode bytes, no allocation per probe
    // (std::unordered_map<std::string, .>::find needs a std::string: a malloc per read for 18-byte barcodes)
    struct BcTable {
        std::vector<uint32_t> slot;            // index + 1; 0 = empty
        const std::vector<std::string>* names = nullptr;
        uint64_t mask = 0;
        static uint64_t hash(const unsigned char* p, size_t n) {
            uint64_t h = 1469598103934665603ull;
            for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
            return h ^ (h >> 29);
        }
        void build(const std::vector<std::string>& v) {
            names = &v;
            size_t cap = 16;
            while (cap < 2 * v.size()) cap <<= 1;
            slot.assign(cap, 0); mask = cap - 1;
            for (size_t i = 0; i < v.size(); ++i) {
                uint64_t k = hash((const unsigned char*)v[i].data(), v[i].size()) & mask;
                while (slot[k]) k = (k + 1) & mask;
                slot[k] = (uint32_t)i + 1;
            }
        }
        bool find(const unsigned char* p, size_t n, uint32_t* out) const {
            for (uint64_t k = hash(p, n) & mask; slot[k]; k = (k + 1) & mask) {
                const std::string& s = (*names)[slot[k] - 1];
                if (s.size() == n && memcmp(s.data(), p, n) == 0) { *out = slot[k] - 1; return true; }
            }
            return false;
        }
    } bc_table;
    bc_table.build(P->barcodes);

    ph.mark("barcodes");
    // ---- VCF records (:221-234) ----
    std::vector<VcfRec> vcf;
    {
        std::string data;
        std::string path = a->vcf;
        if (!read_gz(path, data)) return fail(VTX_E_INVAL, "error opening vcf file %s", path.c_str());
        // bcf::Reader::from_path (src/main.rs:220) reads BCF as well as VCF — by content, not by extension
        if (data.size() >= 9 && memcmp(data.data(), "BCF\2", 4) == 0) {
            std::string msg;
            if (!parse_bcf(data, vcf, msg)) return fail(VTX_E_INVAL, "%s: %s", path.c_str(), msg.c_str());
            for (const VcfRec& r : vcf) P->variant_names.push_back(r.chrom + "_" + std::to_string(r.pos));
            data.clear();
        }
        for (auto& line : split_lines(data)) {
            if (line.empty() || line[0] == '#') continue;
            std::vector<std::string> f;
            size_t i = 0;
            while (f.size() < 5) {
                size_t j = line.find('\t', i);
                if (j == std::string::npos) { f.emplace_back(line, i); break; }
                f.emplace_back(line, i, j - i);
                i = j + 1;
            }
            if (f.size() < 5) return fail(VTX_E_INVAL, "malformed VCF line: %s", line.c_str());
            VcfRec r;
            r.chrom = f[0];
            r.pos = atoll(f[1].c_str()) - 1;
            r.alleles.push_back(f[3]);
            if (f[4] != ".") {
                size_t s = 0;
                while (true) {
                    size_t c = f[4].find(',', s);
                    if (c == std::string::npos) { r.alleles.emplace_back(f[4], s); break; }
                    r.alleles.emplace_back(f[4], s, c - s);
                    s = c + 1;
                }
            }
            P->variant_names.push_back(r.chrom + "_" + std::to_string(r.pos));   // write_variants :1174 (0-based pos)
            vcf.push_back(std::move(r));
        }
        P->n_variants = (uint32_t)vcf.size();
    }

    ph.mark("vcf");
    // ---- FASTA + .fai ----
    Fasta fa;
    {
        std::string fai;
        if (!read_file(std::string(a->fasta) + ".fai", fai)) return fail(VTX_E_INVAL, "error opening fasta index: %s.fai", a->fasta);
        for (auto& line : split_lines(fai)) {
            if (line.empty()) continue;
            FaiEntry e;
            char name[4096];
            unsigned long long len, off, lb, lw;
            if (sscanf(line.c_str(), "%4095[^\t]\t%llu\t%llu\t%llu\t%llu", name, &len, &off, &lb, &lw) != 5)
                return fail(VTX_E_INVAL, "malformed .fai line: %s", line.c_str());
            e.name = name; e.len = len; e.offset = off; e.linebases = lb; e.linewidth = lw;
            fa.by_name.emplace(e.name, fa.seqs.size());
            fa.seqs.push_back(e);
        }
        if (!fa.data.open(a->fasta)) return fail(VTX_E_INVAL, "error opening fasta file %s", a->fasta);
    }

    ph.mark("fasta index");
    // ---- BAM: header ----
    MappedFile& bam_file = *bam_holder;
    bam_thread.join();                                   // (the header walk ran beside the VCF / FASTA work above)
    if (bam_state == -1) return fail(VTX_E_INVAL, "error opening bam file: %s", a->bam);
    if (ends_with(a->bam, ".cram")) return fail(VTX_E_UNSUPPORTED, "CRAM input is not supported");
    if (bam_state != 1) return fail(VTX_E_INVAL, "%s is not a valid BGZF/BAM file", a->bam);

    // streaming inflater over chunks of blocks
    Pool pool(threads);
    ByteBuf buf;                          // decompressed bytes not yet consumed
    size_t buf_pos = 0, next_block = 0;
    size_t chunk_blocks = 512;            // blocks per inflate round; restarts small after an index-guided jump
    size_t max_chunk_blocks = 512;
    if (const char* e = VTXH_DEV_ENV("VTXH_CHUNK_BLOCKS")) chunk_blocks = max_chunk_blocks = std::max<size_t>(1, strtoull(e, nullptr, 10));   // tests: many windows
    if (plan) chunk_blocks = max_chunk_blocks = 2;      // a plan inflates the header's blocks only
    uint64_t buf_origin = 0;              // offset of buf[0] in the inflated stream of the whole file
    size_t chunk_limit_block = SIZE_MAX;  // index-guided sweep: no read-ahead beyond the block the sweep would jump to anyway
    uint64_t n_inflated = 0, n_jumps = 0;
    // the window whose records are indexed but not parsed yet (the parse of window k runs beside the indexing of window
    // k + 1): set while it waits; its bytes move to pend_store when the sweep needs the buffer for more data
    size_t pend_begin = SIZE_MAX;
    bool pend_detached = false;
    ByteBuf pend_store;
    bool refill_failed = false;           // a block did not inflate / the buffer could not grow (as opposed to: the file has no more)
    auto refill = [&](size_t need) -> bool {   // ensure buf has >= need bytes from buf_pos, if the file has them
        while (buf.size() - buf_pos < need && next_block < blocks.size()) {
            size_t chunk = std::min(blocks.size() - next_block, chunk_blocks);
            // a jump is ahead: small rounds, so that the sweep notices the end of its segment before it has inflated its way
            // to the jump target (dense VCFs have no jump ahead and keep the large rounds)
            if (chunk_limit_block != SIZE_MAX) chunk = std::min<size_t>(chunk, 32);
            chunk_blocks = std::min<size_t>(max_chunk_blocks, chunk_blocks * 2);
            if (pend_begin != SIZE_MAX && !pend_detached) {
                // a window is indexed and waits for its parse: its bytes stay where they are (pend_store); the sweep goes on
                // in the spare buffer, which starts with the unconsumed tail (a partial record)
                buf.swap_with(pend_store);
                const size_t tail = pend_store.size() - buf_pos;
                buf.len = 0;
                if (!buf.grow(tail)) { refill_failed = true; return false; }
                memcpy(buf.data(), pend_store.data() + buf_pos, tail);
                buf_pos = 0;
                pend_detached = true;
            } else if (buf_pos) {
                buf.drop_prefix(buf_pos); buf_origin += buf_pos; buf_pos = 0;
            }
            std::vector<size_t> off(chunk + 1, 0);
            for (size_t k = 0; k < chunk; ++k) off[k + 1] = off[k] + blocks[next_block + k].isize;
            const size_t base = buf.size();
            if (!buf.grow(off[chunk])) { refill_failed = true; return false; }
            std::atomic<size_t> nextk{0};
            std::atomic<bool> ok{true};
            pool.run([&](size_t) {
                for (size_t k; (k = nextk.fetch_add(1)) < chunk;)
                    if (!inflate_block(bam_file, blocks[next_block + k], buf.data() + base + off[k])) ok = false;
            });
            if (!ok) { buf.len = base; refill_failed = true; return false; }     // nothing half-inflated is ever indexed as records
            next_block += chunk;
            n_inflated += chunk;
        }
        return buf.size() - buf_pos >= need;
    };
    if (!refill(12) && refill_failed) return fail(VTX_E_INVAL, "%s: a BGZF block does not inflate (or out of memory)", a->bam);
    if (buf.size() - buf_pos < 12 || memcmp(buf.data() + buf_pos, "BAM\1", 4) != 0) return fail(VTX_E_INVAL, "%s: bad BAM magic", a->bam);
    uint32_t l_text = rd32(buf.data() + buf_pos + 4);
    if (!refill(12 + (size_t)l_text)) return fail(VTX_E_INVAL, "%s: truncated BAM header", a->bam);
    buf_pos += 8 + l_text;
    uint32_t n_ref = rd32(buf.data() + buf_pos);
    buf_pos += 4;
    std::vector<std::string> bam_refs;
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (!refill(4)) return fail(VTX_E_INVAL, "%s: truncated BAM header", a->bam);
        uint32_t l_name = rd32(buf.data() + buf_pos);
        if (!refill(8 + (size_t)l_name)) return fail(VTX_E_INVAL, "%s: truncated BAM header", a->bam);
        bam_refs.emplace_back((const char*)buf.data() + buf_pos + 4, l_name ? l_name - 1 : 0);
        buf_pos += 8 + l_name;
    }
    std::unordered_map<std::string, int32_t> tid_of;
    for (size_t i = 0; i < bam_refs.size(); ++i) tid_of.emplace(bam_refs[i], (int32_t)i);

    ph.mark("bam header");
    // ---- validate_inputs (:545-594) + evaluate_rec pre-alignment part (:610-684) ----
    std::vector<LocusBuild> loci;
    std::vector<std::vector<Interval>> by_tid(bam_refs.size());
    std::vector<int64_t> max_span(bam_refs.size(), 1);
    for (size_t i = 0; i < vcf.size(); ++i) {
        const VcfRec& v = vcf[i];
        auto fi = fa.by_name.find(v.chrom);
        if (fi == fa.by_name.end()) return fail(VTX_E_INVAL, "Sequence %s not seen in FASTA", v.chrom.c_str());
        auto ti = tid_of.find(v.chrom);
        if (ti == tid_of.end()) return fail(VTX_E_INVAL, "Sequence %s not seen in BAM", v.chrom.c_str());
        const FaiEntry& fe = fa.seqs[fi->second];
        const int64_t end = v.pos + (int64_t)v.alleles[0].size();
        if ((uint64_t)end > fe.len)
            return fail(VTX_E_INVAL, "Record %s:%lld has end position %lld, which is larger than the chromosome length (%llu). Does your FASTA match your VCF?",
                        v.chrom.c_str(), (long long)v.pos, (long long)end, (unsigned long long)fe.len);
    }
    {
        // haplotypes of all records in parallel (verdict per record), then the loci in VCF order
        std::vector<LocusBuild> built(vcf.size());
        std::vector<uint8_t> verdict(vcf.size(), 0);            // 0 locus, 1 multi-allelic, 2 invalid characters, 3 outside this range of rows
        std::vector<int32_t> tid_i(vcf.size(), 0);
        // every worker appends the haplotypes of ITS records (a contiguous range of the VCF) to a buffer of its own — REF then ALT per
        // valid locus, the arena's layout — and the buffers are then copied behind each other: the arena's pages are touched by all
        // workers at once instead of by one thread appending 200 000 strings (the pages, not the bytes, are what this costs)
        struct HapOut { ByteBuf bytes; };
        std::vector<HapOut> hout((size_t)threads);
        pool.run([&](size_t t) {
            std::string left, right, refh;
            ByteBuf& out = hout[t].bytes;
            for (size_t i = vcf.size() * t / (size_t)threads, e = vcf.size() * (t + 1) / (size_t)threads; i < e; ++i) {
                const VcfRec& v = vcf[i];
                if (i < row_begin || i >= row_end) { verdict[i] = 3; continue; }
                if (v.alleles.size() > 2) { verdict[i] = 1; continue; }                                // :646-653
                static const std::string no_alt;
                const std::string& alt = v.alleles.size() == 2 ? v.alleles[1] : no_alt;                // :656-659
                const FaiEntry& fe = fa.seqs[fa.by_name.find(v.chrom)->second];
                const int64_t start = v.pos, end = v.pos + (int64_t)v.alleles[0].size();
                const int64_t pad = a->padding;
                LocusBuild& L = built[i];
                L.row = (uint32_t)i; L.start = start; L.end = end;
                // construct_haplotypes :958-994
                const int64_t ls = start >= pad ? start - pad : 0;
                const int64_t re = std::min<int64_t>(end + pad, (int64_t)fe.len);
                fa.fetch_upper(fe, (uint64_t)ls, (uint64_t)std::min<int64_t>(start, (int64_t)fe.len), left);
                fa.fetch_upper(fe, (uint64_t)end, (uint64_t)re, right);
                int64_t rs = (int64_t)((int32_t)start - (int32_t)pad);       // i32 casts, :944
                if (rs < 0) rs = 0;
                fa.fetch_upper(fe, (uint64_t)rs, (uint64_t)re, refh);
                bool ok = true;                                                                         // :675-684: the whole ALT haplotype
                for (unsigned char c : left) ok &= valid[c];
                for (unsigned char c : alt) ok &= valid[c];
                for (unsigned char c : right) ok &= valid[c];
                if (!ok) { verdict[i] = 2; continue; }
                tid_i[i] = tid_of.find(v.chrom)->second;
                L.ref_len = (uint32_t)refh.size(); L.alt_len = (uint32_t)(left.size() + alt.size() + right.size());
                L.ref_off = out.size();                                      // (relative to this worker's buffer until the copy below)
                unsigned char* d = out.grow(refh.size() + L.alt_len);
                if (!d) { verdict[i] = 4; continue; }
                memcpy(d, refh.data(), refh.size()); d += refh.size();
                L.alt_off = L.ref_off + refh.size();
                memcpy(d, left.data(), left.size()); d += left.size();
                memcpy(d, alt.data(), alt.size()); d += alt.size();
                memcpy(d, right.data(), right.size());
            }
        });
        for (size_t i = 0; i < vcf.size(); ++i) if (verdict[i] == 4) return fail(VTX_E_NOMEM, "out of memory building the haplotypes");
        {
            std::vector<uint64_t> hbase((size_t)threads + 1, 0);
            for (int t = 0; t < threads; ++t) hbase[(size_t)t + 1] = hbase[(size_t)t] + hout[(size_t)t].bytes.size();
            if (hbase[(size_t)threads] > 0xffffffffull) return fail(VTX_E_UNSUPPORTED, "haplotype arena above 4 GiB: split the VCF");
            if (hbase[(size_t)threads] && !P->hap_arena.grow((size_t)hbase[(size_t)threads])) return fail(VTX_E_NOMEM, "out of memory building the haplotypes");
            pool.run([&](size_t t) {
                if (hout[t].bytes.size()) memcpy(P->hap_arena.data() + hbase[t], hout[t].bytes.data(), hout[t].bytes.size());
                for (size_t i = vcf.size() * t / (size_t)threads, e = vcf.size() * (t + 1) / (size_t)threads; i < e; ++i)
                    if (verdict[i] == 0) { built[i].ref_off += hbase[t]; built[i].alt_off += hbase[t]; }
            });
        }
        for (size_t i = 0; i < vcf.size(); ++i) {
            if (verdict[i] == 1) { ++P->metrics.num_multiallelic_recs; continue; }
            if (verdict[i] == 2) { ++P->metrics.num_invalid_recs; continue; }
            if (verdict[i] == 3) continue;
            const int32_t tid = tid_i[i];
            by_tid[(size_t)tid].push_back(Interval{built[i].start, built[i].end, (uint32_t)loci.size()});
            max_span[(size_t)tid] = std::max(max_span[(size_t)tid], built[i].end - built[i].start);
            loci.push_back(std::move(built[i]));
        }
    }
    for (auto& iv : by_tid)
        std::stable_sort(iv.begin(), iv.end(), [](const Interval& x, const Interval& y) { return x.start < y.start; });

    ph.mark("haplotypes");
    // ---- index-guided skipping (the reference does an indexed fetch per locus, :822-826) ----
    // With a usable .bai the sweep only visits the stretches of the file that can hold reads of a locus: for every locus,
    // in (contig, start) order, the linear index gives the smallest virtual offset of an alignment overlapping its first
    // 16 kb window; the sweep starts there and runs until a record lies at or beyond the end of the loci it is serving
    // (coordinate-sorted file: nothing later can overlap them), then jumps to the next locus' offset — unless that is
    // within a few blocks, where sweeping on is cheaper than a restart.  Dense VCFs degenerate to the single sweep.
    struct Target { int32_t tid; int64_t start, end; uint64_t voff; };
    std::vector<Target> targets;
    bool use_index = false;
    int lin_shift = 14;                   // bases per window of the table below, as a shift (.bai: 16 kb; .csi: its min_shift)
    std::vector<std::vector<uint64_t>> lin;
    {
        // the .bai's linear index, or the same table rebuilt from a .csi's leaf bins (src/main.rs:520-529 accepts either)
        if (!VTXH_DEV_ENV("VTXH_NO_INDEX") && (read_bai_linear(std::string(a->bam) + ".bai", bam_refs.size(), lin) ||
                                                read_csi_linear(std::string(a->bam) + ".csi", bam_refs.size(), lin, &lin_shift))) {
            use_index = true;
            for (size_t t = 0; t < by_tid.size(); ++t) {
                const auto& iv = by_tid[t];
                const auto& li = lin[t];
                for (const Interval& x : iv) {
                    if (li.empty()) continue;                                  // no alignment on this contig
                    size_t w = (size_t)(x.start >> lin_shift);
                    if (w >= li.size()) continue;                              // nothing overlaps this window or any later one
                    uint64_t v = 0;
                    for (size_t k = w + 1; k-- > 0 && !v;) v = li[k];          // 0 = "not recorded": fall back to an earlier window
                    targets.push_back(Target{(int32_t)t, x.start, x.end, v});  // v == 0: from the first alignment of the file
                }
            }
        }
    }
    if (plan) {
        // ---- the plan of a device-side ingest (vtx_submit_bam): the loci, the BGZF blocks that can hold their reads, the record starts
        //      the index names inside them, and where to stop.  No read is touched here. ----
        P->is_plan = true;
        P->pl_mapq = a->mapq; P->pl_primary = a->primary_only; P->pl_nodup = a->no_duplicates;
        P->pl_tag[0] = bam_tag[0]; P->pl_tag[1] = bam_tag[1];
        for (size_t l = 0; l < loci.size(); ++l) {
            const LocusBuild& L = loci[l];
            vtx_locus o{};
            o.row = L.row;
            o.ref_off = (uint32_t)L.ref_off; o.ref_len = L.ref_len;
            o.alt_off = (uint32_t)L.alt_off; o.alt_len = L.alt_len;
            P->loci.push_back(o);
        }
        P->blocks_total = blocks.size();
        auto done = [&](const char* why) { if (why) P->plan_reason = why; else P->planned = true; P->bam_map = std::move(bam_holder); *out = P.release(); return VTX_OK; };
        P->pl_tid_begin.assign(bam_refs.size() + 1, 0);
        P->pl_span.assign(bam_refs.size(), 1);
        for (size_t t = 0; t < by_tid.size(); ++t) {
            P->pl_tid_begin[t] = (uint32_t)P->pl_iv.size();
            for (const Interval& x : by_tid[t]) {
                if (x.start > INT32_MAX || x.end > INT32_MAX) return done("a locus beyond 2^31 on its contig");
                P->pl_iv.push_back(vtx_bam_interval{(int32_t)x.start, (int32_t)x.end, x.locus, 0});
            }
            if (max_span[t] > INT32_MAX) return done("a locus longer than 2^31");
            P->pl_span[t] = (int32_t)max_span[t];
        }
        P->pl_tid_begin[bam_refs.size()] = (uint32_t)P->pl_iv.size();
        if (!use_index) return done("no usable .bai / .csi next to the BAM (the record starts come from the index)");
        if (targets.empty()) return done(nullptr);                   // no locus can have reads: nothing to inflate
        std::vector<uint64_t> ustart(blocks.size() + 1, 0);
        for (size_t b = 0; b < blocks.size(); ++b) ustart[b + 1] = ustart[b] + blocks[b].isize;
        auto blk_of = [&](uint64_t voff) -> size_t {                  // the block that starts at voff's compressed offset (blocks.size(): none)
            const size_t co = (size_t)(voff >> 16);
            size_t lo = 0, hi = blocks.size();
            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (blocks[mid].start < co) lo = mid + 1; else hi = mid; }
            return lo < blocks.size() && blocks[lo].start == co ? lo : blocks.size();
        };
        auto upos_of = [&](uint64_t voff, uint64_t* up) -> bool {
            const size_t b = blk_of(voff);
            if (b == blocks.size() || (voff & 0xffff) > blocks[b].isize) return false;
            *up = ustart[b] + (voff & 0xffff);
            return true;
        };
        auto record_at = [&](uint64_t voff, int32_t* rt, int64_t* rp) -> bool {      // (tid, pos) of the record that starts at voff
            size_t b = blk_of(voff);
            if (b == blocks.size()) return false;
            const size_t within = (size_t)(voff & 0xffff);
            std::vector<unsigned char> tmp;
            while (tmp.size() < within + 12 && b < blocks.size()) {
                const size_t o = tmp.size();
                tmp.resize(o + blocks[b].isize + 8);
                if (!inflate_block(bam_file, blocks[b], tmp.data() + o)) return false;
                tmp.resize(o + blocks[b].isize);
                ++b;
            }
            if (tmp.size() < within + 12) return false;
            *rt = rdi32(tmp.data() + within + 4); *rp = rdi32(tmp.data() + within + 8);
            return true;
        };
        const uint64_t first_rec = buf_origin + buf_pos;             // where the header ends
        uint64_t start_upos = first_rec, start_voff = 0;
        if (targets[0].voff) {
            start_voff = targets[0].voff;
            if (!upos_of(start_voff, &start_upos)) return done("the .bai names an offset that is not in the BAM");
            if (start_upos < first_rec) return done("the .bai names an offset inside the BAM header");
        }
        // where to stop: the first indexed record at or beyond the end of the last contig's last locus (coordinate-sorted file:
        // nothing later can overlap a locus), else the first record of a later contig, else the end of the file
        const int32_t tl = targets.back().tid;
        int64_t seg_end = 0;
        for (const Target& t : targets) if (t.tid == tl) seg_end = std::max(seg_end, t.end);
        uint64_t end_upos = ustart[blocks.size()];
        bool found_end = false;
        {
            const auto& li = lin[(size_t)tl];
            uint64_t prev = 0;
            int probes = 0;
            for (size_t w = (size_t)(seg_end >> lin_shift); w < li.size() && !found_end; ++w) {
                const uint64_t v = li[w];
                if (!v || v == prev || v <= start_voff) continue;
                prev = v;
                int32_t rt; int64_t rp;
                if (!record_at(v, &rt, &rp)) return done("the .bai names an offset that is not a record of the BAM");
                if (rt < 0 || rt > tl || (rt == tl && rp >= seg_end)) {
                    if (!upos_of(v, &end_upos)) return done("the .bai names an offset that is not in the BAM");
                    found_end = true;
                }
                if (++probes > 64) break;                            // (a pile-up of long records over the locus: sweep to the next contig)
            }
            for (size_t t = (size_t)tl + 1; t < lin.size() && !found_end; ++t)
                for (const uint64_t v : lin[t])
                    if (v) { if (!upos_of(v, &end_upos)) return done("the .bai names an offset that is not in the BAM"); found_end = true; break; }
        }
        if (end_upos <= start_upos) return done(nullptr);            // nothing between: no reads
        // blocks [b0, b1) hold [start_upos, end_upos)
        const size_t b0 = (size_t)(std::upper_bound(ustart.begin(), ustart.end(), start_upos) - ustart.begin()) - 1;
        const size_t b1 = (size_t)(std::lower_bound(ustart.begin(), ustart.end(), end_upos) - ustart.begin());
        // sparse loci far apart: the host's index-guided sweep inflates a few blocks per locus; one contiguous range would inflate
        // everything between the first and the last
        if (ustart[b1] - ustart[b0] > ((uint64_t)64 << 20) && (ustart[b1] - ustart[b0]) >> 20 > targets.size())       // (> 1 MiB of BAM per locus)
            return done("sparse loci (an index-guided sweep on the host inflates less)");
        if (ustart[b1] - ustart[b0] > ((uint64_t)48 << 30)) return done("more than 48 GiB of inflated BAM in one range: stream ranges of loci");
        for (size_t b = b0; b < b1; ++b) P->pl_blocks.push_back(vtx_bgzf_block{(uint64_t)blocks[b].coff, blocks[b].clen, blocks[b].isize});
        const uint64_t base = ustart[b0];
        P->pl_seeds.push_back(start_upos - base);
        for (size_t t = (size_t)targets[0].tid; t <= (size_t)tl; ++t)
            for (const uint64_t v : lin[t]) {
                if (!v) continue;
                uint64_t up;
                if (!upos_of(v, &up)) return done("the .bai names an offset that is not in the BAM");
                if (up > start_upos && up < end_upos) P->pl_seeds.push_back(up - base);
            }
        std::sort(P->pl_seeds.begin(), P->pl_seeds.end());
        P->pl_seeds.erase(std::unique(P->pl_seeds.begin(), P->pl_seeds.end()), P->pl_seeds.end());
        P->pl_end = end_upos - base;
        // a record chain between two seeds is walked by ONE lane: a pile-up of hundreds of MB inside one 16 kb window (amplicon data)
        // would serialise the device — the host's sweep indexes records at memory speed
        for (size_t i = 0; i < P->pl_seeds.size(); ++i) {
            const uint64_t stop = i + 1 < P->pl_seeds.size() ? P->pl_seeds[i + 1] : P->pl_end;
            if (stop - P->pl_seeds[i] > ((uint64_t)256 << 20)) { P->pl_blocks.clear(); P->pl_seeds.clear(); return done("more than 256 MiB of BAM between two indexed record starts"); }
        }
        P->blocks_inflated = b1 - b0;
        ph.mark("plan");
        return done(nullptr);
    }
    // ---- sweep the BAM (fetch + filters of evaluate_alns, :822-895) ----
    // Per window of inflated blocks: record boundaries are indexed sequentially (a hop per record), the
    // records are parsed and filtered by `threads` workers over contiguous ranges into thread-local
    // outputs, and the outputs are merged in thread order — so every locus sees its reads in BAM order,
    // exactly like one sequential sweep.
    // rr offsets are relative to the worker's arenas; roff / toff: where those start in the global arenas (64-bit)
    // 32 bytes per surviving (read, locus) pair.  roff: the read's bases; toff: the record's first tag byte — raw mode the barcode,
    // then (umi_len != VTX_TAG_MISSING) the UMI right behind it; cooked mode (bc_len 0) the UMI alone.  (Round 3 kept 48 bytes per
    // pair AND a second copy sorted by locus: every byte of either is a page the process touches for the first time, and
    // those pages — not the work on them — are what the packer's time is made of.)
    struct Hit { uint32_t locus, cell, read_len; uint16_t bc_len, umi_len; uint64_t roff, toff; };
    static_assert(sizeof(Hit) == 32, "compact hit");
    // reads are not copied by the filter pass: it notes where each kept read's packed bases lie (the window's bytes stay
    // put until the parse is over) and the second pass decodes them straight into the global arena
    struct Decode { const unsigned char* sq; uint32_t l_seq; uint32_t off; };
    // (one per worker, written on every record: each on its own cache lines, or the workers fight over them)
    struct alignas(128) WorkerOut { std::vector<Hit> hits; std::vector<Decode> dec; uint64_t reads_size = 0; std::string tags; vtxh_metrics m{}; std::string err; uint64_t rbase = 0; int32_t hint_tid = -1; size_t hint_hi = 0; };
    ByteBuf& reads = P->read_arena;
    // a->read_format VTX_READS_NIBBLES: the arena keeps the BAM's two-bases-per-byte form (half the pages to touch here, half the
    // bytes over PCIe; the device unpacks: vtx_set_read_format); reads_bases counts BASES either way
    const bool nibbles = a->read_format == VTX_READS_NIBBLES;
    P->read_format = nibbles ? VTX_READS_NIBBLES : VTX_READS_BYTES;
    uint64_t reads_bases = 0;
    auto process = [&](const unsigned char* r, uint32_t bs, WorkerOut& o, std::vector<uint32_t>& hits) -> bool {
        const int32_t tid = rdi32(r);
        if (tid < 0 || (size_t)tid >= by_tid.size() || by_tid[(size_t)tid].empty()) return true;
        const int64_t pos = rdi32(r + 4);
        const uint32_t l_rn = r[8], mapq = r[9];
        const uint32_t n_cig = r[12] | (r[13] << 8), flag = r[14] | (r[15] << 8);
        const uint32_t l_seq = rd32(r + 16);
        const unsigned char* cig = r + 32 + l_rn;
        const unsigned char* sq = cig + 4 * (size_t)n_cig;
        const unsigned char* aux = sq + (l_seq + 1) / 2 + l_seq;
        if (aux > r + bs) { o.err = "malformed BAM record"; return false; }
        // bam_endpos: unmapped or no reference-consuming op => pos + 1
        int64_t rlen = 0;
        if (!(flag & FLAG_UNMAP))
            for (uint32_t k = 0; k < n_cig; ++k) {
                uint32_t c = rd32(cig + 4 * k), op = c & 15;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4;
            }
        const int64_t endpos = pos + (rlen > 0 ? rlen : 1);
        // loci of this contig with start < endpos && end > pos, in VCF-independent (start) order
        const auto& iv = by_tid[(size_t)tid];
        hits.clear();
        // hi = first interval with start >= endpos.  Records come in coordinate order, so the answer is almost always within
        // a few intervals of the previous record's: walk from there, binary search only after a long hop.
        size_t hi;
        {
            size_t h = o.hint_tid == tid ? std::min(o.hint_hi, iv.size()) : iv.size() + 1;
            int steps = 0;
            if (h <= iv.size()) {
                while (h < iv.size() && iv[h].start < endpos && steps < 16) { ++h; ++steps; }
                while (h > 0 && iv[h - 1].start >= endpos && steps < 16) { --h; ++steps; }
            }
            if (h > iv.size() || steps >= 16)
                h = (size_t)(std::lower_bound(iv.begin(), iv.end(), endpos, [](const Interval& x, int64_t e) { return x.start < e; }) - iv.begin());
            hi = h; o.hint_tid = tid; o.hint_hi = h;
        }
        for (size_t k = hi; k-- > 0;) {
            if (iv[k].start + max_span[(size_t)tid] <= pos) break;
            if (iv[k].end > pos) hits.push_back(iv[k].locus);
        }
        if (hits.empty()) return true;
        bool seq_ready = false, tags_ready = false;
        vtx_raw_record rr{};
        uint32_t cell = 0;
        bool has_cell = false;
        for (uint32_t li : hits) {
            const LocusBuild& L = loci[li];
            ++o.m.num_reads;                                                     // :831
            if (mapq < a->mapq) { ++o.m.num_low_mapq; continue; }                 // :833
            if (a->primary_only && (flag & (FLAG_SECONDARY | FLAG_SUPP))) { ++o.m.num_non_primary; continue; }   // :841
            if (a->no_duplicates && (flag & FLAG_DUP)) { ++o.m.num_duplicates; continue; }                      // :849
            if (!useful_alignment(cig, n_cig, pos, L.start, L.end)) { ++o.m.num_not_useful; continue; }          // :857
            if (!tags_ready) {
                // tag bytes are copied once per read; raw mode ships them to the device as they are
                tags_ready = true;
                const unsigned char* val; size_t vlen;
                rr.bc_len = VTX_TAG_MISSING; rr.umi_len = VTX_TAG_MISSING;
                if (aux_string(aux, (size_t)(r + bs - aux), bam_tag.c_str(), &val, &vlen) && vlen < VTX_TAG_MISSING) {   // :867
                    if (raw) {
                        rr.bc_off = (uint32_t)o.tags.size(); rr.bc_len = (uint16_t)vlen;
                        o.tags.append((const char*)val, vlen);
                    } else {
                        if (bc_table.find(val, vlen, &cell)) { has_cell = true; rr.bc_len = 0; }
                    }
                }
                if ((raw || a->use_umi) && rr.bc_len != VTX_TAG_MISSING && aux_string(aux, (size_t)(r + bs - aux), "UB", &val, &vlen) == 1 && vlen < VTX_TAG_MISSING) {   // :879
                    rr.umi_off = (uint32_t)o.tags.size(); rr.umi_len = (uint16_t)vlen;
                    o.tags.append((const char*)val, vlen);
                }
            }
            // raw: only a missing / non-Z barcode tag is decided here; cooked: the in-list test too (:870-876)
            if (raw ? rr.bc_len == VTX_TAG_MISSING : !has_cell) { ++o.m.num_not_cell_bc; continue; }
            if (!raw && a->use_umi && rr.umi_len == VTX_TAG_MISSING) { ++o.m.num_non_umi; continue; }           // :879-888
            if (!seq_ready) {                                                               // rec.seq().as_bytes() :896
                seq_ready = true;
                rr.read_off = (uint32_t)o.reads_size;                           // one copy per read, shared by its loci
                o.dec.push_back(Decode{sq, l_seq, rr.read_off});
                o.reads_size += nibbles ? ((uint64_t)l_seq + 1) & ~(uint64_t)1 : l_seq;     // (nibbles: every read starts at an even base)
            }
            rr.read_len = l_seq;
            // (offsets relative to this slice's arenas until the slice is copied out; the UMI, when there is one, was appended right
            //  behind the barcode — or alone, in cooked mode)
            o.hits.push_back(Hit{li, cell, l_seq, raw ? rr.bc_len : (uint16_t)0, rr.umi_len, rr.read_off,
                                 raw ? rr.bc_off : (rr.umi_len != VTX_TAG_MISSING ? rr.umi_off : 0u)});
        }
        return o.reads_size <= 0xffffffffull && o.tags.size() <= 0xffffffffull;
    };
    std::vector<size_t> rec_offs;
    // every window's worker outputs go to the global arrays at prefix offsets (thread order = BAM order), copied by the
    // workers themselves
    // a window is cut into more slices than workers and the workers take slices as they come (a worker that shares its core
    // with the indexing thread, or was descheduled, does not hold the window up); slice order = BAM order
    std::vector<WorkerOut> outs((size_t)threads * (threads > 1 ? 4 : 1));
    ByteBuf hit_store;                         // Hit[]: every surviving (read, locus) pair, BAM order, offsets into the global arenas
    size_t n_hits = 0;
    ByteBuf& tag_store = P->tag_arena;         // raw: barcode + UMI bytes for the device; cooked: UMI bytes until the ids are assigned
    // index-guided sweep state: targets [tg, ...) still to serve; the running segment ends at (seg_tid, seg_end)
    size_t tg = 0;
    int32_t seg_tid = -1;
    int64_t seg_end = 0;
    const size_t kNearBlocks = 64;                    // a next offset this close is reached by sweeping on
    const uint64_t first_voff = ((uint64_t)0);        // (targets with voff 0 start where the header ended: no jump needed)
    (void)first_voff;
    auto open_segment = [&]() {                       // targets[tg] opens a segment; loci starting inside it join it
        seg_tid = targets[tg].tid; seg_end = targets[tg].end;
        size_t k = tg + 1;
        while (k < targets.size() && targets[k].tid == seg_tid && targets[k].start < seg_end) { seg_end = std::max(seg_end, targets[k].end); ++k; }
    };
    auto block_of = [&](uint64_t voff) -> size_t {    // index of the block that starts at the compressed offset of voff
        const size_t co = (size_t)(voff >> 16);
        size_t lo = 0, hi = blocks.size();
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (blocks[mid].start < co) lo = mid + 1; else hi = mid; }
        return lo;
    };
    bool jump_pending = false;
    uint64_t jump_voff = 0;
    size_t far_ptr = 0;                               // first target whose offset lies beyond the near range of the sweep
    auto update_read_ahead = [&]() {                  // the next offset the sweep would JUMP to bounds the read-ahead
        if (far_ptr < tg) far_ptr = tg;
        while (far_ptr < targets.size() && (!targets[far_ptr].voff || block_of(targets[far_ptr].voff) <= next_block + kNearBlocks)) ++far_ptr;
        chunk_limit_block = far_ptr < targets.size() ? block_of(targets[far_ptr].voff) : SIZE_MAX;
    };
    if (use_index) {
        if (targets.empty()) { buf.drop_prefix(buf.size()); buf_pos = 0; next_block = blocks.size(); }   // no locus can have reads
        else {
            open_segment();
            if (targets[0].voff && block_of(targets[0].voff) > next_block + kNearBlocks) { jump_pending = true; jump_voff = targets[0].voff; }
        }
    }
    // ---- parse + filter of one indexed window (pend_offs), all threads; then every worker copies its output to prefix
    //      offsets of the global arrays (thread order = BAM order) ----
    std::vector<size_t> pend_offs;
    int parse_rc = VTX_OK;
    std::string parse_msg;
    double t_index_s = 0, t_parse_s = 0, t_filter_s = 0;               // (VTXH_PROFILE: the serial record index against the parallel parse it overlaps)
    auto parse_pending = [&]() {
        const size_t nrec = pend_offs.size();
        if (!nrec) return;
        const auto t_p0 = std::chrono::steady_clock::now();
        struct Acc { double& d; std::chrono::steady_clock::time_point t0; ~Acc() { d += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } acc{t_parse_s, t_p0};
        const unsigned char* pend_base = pend_detached ? pend_store.data() : buf.data();
        const size_t n_slices = outs.size();
        std::atomic<size_t> next_slice{0};
        pool.run([&](size_t) {
            std::vector<uint32_t> hits;
            for (size_t c; (c = next_slice.fetch_add(1)) < n_slices;) {
                WorkerOut& o = outs[c];
                o.hits.clear(); o.dec.clear(); o.reads_size = 0; o.tags.clear(); o.m = vtxh_metrics{}; o.err.clear();
                for (size_t k = nrec * c / n_slices, e = nrec * (c + 1) / n_slices; k < e; ++k) {
                    const unsigned char* rp = pend_base + pend_offs[k];
                    if (!process(rp + 4, rd32(rp), o, hits)) { if (o.err.empty()) o.err = "one window of the BAM holds more than 4 GiB of read bases"; break; }
                }
            }
        });
        t_filter_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_p0).count();
        std::vector<uint64_t> tbase(n_slices), hbase(n_slices);
        uint64_t rtotal = reads_bases, ttotal = tag_store.size(), htotal = n_hits;
        for (size_t t = 0; t < outs.size(); ++t) {
            WorkerOut& o = outs[t];
            if (!o.err.empty()) { parse_rc = o.err[0] == 'm' ? VTX_E_INVAL : VTX_E_UNSUPPORTED; parse_msg = o.err; return; }
            o.rbase = rtotal; rtotal += o.reads_size;
            tbase[t] = ttotal; ttotal += o.tags.size();
            hbase[t] = htotal; htotal += o.hits.size();
            const uint64_t* src = &o.m.num_reads;
            uint64_t* dst = &P->metrics.num_reads;
            for (int k = 0; k < 9; ++k) dst[k] += src[k];
        }
        if (!reads.grow((size_t)((rtotal - reads_bases) / (nibbles ? 2 : 1))) || !tag_store.grow((size_t)(ttotal - tag_store.size())) ||
            !hit_store.grow((size_t)(htotal - n_hits) * sizeof(Hit))) { parse_rc = VTX_E_NOMEM; parse_msg = "out of memory growing the read arenas"; return; }
        n_hits = (size_t)htotal; reads_bases = rtotal;
        Hit* all = (Hit*)hit_store.data();
        next_slice = 0;
        pool.run([&](size_t) {
          for (size_t t; (t = next_slice.fetch_add(1)) < n_slices;) {
            const WorkerOut& o = outs[t];
            for (const Decode& d : o.dec) {                                                  // rec.seq().as_bytes() :896
                if (nibbles) {                                                               // the record's own packed bases, as they are
                    memcpy(reads.data() + (o.rbase + d.off) / 2, d.sq, ((size_t)d.l_seq + 1) / 2);
                    continue;
                }
                unsigned char* dst = reads.data() + o.rbase + d.off;
                for (uint32_t k = 0; k + 1 < d.l_seq; k += 2) memcpy(dst + k, &kNt16Pair[d.sq[k >> 1]], 2);
                if (d.l_seq & 1) dst[d.l_seq - 1] = (unsigned char)kNt16[d.sq[d.l_seq >> 1] >> 4];
            }
            if (!o.tags.empty()) memcpy(tag_store.data() + tbase[t], o.tags.data(), o.tags.size());
            Hit* dst = all + hbase[t];
            for (size_t k = 0; k < o.hits.size(); ++k) {
                Hit h = o.hits[k];
                h.roff += o.rbase; h.toff += tbase[t];
                dst[k] = h;
            }
          }
        });
        pend_offs.clear();
        pend_begin = SIZE_MAX; pend_detached = false;
    };
    // the pending window is parsed on a helper thread (which drives the pool) while this thread indexes the next one
    std::thread parse_thread;
    auto finish_parse = [&]() -> bool {
        if (parse_thread.joinable()) parse_thread.join();
        return parse_rc == VTX_OK;
    };
    struct ParseJoiner { std::thread& t; ~ParseJoiner() { if (t.joinable()) t.join(); } } parse_joiner{parse_thread};
    while (true) {
        if (jump_pending) {
            // the buffer restarts elsewhere: the window still waiting for its parse goes first
            parse_pending();
            if (parse_rc != VTX_OK) return fail(parse_rc, "%s: %s", a->bam, parse_msg.c_str());
            // restart the record stream at a virtual offset: drop what is buffered, inflate from that block on
            jump_pending = false;
            ++n_jumps;
            const size_t b = block_of(jump_voff);
            if (b >= blocks.size() || blocks[b].start != (size_t)(jump_voff >> 16)) return fail(VTX_E_INVAL, "%s.bai: offset outside the BAM", a->bam);
            buf.drop_prefix(buf.size());
            buf_pos = 0;
            next_block = b; chunk_blocks = std::min<size_t>(32, max_chunk_blocks);
            refill((size_t)(jump_voff & 0xffff) + 1);
            if (refill_failed) return fail(VTX_E_INVAL, "%s: a BGZF block does not inflate (or out of memory)", a->bam);
            buf_pos = (size_t)(jump_voff & 0xffff);
            if (buf_pos > buf.size()) return fail(VTX_E_INVAL, "%s.bai: offset outside its block", a->bam);
        }
        if (use_index) update_read_ahead();
        refill(buf.size() - buf_pos + 1);             // one more chunk of blocks, if the file has one
        if (refill_failed) return fail(VTX_E_INVAL, "%s: a BGZF block does not inflate (or out of memory)", a->bam);
        ph.mark("inflate");
        if (!pend_offs.empty()) parse_thread = std::thread(parse_pending);      // ... beside the indexing below
        const auto t_idx0 = std::chrono::steady_clock::now();
        rec_offs.clear();
        size_t p = buf_pos;
        bool all_served = false;
        while (buf.size() - p >= 4) {
            const uint32_t bs = rd32(buf.data() + p);
            if (buf.size() - p - 4 < bs) break;
            if (bs < 32) return fail(VTX_E_INVAL, "%s: malformed BAM record", a->bam);
            if (use_index) {
                // a record at or beyond the end of the running segment: its loci are served (sorted file)
                const int32_t rt = rdi32(buf.data() + p + 4);
                const int64_t rp = rdi32(buf.data() + p + 8);
                bool moved = false;
                while (tg < targets.size() && (rt < 0 || rt > seg_tid || (rt == seg_tid && rp >= seg_end))) {
                    while (tg < targets.size() && targets[tg].tid == seg_tid && targets[tg].start < seg_end) ++tg;   // served
                    if (tg == targets.size()) break;
                    open_segment();
                    moved = true;
                }
                if (tg == targets.size()) { all_served = true; break; }
                if (moved && targets[tg].voff) {
                    // where does the next segment's first possible record live?  far ahead: jump; near (or behind): sweep on
                    const size_t nb = block_of(targets[tg].voff);
                    if (nb > next_block + kNearBlocks) { jump_pending = true; jump_voff = targets[tg].voff; break; }
                }
            }
            rec_offs.push_back(p);
            p += 4 + (size_t)bs;
            __builtin_prefetch(buf.data() + p + 8 * (4 + (size_t)bs));      // records are of similar size: the chain is predictable
            __builtin_prefetch(buf.data() + p + 8 * (4 + (size_t)bs) + 64);
        }
        t_index_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_idx0).count();
        if (!finish_parse()) return fail(parse_rc, "%s: %s", a->bam, parse_msg.c_str());
        const bool eof = next_block >= blocks.size();
        if (rec_offs.empty() && (all_served || jump_pending)) {
            if (all_served) break;
            continue;                                  // nothing to parse before the jump
        }
        if (rec_offs.empty()) {
            if (!eof) continue;                        // a record larger than the window: load more
            if (buf.size() - buf_pos >= 4) return fail(VTX_E_INVAL, "%s: truncated BAM record", a->bam);
            break;
        }
        ph.mark("record index + parse of the window before");
        // this window's parse runs beside the indexing of the next one
        pend_offs.swap(rec_offs);
        pend_begin = pend_offs.front(); pend_detached = false;
        buf_pos = p;
        if (all_served) break;
        if (jump_pending) continue;
        if (eof && buf.size() - buf_pos < 4) break;
    }
    parse_pending();                                   // the last window
    if (parse_rc != VTX_OK) return fail(parse_rc, "%s: %s", a->bam, parse_msg.c_str());
    if (getenv("VTXH_PROFILE")) fprintf(stderr, "[vtxh]   record index %.3f s (one thread), parse %.3f s (%d threads; filter pass %.3f s), side by side\n", t_index_s, t_parse_s, threads, t_filter_s);
    P->blocks_inflated = n_inflated; P->blocks_total = blocks.size(); P->index_jumps = n_jumps;

    // ---- group the hits by locus: stable counting sort (hits are in BAM order, so every locus keeps it); thread t owns the
    //      t-th slice of the hits, and within a locus the slices land in thread order ----
    const size_t nloc = loci.size();
    const Hit* all_hits = (const Hit*)hit_store.data();
    std::vector<uint64_t> l_begin(nloc + 1, 0);
    // (the sort moves 4-byte indices, not the hits: by_locus(j) is the j-th hit in (locus, BAM) order)
    if (n_hits > 0xffffffffull) return fail(VTX_E_UNSUPPORTED, "more than 2^32 (read, locus) pairs in one pack: pack ranges of VCF rows");
    ByteBuf order_store;
    if (!order_store.grow(n_hits * sizeof(uint32_t))) return fail(VTX_E_NOMEM, "out of memory sorting the reads");
    uint32_t* order = (uint32_t*)order_store.data();
    auto by_locus = [&](uint64_t j) -> const Hit& { return all_hits[order[j]]; };
    {
        // thread t counts its slice over the locus range the slice touches (narrow in a sorted file); the cursors are
        // then handed out in thread order, so within a locus the slices land in BAM order
        const size_t T = (size_t)threads;
        struct Slice { size_t k0, k1; uint32_t lmin, lmax; std::vector<uint64_t> cur; };
        std::vector<Slice> sl(T);
        pool.run([&](size_t t) {
            Slice& S = sl[t];
            S.k0 = n_hits * t / T; S.k1 = n_hits * (t + 1) / T; S.lmin = UINT32_MAX; S.lmax = 0;
            for (size_t k = S.k0; k < S.k1; ++k) { S.lmin = std::min(S.lmin, all_hits[k].locus); S.lmax = std::max(S.lmax, all_hits[k].locus); }
        });
        {
            // an unsorted file spreads every slice over all loci: T full-width counters would not pay — one slice then
            size_t width = 0;
            for (const Slice& S : sl) if (S.k0 < S.k1) width += (size_t)S.lmax - S.lmin + 1;
            if (width > (size_t)(64u << 20) && T > 1) {
                Slice all{0, n_hits, UINT32_MAX, 0, {}};
                for (const Slice& S : sl) if (S.k0 < S.k1) { all.lmin = std::min(all.lmin, S.lmin); all.lmax = std::max(all.lmax, S.lmax); }
                sl.assign(T, Slice{0, 0, UINT32_MAX, 0, {}});
                sl[0] = all;
            }
        }
        pool.run([&](size_t t) {
            Slice& S = sl[t];
            if (S.k0 == S.k1) return;
            S.cur.assign((size_t)S.lmax - S.lmin + 1, 0);
            for (size_t k = S.k0; k < S.k1; ++k) ++S.cur[all_hits[k].locus - S.lmin];
        });
        for (const Slice& S : sl)
            for (size_t i = 0; i < S.cur.size(); ++i) l_begin[S.lmin + i + 1] += S.cur[i];
        for (size_t l = 0; l < nloc; ++l) l_begin[l + 1] += l_begin[l];
        {
            std::vector<uint64_t> taken(nloc, 0);          // slots of locus l handed to the slices before this one
            for (Slice& S : sl)
                for (size_t i = 0; i < S.cur.size(); ++i) {
                    const uint64_t c = S.cur[i];
                    S.cur[i] = l_begin[S.lmin + i] + taken[S.lmin + i];
                    taken[S.lmin + i] += c;
                }
        }
        pool.run([&](size_t t) {
            Slice& S = sl[t];
            for (size_t k = S.k0; k < S.k1; ++k) order[S.cur[all_hits[k].locus - S.lmin]++] = (uint32_t)k;
        });
    }
    const size_t n_sorted = n_hits;
    // ---- batches: consecutive loci whose reads / tags span < 4 GiB (32-bit offsets relative to the batch window) ----
    uint64_t limit = 0xF0000000ull;
    if (const char* e = VTXH_DEV_ENV("VTXH_BATCH_BYTES")) limit = std::max<uint64_t>(1, strtoull(e, nullptr, 10));   // tests
    const bool need_tags = raw;
    {
        vtxh_pack::Batch cur{0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t rlo = UINT64_MAX, rhi = 0, tlo = UINT64_MAX, thi = 0;
        auto close = [&](uint32_t l_end) {
            cur.l1 = l_end; cur.rec1 = l_begin[l_end];
            cur.rbase = rlo == UINT64_MAX ? 0 : rlo; cur.rbytes = rlo == UINT64_MAX ? 0 : rhi - rlo;
            cur.tbase = tlo == UINT64_MAX ? 0 : tlo; cur.tbytes = tlo == UINT64_MAX ? 0 : thi - tlo;
            P->batches.push_back(cur);
            cur = vtxh_pack::Batch{l_end, l_end, l_begin[l_end], l_begin[l_end], 0, 0, 0, 0};
            rlo = tlo = UINT64_MAX; rhi = thi = 0;
        };
        // the byte extents of every locus' reads and tags, loci in parallel
        struct Extent { uint64_t a0, a1, b0, b1; };
        std::vector<Extent> ext(nloc);
        pool.run([&](size_t t) {
            for (size_t l = nloc * t / (size_t)threads, e = nloc * (t + 1) / (size_t)threads; l < e; ++l) {
                uint64_t a0 = UINT64_MAX, a1 = 0, b0 = UINT64_MAX, b1 = 0;
                for (uint64_t j = l_begin[l]; j < l_begin[l + 1]; ++j) {
                    const Hit& h = by_locus(j);
                    a0 = std::min(a0, h.roff); a1 = std::max(a1, h.roff + h.read_len);
                    if (need_tags) {
                        b0 = std::min(b0, h.toff);
                        b1 = std::max(b1, h.toff + h.bc_len + (h.umi_len != VTX_TAG_MISSING ? h.umi_len : 0u));
                    }
                }
                ext[l] = Extent{a0, a1, b0, b1};
            }
        });
        for (size_t l = 0; l < nloc; ++l) {
            const uint64_t a0 = ext[l].a0, a1 = ext[l].a1, b0 = ext[l].b0, b1 = ext[l].b1;
            if (a1 - std::min(a0, a1) > limit || b1 - std::min(b0, b1) > limit || l_begin[l + 1] - l_begin[l] > 0x7fffffffull)
                return fail(VTX_E_UNSUPPORTED, "locus %zu alone needs more than %llu bytes of reads", l, (unsigned long long)limit);
            const uint64_t nr0 = std::min(rlo, a0), nr1 = std::max(rhi, a1), nt0 = std::min(tlo, b0), nt1 = std::max(thi, b1);
            const bool fits = (nr1 <= nr0 || nr1 - nr0 <= limit) && (nt1 <= nt0 || nt1 - nt0 <= limit) &&
                              l_begin[l + 1] - cur.rec0 <= 0x7fffffffull;
            if (!fits && l > cur.l0) close((uint32_t)l);
            rlo = std::min(rlo, a0); rhi = std::max(rhi, a1); tlo = std::min(tlo, b0); thi = std::max(thi, b1);
        }
        close((uint32_t)nloc);
    }
    for (size_t l = 0; l < nloc; ++l) {
        const LocusBuild& L = loci[l];
        vtx_locus o{};
        o.row = L.row; o.rec_count = (uint32_t)(l_begin[l + 1] - l_begin[l]);
        o.ref_off = (uint32_t)L.ref_off; o.ref_len = L.ref_len;
        o.alt_off = (uint32_t)L.alt_off; o.alt_len = L.alt_len;
        P->loci.push_back(o);
    }
    std::vector<uint32_t> batch_of(nloc, 0);
    for (size_t b = 0; b < P->batches.size(); ++b)
        for (uint32_t l = P->batches[b].l0; l < P->batches[b].l1; ++l) {
            batch_of[l] = (uint32_t)b;
            P->loci[l].rec_begin = (uint32_t)(l_begin[l] - P->batches[b].rec0);
        }
    if (raw) {
        if (!P->raw_records.alloc(n_sorted)) return fail(VTX_E_NOMEM, "out of memory for %zu records", n_sorted);
        pool.run([&](size_t t) {
            for (size_t l = nloc * t / (size_t)threads, e = nloc * (t + 1) / (size_t)threads; l < e; ++l) {
                const vtxh_pack::Batch& B = P->batches[batch_of[l]];
                for (uint64_t j = l_begin[l]; j < l_begin[l + 1]; ++j) {
                    const Hit& h = by_locus(j);
                    vtx_raw_record rr{};
                    rr.read_off = (uint32_t)(h.roff - B.rbase);
                    rr.read_len = h.read_len;
                    rr.bc_off = (uint32_t)(h.toff - B.tbase);
                    rr.bc_len = h.bc_len;
                    rr.umi_off = h.umi_len != VTX_TAG_MISSING ? (uint32_t)(h.toff + h.bc_len - B.tbase) : 0u;
                    rr.umi_len = h.umi_len;
                    P->raw_records[j] = rr;
                }
            }
        });
        ph.mark("pack");
        *out = P.release();
        return VTX_OK;
    }
    // ---- cooked: UMI ids by first occurrence, then the stable sort by (cell, umi) (:932 + per-cell UMI grouping),
    //      loci in parallel (disjoint output ranges) ----
    if (!P->records.alloc(n_sorted)) return fail(VTX_E_NOMEM, "out of memory for %zu records", n_sorted);
    {
        std::atomic<size_t> next_locus{0};
        auto pack_loci = [&]() {
            std::unordered_map<std::string, uint32_t> umi_ids;
            std::vector<LocusBuild::Rec> recs;
            for (size_t l; (l = next_locus.fetch_add(16)) < nloc;)
                for (size_t ll = l; ll < std::min(nloc, l + 16); ++ll) {
                    const uint64_t rbase = P->batches[batch_of[ll]].rbase;
                    umi_ids.clear(); recs.clear();
                    for (uint64_t j = l_begin[ll]; j < l_begin[ll + 1]; ++j) {
                        const Hit& h = by_locus(j);
                        uint32_t uid = 0;     // without --umi every read carries the same dummy UMI (:890-894)
                        if (a->use_umi) uid = umi_ids.emplace(std::string((const char*)tag_store.data() + h.toff + h.bc_len, h.umi_len), (uint32_t)umi_ids.size()).first->second;
                        recs.push_back(LocusBuild::Rec{h.cell, uid, h.roff - rbase, h.read_len});
                    }
                    std::stable_sort(recs.begin(), recs.end(), [](const LocusBuild::Rec& x, const LocusBuild::Rec& y) {
                        return x.cell != y.cell ? x.cell < y.cell : x.umi < y.umi;
                    });
                    for (size_t k = 0; k < recs.size(); ++k)
                        P->records[l_begin[ll] + k] = vtx_record{(uint32_t)recs[k].read_off, recs[k].read_len, recs[k].cell, recs[k].umi};
                }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(pack_loci);
        pack_loci();
        for (auto& t : th) t.join();
    }
    tag_store.release();
    ph.mark("sort + pack");
    *out = P.release();
    return VTX_OK;
}

int vtxh_read_format(const vtxh_pack* p) { return p ? p->read_format : VTX_READS_BYTES; }

int vtxh_test_inflate(const uint8_t* in, uint64_t in_len, uint8_t* out, uint64_t out_len) {
    vtxinf::Tables T;
    return vtxinf::inflate_raw(in, (size_t)in_len, out, (size_t)out_len, T) ? 1 : 0;
}

}  // extern "C"
