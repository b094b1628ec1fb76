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
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/vtx_host.h"

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
        for (uint64_t p = start; p < end; ++p) {
            uint64_t off = e.offset + (p / e.linebases) * e.linewidth + p % e.linebases;
            unsigned char c = off < data.size() ? data.data()[off] : 'N';
            if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
            out.push_back((char)c);
        }
    }
};

// ---- VCF (text, optionally gz) -------------------------------------------------
struct VcfRec { std::string chrom; int64_t pos; std::vector<std::string> alleles; };

// ---- BAM ------------------------------------------------------------------------
const char kNt16[] = "=ACMGRSVTWYHKDBN";
enum { FLAG_UNMAP = 0x4, FLAG_SECONDARY = 0x100, FLAG_DUP = 0x400, FLAG_SUPP = 0x800 };

struct BgzfBlock { size_t coff; uint32_t clen; uint32_t isize; };

bool index_bgzf(const MappedFile& file, std::vector<BgzfBlock>& blocks) {
    size_t o = 0;
    while (o + 18 <= file.size()) {
        const unsigned char* h = (const unsigned char*)file.data() + o;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
        uint32_t xlen = h[10] | (h[11] << 8);
        uint32_t bsize = 0;
        bool found = false;
        size_t x = 12;
        while (x + 4 <= 12 + xlen) {
            uint32_t slen = h[x + 2] | (h[x + 3] << 8);
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) { bsize = (h[x + 4] | (h[x + 5] << 8)) + 1u; found = true; }
            x += 4 + slen;
        }
        if (!found || o + bsize > file.size() || bsize < 12 + xlen + 8) return false;
        const unsigned char* t = h + bsize - 4;
        uint32_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        blocks.push_back(BgzfBlock{o + 12 + xlen, bsize - 12 - xlen - 8, isize});
        o += bsize;
    }
    return o == file.size();
}

bool inflate_block(const MappedFile& file, const BgzfBlock& b, unsigned char* dst) {
    if (b.isize == 0) return true;
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

struct LocusBuild {
    uint32_t row;
    int64_t start, end;
    std::string ref_hap, alt_hap;
    struct Rec { uint32_t cell, umi; uint64_t read_off; uint32_t read_len; };
    std::vector<Rec> recs;
    std::vector<vtx_raw_record> raw_recs;      // raw mode: BAM order, tags as bytes
    std::unordered_map<std::string, uint32_t> umi_ids;
};

struct Interval { int64_t start, end; uint32_t locus; };

}  // namespace

struct vtxh_pack {
    std::vector<vtx_locus> loci;
    std::vector<vtx_record> records;
    std::string hap_arena, read_arena;
    vtxh_metrics metrics{};
    uint32_t n_variants = 0;
    std::vector<std::string> barcodes, variant_names;
    // raw mode (vtxh_pack_files_raw)
    std::vector<vtx_raw_record> raw_records;
    std::string tag_arena, bc_bytes;
    std::vector<uint64_t> bc_offsets;
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
    FILE* f = fopen(path, "wb");
    if (!f) return fail(VTX_E_INVAL, "cannot open %s for writing", path);
    std::vector<char> buf(1 << 20);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%% written by sprs\n%u %u %llu\n", n_rows, n_cols,
            (unsigned long long)nnz);
    char line[96], num[40];
    for (uint64_t k = 0; k < nnz; ++k) {
        char* p = line;
        p = std::to_chars(p, p + 12, row[k] + 1u).ptr; *p++ = ' ';
        p = std::to_chars(p, p + 12, col[k] + 1u).ptr; *p++ = ' ';
        int n = vtxh_format_f64(value[k], num);
        memcpy(p, num, (size_t)n); p += n; *p++ = '\n';
        fwrite(line, 1, (size_t)(p - line), f);
    }
    bool ok = fclose(f) == 0;
    return ok ? VTX_OK : fail(VTX_E_INVAL, "error writing %s", path);
}

void vtxh_free(vtxh_pack* p) { delete p; }
void vtxh_get_batch(const vtxh_pack* p, vtx_batch* out) {
    out->loci = p->loci.data(); out->n_loci = (uint32_t)p->loci.size();
    out->records = p->records.data(); out->n_records = (uint32_t)p->records.size();
    out->hap_arena = (const uint8_t*)p->hap_arena.data(); out->hap_bytes = p->hap_arena.size();
    out->read_arena = (const uint8_t*)p->read_arena.data(); out->read_bytes = p->read_arena.size();
}
void vtxh_get_metrics(const vtxh_pack* p, vtxh_metrics* out) { *out = p->metrics; }
uint32_t vtxh_num_variants(const vtxh_pack* p) { return p->n_variants; }
uint32_t vtxh_num_barcodes(const vtxh_pack* p) { return (uint32_t)p->barcodes.size(); }
const char* vtxh_variant_name(const vtxh_pack* p, uint32_t i) { return i < p->variant_names.size() ? p->variant_names[i].c_str() : ""; }
const char* vtxh_barcode(const vtxh_pack* p, uint32_t j) { return j < p->barcodes.size() ? p->barcodes[j].c_str() : ""; }

void vtxh_get_raw_batch(const vtxh_pack* p, vtx_raw_batch* out) {
    out->loci = p->loci.data(); out->n_loci = (uint32_t)p->loci.size();
    out->records = p->raw_records.data(); out->n_records = (uint32_t)p->raw_records.size();
    out->hap_arena = (const uint8_t*)p->hap_arena.data(); out->hap_bytes = p->hap_arena.size();
    out->read_arena = (const uint8_t*)p->read_arena.data(); out->read_bytes = p->read_arena.size();
    out->tag_arena = (const uint8_t*)p->tag_arena.data(); out->tag_bytes = p->tag_arena.size();
}
void vtxh_get_barcode_table(const vtxh_pack* p, const uint8_t** bytes, const uint64_t** offsets, uint32_t* n) {
    *bytes = (const uint8_t*)p->bc_bytes.data(); *offsets = p->bc_offsets.data(); *n = (uint32_t)p->barcodes.size();
}

static int pack_impl(const vtxh_args* a, bool raw, vtxh_pack** out);
int vtxh_pack_files(const vtxh_args* a, vtxh_pack** out) { return pack_impl(a, false, out); }
int vtxh_pack_files_raw(const vtxh_args* a, vtxh_pack** out) { return pack_impl(a, true, out); }

static int pack_impl(const vtxh_args* a, bool raw, vtxh_pack** out) {
    if (!a || !out || !a->vcf || !a->bam || !a->fasta || !a->cell_barcodes) return fail(VTX_E_INVAL, "vtxh_pack_files: null argument");
    *out = nullptr;
    const std::string bam_tag = a->bam_tag ? a->bam_tag : "CB";
    if (bam_tag.size() != 2) return fail(VTX_E_INVAL, "--bam-tag must be two characters");
    bool valid[256] = {false};
    for (const char* c = a->valid_chars ? a->valid_chars : "ATGCatgc"; *c; ++c) valid[(unsigned char)*c] = true;
    const int threads = a->threads > 0 ? a->threads : 1;
    std::unique_ptr<vtxh_pack> P(new vtxh_pack());

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

    // ---- VCF records (:221-234) ----
    std::vector<VcfRec> vcf;
    {
        std::string data;
        std::string path = a->vcf;
        if (ends_with(path, ".bcf")) return fail(VTX_E_UNSUPPORTED, "BCF input is not supported; use text VCF (optionally .gz)");
        if (!read_gz(path, data)) return fail(VTX_E_INVAL, "error opening vcf file %s", path.c_str());
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

    // ---- BAM: header ----
    MappedFile bam_file;
    if (!bam_file.open(a->bam)) return fail(VTX_E_INVAL, "error opening bam file: %s", a->bam);
    if (ends_with(a->bam, ".cram")) return fail(VTX_E_UNSUPPORTED, "CRAM input is not supported");
    std::vector<BgzfBlock> blocks;
    if (!index_bgzf(bam_file, blocks)) return fail(VTX_E_INVAL, "%s is not a valid BGZF/BAM file", a->bam);

    // streaming inflater over chunks of blocks
    std::vector<unsigned char> buf;       // decompressed bytes not yet consumed
    size_t buf_pos = 0, next_block = 0;
    auto refill = [&](size_t need) -> bool {   // ensure buf has >= need bytes from buf_pos, if the file has them
        while (buf.size() - buf_pos < need && next_block < blocks.size()) {
            const size_t chunk = std::min(blocks.size() - next_block, (size_t)2048);
            if (buf_pos) { buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)buf_pos); buf_pos = 0; }
            std::vector<size_t> off(chunk + 1, 0);
            for (size_t k = 0; k < chunk; ++k) off[k + 1] = off[k] + blocks[next_block + k].isize;
            const size_t base = buf.size();
            buf.resize(base + off[chunk]);
            std::atomic<size_t> nextk{0};
            std::atomic<bool> ok{true};
            auto work = [&]() {
                for (size_t k; (k = nextk.fetch_add(1)) < chunk;)
                    if (!inflate_block(bam_file, blocks[next_block + k], buf.data() + base + off[k])) ok = false;
            };
            std::vector<std::thread> th;
            for (int t = 1; t < threads; ++t) th.emplace_back(work);
            work();
            for (auto& t : th) t.join();
            if (!ok) return false;
            next_block += chunk;
        }
        return buf.size() - buf_pos >= need;
    };
    if (!refill(12) || memcmp(buf.data() + buf_pos, "BAM\1", 4) != 0) return fail(VTX_E_INVAL, "%s: bad BAM magic", a->bam);
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

    // ---- validate_inputs (:545-594) + evaluate_rec pre-alignment part (:610-684) ----
    std::vector<LocusBuild> loci;
    std::vector<std::vector<Interval>> by_tid(bam_refs.size());
    std::vector<int64_t> max_span(bam_refs.size(), 1);
    std::string left, right;
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
    for (size_t i = 0; i < vcf.size(); ++i) {
        const VcfRec& v = vcf[i];
        if (v.alleles.size() > 2) { ++P->metrics.num_multiallelic_recs; continue; }          // :646-653
        const std::string alt = v.alleles.size() == 2 ? v.alleles[1] : std::string();          // :656-659
        const FaiEntry& fe = fa.seqs[fa.by_name[v.chrom]];
        const int64_t start = v.pos, end = v.pos + (int64_t)v.alleles[0].size();
        const int64_t pad = a->padding;
        LocusBuild L;
        L.row = (uint32_t)i; L.start = start; L.end = end;
        // construct_haplotypes :958-994
        const int64_t ls = start >= pad ? start - pad : 0;
        const int64_t re = std::min<int64_t>(end + pad, (int64_t)fe.len);
        fa.fetch_upper(fe, (uint64_t)ls, (uint64_t)std::min<int64_t>(start, (int64_t)fe.len), left);
        fa.fetch_upper(fe, (uint64_t)end, (uint64_t)re, right);
        L.alt_hap = left + alt + right;
        int64_t rs = (int64_t)((int32_t)start - (int32_t)pad);       // i32 casts, :944
        if (rs < 0) rs = 0;
        fa.fetch_upper(fe, (uint64_t)rs, (uint64_t)re, L.ref_hap);
        bool ok = true;
        for (unsigned char c : L.alt_hap) if (!valid[c]) { ok = false; break; }                 // :675-684
        if (!ok) { ++P->metrics.num_invalid_recs; continue; }
        const int32_t tid = tid_of[v.chrom];
        by_tid[(size_t)tid].push_back(Interval{start, end, (uint32_t)loci.size()});
        max_span[(size_t)tid] = std::max(max_span[(size_t)tid], end - start);
        loci.push_back(std::move(L));
    }
    for (auto& iv : by_tid)
        std::stable_sort(iv.begin(), iv.end(), [](const Interval& x, const Interval& y) { return x.start < y.start; });

    // ---- sweep the BAM once (fetch + filters of evaluate_alns, :822-895) ----
    std::string& reads = P->read_arena;
    std::string seq;
    std::vector<uint32_t> hits;
    while (true) {
        if (!refill(4)) break;
        const uint32_t bs = rd32(buf.data() + buf_pos);
        if (!refill(4 + (size_t)bs)) return fail(VTX_E_INVAL, "%s: truncated BAM record", a->bam);
        const unsigned char* r = buf.data() + buf_pos + 4;
        buf_pos += 4 + (size_t)bs;
        if (bs < 32) return fail(VTX_E_INVAL, "%s: malformed BAM record", a->bam);
        const int32_t tid = rdi32(r);
        if (tid < 0 || (size_t)tid >= by_tid.size() || by_tid[(size_t)tid].empty()) continue;
        const int64_t pos = rdi32(r + 4);
        const uint32_t l_rn = r[8], mapq = r[9];
        const uint32_t n_cig = r[12] | (r[13] << 8), flag = r[14] | (r[15] << 8);
        const uint32_t l_seq = rd32(r + 16);
        const unsigned char* cig = r + 32 + l_rn;
        const unsigned char* sq = cig + 4 * (size_t)n_cig;
        const unsigned char* aux = sq + (l_seq + 1) / 2 + l_seq;
        if (aux > r + bs) return fail(VTX_E_INVAL, "%s: malformed BAM record", a->bam);
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
        size_t hi = (size_t)(std::lower_bound(iv.begin(), iv.end(), endpos,
                                              [](const Interval& x, int64_t e) { return x.start < e; }) - iv.begin());
        for (size_t k = hi; k-- > 0;) {
            if (iv[k].start + max_span[(size_t)tid] <= pos) break;
            if (iv[k].end > pos) hits.push_back(iv[k].locus);
        }
        if (hits.empty()) continue;
        bool seq_ready = false, tags_ready = false;
        uint64_t seq_off = 0;
        vtx_raw_record rr{};
        for (uint32_t li : hits) {
            LocusBuild& L = loci[li];
            ++P->metrics.num_reads;                                                     // :831
            if (mapq < a->mapq) { ++P->metrics.num_low_mapq; continue; }                 // :833
            if (a->primary_only && (flag & (FLAG_SECONDARY | FLAG_SUPP))) { ++P->metrics.num_non_primary; continue; }   // :841
            if (a->no_duplicates && (flag & FLAG_DUP)) { ++P->metrics.num_duplicates; continue; }                      // :849
            if (!useful_alignment(cig, n_cig, pos, L.start, L.end)) { ++P->metrics.num_not_useful; continue; }          // :857
            const unsigned char* val; size_t vlen;
            if (raw) {
                // the tag bytes go to the device as they are; only a missing / non-Z barcode tag is decided here
                if (!tags_ready) {
                    tags_ready = true;
                    rr = vtx_raw_record{};
                    rr.bc_len = VTX_TAG_MISSING;
                    if (aux_string(aux, (size_t)(r + bs - aux), bam_tag.c_str(), &val, &vlen) && vlen < VTX_TAG_MISSING) {
                        rr.bc_off = (uint32_t)P->tag_arena.size(); rr.bc_len = (uint16_t)vlen;
                        P->tag_arena.append((const char*)val, vlen);
                        rr.umi_len = VTX_TAG_MISSING;
                        if (aux_string(aux, (size_t)(r + bs - aux), "UB", &val, &vlen) == 1 && vlen < VTX_TAG_MISSING) {
                            rr.umi_off = (uint32_t)P->tag_arena.size(); rr.umi_len = (uint16_t)vlen;
                            P->tag_arena.append((const char*)val, vlen);
                        }
                    }
                }
                if (rr.bc_len == VTX_TAG_MISSING) { ++P->metrics.num_not_cell_bc; continue; }
                if (!seq_ready) {
                    seq.resize(l_seq);
                    for (uint32_t k = 0; k < l_seq; ++k) seq[k] = kNt16[(sq[k >> 1] >> ((~k & 1) << 2)) & 15];
                    seq_ready = true;
                    seq_off = reads.size();
                    reads += seq;
                }
                rr.read_off = (uint32_t)seq_off; rr.read_len = l_seq;
                L.raw_recs.push_back(rr);
                continue;
            }
            uint32_t cell = 0;
            bool has_cell = false;
            if (aux_string(aux, (size_t)(r + bs - aux), bam_tag.c_str(), &val, &vlen)) {                                // :867
                auto it = bc_index.find(std::string((const char*)val, vlen));
                if (it != bc_index.end()) { has_cell = true; cell = it->second; }
            }
            if (!has_cell) { ++P->metrics.num_not_cell_bc; continue; }
            std::string umi;
            bool has_umi = aux_string(aux, (size_t)(r + bs - aux), "UB", &val, &vlen) == 1;                            // :879
            if (a->use_umi && !has_umi) { ++P->metrics.num_non_umi; continue; }
            if (a->use_umi) umi.assign((const char*)val, vlen); else umi.assign(1, '\1');                               // :890-894
            if (!seq_ready) {                                                               // rec.seq().as_bytes() :896
                seq.resize(l_seq);
                for (uint32_t k = 0; k < l_seq; ++k) seq[k] = kNt16[(sq[k >> 1] >> ((~k & 1) << 2)) & 15];
                seq_ready = true;
            }
            uint32_t uid = L.umi_ids.emplace(umi, (uint32_t)L.umi_ids.size()).first->second;
            L.recs.push_back(LocusBuild::Rec{cell, uid, reads.size(), l_seq});
            reads += seq;
        }
    }

    if (reads.size() > 0xffffffffull || P->tag_arena.size() > 0xffffffffull)
        return fail(VTX_E_UNSUPPORTED, "read arena above 4 GiB: split the VCF");
    if (raw) {
        for (auto& L : loci) {
            vtx_locus o{};
            o.row = L.row; o.rec_begin = (uint32_t)P->raw_records.size(); o.rec_count = (uint32_t)L.raw_recs.size();
            o.ref_off = (uint32_t)P->hap_arena.size(); o.ref_len = (uint32_t)L.ref_hap.size();
            P->hap_arena += L.ref_hap;
            o.alt_off = (uint32_t)P->hap_arena.size(); o.alt_len = (uint32_t)L.alt_hap.size();
            P->hap_arena += L.alt_hap;
            P->raw_records.insert(P->raw_records.end(), L.raw_recs.begin(), L.raw_recs.end());
            P->loci.push_back(o);
        }
        *out = P.release();
        return VTX_OK;
    }
    // ---- pack: stable sort by (cell, umi) (:932 + the per-cell UMI grouping) ----
    for (auto& L : loci) {
        std::stable_sort(L.recs.begin(), L.recs.end(), [](const LocusBuild::Rec& x, const LocusBuild::Rec& y) {
            return x.cell != y.cell ? x.cell < y.cell : x.umi < y.umi;
        });
        vtx_locus o{};
        o.row = L.row; o.rec_begin = (uint32_t)P->records.size(); o.rec_count = (uint32_t)L.recs.size();
        o.ref_off = (uint32_t)P->hap_arena.size(); o.ref_len = (uint32_t)L.ref_hap.size();
        P->hap_arena += L.ref_hap;
        o.alt_off = (uint32_t)P->hap_arena.size(); o.alt_len = (uint32_t)L.alt_hap.size();
        P->hap_arena += L.alt_hap;
        for (auto& rc : L.recs) {
            if (rc.read_off + rc.read_len > 0xffffffffull) return fail(VTX_E_UNSUPPORTED, "read arena above 4 GiB: split the VCF");
            P->records.push_back(vtx_record{(uint32_t)rc.read_off, rc.read_len, rc.cell, rc.umi});
        }
        P->loci.push_back(o);
    }
    *out = P.release();
    return VTX_OK;
}

}  // extern "C"
