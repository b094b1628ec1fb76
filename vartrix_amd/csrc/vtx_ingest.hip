// vtx_ingest.hip — the BAM ingest on the device (round 6): BGZF inflate, record split, and the fetch + filter half of evaluate_alns.
//
// What it replaces (reference 10XGenomics/vartrix v1.1.22): `bam.fetch(tid, start, end)` + `bam.records()` per locus
// (src/main.rs:822-830; rust-htslib -> htslib bgzf_read -> zlib below them), the read filters in their order with the Metrics
// counters (:831-864), useful_alignment (:790-806), get_cell_barcode / get_umi as tag BYTES (:737-757, :867-888 — the in-list test,
// the UB test, the UMI grouping and the sort by cell follow in vtx_prep.hip, as for vtx_submit_raw) and rec.seq() (:896).
// In rounds 1-5 this was host code (host/vtx_host.cpp: sixteen threads, 1.6 s of a 2.4 s run at config-3 scale, the device 1 % of it).
//
// Pipeline (all on the context's stream; HBM-bound byte work, no MFMA, no host round trip between the kernels):
//   bgzf_inflate_kernel   one LANE per BGZF block (vtx_inflate_core.h: a flat state machine, Huffman codes in registers, the symbol
//                         lists in LDS), every block of the file in flight at once; 0.9 GB -> 3 GB
//   bam_chain_kernel x2   record boundaries: block_size chains are serial, but the .bai's linear index names a record start every
//                         16 kb of genome — one lane per seed walks to the next seed (count, scan, fill)
//   bam_scan_kernel       one lane per BAM record: fixed fields, end position from the CIGAR, the loci it overlaps (binary search in
//                         the contig's sorted intervals), the filters per (read, locus) pair in the reference's order with its
//                         counters, the barcode / UB tag bytes; per record: surviving pairs, read and tag bytes to keep
//   scans                 pair offsets, read-arena and tag-arena offsets (BAM order: the layout the host packer produces)
//   bam_emit_kernel       raw records (vtx_raw_record + locus) per surviving pair, packed bases and tag bytes copied once per read
// and then vtx_prep.hip's barcode lookup / UMI grouping / sort, exactly as after vtx_submit_raw.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "vtx_device.h"
#include "vtx_ingest.h"
#include "vtx_inflate_core.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// BGZF inflate: one lane per block.  LDS: 444 bytes per lane, lane-interleaved (28.4 KB per wavefront: five per CU = 81 920 lanes
// resident on 256 CUs; the first cut kept 16-bit symbols: 47.6 KB, three per CU).
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const vtxg_block* __restrict__ blocks,
                                                          uint32_t n_blocks, uint8_t* __restrict__ out, uint32_t* __restrict__ err,
                                                          uint32_t* __restrict__ status, uint32_t b_base) {
    __shared__ uint8_t s_bytes[vtxi::BYTES * 64];
    __shared__ uint32_t s_hi[vtxi::HI_WORDS * 64];
    __shared__ uint16_t s_cnt[vtxi::CNT_WORDS * 64];
    const vtxi::Scratch sc{s_bytes + threadIdx.x, s_hi + threadIdx.x, s_cnt + threadIdx.x, 64};
    for (uint32_t b = blockIdx.x * 64 + threadIdx.x; b < n_blocks; b += gridDim.x * 64) {
        const vtxg_block B = blocks[b];
        uint32_t st = vtxi::ST_OK;
        if (B.isize) st = vtxi::inflate_block(comp + B.coff, B.clen, out + B.uoff, B.isize, sc, nullptr);
        if (st != vtxi::ST_OK) { atomicMin(&err[1], b_base + b); atomicOr(&err[0], 1u << st); }     // (b_base: the launch covers blocks [b_base, b_base + n_blocks) of the ingest)
        if (status) status[b] = st;                         // (vtx_debug_inflate: the verdict per block)
    }
}

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// ---------------------------------------------------------------------------------------------------------------------------
// Record boundaries.  Seed i (a record start the .bai names) .. seed i + 1: one lane hops from block_size to block_size.
// off == nullptr: count; else: write the offsets.  A chain that does not land on the next seed means the index and the file
// disagree (or a record is malformed): the host packer takes over.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bam_chain_kernel(const uint8_t* __restrict__ data, uint64_t total, const uint64_t* __restrict__ seeds,
                                                       uint32_t n_seeds, uint64_t end_upos, uint32_t* __restrict__ cnt,
                                                       const uint32_t* __restrict__ off, uint64_t* __restrict__ rec_upos,
                                                       uint32_t* __restrict__ err) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_seeds) return;
    uint64_t p = seeds[i];
    const uint64_t stop = i + 1 < n_seeds ? seeds[i + 1] : end_upos;
    uint32_t k = 0;
    const uint32_t base = off ? (i ? off[i - 1] : 0u) : 0u;      // (off: INCLUSIVE scan of the counts)
    bool bad = false;
    while (p < stop) {
        if (p + 36 > total) { bad = true; break; }
        const uint32_t bs = ld32(data + p);
        if (bs < 32 || p + 4 + (uint64_t)bs > total) { bad = true; break; }
        if (off) rec_upos[base + k] = p;
        ++k;
        p += 4 + (uint64_t)bs;
    }
    if (bad || p != stop) atomicOr(&err[0], VTXG_ERR_CHAIN);
    if (!off) cnt[i] = k;
}

// ---------------------------------------------------------------------------------------------------------------------------
// rust-htslib 0.36 CigarStringView::read_pos(ref_pos, include_softclips = false, include_dels = true) as called from
// useful_alignment (src/main.rs:796): 1 = Some, 0 = None, -1 = Err.  Same restatement as host/vtx_host.cpp: cigar_read_pos and
// oracle/vtx_oracle.c: vtxo_cigar_read_pos.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ int cigar_read_pos(const uint8_t* cig, uint32_t n_ops, int64_t pos, int64_t ref_pos) {
    int64_t rpos = pos;
    uint32_t j = 0;
    for (uint32_t i = 0; i < n_ops; ++i) {
        const uint32_t op = ld32(cig + 4 * i) & 15u;
        if (op == 0 || op == 7 || op == 8 || op == 1) { j = i; break; }
        if (op == 4) { j = i; break; }
        if (op == 2 || op == 3) return -1;
        if (op == 5 && i > 0 && i + 1 < n_ops) return -1;
        if ((op == 6 || op == 5) && i + 1 == n_ops) return 0;
    }
    while (rpos <= ref_pos && j < n_ops) {
        const uint32_t c = ld32(cig + 4 * j), op = c & 15u;
        const int64_t l = c >> 4;
        const bool contains = rpos <= ref_pos && rpos + l > ref_pos;
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
// useful_alignment, src/main.rs:790-806 (probes start..=end, inclusive; an invalid CIGAR drops the read, :799-802)
__device__ bool useful_alignment(const uint8_t* cig, uint32_t n_ops, int64_t pos, int64_t start, int64_t end) {
    for (int64_t i = start; i <= end; ++i) {
        const int r = cigar_read_pos(cig, n_ops, pos, i);
        if (r == 1) return true;
        if (r < 0) return false;
    }
    return false;
}
// rec.aux(tag) matched against Aux::String (src/main.rs:742-748, :753-755): type 'Z' only.  Returns the value's offset from aux
// (0xffffffff: no such Z tag) and *len.
__device__ uint32_t aux_string(const uint8_t* aux, uint32_t n, uint32_t tag2, uint32_t* len) {
    uint32_t o = 0;
    while (o + 3 <= n) {
        const uint32_t t2 = ld16(aux + o);
        const uint32_t ty = aux[o + 2];
        o += 3;
        uint32_t size;
        bool is_z = false;
        switch (ty) {
        case 'A': case 'c': case 'C': size = 1; break;
        case 's': case 'S': size = 2; break;
        case 'i': case 'I': case 'f': size = 4; break;
        case 'd': size = 8; break;
        case 'Z': case 'H': {
            uint32_t e = o;
            while (e < n && aux[e]) ++e;
            size = e - o + 1;
            is_z = ty == 'Z';
            break;
        }
        case 'B': {
            if (o + 5 > n) return 0xffffffffu;
            const uint32_t sub = aux[o];
            const uint32_t cnt = ld32(aux + o + 1);
            const uint32_t es = (sub == 'c' || sub == 'C') ? 1u : (sub == 's' || sub == 'S') ? 2u : 4u;
            const uint64_t sz = 5ull + (uint64_t)cnt * es;
            if (sz > n) return 0xffffffffu;
            size = (uint32_t)sz;
            break;
        }
        default: return 0xffffffffu;
        }
        if (t2 == tag2) {
            if (!is_z) return 0xffffffffu;
            *len = size - 1;
            return o;
        }
        o += size;
    }
    return 0xffffffffu;
}

struct RecView {
    const uint8_t* r;          // behind block_size
    uint32_t bs;
    int32_t tid;
    int64_t pos, endpos;
    uint32_t mapq, flag, n_cig, l_seq;
    const uint8_t* cig;
    const uint8_t* sq;
    const uint8_t* aux;
    bool malformed;
};
__device__ __forceinline__ RecView view_record(const uint8_t* data, uint64_t p) {
    RecView v;
    v.bs = ld32(data + p);
    v.r = data + p + 4;
    v.tid = (int32_t)ld32(v.r);
    v.pos = (int32_t)ld32(v.r + 4);
    const uint32_t w2 = ld32(v.r + 8), w3 = ld32(v.r + 12);
    const uint32_t l_rn = w2 & 0xffu;
    v.mapq = (w2 >> 8) & 0xffu;
    v.n_cig = w3 & 0xffffu;
    v.flag = w3 >> 16;
    v.l_seq = ld32(v.r + 16);
    v.cig = v.r + 32 + l_rn;
    v.sq = v.cig + 4 * (size_t)v.n_cig;
    const uint64_t aux_off = 32ull + l_rn + 4ull * v.n_cig + ((uint64_t)v.l_seq + 1) / 2 + v.l_seq;
    v.malformed = aux_off > v.bs;
    v.aux = v.r + (v.malformed ? v.bs : aux_off);
    int64_t rlen = 0;                                     // bam_endpos: unmapped or no reference-consuming op => pos + 1
    if (!v.malformed && !(v.flag & 0x4u))
        for (uint32_t k = 0; k < v.n_cig; ++k) {
            const uint32_t c = ld32(v.cig + 4 * k), op = c & 15u;
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4;
        }
    v.endpos = v.pos + (rlen > 0 ? rlen : 1);
    return v;
}

// hi = first interval of the contig with start >= endpos (the loci that can overlap lie below it)
__device__ __forceinline__ uint32_t first_not_below(const int32_t* __restrict__ iv_start, uint32_t lo, uint32_t hi, int64_t endpos) {
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((int64_t)iv_start[mid] < endpos) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One lane per BAM record.  EMIT = false: counts (pairs that survive, metrics, the bytes to keep); EMIT = true: the raw records.
template <bool EMIT>
__global__ __launch_bounds__(256) void bam_scan_kernel(const uint8_t* __restrict__ data, const uint64_t* __restrict__ rec_upos,
                                                       uint32_t n_rec, vtxg_filter f, const int32_t* __restrict__ iv_start,
                                                       const int32_t* __restrict__ iv_end, const uint32_t* __restrict__ iv_locus,
                                                       const uint32_t* __restrict__ tid_begin, const int32_t* __restrict__ tid_span,
                                                       uint32_t* __restrict__ n_hit, uint32_t* __restrict__ read_sz,
                                                       uint32_t* __restrict__ tag_sz, vtxg_recinfo* __restrict__ info,
                                                       const uint32_t* __restrict__ hit_scan, const uint32_t* __restrict__ read_scan,
                                                       const uint32_t* __restrict__ tag_scan, vtx_raw_record* __restrict__ raw,
                                                       uint32_t* __restrict__ raw_locus, uint8_t* __restrict__ tags,
                                                       uint8_t* __restrict__ reads_packed, unsigned long long* __restrict__ counters,
                                                       uint32_t* __restrict__ err) {
    __shared__ unsigned long long s_cnt[VTXG_N_COUNTERS];
    if (!EMIT) {
        if (threadIdx.x < VTXG_N_COUNTERS) s_cnt[threadIdx.x] = 0;
        __syncthreads();
    }
    for (uint32_t rix = blockIdx.x * 256 + threadIdx.x; rix < n_rec; rix += gridDim.x * 256) {
        if (EMIT && n_hit[rix] == 0) continue;
        const uint64_t p = rec_upos[rix];
        const RecView v = view_record(data, p);
        uint32_t hits = 0, m_reads = 0, m_mapq = 0, m_prim = 0, m_dup = 0, m_useful = 0, m_nobc = 0;
        uint32_t bc_rel = 0, umi_rel = 0, bc_len = VTX_TAG_MISSING, umi_len = VTX_TAG_MISSING;
        bool tags_ready = false;
        uint32_t h_base = 0, roff = 0, toff = 0;
        if (EMIT) {
            h_base = rix ? hit_scan[rix - 1] : 0u;
            roff = rix ? read_scan[rix - 1] : 0u;
            toff = rix ? tag_scan[rix - 1] : 0u;
            const vtxg_recinfo I = info[rix];
            bc_rel = I.bc_rel; umi_rel = I.umi_rel; bc_len = I.lens & 0xffffu; umi_len = I.lens >> 16;
        }
        if (v.malformed) { if (!EMIT) atomicOr(&err[0], VTXG_ERR_RECORD); }
        else if (v.tid >= 0 && (uint32_t)v.tid < f.n_ref) {
            const uint32_t i0 = tid_begin[v.tid], i1 = tid_begin[v.tid + 1];
            if (i1 > i0) {
                const int64_t span = tid_span[v.tid];
                uint32_t k = first_not_below(iv_start, i0, i1, v.endpos);
                // loci of this contig with start < endpos && end > pos (htslib's overlap on [start, end), src/main.rs:822-826)
                while (k-- > i0) {
                    if ((int64_t)iv_start[k] + span <= v.pos) break;
                    if ((int64_t)iv_end[k] <= v.pos) continue;
                    ++m_reads;                                                              // :831
                    if (v.mapq < f.min_mapq) { ++m_mapq; continue; }                        // :833
                    if (f.primary_only && (v.flag & (0x100u | 0x800u))) { ++m_prim; continue; }   // :841
                    if (f.no_duplicates && (v.flag & 0x400u)) { ++m_dup; continue; }        // :849
                    if (!useful_alignment(v.cig, v.n_cig, v.pos, iv_start[k], iv_end[k])) { ++m_useful; continue; }   // :857
                    if (!EMIT && !tags_ready) {
                        tags_ready = true;
                        const uint32_t n_aux = (uint32_t)(v.r + v.bs - v.aux);
                        uint32_t len = 0;
                        uint32_t o = aux_string(v.aux, n_aux, f.bam_tag, &len);            // :867 (the in-list test: vtx_prep.hip)
                        if (o != 0xffffffffu && len < VTX_TAG_MISSING) {
                            bc_rel = (uint32_t)(v.aux - v.r) + o; bc_len = len;
                            o = aux_string(v.aux, n_aux, (uint32_t)'U' | ((uint32_t)'B' << 8), &len);   // :879 (the test itself: vtx_prep.hip)
                            if (o != 0xffffffffu && len < VTX_TAG_MISSING) { umi_rel = (uint32_t)(v.aux - v.r) + o; umi_len = len; }
                        }
                    }
                    if (bc_len == VTX_TAG_MISSING) { ++m_nobc; continue; }
                    if (EMIT) {
                        vtx_raw_record rr;
                        rr.read_off = roff; rr.read_len = v.l_seq;
                        rr.bc_off = toff; rr.umi_off = umi_len != VTX_TAG_MISSING ? toff + bc_len : 0u;
                        rr.bc_len = (uint16_t)bc_len; rr.umi_len = (uint16_t)umi_len;
                        raw[h_base + hits] = rr;
                        raw_locus[h_base + hits] = iv_locus[k];
                    }
                    ++hits;
                }
            }
        }
        if (!EMIT) {
            n_hit[rix] = hits;
            read_sz[rix] = hits ? (v.l_seq + 1u) & ~1u : 0u;           // bases; every read starts at an even one (two per byte)
            tag_sz[rix] = hits ? bc_len + (umi_len != VTX_TAG_MISSING ? umi_len : 0u) : 0u;
            if (hits) info[rix] = vtxg_recinfo{bc_rel, umi_rel, bc_len | (umi_len << 16)};
            if (hits && v.l_seq > 0x7fffffffu) atomicOr(&err[0], VTXG_ERR_RECORD);
            if (m_reads) atomicAdd(&s_cnt[0], (unsigned long long)m_reads);
            if (m_mapq) atomicAdd(&s_cnt[1], (unsigned long long)m_mapq);
            if (m_prim) atomicAdd(&s_cnt[2], (unsigned long long)m_prim);
            if (m_dup) atomicAdd(&s_cnt[3], (unsigned long long)m_dup);
            if (m_useful) atomicAdd(&s_cnt[4], (unsigned long long)m_useful);
            if (m_nobc) atomicAdd(&s_cnt[5], (unsigned long long)m_nobc);
            if (hits) { atomicAdd(&s_cnt[6], (unsigned long long)read_sz[rix]); atomicAdd(&s_cnt[7], (unsigned long long)tag_sz[rix]); atomicAdd(&s_cnt[8], (unsigned long long)hits); }
        } else if (hits) {
            // the read's packed bases and its tag bytes, once per read (its pairs share them): exact byte counts — the neighbours
            // belong to other lanes
            uint8_t* d = tags + toff;
            const uint8_t* s = v.r + bc_rel;
            for (uint32_t i = 0; i < bc_len; ++i) d[i] = s[i];
            if (umi_len != VTX_TAG_MISSING) { d += bc_len; s = v.r + umi_rel; for (uint32_t i = 0; i < umi_len; ++i) d[i] = s[i]; }
            const uint32_t nb = (v.l_seq + 1u) >> 1;
            d = reads_packed + (roff >> 1);
            s = v.sq;
            uint32_t i = 0;
            for (; i + 8 <= nb; i += 8) { const uint64_t w = ld64(s + i); __builtin_memcpy(d + i, &w, 8); }
            for (; i < nb; ++i) d[i] = s[i];
            // (an odd read's last low nibble travels as the BAM holds it, like the host packer's memcpy: it is never read)
        }
    }
    if (!EMIT) {
        __syncthreads();
        if (threadIdx.x < VTXG_N_COUNTERS && s_cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], s_cnt[threadIdx.x]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Matrix-Market text of resident triplets (sprs::io::write_matrix_market, src/main.rs:381-389): "row+1 col+1 value\n" per triplet,
// Rust `{}` of an f64 that holds a non-negative integer = its decimal digits (consensus 1 / 2 / 3, coverage counts).  Any other
// value (alt_frac's fractions, NaN) sets the flag: the host formatter (shortest round-trip digits, vtxh_write_mtx) takes over.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ndigits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u
         : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ __forceinline__ uint8_t* put_u32(uint8_t* p, uint32_t v, uint32_t nd) {      // nd = ndigits(v); returns the end
    for (uint32_t i = nd; i-- > 0;) { p[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
    return p + nd;
}
__global__ __launch_bounds__(256) void mtx_len_kernel(const uint32_t* __restrict__ row, const uint32_t* __restrict__ col,
                                                      const double* __restrict__ val, uint32_t n, uint32_t* __restrict__ len,
                                                      double* __restrict__ sum, uint32_t* __restrict__ flag) {
    __shared__ double s_sum[4];
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (k < n) {
        v = val[k];
        const bool integral = v >= 0.0 && v < 4294967296.0 && v == (double)(uint32_t)v;      // (false for NaN)
        if (!integral) { atomicOr(flag, 1u); v = 0.0; len[k] = 0; }
        else len[k] = ndigits(row[k] + 1u) + ndigits(col[k] + 1u) + ndigits((uint32_t)v) + 3u;
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { const double t = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]; if (t != 0.0) atomicAdd(sum, t); }
}
__global__ __launch_bounds__(256) void mtx_text_kernel(const uint32_t* __restrict__ row, const uint32_t* __restrict__ col,
                                                       const double* __restrict__ val, uint32_t n, const uint32_t* __restrict__ end,
                                                       uint8_t* __restrict__ text) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t r = row[k] + 1u, c = col[k] + 1u, v = (uint32_t)val[k];
    uint8_t* p = text + (k ? end[k - 1] : 0u);
    p = put_u32(p, r, ndigits(r)); *p++ = ' ';
    p = put_u32(p, c, ndigits(c)); *p++ = ' ';
    p = put_u32(p, v, ndigits(v)); *p = '\n';
}

}  // namespace

extern "C" {

hipError_t vtxg_inflate(const uint8_t* comp, const vtxg_block* blocks, uint32_t n_blocks, uint8_t* out, uint32_t* err, uint32_t* status, uint32_t b_base, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    const uint32_t wgs = std::min<uint32_t>((n_blocks + 63) / 64, 256u * 5u);
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(wgs), dim3(64), 0, s, comp, blocks, n_blocks, out, err, status, b_base);
    return hipGetLastError();
}

hipError_t vtxg_chain(const uint8_t* data, uint64_t total, const uint64_t* seeds, uint32_t n_seeds, uint64_t end_upos, uint32_t* cnt,
                      const uint32_t* off, uint64_t* rec_upos, uint32_t* err, hipStream_t s) {
    if (!n_seeds) return hipSuccess;
    hipLaunchKernelGGL(bam_chain_kernel, dim3((n_seeds + 63) / 64), dim3(64), 0, s, data, total, seeds, n_seeds, end_upos, cnt, off, rec_upos, err);
    return hipGetLastError();
}

hipError_t vtxg_scan(int emit, const uint8_t* data, const uint64_t* rec_upos, uint32_t n_rec, vtxg_filter f, const int32_t* iv_start,
                     const int32_t* iv_end, const uint32_t* iv_locus, const uint32_t* tid_begin, const int32_t* tid_span,
                     uint32_t* n_hit, uint32_t* read_sz, uint32_t* tag_sz, vtxg_recinfo* info, const uint32_t* hit_scan,
                     const uint32_t* read_scan, const uint32_t* tag_scan, vtx_raw_record* raw, uint32_t* raw_locus, uint8_t* tags,
                     uint8_t* reads_packed, unsigned long long* counters, uint32_t* err, hipStream_t s) {
    if (!n_rec) return hipSuccess;
    const uint32_t wgs = std::min<uint32_t>((n_rec + 255) / 256, 256u * 32u);
    if (emit)
        hipLaunchKernelGGL(bam_scan_kernel<true>, dim3(wgs), dim3(256), 0, s, data, rec_upos, n_rec, f, iv_start, iv_end, iv_locus, tid_begin,
                           tid_span, n_hit, read_sz, tag_sz, info, hit_scan, read_scan, tag_scan, raw, raw_locus, tags, reads_packed, counters, err);
    else
        hipLaunchKernelGGL(bam_scan_kernel<false>, dim3(wgs), dim3(256), 0, s, data, rec_upos, n_rec, f, iv_start, iv_end, iv_locus, tid_begin,
                           tid_span, n_hit, read_sz, tag_sz, info, hit_scan, read_scan, tag_scan, raw, raw_locus, tags, reads_packed, counters, err);
    return hipGetLastError();
}

hipError_t vtxg_mtx_len(const uint32_t* row, const uint32_t* col, const double* val, uint32_t n, uint32_t* len, double* sum, uint32_t* flag, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(mtx_len_kernel, dim3((n + 255) / 256), dim3(256), 0, s, row, col, val, n, len, sum, flag);
    return hipGetLastError();
}
hipError_t vtxg_mtx_text(const uint32_t* row, const uint32_t* col, const double* val, uint32_t n, const uint32_t* end, uint8_t* text, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(mtx_text_kernel, dim3((n + 255) / 256), dim3(256), 0, s, row, col, val, n, end, text);
    return hipGetLastError();
}

}  // extern "C"
