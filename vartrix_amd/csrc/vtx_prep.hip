// vtx_prep.hip — device-side preparation of RAW batches (include/vtx.h: vtx_submit_raw).
//
// What the reference does per read on the CPU — barcode dictionary lookup (get_cell_barcode,
// src/main.rs:737-750 and :867-876), the UB-present test (:879-888), grouping of a cell's reads by UMI
// byte string (parse_scores, :1047-1057) and the sort by cell (:932) — done here for a whole batch:
//
//   resolve   one lane per raw record: open-addressing probe of the barcode table (hash + byte
//             verification), 64-bit hash of the UMI bytes, sort keys
//   sort      two stable LSD radix sorts (hipcub): by UMI hash, then by (locus, cell)
//   finalize  gather into vtx_record order, UMI group heads with BYTE verification of equal hashes
//             (a collision between two UMIs of one cell is detected, never silently merged: the caller
//             re-runs with another seed), per-locus counts, kernel-shape histogram
//
// All of it is HBM-bound integer/byte work: one pass over the records per step, coalesced except the
// tag-byte and table probes (L2-resident: the barcode table of 10^4..10^6 entries is a few MB).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "vtx_device.h"

namespace {

// rec_locus for the raw records: one wavefront per locus
__global__ __launch_bounds__(256) void prep_rec_locus_kernel(const vtx_locus* __restrict__ loci, uint32_t n_loci,
                                                             uint32_t* __restrict__ rec_locus) {
    const uint32_t l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= n_loci) return;
    const uint32_t b = loci[l].rec_begin, e = b + loci[l].rec_count;
    for (uint32_t r = b + (threadIdx.x & 63); r < e; r += 64) rec_locus[r] = l;
}

// counters: [0] not_cell_bc, [1] non_umi, [2] error flags, [3] cells (sum of DP cells), [4] collision, [5] max read len
// Both record kernels walk the records with a grid-stride loop and keep their counters per workgroup
// (registers -> LDS -> one global atomic per workgroup): a global atomic per wavefront on one address
// costs more than the rest of the kernel.
__global__ __launch_bounds__(256) void prep_resolve_kernel(
    const vtx_raw_record* __restrict__ raw, uint32_t n, const uint32_t* __restrict__ rec_locus,
    const uint8_t* __restrict__ tags, uint64_t tag_bytes, uint64_t read_bytes, uint32_t max_read_len_fmt,
    const uint32_t* __restrict__ bc_slots, uint32_t bc_mask, const uint64_t* __restrict__ bc_hash,
    const uint64_t* __restrict__ bc_off, const uint8_t* __restrict__ bc_bytes,
    int use_umi, uint64_t seed, uint64_t hash_mask, uint32_t cell_bits, uint32_t n_loci,
    uint64_t* __restrict__ key_lc, uint64_t* __restrict__ key_umi, uint32_t* __restrict__ idx,
    unsigned long long* __restrict__ counters) {
    __shared__ uint32_t s_cnt[3];
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // bit 31 of the length limit: the reads arrive two bases per byte (VTX_READS_NIBBLES) — every read must start at an even base
    const uint32_t max_read_len = max_read_len_fmt & 0x7fffffffu, odd_mask = max_read_len_fmt >> 31;
    uint32_t n_not_bc = 0, n_no_umi = 0, n_bad = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint64_t klc = (uint64_t)n_loci << cell_bits, ku = 0;
        const vtx_raw_record r = raw[i];
        const bool umi_missing = r.umi_len == VTX_TAG_MISSING;
        const bool bad = (uint64_t)r.bc_off + r.bc_len > tag_bytes || (uint64_t)r.read_off + r.read_len > read_bytes ||
                         r.read_len > max_read_len || (r.read_off & odd_mask) || (use_umi && !umi_missing && (uint64_t)r.umi_off + r.umi_len > tag_bytes);
        if (bad) ++n_bad;
        else {
            const uint8_t* b = tags + r.bc_off;
            const uint64_t h = vtx_hash_bytes(b, r.bc_len, 0);
            uint32_t cell = 0xffffffffu;
            for (uint32_t s = (uint32_t)h & bc_mask;; s = (s + 1) & bc_mask) {
                const uint32_t e = bc_slots[s];
                if (!e) break;
                const uint32_t j = e - 1;
                if (bc_hash[j] != h) continue;
                const uint64_t o = bc_off[j];
                if (bc_off[j + 1] - o != r.bc_len) continue;
                if (vtx_bytes_equal(bc_bytes + o, b, r.bc_len)) { cell = j; break; }
            }
            if (cell == 0xffffffffu) ++n_not_bc;                     // :870-876
            else if (use_umi && umi_missing) ++n_no_umi;             // :879-888
            else {
                klc = ((uint64_t)rec_locus[i] << cell_bits) | cell;
                if (use_umi) ku = vtx_hash_bytes(tags + r.umi_off, r.umi_len, seed) & hash_mask;
            }
        }
        key_lc[i] = klc; key_umi[i] = ku; idx[i] = i;
    }
    if (n_not_bc) atomicAdd(&s_cnt[0], n_not_bc);
    if (n_no_umi) atomicAdd(&s_cnt[1], n_no_umi);
    if (n_bad) atomicAdd(&s_cnt[2], n_bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_cnt[0]) atomicAdd(&counters[0], (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&counters[1], (unsigned long long)s_cnt[1]);
        if (s_cnt[2]) atomicOr(&counters[2], 1ull);
    }
}

__global__ __launch_bounds__(256) void prep_gather_u64_kernel(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx,
                                                              uint32_t n, uint64_t* __restrict__ dst) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) dst[j] = src[idx[j]];
}

__constant__ uint32_t c_shape_cap[16];
__constant__ uint32_t c_fast_limit[2];     // longest read / haplotype the fast kernels hold; beyond: shape n_shapes = the slow list

__global__ __launch_bounds__(256) void prep_finalize_kernel(
    uint32_t n_kept, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ key_lc_sorted,
    const uint64_t* __restrict__ key_umi, const vtx_raw_record* __restrict__ raw, const uint8_t* __restrict__ tags,
    const vtx_locus* __restrict__ loci, uint32_t cell_bits, int use_umi, uint32_t n_shapes,
    vtx_record* __restrict__ records, uint32_t* __restrict__ rec_locus, uint32_t* __restrict__ umi_head,
    uint8_t* __restrict__ shape, uint32_t* __restrict__ seq, uint32_t* __restrict__ locus_first,
    uint32_t* __restrict__ locus_end, uint32_t* __restrict__ shape_cnt, unsigned long long* __restrict__ counters) {
    __shared__ uint32_t s_shape[16];
    __shared__ unsigned long long s_cells;
    __shared__ uint32_t s_maxlen, s_collide;
    if (threadIdx.x < 16) s_shape[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_cells = 0; s_maxlen = 0; s_collide = 0; }
    __syncthreads();
    unsigned long long cells = 0;
    uint32_t max_rl = 0, max_all = 0;
    bool collide = false;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n_kept; j += gridDim.x * 256) {
        const uint32_t i = perm[j];
        const uint64_t k = key_lc_sorted[j];
        const uint32_t locus = (uint32_t)(k >> cell_bits), cell = (uint32_t)(k & ((1ull << cell_bits) - 1));
        const vtx_raw_record r = raw[i];
        const uint64_t kprev = j ? key_lc_sorted[j - 1] : ~0ull;
        bool head = kprev != k;
        if (use_umi && !head) {
            const uint32_t ip = perm[j - 1];
            if (key_umi[ip] != key_umi[i]) head = true;
            else {
                // equal hashes inside one (locus, cell): must be the same bytes, else this seed collides
                const vtx_raw_record q = raw[ip];
                if (q.umi_len != r.umi_len || !vtx_bytes_equal(tags + q.umi_off, tags + r.umi_off, r.umi_len)) collide = true;
            }
        }
        records[j] = vtx_record{r.read_off, r.read_len, cell, 0};
        rec_locus[j] = locus;
        umi_head[j] = head ? 1u : 0u;
        seq[j] = j;
        // records are sorted by locus: the run boundaries give every locus its record range without atomics
        if (j == 0 || (uint32_t)(kprev >> cell_bits) != locus) locus_first[locus] = j;
        if (j + 1 == n_kept || (uint32_t)(key_lc_sorted[j + 1] >> cell_bits) != locus) locus_end[locus] = j + 1;
        const uint32_t rl = r.read_len;
        const bool slow = rl > c_fast_limit[0] || max(loci[locus].ref_len, loci[locus].alt_len) > c_fast_limit[1];
        uint32_t my_shape = 0;
        while (my_shape + 1 < n_shapes && c_shape_cap[my_shape] < rl) ++my_shape;
        if (slow) my_shape = n_shapes;                // the slow list (sorted last)
        shape[j] = (uint8_t)my_shape;
        atomicAdd(&s_shape[my_shape], 1u);            // LDS atomic: mostly one address, the LDS unit coalesces equal-address adds
        cells += (unsigned long long)rl * ((unsigned long long)loci[locus].ref_len + loci[locus].alt_len);
        if (!slow) max_rl = max(max_rl, rl);
        else max_all = max(max_all, rl);              // longest read among the SLOW records (long read or long haplotype)
    }
    for (int o = 32; o > 0; o >>= 1) { cells += __shfl_down(cells, o); max_rl = max(max_rl, __shfl_down(max_rl, o)); max_all = max(max_all, __shfl_down(max_all, o)); }
    if ((threadIdx.x & 63) == 0 && max_all) atomicMax(&counters[7], (unsigned long long)max_all);   // sizes slow_align_kernel's DP columns: a record is slow for a long read OR a long haplotype
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cells, cells); atomicMax(&s_maxlen, max_rl); }
    if (collide) s_collide = 1;
    __syncthreads();
    if (threadIdx.x <= n_shapes && s_shape[threadIdx.x]) atomicAdd(&shape_cnt[threadIdx.x], s_shape[threadIdx.x]);
    if (threadIdx.x == 0) {
        if (s_cells) atomicAdd(&counters[3], s_cells);
        if (s_maxlen) atomicMax(&counters[5], (unsigned long long)s_maxlen);
        if (s_collide) atomicOr(&counters[4], 1ull);
    }
}

// umi_id = dense group number (inclusive scan of the heads); locus record ranges from the count scan
__global__ __launch_bounds__(256) void prep_umi_id_kernel(vtx_record* __restrict__ records, const uint32_t* __restrict__ umi_scan, uint32_t n) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) records[j].umi_id = umi_scan[j] - 1;
}
__global__ __launch_bounds__(256) void prep_locus_counts_kernel(uint32_t* __restrict__ first_to_count, const uint32_t* __restrict__ end,
                                                                uint32_t n_loci) {
    const uint32_t l = blockIdx.x * 256 + threadIdx.x;
    if (l < n_loci) first_to_count[l] = end[l] - first_to_count[l];      // both 0 for a locus without records
}
__global__ __launch_bounds__(256) void prep_locus_ranges_kernel(vtx_locus* __restrict__ loci, const uint32_t* __restrict__ cnt,
                                                                const uint32_t* __restrict__ cnt_scan, uint32_t n_loci) {
    const uint32_t l = blockIdx.x * 256 + threadIdx.x;
    if (l < n_loci) { loci[l].rec_count = cnt[l]; loci[l].rec_begin = cnt_scan[l] - cnt[l]; }
}

// Per-record half of vtx_submit's validation, on the device (the host only checks the loci): bounds of the read,
// length limit, cell index range, order (cell_index, umi_id) inside the locus; plus what the host loop used to derive —
// kernel shape per record, shape histogram, DP-cell count, longest read.  counters[6] = min over offending records of
// (record << 3 | code): 1 read outside the arena, 2 read too long, 3 cell_index >= n_barcodes, 4 order.
__global__ __launch_bounds__(256) void prep_check_kernel(
    const vtx_record* __restrict__ records, uint32_t n, const uint32_t* __restrict__ rec_locus,
    const vtx_locus* __restrict__ loci, uint64_t read_bytes, uint32_t max_read_len_fmt, uint32_t n_barcodes, uint32_t n_shapes,
    uint8_t* __restrict__ shape, uint32_t* __restrict__ seq, uint32_t* __restrict__ shape_cnt,
    unsigned long long* __restrict__ counters) {
    const uint32_t max_read_len = max_read_len_fmt & 0x7fffffffu, odd_mask = max_read_len_fmt >> 31;   // (bit 31: VTX_READS_NIBBLES, as in prep_resolve_kernel)
    __shared__ uint32_t s_shape[16];
    __shared__ unsigned long long s_cells, s_bad;
    __shared__ uint32_t s_maxlen;
    if (threadIdx.x < 16) s_shape[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_cells = 0; s_maxlen = 0; s_bad = ~0ull; }
    __syncthreads();
    unsigned long long cells = 0, bad = ~0ull;
    uint32_t max_rl = 0, max_all = 0;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const vtx_record r = records[j];
        const uint32_t locus = rec_locus[j];
        uint32_t code = 0;
        if ((uint64_t)r.read_off + r.read_len > read_bytes) code = 1;
        else if (r.read_len > max_read_len) code = 2;
        else if (r.cell_index >= n_barcodes) code = 3;
        else if (r.read_off & odd_mask) code = 5;
        else if (j > 0 && rec_locus[j - 1] == locus) {
            const vtx_record q = records[j - 1];
            if (q.cell_index > r.cell_index || (q.cell_index == r.cell_index && q.umi_id > r.umi_id)) code = 4;
        }
        if (code) bad = min(bad, ((unsigned long long)j << 3) | code);
        const uint32_t rl = min(r.read_len, max_read_len);
        const bool slow = rl > c_fast_limit[0] || max(loci[locus].ref_len, loci[locus].alt_len) > c_fast_limit[1];
        uint32_t my_shape = 0;
        while (my_shape + 1 < n_shapes && c_shape_cap[my_shape] < rl) ++my_shape;
        if (slow) my_shape = n_shapes;                // the slow list (sorted last)
        shape[j] = (uint8_t)my_shape;
        seq[j] = j;
        atomicAdd(&s_shape[my_shape], 1u);
        cells += (unsigned long long)r.read_len * ((unsigned long long)loci[locus].ref_len + loci[locus].alt_len);
        if (!slow) max_rl = max(max_rl, r.read_len);
        else max_all = max(max_all, min(r.read_len, max_read_len));   // longest read among the SLOW records
    }
    for (int o = 32; o > 0; o >>= 1) {
        cells += __shfl_down(cells, o); max_rl = max(max_rl, __shfl_down(max_rl, o)); max_all = max(max_all, __shfl_down(max_all, o));
        bad = min(bad, (unsigned long long)__shfl_down(bad, o));
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cells, cells); atomicMax(&s_maxlen, max_rl); atomicMin(&s_bad, bad); }
    if ((threadIdx.x & 63) == 0 && max_all) atomicMax(&counters[7], (unsigned long long)max_all);   // sizes slow_align_kernel's DP columns: a record is slow for a long read OR a long haplotype
    __syncthreads();
    if (threadIdx.x <= n_shapes && s_shape[threadIdx.x]) atomicAdd(&shape_cnt[threadIdx.x], s_shape[threadIdx.x]);
    if (threadIdx.x == 0) {
        if (s_cells) atomicAdd(&counters[3], s_cells);
        if (s_maxlen) atomicMax(&counters[5], (unsigned long long)s_maxlen);
        if (s_bad != ~0ull) atomicMin(&counters[6], s_bad);
    }
}

// LUT-kernel eligibility of one work list: no `group`-record workgroup may span more than `cap` loci
__global__ __launch_bounds__(256) void prep_lut_check_kernel(const uint32_t* __restrict__ work, uint32_t count,
                                                             const uint32_t* __restrict__ rec_locus, uint32_t cap,
                                                             uint32_t group, uint32_t* __restrict__ flag) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if ((uint64_t)g * group >= count) return;
    const uint32_t a = work[g * group], b = work[min(count, g * group + group) - 1];
    if (rec_locus[b] - rec_locus[a] + 1 > cap) *flag = 1;
}

}  // namespace

extern "C" {

hipError_t vtxk_prep_set_shapes(const uint32_t* caps, uint32_t n, uint32_t fast_read_len, uint32_t fast_hap_len) {
    const uint32_t lim[2] = {fast_read_len, fast_hap_len};
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_fast_limit), lim, sizeof lim);
    if (e != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(c_shape_cap), caps, n * sizeof(uint32_t));
}

size_t vtxk_prep_sort_temp_bytes(uint32_t n) {
    size_t a = 0, b = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                       (uint32_t*)nullptr, (int)std::max(n, 1u));
    hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint8_t*)nullptr, (uint8_t*)nullptr, (const uint32_t*)nullptr,
                                       (uint32_t*)nullptr, (int)std::max(n, 1u));
    return std::max(a, b);
}

hipError_t vtxk_prep_sort_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                              uint32_t n, int end_bit, void* temp, size_t temp_bytes, hipStream_t s) {
    if (!n) return hipSuccess;
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, s);
}

size_t vtxk_sort_keys_u32_temp_bytes(uint32_t n) {
    size_t a = 0;
    hipcub::DeviceRadixSort::SortKeys(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)std::max(n, 1u));
    return a;
}

hipError_t vtxk_sort_keys_u32(const uint32_t* keys_in, uint32_t* keys_out, uint32_t n, void* temp, size_t temp_bytes, hipStream_t s) {
    if (!n) return hipSuccess;
    return hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys_in, keys_out, (int)n, 0, 32, s);
}

hipError_t vtxk_prep_sort_u8(const uint8_t* keys_in, uint8_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                             uint32_t n, void* temp, size_t temp_bytes, hipStream_t s) {
    if (!n) return hipSuccess;
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 4, s);
}

hipError_t vtxk_prep_rec_locus(const vtx_locus* loci, uint32_t n_loci, uint32_t* rec_locus, hipStream_t s) {
    if (!n_loci) return hipSuccess;
    hipLaunchKernelGGL(prep_rec_locus_kernel, dim3((n_loci + 3) / 4), dim3(256), 0, s, loci, n_loci, rec_locus);
    return hipGetLastError();
}

hipError_t vtxk_prep_resolve(const vtx_raw_record* raw, uint32_t n, const uint32_t* rec_locus, const uint8_t* tags,
                             uint64_t tag_bytes, uint64_t read_bytes, uint32_t max_read_len, const uint32_t* bc_slots,
                             uint32_t bc_mask, const uint64_t* bc_hash, const uint64_t* bc_off, const uint8_t* bc_bytes,
                             int use_umi, uint64_t seed, uint64_t hash_mask, uint32_t cell_bits, uint32_t n_loci,
                             uint64_t* key_lc, uint64_t* key_umi, uint32_t* idx, unsigned long long* counters, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(prep_resolve_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 256 * 16)), dim3(256), 0, s, raw, n, rec_locus, tags, tag_bytes,
                       read_bytes, max_read_len, bc_slots, bc_mask, bc_hash, bc_off, bc_bytes, use_umi, seed, hash_mask,
                       cell_bits, n_loci, key_lc, key_umi, idx, counters);
    return hipGetLastError();
}

hipError_t vtxk_prep_gather_u64(const uint64_t* src, const uint32_t* idx, uint32_t n, uint64_t* dst, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(prep_gather_u64_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, idx, n, dst);
    return hipGetLastError();
}

hipError_t vtxk_prep_finalize(uint32_t n_kept, const uint32_t* perm, const uint64_t* key_lc_sorted, const uint64_t* key_umi,
                              const vtx_raw_record* raw, const uint8_t* tags, const vtx_locus* loci, uint32_t cell_bits,
                              int use_umi, uint32_t n_shapes, vtx_record* records, uint32_t* rec_locus, uint32_t* umi_head,
                              uint8_t* shape, uint32_t* seq, uint32_t* locus_first, uint32_t* locus_end, uint32_t* shape_cnt,
                              unsigned long long* counters, hipStream_t s) {
    if (!n_kept) return hipSuccess;
    hipLaunchKernelGGL(prep_finalize_kernel, dim3(std::min<uint32_t>((n_kept + 255) / 256, 256 * 16)), dim3(256), 0, s, n_kept, perm, key_lc_sorted,
                       key_umi, raw, tags, loci, cell_bits, use_umi, n_shapes, records, rec_locus, umi_head, shape, seq,
                       locus_first, locus_end, shape_cnt, counters);
    return hipGetLastError();
}

hipError_t vtxk_prep_umi_ids(vtx_record* records, const uint32_t* umi_scan, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(prep_umi_id_kernel, dim3((n + 255) / 256), dim3(256), 0, s, records, umi_scan, n);
    return hipGetLastError();
}

hipError_t vtxk_prep_locus_counts(uint32_t* first_to_count, const uint32_t* end, uint32_t n_loci, hipStream_t s) {
    if (!n_loci) return hipSuccess;
    hipLaunchKernelGGL(prep_locus_counts_kernel, dim3((n_loci + 255) / 256), dim3(256), 0, s, first_to_count, end, n_loci);
    return hipGetLastError();
}

hipError_t vtxk_prep_locus_ranges(vtx_locus* loci, const uint32_t* cnt, const uint32_t* cnt_scan, uint32_t n_loci, hipStream_t s) {
    if (!n_loci) return hipSuccess;
    hipLaunchKernelGGL(prep_locus_ranges_kernel, dim3((n_loci + 255) / 256), dim3(256), 0, s, loci, cnt, cnt_scan, n_loci);
    return hipGetLastError();
}

hipError_t vtxk_prep_lut_check(const uint32_t* work, uint32_t count, const uint32_t* rec_locus, uint32_t cap, uint32_t group,
                               uint32_t* flag, hipStream_t s) {
    if (!count) return hipSuccess;
    const uint32_t groups = (count + group - 1) / group;
    hipLaunchKernelGGL(prep_lut_check_kernel, dim3((groups + 255) / 256), dim3(256), 0, s, work, count, rec_locus, cap, group, flag);
    return hipGetLastError();
}

hipError_t vtxk_prep_check(const vtx_record* records, uint32_t n, const uint32_t* rec_locus, const vtx_locus* loci,
                           uint64_t read_bytes, uint32_t max_read_len, uint32_t n_barcodes, uint32_t n_shapes, uint8_t* shape,
                           uint32_t* seq, uint32_t* shape_cnt, unsigned long long* counters, hipStream_t s) {
    if (!n) return hipSuccess;
    const uint32_t blocks = std::min<uint32_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(prep_check_kernel, dim3(blocks), dim3(256), 0, s, records, n, rec_locus, loci, read_bytes, max_read_len,
                       n_barcodes, n_shapes, shape, seq, shape_cnt, counters);
    return hipGetLastError();
}

}  // extern "C"
