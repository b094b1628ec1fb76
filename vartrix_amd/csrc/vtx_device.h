// vtx_device.h — internal declarations shared by the kernels and the C-ABI layer.
#ifndef VTX_DEVICE_H
#define VTX_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vtx.h"

// Experiment / test hooks.  The PRODUCTION library (libvtx.so) reads one environment variable, VTX_DEBUG (stderr diagnostics; changes
// no result).  Every other knob — stage on/off switches, ablations ("results wrong by design"), buffer caps that force the overflow
// paths, kernel variants, the socket transport that stands in for RCCL in tests — exists only in libvtx_dev.so, the same sources
// compiled with -DVTX_DEVTOOLS (make dev; loaded by tests/ and tools/ through VTX_LIB_VARIANT=dev).  In the production build
// VTX_DEV_ENV("...") is the constant nullptr: the branches fold away and the strings are not in the binary.
#ifdef VTX_DEVTOOLS
#include <cstdlib>
#define VTX_DEV_ENV(name) getenv(name)
#define VTX_DEVTOOLS_ON 1
#else
#define VTX_DEV_ENV(name) ((const char*)nullptr)
#define VTX_DEVTOOLS_ON 0
#endif
// a kernel's profiling-ablation selector: the constant 0 in the production build (the early exits compile out of the hot bodies)
#define VTX_ABLATE(x) (VTX_DEVTOOLS_ON ? (uint32_t)(x) : 0u)

// 64-bit hash of a tag byte string (FNV-1a style over little-endian 8-byte words + fmix64), identical on the
// host (barcode table build) and the device.  The words are read with memcpy: gfx950 global memory takes
// unaligned dwordx2 loads, so a lane hashes an 18-byte barcode with 3 loads instead of 18.
static __host__ __device__ inline uint64_t vtx_load_le(const uint8_t* p, uint32_t n) {   // n <= 8 bytes, zero-extended
    uint64_t w = 0;
    if (n == 8) { __builtin_memcpy(&w, p, 8); return w; }
    for (uint32_t i = 0; i < n; ++i) w |= (uint64_t)p[i] << (8 * i);
    return w;
}
static __host__ __device__ inline uint64_t vtx_hash_bytes(const uint8_t* p, uint32_t n, uint64_t seed) {
    uint64_t h = (0xcbf29ce484222325ull ^ seed) + n;
    for (uint32_t i = 0; i < n; i += 8) { h ^= vtx_load_le(p + i, n - i < 8 ? n - i : 8); h *= 0x100000001b3ull; h ^= h >> 29; }
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return h;
}
static __host__ __device__ inline bool vtx_bytes_equal(const uint8_t* a, const uint8_t* b, uint32_t n) {
    bool eq = true;
    for (uint32_t i = 0; i < n; i += 8) { const uint32_t m = n - i < 8 ? n - i : 8; eq &= vtx_load_le(a + i, m) == vtx_load_le(b + i, m); }
    return eq;
}

extern "C" {
hipError_t vtxk_launch_sw_full(int R, int GL, uint32_t n_work, const uint32_t* work, const vtx_record* records,
                               const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                               const uint8_t* hap_arena, int32_t* ref_score, int32_t* alt_score,
                               uint32_t max_hap_len, hipStream_t stream);
hipError_t vtxk_launch_sw_full_lut(int R, uint32_t n_work, const uint32_t* work, const vtx_record* records,
                                   const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                   const uint8_t* hap_arena, int32_t* ref_score, int32_t* alt_score, uint32_t max_hap_len,
                                   uint32_t loci_cap, uint32_t* redo, uint32_t* redo_count, hipStream_t stream);
hipError_t vtxk_launch_sw_full_duo(int R, uint32_t n_work, const uint32_t* work, const vtx_record* records,
                                   const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                   const uint8_t* hap_arena, int32_t* ref_score, int32_t* alt_score, uint32_t max_hap_len,
                                   uint32_t loci_cap, uint32_t* redo, uint32_t* redo_count, uint32_t pair_cols,
                                   hipStream_t stream);
hipError_t vtxk_launch_sw_banded(int R, int GL, uint32_t n_hard, const uint32_t* hard, const vtx_record* records,
                                 const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                 const uint8_t* hap_arena, const uint16_t* band, uint32_t band_stride, int32_t* ref_score,
                                 int32_t* alt_score, uint32_t max_hap_len, hipStream_t stream);
size_t vtxk_band_ws_stride(uint32_t m_cap, uint32_t max_hap);
size_t vtxk_band_lds_stride(uint32_t m_cap, uint32_t max_hap, uint32_t max_read);
hipError_t vtxk_launch_band(const uint32_t* tasks, uint32_t n_tasks, uint32_t task_base, const vtx_record* records,
                            const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                            const uint8_t* hap_arena, uint8_t* workspace, uint64_t ws_stride, uint32_t m_cap,
                            uint32_t max_hap, int32_t* ref_score, int32_t* alt_score, uint16_t* band, uint32_t band_stride,
                            uint32_t* hard_list, uint32_t* overflow_list, uint32_t* counters, int in_lds, uint32_t max_read,
                            hipStream_t s);
hipError_t vtxk_launch_band_run(uint32_t n_tasks, uint32_t task_base, const vtx_record* records,
                                const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena,
                                const uint8_t* hap_arena, uint32_t max_hap, uint32_t min_hap, int32_t* ref_score,
                                int32_t* alt_score, uint32_t* logbuf, uint16_t* band, uint32_t band_stride,
                                uint32_t* hard_list, uint32_t* overflow_list, uint32_t* pending_list, uint32_t* pend_buf,
                                uint32_t hard_cap, uint32_t pend_cap, uint32_t* counters, uint32_t tasks_per_locus,
                                uint32_t gt_l0, uint32_t n_loci, uint8_t* gtables, size_t gtables_bytes, const uint32_t* task_list,
                                int long_lists, hipStream_t s);
hipError_t vtxk_launch_band_diag(uint32_t n_tasks, uint32_t task_base, const vtx_record* records, const uint32_t* rec_locus,
                                 const vtx_locus* loci, const uint8_t* read_arena, const uint8_t* hap_arena, uint32_t max_hap, uint32_t min_hap,
                                 int32_t* ref_score, int32_t* alt_score, uint32_t* fail_list, uint32_t* refine_rec,
                                 uint32_t refine_cap, uint32_t* counters, uint32_t tasks_per_locus, uint32_t gt_l0, uint32_t n_loci,
                                 uint8_t* gtables, size_t gtables_bytes, int stats, uint32_t* tight_list, uint32_t* tight_pack, uint8_t* stage,
                                 uint32_t* dense_list, uint32_t dense_mask, uint32_t max_read, hipStream_t s);
uint32_t vtxk_band_refine_words(void);
hipError_t vtxk_launch_band_corridor(const uint32_t* recs, uint32_t n_recs, const vtx_record* records, const uint32_t* rec_locus,
                                     const vtx_locus* loci, const uint8_t* read_arena, const uint8_t* hap_arena, int32_t* ref_score,
                                     int32_t* alt_score, uint32_t* counters, int stats, uint32_t* tight_list, uint32_t* tight_pack,
                                     uint8_t* stage, const uint32_t* n_dev, hipStream_t s);
hipError_t vtxk_launch_band_refine(const uint32_t* recs, uint32_t n_recs, const vtx_record* records, const uint32_t* rec_locus,
                                   const vtx_locus* loci, const uint8_t* read_arena, uint32_t max_hap, int32_t* ref_score,
                                   int32_t* alt_score, uint32_t* fail_list, uint32_t* counters, uint32_t tasks_per_locus,
                                   uint32_t gt_l0, const uint8_t* gtables, int stats, uint32_t* tight_list, uint32_t* tight_pack,
                                   uint8_t* stage, const uint32_t* n_dev, hipStream_t s);
// the second stage over the tasks band_diag_kernel left with too many off-diagonal matches (vtx_band.hip: band_diag2_kernel, and
// band_stream_kernel for what exceeds its list when stream_list != nullptr)
hipError_t vtxk_launch_band_diag2(const uint32_t* tasks, uint32_t n_tasks, const vtx_record* records, const uint32_t* rec_locus,
                                  const vtx_locus* loci, const uint8_t* read_arena, uint32_t max_hap, int32_t* ref_score,
                                  int32_t* alt_score, uint32_t tasks_per_locus, uint32_t gt_l0, const uint8_t* gtables,
                                  uint32_t* sweep_list, uint32_t* tight_list, uint32_t* tight_pack, uint32_t* counters,
                                  uint32_t* stream_list, uint32_t* stream_diag, uint32_t* stream_cnt, uint8_t* stage, hipStream_t s);
// ---- the band for any task (vtx_sweep.hip), the masked DP over a device-counted list, the full-matrix check ----
hipError_t vtxk_launch_band_sweep(const uint32_t* tasks, uint32_t n_tasks, const uint32_t* n_dev, const vtx_record* records,
                                  const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena, const uint8_t* hap_arena,
                                  uint16_t* band, uint32_t band_stride, uint32_t* hard_list, uint32_t* overflow_list, uint32_t* counters,
                                  uint32_t* stat_counters, uint8_t* stage, uint32_t* dbg, uint32_t* glog, hipStream_t s);
size_t vtxk_band_sweep_log_bytes(void);            // glog: the section logs of the resident workgroups
uint32_t vtxk_band_sweep_grid(uint32_t n_tasks);
#ifdef VTX_DEVTOOLS
// round 4's kernel (vtx_sweep_v1.hip, libvtx_dev.so only: VTX_SWEEP_V1=1), the reference of the A/B tests
hipError_t vtxk_launch_band_sweep_v1(int tier, const uint32_t* tasks, uint32_t n_tasks, const uint32_t* n_dev, const vtx_record* records,
                                     const uint32_t* rec_locus, const vtx_locus* loci, const uint8_t* read_arena, const uint8_t* hap_arena,
                                     uint16_t* band, uint32_t band_stride, uint32_t* hard_list, uint32_t* overflow_list, uint32_t* counters,
                                     uint32_t* stat_counters, uint8_t* stage, uint32_t* dbg, hipStream_t s);
#endif
uint32_t vtxk_band_sweep_max_len(void);
hipError_t vtxk_launch_sw_banded_dev(int R, int GL, uint32_t n_cap, const uint32_t* hard, const uint32_t* n_dev,
                                     const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                     const uint8_t* read_arena, const uint8_t* hap_arena, const uint16_t* band, uint32_t band_stride,
                                     int32_t* ref_score, int32_t* alt_score, uint32_t max_hap_len, hipStream_t stream);
hipError_t vtxk_launch_sw_check(int R, int GL, uint32_t n_cap, const uint32_t* list, const uint32_t* packs, const uint32_t* n_dev,
                                const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                const uint8_t* read_arena, const uint8_t* hap_arena, int32_t* ref_score, int32_t* alt_score,
                                uint32_t max_hap_len, uint32_t* recheck_list, uint32_t* recheck_pack, uint32_t* recheck_count,
                                uint8_t* stage, hipStream_t stream);
hipError_t vtxk_launch_sw_diag_band(int R, int GL, uint32_t n_cap, const uint32_t* list, const uint32_t* packs, const uint32_t* n_dev,
                                    const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                    const uint8_t* read_arena, const uint8_t* hap_arena, int32_t* ref_score, int32_t* alt_score,
                                    uint32_t max_hap_len, uint8_t* stage, hipStream_t stream);
hipError_t vtxk_mark_stage(const uint32_t* list, uint32_t n, const uint32_t* n_dev, uint8_t code, uint8_t* stage, hipStream_t s);
hipError_t vtxk_mark_stage_records(const uint32_t* recs, uint32_t n, uint8_t code, uint8_t* stage, hipStream_t s);
hipError_t vtxk_fill_i32(int32_t* p, uint32_t n, int32_t v, hipStream_t s);
int vtxk_band_second_chance(uint32_t tasks_per_locus, int long_lists);
uint32_t vtxk_band_lanes(uint32_t n_tasks);
size_t vtxk_band_coop_lds(uint32_t max_hap, uint32_t mc);
hipError_t vtxk_launch_band_coop(int tier, const uint32_t* tasks, uint32_t n_tasks, const vtx_record* records, const uint32_t* rec_locus,
                                 const vtx_locus* loci, const uint8_t* read_arena, const uint8_t* hap_arena, uint32_t max_hap,
                                 int32_t* ref_score, int32_t* alt_score, uint16_t* band, uint32_t band_stride, uint32_t* hard_list,
                                 uint32_t* overflow_list, uint32_t* counters, hipStream_t s);
size_t vtxk_band_gtables_bytes(uint32_t n_loci, uint32_t max_hap, uint32_t tasks_per_locus, uint32_t* loci_cap);
uint32_t vtxk_band_task_words(void);
uint32_t vtxk_band_pend_words(void);
uint32_t vtxk_band_run_grid(uint32_t nt, uint32_t wpe);
uint32_t vtxk_band_run_lanes(void);
hipError_t vtxk_launch_band_pending(const uint32_t* pending, uint32_t n_pending, const uint32_t* pend_buf,
                                    int32_t* ref_score, int32_t* alt_score, uint16_t* band, uint32_t band_stride,
                                    uint32_t* hard_list, uint32_t* counters, hipStream_t s);
uint32_t vtxk_band_poly_stride(void);
hipError_t vtxk_launch_band_expand(const uint32_t* hard_list, uint32_t n_hard, const vtx_record* records,
                                   const uint32_t* rec_locus, const vtx_locus* loci, const uint16_t* src,
                                   uint32_t src_stride, uint16_t* band, uint32_t band_stride, hipStream_t s);
/* Records the fast kernels hold: reads up to VTX_FAST_READ_LEN bases (16 rows x 64 lanes), haplotypes up to
 * VTX_FAST_HAP_LEN (LDS tables).  Longer ones take slow_align_kernel (exact, one lane per alignment, global scratch). */
#define VTX_FAST_READ_LEN 1024u
#define VTX_FAST_HAP_LEN 2400u
size_t vtxk_slow_ws_stride(uint32_t m_cap, uint32_t max_hap, uint32_t max_read);
hipError_t vtxk_launch_slow_align(const uint32_t* recs, const uint32_t* tasks, uint32_t n_tasks, int banded,
                                  const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci,
                                  const uint8_t* read_arena, const uint8_t* hap_arena, uint8_t* workspace, uint64_t ws_stride,
                                  uint32_t m_cap, uint32_t max_hap, uint32_t max_read, int32_t* ref_score, int32_t* alt_score,
                                  uint32_t* retry_list, uint32_t* counters, hipStream_t s);
hipError_t vtxk_values_from_counts(const uint32_t* alt, const uint32_t* ref, const uint32_t* unk, uint32_t n, int mode,
                                   double* o_val, double* o_refval, hipStream_t s);
hipError_t vtxk_group_heads(const vtx_record* records, const uint32_t* rec_locus, uint32_t n, uint32_t* head_cell,
                            uint32_t* head_umi, hipStream_t s);
hipError_t vtxk_group_table(const vtx_record* records, const uint32_t* rec_locus, const vtx_locus* loci, uint32_t n,
                            const uint32_t* head_cell, const uint32_t* head_umi, const uint32_t* cell_scan,
                            const uint32_t* umi_scan, uint32_t* grp_row, uint32_t* grp_col, uint32_t* umi_cellgrp,
                            hipStream_t s);
hipError_t vtxk_count_calls(const int32_t* ref_score, const int32_t* alt_score, uint32_t n, int32_t min_score,
                            const uint32_t* gscan, uint32_t* cnt, hipStream_t s);
hipError_t vtxk_umi_collapse(const uint32_t* umi_cnt, uint32_t n_umi, const uint32_t* umi_cellgrp,
                             uint32_t* cell_cnt, hipStream_t s);
hipError_t vtxk_keep_flags(const uint32_t* cell_cnt, uint32_t n_grp, int mode, uint32_t* keep, hipStream_t s);
hipError_t vtxk_emit_coo(const uint32_t* cell_cnt, uint32_t n_grp, int mode, const uint32_t* keep,
                         const uint32_t* keep_scan, const uint32_t* grp_row, const uint32_t* grp_col, uint32_t* o_row,
                         uint32_t* o_col, uint32_t* o_alt, uint32_t* o_ref, uint32_t* o_unk, double* o_val,
                         double* o_refval, hipStream_t s);
hipError_t vtxk_inclusive_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, void* temp, size_t temp_bytes,
                                   hipStream_t s);
size_t vtxk_scan_temp_bytes(uint32_t n);
// ---- vtx_prep.hip: device-side preparation of raw batches ----
hipError_t vtxk_prep_set_shapes(const uint32_t* caps, uint32_t n, uint32_t fast_read_len, uint32_t fast_hap_len);
size_t vtxk_prep_sort_temp_bytes(uint32_t n);
hipError_t vtxk_prep_sort_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                              uint32_t n, int end_bit, void* temp, size_t temp_bytes, hipStream_t s);
size_t vtxk_sort_keys_u32_temp_bytes(uint32_t n);
hipError_t vtxk_sort_keys_u32(const uint32_t* keys_in, uint32_t* keys_out, uint32_t n, void* temp, size_t temp_bytes, hipStream_t s);
hipError_t vtxk_prep_sort_u8(const uint8_t* keys_in, uint8_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                             uint32_t n, void* temp, size_t temp_bytes, hipStream_t s);
hipError_t vtxk_prep_rec_locus(const vtx_locus* loci, uint32_t n_loci, uint32_t* rec_locus, hipStream_t s);
hipError_t vtxk_prep_resolve(const vtx_raw_record* raw, uint32_t n, const uint32_t* rec_locus, const uint8_t* tags,
                             uint64_t tag_bytes, uint64_t read_bytes, uint32_t max_read_len, const uint32_t* bc_slots,
                             uint32_t bc_mask, const uint64_t* bc_hash, const uint64_t* bc_off, const uint8_t* bc_bytes,
                             int use_umi, uint64_t seed, uint64_t hash_mask, uint32_t cell_bits, uint32_t n_loci,
                             uint64_t* key_lc, uint64_t* key_umi, uint32_t* idx, unsigned long long* counters, hipStream_t s);
hipError_t vtxk_prep_gather_u64(const uint64_t* src, const uint32_t* idx, uint32_t n, uint64_t* dst, hipStream_t s);
hipError_t vtxk_prep_finalize(uint32_t n_kept, const uint32_t* perm, const uint64_t* key_lc_sorted, const uint64_t* key_umi,
                              const vtx_raw_record* raw, const uint8_t* tags, const vtx_locus* loci, uint32_t cell_bits,
                              int use_umi, uint32_t n_shapes, vtx_record* records, uint32_t* rec_locus, uint32_t* umi_head,
                              uint8_t* shape, uint32_t* seq, uint32_t* locus_first, uint32_t* locus_end, uint32_t* shape_cnt,
                              unsigned long long* counters, hipStream_t s);
hipError_t vtxk_prep_locus_counts(uint32_t* first_to_count, const uint32_t* end, uint32_t n_loci, hipStream_t s);
hipError_t vtxk_prep_umi_ids(vtx_record* records, const uint32_t* umi_scan, uint32_t n, hipStream_t s);
hipError_t vtxk_prep_locus_ranges(vtx_locus* loci, const uint32_t* cnt, const uint32_t* cnt_scan, uint32_t n_loci, hipStream_t s);
hipError_t vtxk_prep_lut_check(const uint32_t* work, uint32_t count, const uint32_t* rec_locus, uint32_t cap, uint32_t group,
                               uint32_t* flag, hipStream_t s);
hipError_t vtxk_unpack_nibbles(const uint8_t* in, uint64_t n_in, uint8_t* out, hipStream_t s);
hipError_t vtxk_prep_check(const vtx_record* records, uint32_t n, const uint32_t* rec_locus, const vtx_locus* loci,
                           uint64_t read_bytes, uint32_t max_read_len, uint32_t n_barcodes, uint32_t n_shapes, uint8_t* shape,
                           uint32_t* seq, uint32_t* shape_cnt, unsigned long long* counters, hipStream_t s);
}
#endif
