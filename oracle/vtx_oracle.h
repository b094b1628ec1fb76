/*
 * vtx_oracle.h — CPU ORACLE, TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under oracle/ is part of the product.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load this library, and
 * only as the checker / the reported CPU baseline — never as the thing that is
 * shipped or measured as "the GPU path".
 *
 * It restates, in plain C, the reference's per-locus genotyping path
 * (10XGenomics/vartrix v1.1.22, src/main.rs) and the algorithm of the
 * third-party crate the reference calls for its arithmetic:
 *
 *     bio 0.30.0  (Cargo.lock:175-178, checksum cc0376f0...d449bd)
 *       alignment::pairwise::banded::Aligner::{new, local}   src/main.rs:899-901
 *       alignment::sparse::{find_kmer_matches, sdpkpp}
 *       alignment::pairwise::{Scoring, MIN_SCORE}
 *
 * The crate source is NOT in /root/reference, is not vendored, and cannot be
 * fetched (no network); there is no Rust toolchain in the image, so the
 * reference cannot be compiled (`oracle/_ref` does not exist for this
 * project).  PINNING STATUS:
 *   - full pipeline (filters -> haplotypes -> SW -> calls -> UMI collapse ->
 *     matrix modes) is pinned by all six runnable .mtx fixtures of the
 *     reference's own tests (tests/golden/, checked in tests/test_golden.py)
 *     for BOTH aligner flavours;
 *   - the band geometry of bio's banded aligner (seed chaining + band
 *     extension, functions vtxo_sdpkpp / vtxo_band_create below) is restated
 *     from the crate's published algorithm and documentation; the reference's
 *     fixtures exercise only 15 clean reads and do not distinguish it from
 *     full Smith-Waterman:  **parity unpinned** for band geometry and indels.
 */
#ifndef VTX_ORACLE_H
#define VTX_ORACLE_H

#include <stdint.h>
#include "../include/vtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* bio::alignment::pairwise::MIN_SCORE */
#define VTXO_MIN_SCORE (-858993459)
/* bio::alignment::pairwise::banded::MAX_CELLS */
#define VTXO_MAX_CELLS 5000000

/* Per-read call codes, src/main.rs:27-31 (0 = None of evaluate_scores). */
#define VTXO_CALL_NONE 0
#define VTXO_CALL_REF 1
#define VTXO_CALL_ALT 2
#define VTXO_CALL_UNKNOWN (-1)

/* Full-matrix affine local alignment score of x (read) vs y (haplotype). */
int32_t vtxo_sw_full(const uint8_t* x, int m, const uint8_t* y, int n,
                     int match, int mismatch, int gap_open, int gap_extend);

/* sparse::find_kmer_matches: all (i, j) with x[i..i+k] == y[j..j+k], sorted
 * lexicographically.  Returns the count; *out is malloc'ed (2 x u32 each).   */
int64_t vtxo_find_kmer_matches(const uint8_t* x, int m, const uint8_t* y, int n,
                               int k, uint32_t** out);

/* sparse::sdpkpp: best chain of k-mer matches.  path_out (capacity
 * n_matches) receives match indices in chain order; returns path length.     */
int64_t vtxo_sdpkpp(const uint32_t* matches, int64_t n_matches, int k,
                    int match_score, int gap_open, int gap_extend,
                    int64_t* path_out, int64_t* score_out);

/* banded::Band::create: per-column row ranges [lo[j], hi[j]) for j in 0..=n
 * (rows 0..=m).  Empty column: lo > hi.  Returns number of cells in band.    */
int64_t vtxo_band_create(const uint8_t* x, int m, const uint8_t* y, int n,
                         int k, int w, int32_t* lo, int32_t* hi);

/* Test hook: lazy-extension length of set_boundaries for the NEXT band constructions (any value >= 0;
 * VTX_BAND_EXT_TO_EDGE = to the matrix corner); negative restores the constant of include/vtx_band_semantics.h.   */
void vtxo_set_lazy_extension(int ext);
/* Test hook: one of the recollected details of the crate (see vtx_oracle.c: VTXO_VAR_*), value < 0 restores the default.    */
void vtxo_set_variant(int which, int value);

/* banded::Aligner::local(x, y).score */
int32_t vtxo_sw_banded(const uint8_t* x, int m, const uint8_t* y, int n,
                       int match, int mismatch, int gap_open, int gap_extend,
                       int k, int w);

/* banded DP given explicit ranges (used to cross-check device band output).  */
int32_t vtxo_sw_ranges(const uint8_t* x, int m, const uint8_t* y, int n,
                       int match, int mismatch, int gap_open, int gap_extend,
                       const int32_t* lo, const int32_t* hi);

/* evaluate_scores, src/main.rs:1019-1030 */
int vtxo_evaluate_scores(int32_t ref_score, int32_t alt_score, int32_t min_score);

/* evaluate_chunk over a packed batch (src/main.rs:596-607 + :898-901):
 * scores for every record, loci split into static contiguous chunks of
 * max(n_loci / threads, 1) like src/main.rs:250-254.  Returns 0.             */
int vtxo_batch_scores(const vtx_batch* b, const vtx_config* cfg,
                      int32_t* ref_score, int32_t* alt_score, int threads);

/* Number of DP cells (i>=1, j>=1) evaluate_chunk evaluates for the batch.    */
uint64_t vtxo_batch_cells(const vtx_batch* b, const vtx_config* cfg, int threads);

/* The merge loop src/main.rs:320-348 over a packed batch: parse_scores +
 * consensus_scoring / alt_frac / coverage.  Output arrays have capacity
 * n_records.  Returns nnz.                                                   */
int64_t vtxo_batch_reduce(const vtx_batch* b, const vtx_config* cfg,
                          const int32_t* ref_score, const int32_t* alt_score,
                          uint32_t* row, uint32_t* col, uint32_t* alt, uint32_t* ref,
                          uint32_t* unk, double* value, double* ref_value);

/* rust-htslib 0.36 CigarStringView::read_pos(ref_pos, include_softclips,
 * include_dels) as used by useful_alignment, src/main.rs:790-806.
 * cigar = BAM-encoded ops (len<<4 | op).  Returns 1 = Some, 0 = None, -1 = Err. */
int vtxo_cigar_read_pos(const uint32_t* cigar, int n_ops, int64_t pos,
                        int64_t ref_pos, int include_softclips, int include_dels,
                        int64_t* qpos);

/* useful_alignment, src/main.rs:790-806: 1 useful, 0 not.                    */
int vtxo_useful_alignment(const uint32_t* cigar, int n_ops, int64_t pos,
                          int64_t locus_start, int64_t locus_end);

/* construct_haplotypes, src/main.rs:958-994, on an in-memory contig.  ref_out /
 * alt_out need capacity (ref_len + alt_len + 2*padding).                      */
void vtxo_construct_haplotypes(const uint8_t* contig, int64_t contig_len,
                               int64_t start, int64_t end,
                               const uint8_t* alt, int64_t alt_len, int64_t padding,
                               uint8_t* ref_out, int64_t* ref_out_len,
                               uint8_t* alt_out, int64_t* alt_out_len);

/* Rust `{}` Display of an f64 as sprs::io::write_matrix_market prints it
 * (src/main.rs:381).  buf needs >= 32 bytes.  Returns length.                 */
int vtxo_format_f64(double v, char* buf);

/* ---- vtx_certify.c: CPU restatement of the device's DP-free certificate (lower bound = score of the
 * chain's anchor staircase, upper bound = chain of exact-match runs); see the proof in that file.   */
int32_t vtxo_chain_cert(const uint8_t* x, int m, const uint8_t* y, int n, int k);
int vtxo_join_same(int D);
int vtxo_join_gap(int D);
void vtxo_set_join_exact(int on);
void vtxo_set_corridor(int kc);
int32_t vtxo_runs_ub_exact(const uint8_t* x, int m, const uint8_t* y, int n, int k);
int32_t vtxo_runs_ub(const uint8_t* x, int m, const uint8_t* y, int n, int k, int* passes_out, int* npieces_out);
int vtxo_batch_certify(const vtx_batch* b, const vtx_config* cfg, int32_t* full, int32_t* banded, int32_t* cert,
                       int32_t* ub_exact, int32_t* ub, int32_t* passes, int32_t* npieces, int threads);

#ifdef __cplusplus
}
#endif
#endif
