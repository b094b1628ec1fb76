/*
 * vtx_host.h — C ABI of the host-side packer (libvtxhost.so): ingest + read
 * filtering + haplotype construction, i.e. everything of the reference that
 * sits *above* the hot-path seam and produces the packed batch of vtx.h.
 *
 * Restates (reference 10XGenomics/vartrix v1.1.22, src/main.rs):
 *   load_barcodes :697-718 / open_with_gz :721-735, the VCF loop :221-234,
 *   validate_inputs :545-594, evaluate_rec :610-695 (multi-allelic skip,
 *   empty ALT, valid_chars), construct_haplotypes :958-994 + read_locus
 *   :936-954, evaluate_alns fetch + filters :822-895, useful_alignment
 *   :790-806, get_cell_barcode :737-750, get_umi :752-757, the stable sort by
 *   cell :932, Metrics :449-459, write_matrix_market :381-389,
 *   write_variants :1166-1179, write_barcodes :1181-1195.
 * Library code replaced: rust-htslib/htslib (BGZF, BAM, text VCF), rust-bio
 * fasta::IndexedReader, flate2, sprs — re-implemented on zlib only.
 * Inputs: text VCF (plain or gzip) and BCF2 (by content, like bcf::Reader::from_path :220); BAM with a .bai or a .csi (:520-529;
 * the window table the sweep and the device's plan use is the .bai's linear index or is rebuilt from the .csi's leaf bins);
 * FASTA + .fai.  Not supported (fails loudly): CRAM.
 *
 * Pure CPU code, no GPU needed: `tests/test_host.py` checks it against the
 * Python restatement on the reference's own fixtures.
 */
#ifndef VTX_HOST_H
#define VTX_HOST_H

#include <stdint.h>
#include "vtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Arguments struct src/main.rs:420-427 plus the file paths / --padding. */
typedef struct vtxh_args {
    const char* vcf;
    const char* bam;
    const char* fasta;
    const char* cell_barcodes;
    uint32_t padding;        /* --padding, default 100 (:88)                    */
    uint32_t mapq;           /* --mapq (:112-116)                               */
    int32_t primary_only;    /* --primary-alignments (:117-119)                 */
    int32_t no_duplicates;   /* --no-duplicates (:120-122)                      */
    int32_t use_umi;         /* --umi (:123-125)                                */
    const char* bam_tag;     /* --bam-tag, default "CB" (:126-129)              */
    const char* valid_chars; /* --valid-chars, default "ATGCatgc" (:130-133)    */
    int32_t threads;         /* BGZF inflate threads (does not affect results)  */
    int32_t read_format;     /* VTX_READS_BYTES (0): read arenas hold rec.seq().as_bytes() (:896); VTX_READS_NIBBLES: the BAM's own
                                two bases per byte — hand vtxh_read_format() to vtx_set_read_format before the submit            */
} vtxh_args;

/* Metrics, src/main.rs:449-459 */
typedef struct vtxh_metrics {
    uint64_t num_reads, num_low_mapq, num_non_primary, num_duplicates, num_not_cell_bc,
             num_not_useful, num_non_umi, num_invalid_recs, num_multiallelic_recs;
} vtxh_metrics;

typedef struct vtxh_pack vtxh_pack;

/* Ingest + filter + pack.  Returns 0 or a negative vtx_status; *out owns all
 * arrays until vtxh_free.  On failure vtxh_last_error() has the message.      */
int vtxh_pack_files(const vtxh_args* args, vtxh_pack** out);
void vtxh_free(vtxh_pack* p);
const char* vtxh_last_error(void);
/* A freed pack's large buffers (>= 1 MiB) are kept for the next pack of the process — at most 12 buffers / 2 GiB; their pages have
 * been touched, which is what a streamed run's next range gains from.  vtxh_trim() returns them to the allocator (a long-lived
 * process that packed once calls it after vtxh_free).
 * Limit: one pack holds at most 2^32 (read, locus) pairs (VTX_E_UNSUPPORTED beyond; pack ranges of VCF rows: vtxh_pack_files_range). */
void vtxh_trim(void);

/* Same ingest, but everything per-read that follows the alignment-level filters — barcode
 * dictionary lookup (:867-876), UB test (:879-888), UMI grouping (:1047-1057) and the sort by
 * cell (:932) — is left to the device (vtx_submit_raw in vtx.h): records stay in BAM order inside
 * a locus, barcode / UMI travel as tag bytes.  Metrics: num_not_cell_bc counts only the reads
 * WITHOUT a usable barcode tag (the in-list test happens on the device) and num_non_umi is 0;
 * add vtx_raw_stats to both.                                                                    */
int vtxh_pack_files_raw(const vtxh_args* args, vtxh_pack** out);
void vtxh_get_raw_batch(const vtxh_pack* p, vtx_raw_batch* out);
/* The barcode list in the layout vtx_set_barcodes takes (n = vtxh_num_barcodes). */
void vtxh_get_barcode_table(const vtxh_pack* p, const uint8_t** bytes, const uint64_t** offsets, uint32_t* n);

/* Streaming: the same ingest restricted to the VCF records (matrix rows) [row_begin, row_end).  Rows outside the range keep their
 * place in the matrix (vtxh_num_variants, names) but bring no loci, no reads and no metrics; with a usable .bai only the stretches
 * of the BAM that can hold reads of the range are inflated.  The packs of consecutive ranges add up to vtxh_pack_files of the whole
 * input (same triplets in range order, metrics summed), so a host can hold ONE range in memory at a time — the reference itself
 * holds one locus' reads at a time (src/main.rs:822-830) — and pack range k + 1 while the device works on range k.           */
int vtxh_pack_files_range(const vtxh_args* args, int raw, uint32_t row_begin, uint32_t row_end, vtxh_pack** out);

/* Device-side ingest (vtx_submit_bam in vtx.h): the PLAN is everything vtxh_pack_files does before it touches a read — barcodes, VCF
 * rows [row_begin, row_end), haplotypes, loci — plus an index of the BAM for the device: the BGZF blocks that can hold reads of those
 * loci (a header walk; the file stays mapped, nothing but its header is inflated), the record starts the .bai's linear index names
 * inside them, and the offset of the first record that lies beyond the last locus.  vtxh_get_ingest fills the struct vtx_submit_bam
 * takes (pointers valid until vtxh_free); VTX_E_UNSUPPORTED — no usable .bai, loci so sparse that an index-guided sweep inflates far
 * less, an index that does not match the file — means: pack on the host (vtxh_pack_files_range), as before.  The returned pack
 * carries the loci, names, barcode table and the two VCF-level Metrics like any raw pack; it has no batches.                    */
int vtxh_plan_ingest(const vtxh_args* args, uint32_t row_begin, uint32_t row_end, vtxh_pack** out);
int vtxh_get_ingest(const vtxh_pack* p, vtx_bam_ingest* out);
int vtxh_is_plan(const vtxh_pack* p);

/* A pack holds one or more BATCHES: consecutive loci whose reads span less than 4 GiB of the arenas, so that the
 * 32-bit offsets of vtx.h hold relative to the batch (the reference has no such limit: it streams per locus).
 * Loci keep their global `row`; feed the batches to vtx_submit / vtx_submit_raw one after the other (or to different
 * devices) and append their triplets in batch order.  vtxh_get_batch / vtxh_get_raw_batch return batch 0.            */
uint32_t vtxh_num_batches(const vtxh_pack* p);
void vtxh_get_batch_at(const vtxh_pack* p, uint32_t i, vtx_batch* out);
void vtxh_get_raw_batch_at(const vtxh_pack* p, uint32_t i, vtx_raw_batch* out);

/* The packed batch (pointers valid until vtxh_free). */
void vtxh_get_batch(const vtxh_pack* p, vtx_batch* out);
void vtxh_get_metrics(const vtxh_pack* p, vtxh_metrics* out);
/* Ingest statistics: {BGZF blocks inflated, BGZF blocks in the file, index-guided jumps}.  With a usable .bai only the
 * stretches of the BAM that can hold reads of a locus are inflated (the reference's indexed fetch, src/main.rs:822-826). */
void vtxh_get_ingest_stats(const vtxh_pack* p, uint64_t out[3]);
uint32_t vtxh_num_variants(const vtxh_pack* p);   /* matrix rows = VCF records (:237)      */
uint32_t vtxh_num_barcodes(const vtxh_pack* p);   /* matrix cols = distinct barcodes (:245) */
/* "{chrom}_{pos0}" of VCF record i (write_variants :1174); barcode of column j. */
const char* vtxh_variant_name(const vtxh_pack* p, uint32_t i);
const char* vtxh_barcode(const vtxh_pack* p, uint32_t j);

/* sprs::io::write_matrix_market of a TriMat<f64> (:381): 3 header lines, then
 * "row+1 col+1 value" in the given order with Rust `{}` float text.           */
int vtxh_write_mtx(const char* path, uint32_t n_rows, uint32_t n_cols, uint64_t nnz,
                   const uint32_t* row, const uint32_t* col, const double* value);
int vtxh_format_f64(double v, char* buf32);
/* the format of the pack's read arenas (vtxh_args.read_format as honoured) */
int vtxh_read_format(const vtxh_pack* p);

/* Test hook: the packer's own raw-DEFLATE decoder (vartrix_amd/csrc/host/vtx_inflate.h; the blocks htslib's bgzf_read hands
 * to zlib behind src/main.rs:822-830) on one stream whose output size is known.  1: accepted, out holds out_len bytes; 0: the
 * decoder declined (the packer then gives the block to zlib).  Nothing is written outside [out, out + out_len).               */
int vtxh_test_inflate(const uint8_t* in, uint64_t in_len, uint8_t* out, uint64_t out_len);

#ifdef __cplusplus
}
#endif
#endif
