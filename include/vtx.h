/*
 * vtx.h — C ABI of the MI355X-native VarTrix genotyping hot path.
 *
 * This is the drop-in boundary for the reference's per-locus hot path.  The
 * reference (10XGenomics/vartrix v1.1.22, one Rust file) has no FFI of its own;
 * the seam this library replaces is the rayon map over chunks of variant loci
 *
 *     src/main.rs:279-291   pool.install(|| rec_chunks.par_iter()
 *                               .map_with(rdr, |r, c| evaluate_chunk(c, r, &args)).collect())
 *     src/main.rs:596-607   evaluate_chunk -> Vec<(usize, EvaluateAlnResults)>
 *
 * plus its consumer, the single-threaded merge loop src/main.rs:320-348 that
 * turns per-locus Scores into matrix triplets.  The host keeps BAM/VCF/FASTA
 * ingest, read filtering (src/main.rs:829-895) and haplotype construction
 * (src/main.rs:958-994) and hands this library *packed batches*:
 *
 *   hap arena   : REF and ALT haplotype byte strings of every locus
 *   read arena  : bases of every read that reached the aligner (src/main.rs:896)
 *   records     : one per (locus, read) = one `Scores` entry of the reference
 *                 (src/main.rs:996-1001), carrying cell_index and an interned
 *                 UMI id instead of the UMI bytes (only UMI equality is used,
 *                 src/main.rs:1053-1056)
 *   loci        : matrix row + the record range + hap offsets of each locus
 *
 * and receives (a) the two Smith-Waterman scores per record — the inner seam
 * src/main.rs:898-901/926-927 — and (b) the matrix triplets of
 * consensus_scoring / alt_frac / coverage (src/main.rs:1111-1164) in the
 * insertion order of the merge loop (row ascending, then cell_index ascending,
 * src/main.rs:320-348 + :932).
 *
 * Conventions: plain C, no exceptions cross the boundary, every entry point
 * returns 0 (VTX_OK) or a negative vtx_status; vtx_strerror gives the message.
 * Inputs are borrowed for the duration of the call only.  Outputs returned by
 * pointer are owned by the context and stay valid until the next vtx_run /
 * vtx_destroy on that context.  A context is bound to one HIP device and is
 * not thread-safe; distinct contexts are independent (one per GPU / per host
 * thread — the analogue of one rayon worker, src/main.rs:466-473).
 *
 * There is NO CPU fallback in this library: every compute entry point needs a
 * gfx950 device and fails with VTX_E_NODEVICE otherwise.
 */
#ifndef VTX_H
#define VTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTX_ABI_VERSION 6

typedef enum vtx_status {
    VTX_OK = 0,
    VTX_E_INVAL = -1,      /* bad argument / malformed batch                      */
    VTX_E_NODEVICE = -2,   /* no usable HIP device (the library has no CPU path)  */
    VTX_E_HIP = -3,        /* a HIP runtime call failed                           */
    VTX_E_NOMEM = -4,      /* host or device allocation failed                    */
    VTX_E_UNSUPPORTED = -5,/* valid request this build cannot serve               */
    VTX_E_STATE = -6,      /* call sequence error (e.g. fetch before run)         */
    VTX_E_PEER = -7        /* a collective was abandoned because another rank reported an error */
} vtx_status;

/* Which restatement of bio::alignment::pairwise::banded::Aligner::local
 * (crate bio 0.30.0, called at src/main.rs:899-901) is computed.            */
typedef enum vtx_aligner {
    VTX_ALIGNER_BANDED = 0, /* k-mer seeded band (K=6, W=20, src/main.rs:33-34) */
    VTX_ALIGNER_FULL = 1    /* full-matrix affine local SW                      */
} vtx_aligner;

/* --scoring-method, src/main.rs:89-94 / :323-346 */
typedef enum vtx_scoring_mode {
    VTX_MODE_CONSENSUS = 0, /* consensus_scoring src/main.rs:1111-1129 */
    VTX_MODE_ALT_FRAC = 1,  /* alt_frac          src/main.rs:1131-1145 */
    VTX_MODE_COVERAGE = 2   /* coverage          src/main.rs:1147-1164 */
} vtx_scoring_mode;

/* Compile-time constants of the reference (src/main.rs:27-38) made explicit.
 * vtx_config_default() fills the reference's values.                        */
typedef struct vtx_config {
    int32_t abi_version;   /* must be VTX_ABI_VERSION                           */
    int32_t device;        /* HIP device ordinal                                */
    int32_t aligner;       /* vtx_aligner                                       */
    int32_t scoring_mode;  /* vtx_scoring_mode                                  */
    int32_t use_umi;       /* --umi, src/main.rs:121-123                        */
    int32_t match_score;   /* MATCH      =  1  src/main.rs:35                   */
    int32_t mismatch_score;/* MISMATCH   = -5  src/main.rs:36                   */
    int32_t gap_open;      /* GAP_OPEN   = -5  src/main.rs:37                   */
    int32_t gap_extend;    /* GAP_EXTEND = -1  src/main.rs:38                   */
    int32_t min_score;     /* MIN_SCORE  = 25  src/main.rs:30                   */
    int32_t kmer_k;        /* K = 6            src/main.rs:33                   */
    int32_t band_w;        /* W = 20           src/main.rs:34                   */
    uint32_t n_barcodes;   /* matrix columns (cell_barcodes.len(), :247)        */
    uint32_t reserved;
} vtx_config;

/* One variant locus = one VCF record that reached evaluate_alns
 * (src/main.rs:686).  Skipped loci (multi-allelic :646-653, invalid ALT
 * haplotype :675-684) are simply not submitted; their matrix row stays empty. */
typedef struct vtx_locus {
    uint32_t row;        /* RecHolder.i — matrix row, src/main.rs:226-228        */
    uint32_t rec_begin;  /* first record of this locus in records[]             */
    uint32_t rec_count;  /* number of records (reads that reached the aligner)  */
    uint32_t ref_off;    /* REF haplotype bytes in hap_arena (src/main.rs:984)  */
    uint32_t ref_len;
    uint32_t alt_off;    /* ALT haplotype bytes in hap_arena (src/main.rs:977-981) */
    uint32_t alt_len;
    uint32_t reserved;
} vtx_locus;

/* One scored read at one locus = one `Scores` of the reference
 * (src/main.rs:923-930) before alignment.  Within a locus, records MUST be
 * ordered by (cell_index, umi_id) ascending — the reference's stable sort by
 * cell_index (src/main.rs:932) followed by the per-cell UMI HashMap
 * (src/main.rs:1047-1057), whose iteration order never reaches the output.   */
typedef struct vtx_record {
    uint32_t read_off;   /* read bases in read_arena (rec.seq().as_bytes(), :896) */
    uint32_t read_len;
    uint32_t cell_index; /* column, get_cell_barcode src/main.rs:737-750         */
    uint32_t umi_id;     /* interned UB bytes; any value when !use_umi           */
} vtx_record;

typedef struct vtx_batch {
    const vtx_locus* loci;
    uint32_t n_loci;
    const vtx_record* records;
    uint32_t n_records;
    const uint8_t* hap_arena;   /* ASCII bytes; compared by byte equality (:898)  */
    uint64_t hap_bytes;
    const uint8_t* read_arena;  /* ASCII bytes as produced by BAM seq decoding    */
    uint64_t read_bytes;
} vtx_batch;

/* Matrix triplets in merge-loop order (src/main.rs:320-348).  `alt`, `ref`,
 * `unk` are the CellCounts of convert_to_counts (src/main.rs:1032-1039) for the
 * (row, col) group after optional UMI collapse; `value` is the f64 the
 * reference adds to `matrix`, `ref_value` the one it adds to `ref_matrix`
 * (coverage mode only, src/main.rs:336-344; 0 otherwise).                     */
typedef struct vtx_coo {
    const uint32_t* row;
    const uint32_t* col;
    const uint32_t* alt;
    const uint32_t* ref;
    const uint32_t* unk;
    const double* value;
    const double* ref_value;
    uint64_t nnz;
} vtx_coo;

/* Device-side timing of the last vtx_run, from hipEvents on the context's
 * stream (ms).  `sw_ms` covers the alignment kernels only: the full-matrix DP
 * (full flavour) or the band kernels + band-masked DP (banded flavour — it
 * never runs the full-matrix DP).                                            */
typedef struct vtx_timing {
    float total_ms;
    float sw_ms;
    float reduce_ms;
    uint32_t sw_launches;
    uint32_t hard_tasks;   /* banded flavour: alignments that needed the band-masked DP */
    float full_ms;         /* sw_full_kernel launches only (part of sw_ms)                */
    float band_ms;         /* band kernels + band-masked DP (banded flavour; part of sw_ms) */
    float band_run_ms;     /* band_run_kernel launches only (seeds, chain, certificate; part of band_ms) */
    uint32_t overflow_tasks; /* banded flavour: alignments handed to the general band kernel        */
    float diag_ms;         /* band_tables_kernel + band_diag_kernel (single-diagonal stage; part of band_run_ms)   */
    uint32_t diag_left;    /* alignments the certificate stages left: band_sweep_kernel + masked DP take them       */
    float check_ms;        /* band-masked DP (and, with VTX_BAND_CHECK, the full-matrix check) of the tasks that left with a certificate */
    float sweep_ms;        /* band_sweep_kernel + band-masked DP over what is left (part of band_ms)                 */
    uint32_t checked_tasks; /* alignments that left the certificate stages WITH a certificate: one-diagonal band + masked DP */
    uint32_t swept_tasks;  /* alignments handed to band_sweep_kernel                                                 */
    uint32_t resweep_tasks; /* (round 4's sweep kernel, developer build only: tasks of its second pass)                     */
    uint32_t diag2_tasks;  /* alignments the second single-diagonal stage looked at (band_diag2_kernel: what the first left with
                              more off-diagonal matches than its list holds) ...                                              */
    uint32_t diag2_scored; /* ... of which it decided outright (cert == ub); the rest of them is in checked_tasks (one-diagonal
                              band) or swept_tasks                                                                            */
    uint32_t diag2_streamed; /* ... and how many exceeded its list and took band_stream_kernel (the harmless test over a window)       */
} vtx_timing;

typedef struct vtx_ctx vtx_ctx;

/* Reference constants (src/main.rs:27-38), banded aligner, consensus mode.   */
void vtx_config_default(vtx_config* cfg);

/* Replaces the per-run setup of src/main.rs:266-283 (Arguments + thread pool). */
int vtx_create(const vtx_config* cfg, vtx_ctx** out);
void vtx_destroy(vtx_ctx* ctx);

/* Upload one packed batch into HBM (validates offsets/ordering first).
 * Replaces handing `rec_chunk` to a worker, src/main.rs:285-289.  The batch
 * stays resident until the next vtx_submit, so vtx_run may be repeated.      */
int vtx_submit(vtx_ctx* ctx, const vtx_batch* batch);

/* Format of the read arena of the batches submitted from now on (vtx_submit and vtx_submit_raw; sticky until set again):
 *   VTX_READS_BYTES    one ASCII byte per base, what rec.seq().as_bytes() returns (src/main.rs:896) — the default;
 *   VTX_READS_NIBBLES  two bases per byte as the BAM record holds them (high nibble first, "=ACMGRSVTWYHKDBN"): half the bytes to
 *                      write on the host and to move over PCIe.  Offsets and lengths still count BASES of the unpacked arena:
 *                      read i's packed bytes start at read_arena + read_off / 2, so every read_off must be even (a read of odd
 *                      length is followed by one unused base), read_bytes is even and the packed arena holds read_bytes / 2
 *                      bytes; the device unpacks it once per submit (unpack_nibbles_kernel) into the arena the kernels read. */
#define VTX_READS_BYTES 0
#define VTX_READS_NIBBLES 1
int vtx_set_read_format(vtx_ctx* ctx, int format);

/* ---- raw batches: barcode lookup, UMI grouping and the sort done on the device ----------------
 * A raw record is a read that passed the alignment-level filters of evaluate_alns
 * (mapq / flags / useful_alignment, src/main.rs:833-864) but whose cell barcode and UMI are
 * still the tag BYTES of the BAM record.  The device then does what the reference does per read
 * on the CPU: `cell_barcodes.get(..)` (get_cell_barcode, src/main.rs:737-750, :867-876), the
 * UB-present test (:879-888), the grouping of reads by UMI byte string (parse_scores' per-cell
 * HashMap, :1047-1057) and the sort by cell (:932).                                               */
#define VTX_TAG_MISSING 0xffffu   /* umi_len of a read without the UB tag */

typedef struct vtx_raw_record {
    uint32_t read_off;   /* into read_arena */
    uint32_t read_len;
    uint32_t bc_off;     /* cell-barcode tag bytes in tag_arena (the whole Z string, e.g. "ACGT...-1") */
    uint32_t umi_off;    /* UB tag bytes in tag_arena; ignored unless cfg.use_umi                       */
    uint16_t bc_len;
    uint16_t umi_len;    /* VTX_TAG_MISSING: no UB tag (dropped and counted when cfg.use_umi)            */
} vtx_raw_record;

typedef struct vtx_raw_batch {
    const vtx_locus* loci;          /* rec_begin/rec_count delimit RAW records (any order inside a locus) */
    uint32_t n_loci;
    const vtx_raw_record* records;
    uint32_t n_records;
    const uint8_t* hap_arena;
    uint64_t hap_bytes;
    const uint8_t* read_arena;
    uint64_t read_bytes;
    const uint8_t* tag_arena;
    uint64_t tag_bytes;
} vtx_raw_batch;

typedef struct vtx_raw_stats {
    uint64_t num_not_cell_bc;   /* barcode not in the list (Metrics.num_not_cell_bc, src/main.rs:870-876) */
    uint64_t num_non_umi;       /* use_umi and no UB tag, after the barcode test (:879-888)               */
    uint64_t kept;              /* records that reach the aligner                                         */
    float prep_ms;              /* device time of lookup + sort + regrouping                              */
    uint32_t hash_rounds;       /* UMI hash seeds tried (1 unless two UMIs of one cell collided)          */
} vtx_raw_stats;

/* Upload the barcode list (load_barcodes, src/main.rs:697-718): barcode j is
 * bytes[offsets[j] .. offsets[j+1]); j is the matrix column.  A byte string listed twice keeps
 * its FIRST index (:704-710).  n must equal cfg.n_barcodes.                                       */
int vtx_set_barcodes(vtx_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets, uint32_t n);

/* vtx_submit for a raw batch: same resident state afterwards (vtx_run / vtx_fetch_* unchanged).
 * Records are resolved, filtered and ordered by (row, cell_index, UMI) on the device; the order
 * of reads inside one (row, cell, UMI) group is unspecified (no result depends on it).           */
int vtx_submit_raw(vtx_ctx* ctx, const vtx_raw_batch* batch, vtx_raw_stats* stats);

/* Parity/debug: the resolved, sorted records of the resident batch and the per-locus record
 * ranges (n_records entries of vtx_record — umi_id is a dense group number — and n_loci
 * (rec_begin, rec_count) pairs).  Buffers are caller-allocated; n_records = stats.kept.           */
int vtx_fetch_records(vtx_ctx* ctx, vtx_record* records, uint32_t* rec_begin, uint32_t* rec_count);

/* ---- vtx_submit_bam: the ingest itself on the device (round 6) -------------------------------------------------------------
 * Replaces, for one contiguous stretch of a coordinate-sorted BAM, what the reference does per locus on the CPU before the
 * aligner: `bam.fetch(tid, start, end)` + `bam.records()` (src/main.rs:822-830; rust-htslib -> htslib bgzf_read -> zlib below
 * them), the read filters in their order with the Metrics counters (:831-864: num_reads, mapq, primary, duplicates,
 * useful_alignment :790-806), the barcode / UB tags as bytes (:737-757) and rec.seq() (:896); then, as after vtx_submit_raw, the
 * in-list test, the UB test, the UMI grouping and the sort by cell.  The host hands over the FILE BYTES and an index of them:
 *   blocks     consecutive BGZF blocks (payload offset / size, ISIZE) — a header walk, no inflate;
 *   seeds      ascending offsets into the inflated stream of those blocks at which a BAM record starts: the first record to look at
 *              and whatever record starts the .bai's linear index names (SAM spec 5.2: one per 16 kb window with alignments).
 *              Record boundaries are a serial block_size chain; the seeds cut it into independent pieces.  A chain that does not
 *              land exactly on the next seed (index and file disagree) is an error, not a guess;
 *   end_upos   records starting at or beyond this offset are not looked at (a record start, or the end of the stream);
 *   intervals  the loci as the fetch sees them, per contig sorted by start (Locus {chrom, start, end}, src/main.rs:619-623);
 *   loci / hap_arena   as in vtx_raw_batch; rec_begin / rec_count are ignored (the device fills them).
 * The same resident state afterwards as after vtx_submit_raw (vtx_run / vtx_fetch_* unchanged); stats carries the Metrics
 * counters the filters produce.  VTX_E_UNSUPPORTED (with the reason in vtx_strerror) when the device declines — a block its
 * inflater does not accept, reads above one batch's 4 GiB arena: the caller then packs on the host (libvtxhost), where zlib and the
 * multi-batch logic live.  Nothing is read from the file but [blocks[0].coff, blocks[n - 1].coff + clen).                         */
typedef struct vtx_bgzf_block {
    uint64_t coff;       /* file offset of the block's raw-DEFLATE payload (behind the gzip header and its extra field) */
    uint32_t clen;       /* payload bytes (BSIZE + 1 - header - 8)                                                      */
    uint32_t isize;      /* ISIZE: bytes it inflates to (<= 65536)                                                      */
} vtx_bgzf_block;

typedef struct vtx_bam_interval {
    int32_t start, end;  /* [start, end) on its contig, 0-based */
    uint32_t locus;      /* index into vtx_bam_ingest.loci      */
    uint32_t reserved;
} vtx_bam_interval;

typedef struct vtx_bam_ingest {
    const uint8_t* file;                 /* the BAM file's bytes (e.g. a read-only mapping) */
    uint64_t file_bytes;
    const vtx_bgzf_block* blocks;
    uint32_t n_blocks;
    uint32_t n_ref;                      /* contigs of the BAM header */
    const uint64_t* seeds;
    uint32_t n_seeds;
    uint32_t n_intervals;
    uint64_t end_upos;
    const vtx_bam_interval* intervals;   /* contig t owns [tid_begin[t], tid_begin[t + 1]) */
    const uint32_t* tid_begin;           /* n_ref + 1 entries */
    const int32_t* tid_max_span;         /* n_ref entries: the longest end - start among the contig's intervals */
    const vtx_locus* loci;
    uint32_t n_loci;
    uint32_t min_mapq;                   /* --mapq (src/main.rs:112-116)            */
    int32_t primary_only;                /* --primary-alignments (:117-119)         */
    int32_t no_duplicates;               /* --no-duplicates (:120-122)              */
    const uint8_t* hap_arena;
    uint64_t hap_bytes;
    char bam_tag[2];                     /* --bam-tag, default "CB" (:126-129)      */
    char reserved[6];
} vtx_bam_ingest;

typedef struct vtx_ingest_stats {
    uint64_t num_reads, num_low_mapq, num_non_primary, num_duplicates, num_not_useful;   /* Metrics, src/main.rs:449-459 */
    uint64_t num_no_barcode_tag;         /* reads without a usable barcode tag (part of num_not_cell_bc; raw.num_not_cell_bc has the rest) */
    uint64_t bam_records, raw_records;   /* BAM records looked at; (read, locus) pairs handed to the preparation */
    uint64_t compressed_bytes, inflated_bytes;
    vtx_raw_stats raw;
    float h2d_ms, inflate_ms, index_ms, filter_ms;   /* host wall time of the upload; device time of the inflate, the record chains, the filter passes */
    float prefetch_ms, prefetch_wait_ms;             /* vtx_prefetch_file: how long its copy took; how long vtx_submit_bam waited for it (0: not used) */
} vtx_ingest_stats;

/* Optional, before vtx_submit_bam: start moving bytes [file_off, file_off + n) of the file at `path` (n = 0: to its end) to the device
 * NOW — while the host still parses the VCF and walks the BGZF headers.  Returns at once; library threads copy the file (through a
 * read-only mapping of their own) into their pinned buffers and push them.  A later
 * vtx_submit_bam whose blocks lie inside the range uses the bytes (it waits for the copy first), any other uploads its own.  A
 * context may be created for this before the barcode list is known: cfg.n_barcodes = 0 means "taken from the first
 * vtx_set_barcodes".                                                                                                              */
int vtx_prefetch_file(vtx_ctx* ctx, const char* path, uint64_t file_off, uint64_t n);
int vtx_submit_bam(vtx_ctx* ctx, const vtx_bam_ingest* ingest, vtx_ingest_stats* stats);

/* Test / audit hook: intermediate arrays of the last vtx_submit_bam.  *bytes = the array's size, min(cap, *bytes) bytes go to dst. */
#define VTX_INGEST_INFLATED 0        /* the inflated stream of the submitted blocks                           */
#define VTX_INGEST_RECORD_OFFSETS 1  /* uint64 per BAM record: where its block_size word lies in that stream */
#define VTX_INGEST_RAW_RECORDS 2     /* vtx_raw_record per surviving (read, locus) pair, BAM order            */
#define VTX_INGEST_RAW_LOCUS 3       /* uint32 per pair: its locus                                            */
#define VTX_INGEST_TAGS 4            /* the tag arena those records point into                                */
#define VTX_INGEST_READS_PACKED 5    /* the read arena, two bases per byte                                    */
int vtx_debug_ingest(vtx_ctx* ctx, int what, void* dst, uint64_t cap, uint64_t* bytes);
/* Test hook: the device's BGZF inflater on arbitrary raw-DEFLATE payloads (block i = file[coff .. coff + clen), expected to inflate
 * to isize bytes): status[i] = 0 accepted, its bytes at out + (sum of the isizes before it); else the decoder's reason.  The product
 * path (vtx_submit_bam) treats any non-zero status as "the host packer decides".                                              */
int vtx_debug_inflate(vtx_ctx* ctx, const uint8_t* file, uint64_t file_bytes, const vtx_bgzf_block* blocks, uint32_t n_blocks,
                      uint8_t* out, uint64_t out_cap, uint32_t* status);

/* Run the hot path on the resident batch: Smith-Waterman of every record
 * against both haplotypes (src/main.rs:898-901), per-read call
 * (evaluate_scores :1019-1030), UMI collapse (parse_scores :1041-1109) and the
 * per-(row, cell) count histogram; results stay in HBM.  Synchronous: returns
 * after the device finished.  Replaces evaluate_chunk + the scoring half of
 * the merge loop.                                                            */
int vtx_run(vtx_ctx* ctx);

/* Copy the per-record scores of the last vtx_run to host arrays of n_records
 * int32 each (Scores.ref_score / alt_score, src/main.rs:926-927).            */
int vtx_fetch_scores(vtx_ctx* ctx, int32_t* ref_score, int32_t* alt_score);

/* Copy the triplets of the last vtx_run to context-owned host memory.        */
int vtx_fetch_coo(vtx_ctx* ctx, vtx_coo* out);

/* Device pointers of the last vtx_run's per-record scores (for callers that
 * keep results on the GPU, e.g. an RCCL gather).  int32[n_records] each.     */
int vtx_device_scores(vtx_ctx* ctx, const int32_t** d_ref, const int32_t** d_alt);

/* sprs::io::write_matrix_market of the last vtx_run's triplets (src/main.rs:381-389: three header lines, then "row+1 col+1 value"
 * per triplet in insertion order) — formatted on the device and streamed into `path` by the copy workers; the triplets never become
 * host arrays.  which: 0 = `value` (the matrix), 1 = `ref_value` (coverage mode's ref matrix, :385-389).  Integral values only
 * (consensus 1 / 2 / 3, coverage counts: Rust's `{}` of such an f64 is its digits); alt_frac's fractions / NaN return
 * VTX_E_UNSUPPORTED and leave nothing at `path` — format those on the host (vtx_fetch_coo + vtxh_write_mtx: shortest round-trip
 * digits).  *sum (optional): the sum of the values written (the reference's "sum of 0" warning, :410-415).                      */
int vtx_write_mtx(vtx_ctx* ctx, const char* path, uint32_t n_rows, uint32_t n_cols, int which, double* sum);

/* Device pointers of the last vtx_run's triplets (same layout as vtx_coo, all
 * arrays resident in HBM) — the payload of the multi-GPU row gather.          */
int vtx_device_coo(vtx_ctx* ctx, vtx_coo* out);

/* ---- Multi-GPU: one process (one ctx) per GPU, loci sharded over the ranks (src/main.rs:284-291: chunks of loci
 * are independent), and ONE exchange at the end — every rank's triplets travel to rank `dst` over RCCL (xGMI inside
 * a node), concatenated in rank order, which is row order when rank r holds the r-th contiguous range of loci
 * (the merge loop's order, src/main.rs:320-348).
 *   vtx_comm_id      rank 0 only: fills the 128-byte RCCL unique id; the host distributes it to the other ranks by
 *                    its own means (MPI, a socket, a file — it is plain bytes).
 *   vtx_comm_init    every rank, same id: joins the communicator (collective).
 *   vtx_gather_coo   every rank, after its own vtx_run (collective).  The counts go round with one all-gather of a
 *                    (count, status) pair per rank — any rank in error makes EVERY rank return VTX_E_PEER before the data
 *                    exchange, so nobody waits for a peer that left; then each rank sends its (row, col, alt, ref, unk) arrays to `dst`, which
 *                    receives every block at its final offset (grouped point-to-point: no padding, each block crosses
 *                    its own link once) and recomputes the f64 values from the counts (the same arithmetic as
 *                    vtx_run's emit step).  On `dst`, *out holds DEVICE pointers to the gathered arrays (valid until
 *                    the next vtx_gather_coo / vtx_destroy); on the other ranks out->nnz = 0.
 *   vtx_fetch_gathered  `dst` only: copies the gathered triplets to context-owned host memory.
 * librccl is opened on first use (dlopen): a single-GPU process never loads it.                                  */
#define VTX_COMM_ID_BYTES 128
int vtx_comm_id(uint8_t id[VTX_COMM_ID_BYTES]);
int vtx_comm_init(vtx_ctx* ctx, const uint8_t id[VTX_COMM_ID_BYTES], int rank, int world);
int vtx_gather_coo(vtx_ctx* ctx, int dst, vtx_coo* out);
int vtx_fetch_gathered(vtx_ctx* ctx, vtx_coo* out);
/* The number of ranks the communicator itself reports (ncclCommCount) — a cross-check of the launcher's world size. */
int vtx_comm_ranks(vtx_ctx* ctx, int* ranks);
/* A rank whose own work failed after vtx_comm_init (vtx_submit / vtx_run error) calls this INSTEAD of vtx_gather_coo: it takes
 * part in the status round, and every other rank's vtx_gather_coo returns VTX_E_PEER instead of waiting for ever.          */
int vtx_gather_abort(vtx_ctx* ctx);
/* The layout of the exchange as a pure function (no device): offsets[r] = where rank r's counts[r] triplets land, *total =
 * their sum; VTX_E_UNSUPPORTED when the total exceeds 2^32 - 1.  vtx_gather_coo follows exactly this plan.                  */
int vtx_gather_plan(int world, const uint64_t* counts, uint64_t* offsets, uint64_t* total);

int vtx_last_timing(vtx_ctx* ctx, vtx_timing* out);

/* ---- audit / test hooks: none changes a result, none is needed in production --------------------------------------------
 * vtx_set_debug(ctx, VTX_DEBUG_STAGE_TRACE, 0 | 1): the next vtx_run records, one byte per task (task = 2 * record + haplotype,
 *   0 = REF), which stage decided that alignment's score (Scores.ref_score / alt_score, src/main.rs:926-927): a certificate
 *   (no DP cell touched), the full-matrix check of a certificate, or a DP.  vtx_fetch_stage copies the 2 * n_records bytes.
 *   The invariant it lets a test assert over EVERY alignment: a banded score that differs from the full-matrix score was
 *   produced by a DP stage, never by a certificate.
 * vtx_set_debug(ctx, VTX_DEBUG_POISON_SCORES, 1) / (.., VTX_DEBUG_POISON_VALUE, v): every later vtx_run first fills both score
 *   arrays with v, so that a stage that fails to write a task's score cannot hide behind the previous run's value.
 * vtx_debug_bands: the band (banded::Aligner's Band: per column of the DP matrix the row range [lo, hi), columns 0 .. hap_len,
 *   rows 0 .. read_len) band_sweep_kernel builds for n_tasks tasks of the resident batch — lo / hi hold `stride` uint16 per
 *   task (stride >= longest haplotype + 1; empty column: lo 0x7fff, hi 0); status[i] = 0 band written, != 0 declined (the
 *   general kernel takes such a task in vtx_run).  Parity of the BAND, not only of the score it leads to.
 * vtx_debug_tables: the haplotype k-mer tables (the device form of the reference's per-haplotype hash index, bio 0.30.0
 *   sparse::hash_kmers as called from banded::Aligner::local, src/main.rs:898-901) the last banded vtx_run left in global
 *   memory: *bytes = their size (0: that run kept its tables in LDS), min(cap, *bytes) bytes are copied to dst (dst may be
 *   null with cap 0).  Lets a test compare two ways of BUILDING the tables byte for byte (VTX_BAND_TABLES_V1).           */
#define VTX_STAGE_UNKNOWN 0        /* band_run_kernel's / band_pending_kernel's certificate (cert == ub over all pieces: chains over several diagonals) */
#define VTX_STAGE_DIAG_CERT 1      /* band_diag_kernel: cert == ub */
#define VTX_STAGE_REFINE_CERT 2    /* band_refine_kernel: cert == refined ub */
#define VTX_STAGE_FULL_CHECK 3     /* full-matrix score == certificate (cert <= banded <= full); experiment hook VTX_BAND_CHECK only */
#define VTX_STAGE_SWEEP_DP 4       /* band_sweep_kernel's band + band-masked DP */
#define VTX_STAGE_GENERAL_DP 5     /* general band kernel (tasks the sweep declined) + band-masked DP */
#define VTX_STAGE_RUN_DP 6         /* VTX_BAND_LEGACY path: band_run_kernel's staircase + band-masked DP */
#define VTX_STAGE_DIAG_DP 7        /* the certificate stages' one-diagonal band (every off-diagonal match harmless) + band-masked DP */
#define VTX_STAGE_SLOW 8           /* slow_align_kernel (records beyond the fast kernels' limits): exact DP */
#define VTX_STAGE_FULL_DP 9        /* full flavour: the full-matrix DP */
#define VTX_STAGE_BAND_CERT 10     /* band_refine_kernel: cert == the run bound over the pieces trimmed to the one-diagonal band — a bound of the BANDED
                                      score (vtx_band_trim.h): like the DP stages, this one may decide a task whose banded score is below the full one */
#define VTX_STAGE_CORRIDOR_CERT 11 /* band_corridor_kernel (round 6): an exact DP over the cells of the one-diagonal band within 8 diagonals of the main one,
                                      with edges that bound every way around them; decided when the maximum needs none of those edges.  A bound of
                                      the BANDED score as well */
#define VTX_DEBUG_STAGE_TRACE 1
#define VTX_DEBUG_POISON_SCORES 2
#define VTX_DEBUG_POISON_VALUE 3
int vtx_set_debug(vtx_ctx* ctx, int key, int64_t value);
int vtx_fetch_stage(vtx_ctx* ctx, uint8_t* stage);
int vtx_debug_bands(vtx_ctx* ctx, const uint32_t* tasks, uint32_t n_tasks, uint32_t stride, uint16_t* lo, uint16_t* hi, uint8_t* status);
int vtx_debug_tables(vtx_ctx* ctx, void* dst, uint64_t cap, uint64_t* bytes);

/* Number of DP cells the last vtx_run evaluated (sum over records and both
 * haplotypes of rows x columns actually computed) — the roofline numerator.  */
int vtx_last_cells(vtx_ctx* ctx, uint64_t* cells);

/* Message for the last failing call on ctx (ctx may be NULL for create).     */
const char* vtx_strerror(const vtx_ctx* ctx);
const char* vtx_status_name(int status);

/* sizeof() of {vtx_config, vtx_locus, vtx_record, vtx_batch, vtx_coo,
 * vtx_timing, vtx_raw_record, vtx_raw_batch, vtx_raw_stats, vtx_bgzf_block, vtx_bam_interval,
 * vtx_bam_ingest, vtx_ingest_stats} as compiled into the library, for binding self-checks.
 * Writes min(n, 13) entries; returns VTX_ABI_VERSION.                        */
int vtx_abi_sizes(uint32_t* out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* VTX_H */
