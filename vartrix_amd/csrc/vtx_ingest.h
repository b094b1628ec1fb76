// vtx_ingest.h — internal declarations of the device-side BAM ingest (vtx_ingest.hip), shared with the C-ABI layer (vtx_api.hip).
#ifndef VTX_INGEST_H
#define VTX_INGEST_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vtx.h"

// a BGZF block on the device: offsets into the uploaded compressed range / the inflated buffer
struct vtxg_block { uint64_t coff, uoff; uint32_t clen, isize; };
// the read filters of evaluate_alns that need no dictionary (src/main.rs:833-864) + the tag to look for (--bam-tag, :126-129)
struct vtxg_filter { uint32_t n_ref, min_mapq, primary_only, no_duplicates, bam_tag; };
// per BAM record with at least one surviving pair: where its tags lie (relative to the record body) and their lengths
struct vtxg_recinfo { uint32_t bc_rel, umi_rel, lens; };

// counters[]: 0 num_reads, 1 num_low_mapq, 2 num_non_primary, 3 num_duplicates, 4 num_not_useful, 5 reads without a usable barcode tag
// (Metrics, src/main.rs:449-459), 6 read bases kept (padded to even), 7 tag bytes kept, 8 surviving (read, locus) pairs
#define VTXG_N_COUNTERS 9
// err[0] bits: 1 << vtxi::Status of a block that did not inflate (bits 1..8); err[1]: the first such block
#define VTXG_ERR_CHAIN (1u << 16)      // a record chain did not land on the next seed / ran off the data
#define VTXG_ERR_RECORD (1u << 17)     // a record whose fields run past its block_size

extern "C" {
hipError_t vtxg_inflate(const uint8_t* comp, const vtxg_block* blocks, uint32_t n_blocks, uint8_t* out, uint32_t* err, uint32_t* status, uint32_t b_base, hipStream_t s);
hipError_t vtxg_chain(const uint8_t* data, uint64_t total, const uint64_t* seeds, uint32_t n_seeds, uint64_t end_upos, uint32_t* cnt,
                      const uint32_t* off, uint64_t* rec_upos, uint32_t* err, hipStream_t s);
hipError_t vtxg_scan(int emit, const uint8_t* data, const uint64_t* rec_upos, uint32_t n_rec, vtxg_filter f, const int32_t* iv_start,
                     const int32_t* iv_end, const uint32_t* iv_locus, const uint32_t* tid_begin, const int32_t* tid_span,
                     uint32_t* n_hit, uint32_t* read_sz, uint32_t* tag_sz, vtxg_recinfo* info, const uint32_t* hit_scan,
                     const uint32_t* read_scan, const uint32_t* tag_scan, vtx_raw_record* raw, uint32_t* raw_locus, uint8_t* tags,
                     uint8_t* reads_packed, unsigned long long* counters, uint32_t* err, hipStream_t s);
// Matrix-Market lines of n triplets with integral values: byte length per line (0 + flag when a value is not a non-negative integer
// below 2^32), sum of the values; then the text at the inclusive scan `end` of the lengths
hipError_t vtxg_mtx_len(const uint32_t* row, const uint32_t* col, const double* val, uint32_t n, uint32_t* len, double* sum, uint32_t* flag, hipStream_t s);
hipError_t vtxg_mtx_text(const uint32_t* row, const uint32_t* col, const double* val, uint32_t n, const uint32_t* end, uint8_t* text, hipStream_t s);
}
#endif
