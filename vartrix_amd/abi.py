"""ctypes mirror of ``include/vtx.h`` — the C-ABI boundary of the hot path.

Field order and widths must match the header exactly; ``tests/test_abi.py``
checks the struct sizes against ``vtx_abi_sizes()`` exported by the library.
The packed batch mirrors what a reference worker receives in
``evaluate_chunk`` (reference ``src/main.rs:596-607``) after read filtering.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

VTX_ABI_VERSION = 6
READS_BYTES, READS_NIBBLES = 0, 1          # vtx_set_read_format

VTX_OK = 0
VTX_E_INVAL = -1
VTX_E_NODEVICE = -2
VTX_E_HIP = -3
VTX_E_NOMEM = -4
VTX_E_UNSUPPORTED = -5
VTX_E_STATE = -6
VTX_E_PEER = -7

ALIGNER_BANDED = 0
ALIGNER_FULL = 1
ALIGNERS = {"banded": ALIGNER_BANDED, "full": ALIGNER_FULL}

MODE_CONSENSUS = 0
MODE_ALT_FRAC = 1
MODE_COVERAGE = 2
# --scoring-method values of the reference CLI, src/main.rs:89-94
MODES = {"consensus": MODE_CONSENSUS, "alt_frac": MODE_ALT_FRAC, "coverage": MODE_COVERAGE}


class VtxConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("device", C.c_int32),
        ("aligner", C.c_int32),
        ("scoring_mode", C.c_int32),
        ("use_umi", C.c_int32),
        ("match_score", C.c_int32),
        ("mismatch_score", C.c_int32),
        ("gap_open", C.c_int32),
        ("gap_extend", C.c_int32),
        ("min_score", C.c_int32),
        ("kmer_k", C.c_int32),
        ("band_w", C.c_int32),
        ("n_barcodes", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


def default_config(**overrides) -> VtxConfig:
    """Reference constants, src/main.rs:27-38 (same as vtx_config_default)."""
    cfg = VtxConfig(
        abi_version=VTX_ABI_VERSION, device=0, aligner=ALIGNER_BANDED,
        scoring_mode=MODE_CONSENSUS, use_umi=0, match_score=1, mismatch_score=-5,
        gap_open=-5, gap_extend=-1, min_score=25, kmer_k=6, band_w=20,
        n_barcodes=0, reserved=0)
    for k, v in overrides.items():
        if k == "aligner" and isinstance(v, str):
            v = ALIGNERS[v]
        if k == "scoring_mode" and isinstance(v, str):
            v = MODES[v]
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, int(v))
    return cfg


LOCUS_DTYPE = np.dtype([
    ("row", "<u4"), ("rec_begin", "<u4"), ("rec_count", "<u4"),
    ("ref_off", "<u4"), ("ref_len", "<u4"), ("alt_off", "<u4"), ("alt_len", "<u4"),
    ("reserved", "<u4")])
RECORD_DTYPE = np.dtype([
    ("read_off", "<u4"), ("read_len", "<u4"), ("cell_index", "<u4"), ("umi_id", "<u4")])
assert LOCUS_DTYPE.itemsize == 32 and RECORD_DTYPE.itemsize == 16


class VtxBatch(C.Structure):
    _fields_ = [
        ("loci", C.c_void_p),
        ("n_loci", C.c_uint32),
        ("records", C.c_void_p),
        ("n_records", C.c_uint32),
        ("hap_arena", C.c_void_p),
        ("hap_bytes", C.c_uint64),
        ("read_arena", C.c_void_p),
        ("read_bytes", C.c_uint64),
    ]


class VtxCoo(C.Structure):
    _fields_ = [
        ("row", C.POINTER(C.c_uint32)),
        ("col", C.POINTER(C.c_uint32)),
        ("alt", C.POINTER(C.c_uint32)),
        ("ref", C.POINTER(C.c_uint32)),
        ("unk", C.POINTER(C.c_uint32)),
        ("value", C.POINTER(C.c_double)),
        ("ref_value", C.POINTER(C.c_double)),
        ("nnz", C.c_uint64),
    ]


class VtxTiming(C.Structure):
    _fields_ = [
        ("total_ms", C.c_float),
        ("sw_ms", C.c_float),
        ("reduce_ms", C.c_float),
        ("sw_launches", C.c_uint32),
        ("hard_tasks", C.c_uint32),
        ("full_ms", C.c_float),
        ("band_ms", C.c_float),
        ("band_run_ms", C.c_float),
        ("overflow_tasks", C.c_uint32),
        ("diag_ms", C.c_float),
        ("diag_left", C.c_uint32),
        ("check_ms", C.c_float),
        ("sweep_ms", C.c_float),
        ("checked_tasks", C.c_uint32),
        ("swept_tasks", C.c_uint32),
        ("resweep_tasks", C.c_uint32),
        ("diag2_tasks", C.c_uint32),
        ("diag2_scored", C.c_uint32),
        ("diag2_streamed", C.c_uint32),
    ]


# vtx_fetch_stage: which stage decided an alignment's score (include/vtx.h)
STAGE_UNKNOWN, STAGE_DIAG_CERT, STAGE_REFINE_CERT, STAGE_FULL_CHECK, STAGE_SWEEP_DP, STAGE_GENERAL_DP, STAGE_RUN_DP = range(7)
STAGE_DIAG_DP, STAGE_SLOW, STAGE_FULL_DP, STAGE_BAND_CERT, STAGE_CORRIDOR_CERT = 7, 8, 9, 10, 11
STAGE_NAMES = {0: "band_run certificate", 1: "diag certificate", 2: "refine certificate", 3: "full-matrix check", 4: "sweep + masked DP",
               5: "general kernel + masked DP", 6: "band_run + masked DP", 7: "diagonal band + masked DP", 8: "slow path", 9: "full DP",
               10: "band-restricted certificate", 11: "corridor certificate"}
DP_STAGES = (STAGE_SWEEP_DP, STAGE_GENERAL_DP, STAGE_RUN_DP, STAGE_DIAG_DP, STAGE_SLOW, STAGE_FULL_DP)
# stages that may decide an alignment whose banded score is BELOW the full-matrix one: the DPs, and the certificate against the
# bounds of the banded score (vtx_band_trim.h; the corridor certificate of vtx_fast_core.h)
BANDED_STAGES = DP_STAGES + (STAGE_BAND_CERT, STAGE_CORRIDOR_CERT)
DEBUG_STAGE_TRACE, DEBUG_POISON_SCORES, DEBUG_POISON_VALUE = 1, 2, 3


TAG_MISSING = 0xFFFF
RAW_RECORD_DTYPE = np.dtype([
    ("read_off", "<u4"), ("read_len", "<u4"), ("bc_off", "<u4"), ("umi_off", "<u4"),
    ("bc_len", "<u2"), ("umi_len", "<u2")])
assert RAW_RECORD_DTYPE.itemsize == 20


class VtxRawBatch(C.Structure):
    _fields_ = [
        ("loci", C.c_void_p),
        ("n_loci", C.c_uint32),
        ("records", C.c_void_p),
        ("n_records", C.c_uint32),
        ("hap_arena", C.c_void_p),
        ("hap_bytes", C.c_uint64),
        ("read_arena", C.c_void_p),
        ("read_bytes", C.c_uint64),
        ("tag_arena", C.c_void_p),
        ("tag_bytes", C.c_uint64),
    ]


class VtxRawStats(C.Structure):
    _fields_ = [
        ("num_not_cell_bc", C.c_uint64),
        ("num_non_umi", C.c_uint64),
        ("kept", C.c_uint64),
        ("prep_ms", C.c_float),
        ("hash_rounds", C.c_uint32),
    ]


# ---- vtx_submit_bam (device-side ingest) ----
BGZF_BLOCK_DTYPE = np.dtype([("coff", "<u8"), ("clen", "<u4"), ("isize", "<u4")])
BAM_INTERVAL_DTYPE = np.dtype([("start", "<i4"), ("end", "<i4"), ("locus", "<u4"), ("reserved", "<u4")])
assert BGZF_BLOCK_DTYPE.itemsize == 16 and BAM_INTERVAL_DTYPE.itemsize == 16
INGEST_INFLATED, INGEST_RECORD_OFFSETS, INGEST_RAW_RECORDS, INGEST_RAW_LOCUS, INGEST_TAGS, INGEST_READS_PACKED = range(6)


class VtxBamIngest(C.Structure):
    _fields_ = [
        ("file", C.c_void_p), ("file_bytes", C.c_uint64),
        ("blocks", C.c_void_p), ("n_blocks", C.c_uint32), ("n_ref", C.c_uint32),
        ("seeds", C.c_void_p), ("n_seeds", C.c_uint32), ("n_intervals", C.c_uint32),
        ("end_upos", C.c_uint64),
        ("intervals", C.c_void_p), ("tid_begin", C.c_void_p), ("tid_max_span", C.c_void_p),
        ("loci", C.c_void_p), ("n_loci", C.c_uint32), ("min_mapq", C.c_uint32),
        ("primary_only", C.c_int32), ("no_duplicates", C.c_int32),
        ("hap_arena", C.c_void_p), ("hap_bytes", C.c_uint64),
        ("bam_tag", C.c_char * 2), ("reserved", C.c_char * 6),
    ]


class VtxIngestStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("num_reads", "num_low_mapq", "num_non_primary", "num_duplicates", "num_not_useful",
                                          "num_no_barcode_tag", "bam_records", "raw_records", "compressed_bytes", "inflated_bytes")] + \
               [("raw", VtxRawStats)] + [(n, C.c_float) for n in ("h2d_ms", "inflate_ms", "index_ms", "filter_ms", "prefetch_ms", "prefetch_wait_ms")]


def pack_nibbles(arena: np.ndarray) -> np.ndarray:
    """One ASCII byte per base -> two bases per byte, high nibble first, the BAM's code "=ACMGRSVTWYHKDBN" (SAM spec 4.2.3)."""
    code = np.full(256, 255, np.uint8)
    for k, ch in enumerate(b"=ACMGRSVTWYHKDBN"):
        code[ch] = k
    a = np.ascontiguousarray(arena, np.uint8)
    if a.size & 1:
        a = np.concatenate([a, np.frombuffer(b"=", np.uint8)])
    c = code[a]
    if (c == 255).any():
        raise ValueError("pack_nibbles: a base outside =ACMGRSVTWYHKDBN")
    return (c[0::2] << 4) | c[1::2]


@dataclass
class RawBatch:
    """Host-side raw batch — the payload of ``vtx_submit_raw``: reads that passed the alignment-level
    filters (src/main.rs:833-864), in any order inside a locus, barcode / UMI still as tag bytes."""
    loci: np.ndarray
    records: np.ndarray
    hap_arena: np.ndarray
    read_arena: np.ndarray
    tag_arena: np.ndarray
    read_format: int = 0          # READS_BYTES | READS_NIBBLES (vtx_set_read_format): read_arena then holds two bases per byte

    def __post_init__(self):
        self.loci = np.ascontiguousarray(self.loci, dtype=LOCUS_DTYPE)
        self.records = np.ascontiguousarray(self.records, dtype=RAW_RECORD_DTYPE)
        self.hap_arena = np.ascontiguousarray(self.hap_arena, dtype=np.uint8)
        self.read_arena = np.ascontiguousarray(self.read_arena, dtype=np.uint8)
        self.tag_arena = np.ascontiguousarray(self.tag_arena, dtype=np.uint8)

    @property
    def n_loci(self) -> int:
        return int(self.loci.shape[0])

    @property
    def n_records(self) -> int:
        return int(self.records.shape[0])

    def as_struct(self) -> VtxRawBatch:
        def ptr(a):
            return a.ctypes.data if a.size else None
        return VtxRawBatch(
            loci=ptr(self.loci), n_loci=self.n_loci, records=ptr(self.records), n_records=self.n_records,
            hap_arena=ptr(self.hap_arena), hap_bytes=int(self.hap_arena.size),
            read_arena=ptr(self.read_arena), read_bytes=int(self.read_arena.size) * (2 if self.read_format == READS_NIBBLES else 1),
            tag_arena=ptr(self.tag_arena), tag_bytes=int(self.tag_arena.size))


@dataclass
class PackedBatch:
    """Host-side packed batch (numpy) — the payload of ``vtx_submit``.

    ``loci``/``records`` use LOCUS_DTYPE / RECORD_DTYPE; arenas are uint8.
    Records of a locus are contiguous and ordered by (cell_index, umi_id),
    which is the reference's stable sort by cell (src/main.rs:932) refined by
    UMI (the per-cell HashMap of src/main.rs:1047-1057 is order-free).
    """
    loci: np.ndarray
    records: np.ndarray
    hap_arena: np.ndarray
    read_arena: np.ndarray
    read_format: int = 0          # READS_BYTES | READS_NIBBLES (vtx_set_read_format): read_arena then holds two bases per byte

    def __post_init__(self):
        self.loci = np.ascontiguousarray(self.loci, dtype=LOCUS_DTYPE)
        self.records = np.ascontiguousarray(self.records, dtype=RECORD_DTYPE)
        self.hap_arena = np.ascontiguousarray(self.hap_arena, dtype=np.uint8)
        self.read_arena = np.ascontiguousarray(self.read_arena, dtype=np.uint8)

    @property
    def n_loci(self) -> int:
        return int(self.loci.shape[0])

    @property
    def n_records(self) -> int:
        return int(self.records.shape[0])

    def as_struct(self) -> VtxBatch:
        def ptr(a):
            return a.ctypes.data if a.size else None
        return VtxBatch(
            loci=ptr(self.loci), n_loci=self.n_loci,
            records=ptr(self.records), n_records=self.n_records,
            hap_arena=ptr(self.hap_arena), hap_bytes=int(self.hap_arena.size),
            read_arena=ptr(self.read_arena), read_bytes=int(self.read_arena.size) * (2 if self.read_format == READS_NIBBLES else 1))

    def to_nibbles(self) -> "PackedBatch":
        """The same batch with its read arena as the BAM holds bases: two per byte, high nibble first (READS_NIBBLES).  Every
        read must start at an even offset (the packer lays them out so; the synthetic generators do when read lengths are even)."""
        if self.read_format == READS_NIBBLES:
            return self
        if self.n_records and (self.records["read_off"] & 1).any():
            raise ValueError("to_nibbles: a read starts at an odd offset")
        return PackedBatch(self.loci, self.records, self.hap_arena, pack_nibbles(self.read_arena), READS_NIBBLES)

    def to_bytes(self) -> "PackedBatch":
        """Inverse of to_nibbles (what unpack_nibbles_kernel writes on the device)."""
        if self.read_format != READS_NIBBLES:
            return self
        lut = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)
        out = np.empty(2 * self.read_arena.size, np.uint8)
        out[0::2] = lut[self.read_arena >> 4]
        out[1::2] = lut[self.read_arena & 15]
        return PackedBatch(self.loci, self.records, self.hap_arena, out, READS_BYTES)

    @staticmethod
    def concat(parts) -> "PackedBatch":
        """The loci of several self-contained batches one after the other (rows renumbered 0, 1, ...): how tests put ONE unusual
        locus into the middle of an ordinary batch.  Byte-per-base arenas only."""
        parts = [p for p in parts if p.n_loci]
        assert all(p.read_format == READS_BYTES for p in parts)
        loci, recs = [], []
        r0 = h0 = a0 = 0
        for p in parts:
            l, r = p.loci.copy(), p.records.copy()
            l["rec_begin"] += r0
            l["ref_off"] += h0
            l["alt_off"] += h0
            r["read_off"] += a0
            loci.append(l)
            recs.append(r)
            r0 += p.n_records
            h0 += int(p.hap_arena.size)
            a0 += int(p.read_arena.size)
        out = PackedBatch(np.concatenate(loci), np.concatenate(recs), np.concatenate([p.hap_arena for p in parts]),
                          np.concatenate([p.read_arena for p in parts]))
        out.loci["row"] = np.arange(out.n_loci, dtype=np.uint32)
        return out

    def slice_loci(self, lo: int, hi: int) -> "PackedBatch":
        """Contiguous sub-batch [lo, hi) of loci, re-based so it is self-contained.

        This is how loci shard across GPUs (rows are independent,
        src/main.rs:284-291): each rank receives only its own loci, records,
        read bases and haplotypes.
        """
        loci = self.loci[lo:hi].copy()
        if loci.shape[0] == 0:
            return PackedBatch(loci, self.records[:0], self.hap_arena[:0], self.read_arena[:0], self.read_format)
        r0 = int(loci["rec_begin"][0])
        r1 = int(loci["rec_begin"][-1] + loci["rec_count"][-1])
        recs = self.records[r0:r1].copy()
        loci["rec_begin"] -= r0
        # haplotypes: contiguous span covering the slice
        h0 = int(min(loci["ref_off"].min(), loci["alt_off"].min()))
        h1 = int(max((loci["ref_off"] + loci["ref_len"]).max(), (loci["alt_off"] + loci["alt_len"]).max()))
        loci["ref_off"] -= h0
        loci["alt_off"] -= h0
        haps = self.hap_arena[h0:h1].copy()
        if recs.shape[0]:
            a0 = int(recs["read_off"].min())
            a1 = int((recs["read_off"] + recs["read_len"]).max())
            recs["read_off"] -= a0
            if self.read_format == READS_NIBBLES:                # (offsets are even: a0 is)
                reads = self.read_arena[a0 // 2:(a1 + 1) // 2].copy()
            else:
                reads = self.read_arena[a0:a1].copy()
        else:
            reads = self.read_arena[:0]
        return PackedBatch(loci, recs, haps, reads, self.read_format)
