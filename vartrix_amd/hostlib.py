"""ctypes binding of libvtxhost.so (include/vtx_host.h): ingest + read filters +
haplotype construction -> packed batch.  CPU-only code; the packed batch then goes
to ``vartrix_amd.lib.Context.submit``."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# VTX_LIB_VARIANT=dev: the developer build of the same sources with the test hooks (VTXH_BATCH_BYTES, VTXH_CHUNK_BLOCKS, ...) compiled
# in (make dev); the production library and CLI read VTXH_PROFILE only
LIB_PATH = CLI_PATH = None


def cli_path(variant="") -> str:
    return os.path.join(_HERE, "bin", "vartrix" + ("_dev" if variant == "dev" else ""))


def use_variant(variant=""):
    """Select the production ("") or developer ("dev") build for the NEXT load() (tests/test_host.py switches to dev for its hooks)."""
    global LIB_PATH, CLI_PATH, _lib
    LIB_PATH = os.path.join(_HERE, "libvtxhost%s.so" % ("_dev" if variant == "dev" else ""))
    CLI_PATH = cli_path(variant)
    _lib = None



SYMBOLS = ("vtxh_pack_files", "vtxh_free", "vtxh_last_error", "vtxh_get_batch", "vtxh_get_metrics", "vtxh_get_ingest_stats",
           "vtxh_num_variants", "vtxh_num_barcodes", "vtxh_variant_name", "vtxh_barcode", "vtxh_write_mtx",
           "vtxh_format_f64", "vtxh_pack_files_raw", "vtxh_get_raw_batch", "vtxh_get_barcode_table", "vtxh_num_batches",
           "vtxh_get_batch_at", "vtxh_get_raw_batch_at", "vtxh_pack_files_range", "vtxh_test_inflate", "vtxh_read_format",
           "vtxh_trim", "vtxh_plan_ingest", "vtxh_get_ingest", "vtxh_is_plan")
METRIC_NAMES = ("num_reads", "num_low_mapq", "num_non_primary", "num_duplicates", "num_not_cell_bc",
                "num_not_useful", "num_non_umi", "num_invalid_recs", "num_multiallelic_recs")


class VtxhArgs(C.Structure):
    _fields_ = [("vcf", C.c_char_p), ("bam", C.c_char_p), ("fasta", C.c_char_p), ("cell_barcodes", C.c_char_p),
                ("padding", C.c_uint32), ("mapq", C.c_uint32), ("primary_only", C.c_int32),
                ("no_duplicates", C.c_int32), ("use_umi", C.c_int32), ("bam_tag", C.c_char_p),
                ("valid_chars", C.c_char_p), ("threads", C.c_int32), ("read_format", C.c_int32)]


class VtxhMetrics(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in METRIC_NAMES]


_lib = None
use_variant("dev" if os.environ.get("VTX_LIB_VARIANT") == "dev" else "")


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built (make -C vartrix_amd/csrc all)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.vtxh_pack_files.restype = C.c_int
        L.vtxh_pack_files.argtypes = [C.POINTER(VtxhArgs), C.POINTER(C.c_void_p)]
        L.vtxh_pack_files_raw.restype = C.c_int
        L.vtxh_pack_files_raw.argtypes = [C.POINTER(VtxhArgs), C.POINTER(C.c_void_p)]
        L.vtxh_pack_files_range.restype = C.c_int
        L.vtxh_pack_files_range.argtypes = [C.POINTER(VtxhArgs), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.vtxh_get_raw_batch.argtypes = [C.c_void_p, C.POINTER(abi.VtxRawBatch)]
        L.vtxh_get_barcode_table.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.vtxh_num_batches.restype = C.c_uint32
        L.vtxh_num_batches.argtypes = [C.c_void_p]
        L.vtxh_get_batch_at.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(abi.VtxBatch)]
        L.vtxh_get_raw_batch_at.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(abi.VtxRawBatch)]
        L.vtxh_test_inflate.restype = C.c_int
        L.vtxh_test_inflate.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.vtxh_read_format.restype = C.c_int
        L.vtxh_read_format.argtypes = [C.c_void_p]
        L.vtxh_free.argtypes = [C.c_void_p]
        L.vtxh_last_error.restype = C.c_char_p
        L.vtxh_get_batch.argtypes = [C.c_void_p, C.POINTER(abi.VtxBatch)]
        L.vtxh_get_metrics.argtypes = [C.c_void_p, C.POINTER(VtxhMetrics)]
        L.vtxh_num_variants.restype = C.c_uint32
        L.vtxh_num_variants.argtypes = [C.c_void_p]
        L.vtxh_num_barcodes.restype = C.c_uint32
        L.vtxh_num_barcodes.argtypes = [C.c_void_p]
        L.vtxh_variant_name.restype = C.c_char_p
        L.vtxh_variant_name.argtypes = [C.c_void_p, C.c_uint32]
        L.vtxh_barcode.restype = C.c_char_p
        L.vtxh_barcode.argtypes = [C.c_void_p, C.c_uint32]
        L.vtxh_write_mtx.restype = C.c_int
        L.vtxh_write_mtx.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vtxh_format_f64.restype = C.c_int
        L.vtxh_format_f64.argtypes = [C.c_double, C.c_char_p]
        L.vtxh_plan_ingest.restype = C.c_int
        L.vtxh_plan_ingest.argtypes = [C.POINTER(VtxhArgs), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.vtxh_get_ingest.restype = C.c_int
        L.vtxh_get_ingest.argtypes = [C.c_void_p, C.POINTER(abi.VtxBamIngest)]
        L.vtxh_is_plan.restype = C.c_int
        L.vtxh_is_plan.argtypes = [C.c_void_p]
        L.vtxh_get_ingest_stats.restype = None
        L.vtxh_get_ingest_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 3)]
        _lib = L
    return _lib


last_ingest_stats = {}        # of the most recent pack_files call: BGZF blocks inflated / in the file, index-guided jumps


class HostError(RuntimeError):
    pass


def pack_files(vcf, bam, fasta, cell_barcodes, padding=100, mapq=0, primary_only=False, no_duplicates=False,
               use_umi=False, bam_tag="CB", valid_chars="ATGCatgc", threads=1, raw=False, all_batches=False, rows=None,
               nibbles=False):
    """-> (PackedBatch, metrics dict, n_variants, barcodes list, variant names); with ``raw`` the batch is a
    RawBatch for ``Context.submit_raw`` (vtxh_pack_files_raw: tags as bytes, BAM order inside a locus).
    A pack whose reads span more than 4 GiB comes as several batches: ``all_batches`` returns the list of them
    (loci keep their global row; triplets of the batches are appended in order)."""
    L = load()
    args = VtxhArgs(vcf.encode(), bam.encode(), fasta.encode(), cell_barcodes.encode(), padding, mapq,
                    int(primary_only), int(no_duplicates), int(use_umi), bam_tag.encode(), valid_chars.encode(), threads,
                    abi.READS_NIBBLES if nibbles else abi.READS_BYTES)
    h = C.c_void_p()
    if rows is not None:      # streaming: VCF records [rows[0], rows[1]) only (vtxh_pack_files_range)
        rc = L.vtxh_pack_files_range(C.byref(args), int(raw), int(rows[0]), int(rows[1]), C.byref(h))
    else:
        rc = (L.vtxh_pack_files_raw if raw else L.vtxh_pack_files)(C.byref(args), C.byref(h))
    if rc != 0:
        raise HostError(L.vtxh_last_error().decode())
    try:
        def arr(ptr, n, dt):
            if not n:
                return np.zeros(0, dt)
            return np.frombuffer(C.string_at(ptr, n * np.dtype(dt).itemsize), dtype=dt).copy()
        nb_batches = L.vtxh_num_batches(h)
        fmt = int(L.vtxh_read_format(h))
        rdiv = 2 if fmt == abi.READS_NIBBLES else 1
        if nb_batches != 1 and not all_batches:
            raise HostError("the pack has %d batches: call pack_files(..., all_batches=True)" % nb_batches)
        batches = []
        for i in range(nb_batches):
            if raw:
                b = abi.VtxRawBatch()
                L.vtxh_get_raw_batch_at(h, i, C.byref(b))
                batches.append(abi.RawBatch(arr(b.loci, b.n_loci, abi.LOCUS_DTYPE), arr(b.records, b.n_records, abi.RAW_RECORD_DTYPE),
                                            arr(b.hap_arena, b.hap_bytes, np.uint8), arr(b.read_arena, b.read_bytes // rdiv, np.uint8),
                                            arr(b.tag_arena, b.tag_bytes, np.uint8), fmt))
            else:
                b = abi.VtxBatch()
                L.vtxh_get_batch_at(h, i, C.byref(b))
                batches.append(abi.PackedBatch(arr(b.loci, b.n_loci, abi.LOCUS_DTYPE), arr(b.records, b.n_records, abi.RECORD_DTYPE),
                                               arr(b.hap_arena, b.hap_bytes, np.uint8), arr(b.read_arena, b.read_bytes // rdiv, np.uint8), fmt))
        batch = batches if all_batches else batches[0]
        m = VtxhMetrics()
        L.vtxh_get_metrics(h, C.byref(m))
        metrics = {n: int(getattr(m, n)) for n in METRIC_NAMES}
        st = (C.c_uint64 * 3)()
        L.vtxh_get_ingest_stats(h, C.byref(st))
        global last_ingest_stats
        last_ingest_stats = {"blocks_inflated": int(st[0]), "blocks_total": int(st[1]), "index_jumps": int(st[2])}
        nv, nb = L.vtxh_num_variants(h), L.vtxh_num_barcodes(h)
        barcodes = [L.vtxh_barcode(h, j) for j in range(nb)]
        variants = [L.vtxh_variant_name(h, i).decode() for i in range(nv)]
    finally:
        L.vtxh_free(h)
    return batch, metrics, nv, barcodes, variants


class IngestPlan:
    """The plan of a device-side ingest (vtxh_plan_ingest): ``ingest`` is the struct ``Context.submit_bam`` takes (its pointers live as
    long as this object), ``reason`` says why there is none.  Also the loci, the VCF-level metrics, the barcode list and the names."""

    def __init__(self, h, L):
        self._h, self._L = h, L
        self.ingest = abi.VtxBamIngest()
        rc = L.vtxh_get_ingest(h, C.byref(self.ingest))
        self.reason = None if rc == 0 else L.vtxh_last_error().decode()
        if rc != 0:
            self.ingest = None
        m = VtxhMetrics()
        L.vtxh_get_metrics(h, C.byref(m))
        self.metrics = {n: int(getattr(m, n)) for n in METRIC_NAMES}
        self.n_variants, nb = L.vtxh_num_variants(h), L.vtxh_num_barcodes(h)
        self.barcodes = [L.vtxh_barcode(h, j) for j in range(nb)]
        self.variants = [L.vtxh_variant_name(h, i).decode() for i in range(self.n_variants)]
        st = (C.c_uint64 * 3)()
        L.vtxh_get_ingest_stats(h, C.byref(st))
        self.blocks_planned, self.blocks_total = int(st[0]), int(st[1])

    @property
    def n_loci(self):
        return int(self.ingest.n_loci) if self.ingest is not None else 0

    def arrays(self):
        """numpy copies of the plan's arrays (tests)."""
        g = self.ingest

        def arr(ptr, n, dt):
            return np.frombuffer(C.string_at(ptr, n * np.dtype(dt).itemsize), dtype=dt).copy() if n else np.zeros(0, dt)
        return dict(blocks=arr(g.blocks, g.n_blocks, abi.BGZF_BLOCK_DTYPE), seeds=arr(g.seeds, g.n_seeds, np.uint64),
                    intervals=arr(g.intervals, g.n_intervals, abi.BAM_INTERVAL_DTYPE), tid_begin=arr(g.tid_begin, g.n_ref + 1, np.uint32),
                    tid_max_span=arr(g.tid_max_span, g.n_ref, np.int32), loci=arr(g.loci, g.n_loci, abi.LOCUS_DTYPE),
                    hap_arena=arr(g.hap_arena, g.hap_bytes, np.uint8), end_upos=int(g.end_upos))

    def close(self):
        if self._h:
            self._L.vtxh_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_ingest(vcf, bam, fasta, cell_barcodes, padding=100, mapq=0, primary_only=False, no_duplicates=False, use_umi=False,
                bam_tag="CB", valid_chars="ATGCatgc", threads=1, rows=None) -> IngestPlan:
    L = load()
    args = VtxhArgs(vcf.encode(), bam.encode(), fasta.encode(), cell_barcodes.encode(), padding, mapq, int(primary_only),
                    int(no_duplicates), int(use_umi), bam_tag.encode(), valid_chars.encode(), threads, abi.READS_NIBBLES)
    h = C.c_void_p()
    r0, r1 = (0, 0xFFFFFFFF) if rows is None else (int(rows[0]), int(rows[1]))
    if L.vtxh_plan_ingest(C.byref(args), r0, r1, C.byref(h)) != 0:
        raise HostError(L.vtxh_last_error().decode())
    return IngestPlan(h, L)


def write_mtx(path, n_rows, n_cols, row, col, value):
    row = np.ascontiguousarray(row, np.uint32)
    col = np.ascontiguousarray(col, np.uint32)
    value = np.ascontiguousarray(value, np.float64)
    rc = load().vtxh_write_mtx(path.encode(), n_rows, n_cols, len(row), row.ctypes.data, col.ctypes.data, value.ctypes.data)
    if rc != 0:
        raise HostError(load().vtxh_last_error().decode())


def format_f64(v: float) -> str:
    buf = C.create_string_buffer(40)
    n = load().vtxh_format_f64(v, buf)
    return buf.raw[:n].decode()
