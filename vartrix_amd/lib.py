"""ctypes binding of libvtx.so — the host side of the C-ABI in ``include/vtx.h``.

There is no fallback: if the shared library is missing or no gfx950 device is
present, the calls raise.  The method names follow the C entry points, which in
turn name the reference seam they replace (``evaluate_chunk`` +
the merge loop, reference ``src/main.rs:596-607`` / ``:320-348``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# Build variants of the SAME sources (vartrix_amd/csrc/Makefile), never loaded in production:
#   dev    -DVTX_DEVTOOLS: the experiment / test hooks (stage switches, ablations, buffer caps, the socket transport standing in for
#          RCCL) exist only there; the production libvtx.so reads VTX_DEBUG and nothing else
#   lazy0, lazy40, anchor5, noseed0   one recollected detail of the crate's band switched (tests/test_gpu_variants.py)
# VTX_LIB_VARIANT=<name> makes a variant the process default; load(variant) / Context(cfg, variant=...) pick one explicitly.
DEFAULT_VARIANT = os.environ.get("VTX_LIB_VARIANT", "")


def lib_path(variant=None) -> str:
    v = DEFAULT_VARIANT if variant is None else variant
    return os.path.join(_HERE, "libvtx%s.so" % ("_" + v if v else ""))


LIB_PATH = lib_path()

SYMBOLS = (
    "vtx_config_default", "vtx_create", "vtx_destroy", "vtx_submit", "vtx_run", "vtx_fetch_scores",
    "vtx_fetch_coo", "vtx_device_scores", "vtx_device_coo", "vtx_last_timing", "vtx_last_cells", "vtx_strerror",
    "vtx_status_name", "vtx_abi_sizes", "vtx_set_barcodes", "vtx_submit_raw", "vtx_fetch_records",
    "vtx_comm_id", "vtx_comm_init", "vtx_gather_coo", "vtx_fetch_gathered", "vtx_gather_abort", "vtx_gather_plan",
    "vtx_set_debug", "vtx_fetch_stage", "vtx_debug_bands", "vtx_debug_tables", "vtx_set_read_format",
    "vtx_submit_bam", "vtx_debug_ingest", "vtx_debug_inflate", "vtx_comm_ranks", "vtx_write_mtx", "vtx_prefetch_file",
)


class VtxError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__("%s: %s" % (status_name(status), message))
        self.status = status


_libs = {}


def load(variant=None):
    """dlopen libvtx.so (built in-tree by ``__graft_entry__.build()``); raises if absent."""
    path = lib_path(variant)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError("%s not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C vartrix_amd/csrc). There is no CPU fallback." % path)
    L = C.CDLL(path)
    ctxp = C.c_void_p
    L.vtx_config_default.restype = None
    L.vtx_config_default.argtypes = [C.POINTER(abi.VtxConfig)]
    L.vtx_create.restype = C.c_int
    L.vtx_create.argtypes = [C.POINTER(abi.VtxConfig), C.POINTER(ctxp)]
    L.vtx_destroy.restype = None
    L.vtx_destroy.argtypes = [ctxp]
    L.vtx_submit.restype = C.c_int
    L.vtx_submit.argtypes = [ctxp, C.POINTER(abi.VtxBatch)]
    L.vtx_run.restype = C.c_int
    L.vtx_run.argtypes = [ctxp]
    L.vtx_set_barcodes.restype = C.c_int
    L.vtx_set_barcodes.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_uint32]
    L.vtx_submit_raw.restype = C.c_int
    L.vtx_submit_raw.argtypes = [ctxp, C.POINTER(abi.VtxRawBatch), C.POINTER(abi.VtxRawStats)]
    L.vtx_fetch_records.restype = C.c_int
    L.vtx_fetch_records.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vtx_fetch_scores.restype = C.c_int
    L.vtx_fetch_scores.argtypes = [ctxp, C.c_void_p, C.c_void_p]
    L.vtx_fetch_coo.restype = C.c_int
    L.vtx_fetch_coo.argtypes = [ctxp, C.POINTER(abi.VtxCoo)]
    L.vtx_device_scores.restype = C.c_int
    L.vtx_device_scores.argtypes = [ctxp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.vtx_device_coo.restype = C.c_int
    L.vtx_device_coo.argtypes = [ctxp, C.POINTER(abi.VtxCoo)]
    L.vtx_last_timing.restype = C.c_int
    L.vtx_last_timing.argtypes = [ctxp, C.POINTER(abi.VtxTiming)]
    L.vtx_last_cells.restype = C.c_int
    L.vtx_last_cells.argtypes = [ctxp, C.POINTER(C.c_uint64)]
    L.vtx_strerror.restype = C.c_char_p
    L.vtx_strerror.argtypes = [ctxp]
    L.vtx_status_name.restype = C.c_char_p
    L.vtx_status_name.argtypes = [C.c_int]
    L.vtx_abi_sizes.restype = C.c_int
    L.vtx_abi_sizes.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
    L.vtx_comm_id.restype = C.c_int
    L.vtx_comm_id.argtypes = [C.c_void_p]
    L.vtx_comm_init.restype = C.c_int
    L.vtx_comm_init.argtypes = [ctxp, C.c_void_p, C.c_int, C.c_int]
    L.vtx_gather_coo.restype = C.c_int
    L.vtx_gather_coo.argtypes = [ctxp, C.c_int, C.POINTER(abi.VtxCoo)]
    L.vtx_fetch_gathered.restype = C.c_int
    L.vtx_fetch_gathered.argtypes = [ctxp, C.POINTER(abi.VtxCoo)]
    L.vtx_gather_abort.restype = C.c_int
    L.vtx_gather_abort.argtypes = [ctxp]
    L.vtx_gather_plan.restype = C.c_int
    L.vtx_gather_plan.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.vtx_set_debug.restype = C.c_int
    L.vtx_set_debug.argtypes = [ctxp, C.c_int, C.c_int64]
    L.vtx_fetch_stage.restype = C.c_int
    L.vtx_fetch_stage.argtypes = [ctxp, C.c_void_p]
    L.vtx_set_read_format.restype = C.c_int
    L.vtx_set_read_format.argtypes = [ctxp, C.c_int]
    L.vtx_debug_tables.restype = C.c_int
    L.vtx_debug_tables.argtypes = [ctxp, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.vtx_debug_bands.restype = C.c_int
    L.vtx_debug_bands.argtypes = [ctxp, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vtx_submit_bam.restype = C.c_int
    L.vtx_submit_bam.argtypes = [ctxp, C.POINTER(abi.VtxBamIngest), C.POINTER(abi.VtxIngestStats)]
    L.vtx_prefetch_file.restype = C.c_int
    L.vtx_prefetch_file.argtypes = [ctxp, C.c_char_p, C.c_uint64, C.c_uint64]
    L.vtx_write_mtx.restype = C.c_int
    L.vtx_write_mtx.argtypes = [ctxp, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
    L.vtx_comm_ranks.restype = C.c_int
    L.vtx_comm_ranks.argtypes = [ctxp, C.POINTER(C.c_int)]
    L.vtx_debug_inflate.restype = C.c_int
    L.vtx_debug_inflate.argtypes = [ctxp, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    L.vtx_debug_ingest.restype = C.c_int
    L.vtx_debug_ingest.argtypes = [ctxp, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    _libs[path] = L
    return L


def status_name(status: int) -> str:
    return load().vtx_status_name(status).decode()


COMM_ID_BYTES = 128


def gather_plan(counts):
    """(status, offsets, total) of vtx_gather_plan: where every rank's triplets land in the gathered arrays (pure; no GPU)."""
    L = load()
    n = len(counts)
    c = (C.c_uint64 * n)(*[int(v) for v in counts])
    off = (C.c_uint64 * n)()
    tot = C.c_uint64(0)
    rc = L.vtx_gather_plan(n, c, off, C.byref(tot))
    return rc, list(off), int(tot.value)


def comm_id() -> bytes:
    """RCCL unique id (rank 0 creates it; the host hands the 128 bytes to the other ranks)."""
    L = load()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = L.vtx_comm_id(buf)
    if rc != abi.VTX_OK:
        raise VtxError(rc, L.vtx_strerror(None).decode())
    return buf.raw


class Context:
    """One vtx_ctx: bound to one GPU, holds one resident batch."""

    def __init__(self, cfg: abi.VtxConfig, variant=None):
        self._L = load(variant)
        self._h = C.c_void_p()
        self.cfg = cfg
        rc = self._L.vtx_create(C.byref(cfg), C.byref(self._h))
        if rc != abi.VTX_OK:
            raise VtxError(rc, self._L.vtx_strerror(None).decode())
        self.n_records = 0

    def _check(self, rc: int):
        if rc != abi.VTX_OK:
            raise VtxError(rc, self._L.vtx_strerror(self._h).decode())

    def close(self):
        if self._h:
            self._L.vtx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_read_format(self, fmt: int):
        self._check(self._L.vtx_set_read_format(self._h, int(fmt)))

    def submit(self, batch: abi.PackedBatch):
        self.set_read_format(getattr(batch, "read_format", 0))
        st = batch.as_struct()
        self._check(self._L.vtx_submit(self._h, C.byref(st)))
        self.n_records = batch.n_records
        self._n_loci = batch.n_loci

    def set_barcodes(self, barcodes):
        """Barcode list (bytes objects, index = matrix column) for ``submit_raw``; load_barcodes, src/main.rs:697-718."""
        blobs = [b if isinstance(b, bytes) else b.encode() for b in barcodes]
        offsets = np.zeros(len(blobs) + 1, np.uint64)
        np.cumsum([len(b) for b in blobs], out=offsets[1:])
        data = np.frombuffer(b"".join(blobs), np.uint8) if blobs else np.zeros(0, np.uint8)
        data = np.ascontiguousarray(data)
        self._check(self._L.vtx_set_barcodes(self._h, data.ctypes.data if data.size else None, offsets.ctypes.data, len(blobs)))

    def submit_raw(self, raw: abi.RawBatch) -> abi.VtxRawStats:
        self.set_read_format(getattr(raw, "read_format", 0))
        st = raw.as_struct()
        stats = abi.VtxRawStats()
        self._check(self._L.vtx_submit_raw(self._h, C.byref(st), C.byref(stats)))
        self.n_records = int(stats.kept)
        self._n_loci = raw.n_loci
        return stats

    def prefetch_file(self, path: str, file_off: int = 0, n: int = 0):
        """Start uploading bytes [file_off, file_off + n) of the file (n = 0: to its end) for a later submit_bam."""
        self._check(self._L.vtx_prefetch_file(self._h, path.encode(), file_off, n))

    def submit_bam(self, ingest: abi.VtxBamIngest, n_loci: int) -> abi.VtxIngestStats:
        """Device-side ingest of a BAM range (vtx_submit_bam): ``ingest`` comes from ``hostlib.plan_ingest`` (or is built by hand in the
        tests); the barcode list must be set.  Afterwards the context is in the state ``submit_raw`` leaves."""
        stats = abi.VtxIngestStats()
        self._check(self._L.vtx_submit_bam(self._h, C.byref(ingest), C.byref(stats)))
        self.n_records = int(stats.raw.kept)
        self._n_loci = n_loci
        return stats

    def debug_inflate(self, data: bytes, blocks):
        """bgzf_inflate_kernel on raw-DEFLATE payloads: blocks = [(offset, clen, isize)] into ``data``.  -> (status array, [bytes per block])"""
        blk = np.array(blocks, dtype=abi.BGZF_BLOCK_DTYPE) if len(blocks) else np.zeros(0, abi.BGZF_BLOCK_DTYPE)
        total = int(blk["isize"].sum())
        out = np.zeros(total + 64, np.uint8)
        status = np.zeros(max(len(blk), 1), np.uint32)
        self._check(self._L.vtx_debug_inflate(self._h, data, len(data), blk.ctypes.data if len(blk) else None, len(blk),
                                              out.ctypes.data, total, status.ctypes.data))
        offs = np.concatenate([[0], np.cumsum(blk["isize"])]).astype(np.int64)
        return status[:len(blk)], [bytes(out[offs[i]:offs[i + 1]]) for i in range(len(blk))]

    def debug_ingest(self, what: int, dtype=np.uint8) -> np.ndarray:
        """An intermediate array of the last submit_bam (abi.INGEST_*)."""
        n = C.c_uint64(0)
        self._check(self._L.vtx_debug_ingest(self._h, what, None, 0, C.byref(n)))
        out = np.zeros(int(n.value), np.uint8)
        if n.value:
            self._check(self._L.vtx_debug_ingest(self._h, what, out.ctypes.data, n.value, C.byref(n)))
        return out.view(dtype)

    def fetch_records(self):
        """Resolved, sorted records of the resident batch + per-locus (rec_begin, rec_count)."""
        recs = np.zeros(self.n_records, abi.RECORD_DTYPE)
        nl = getattr(self, "_n_loci", 0)
        begin = np.zeros(nl, np.uint32)
        count = np.zeros(nl, np.uint32)
        self._check(self._L.vtx_fetch_records(self._h, recs.ctypes.data, begin.ctypes.data, count.ctypes.data))
        return recs, begin, count

    def run(self):
        self._check(self._L.vtx_run(self._h))

    def fetch_scores(self):
        ref = np.zeros(self.n_records, np.int32)
        alt = np.zeros(self.n_records, np.int32)
        self._check(self._L.vtx_fetch_scores(self._h, ref.ctypes.data, alt.ctypes.data))
        return ref, alt

    def fetch_coo(self) -> dict:
        coo = abi.VtxCoo()
        self._check(self._L.vtx_fetch_coo(self._h, C.byref(coo)))
        n = int(coo.nnz)
        out = {}
        for k, dt in (("row", np.uint32), ("col", np.uint32), ("alt", np.uint32), ("ref", np.uint32),
                      ("unk", np.uint32), ("value", np.float64), ("ref_value", np.float64)):
            out[k] = np.ctypeslib.as_array(getattr(coo, k), shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
        return out

    def device_scores(self):
        r, a = C.c_void_p(), C.c_void_p()
        self._check(self._L.vtx_device_scores(self._h, C.byref(r), C.byref(a)))
        return r.value, a.value

    def device_coo(self) -> dict:
        """Raw device addresses of the triplet arrays + nnz (for torch/RCCL interop)."""
        coo = abi.VtxCoo()
        self._check(self._L.vtx_device_coo(self._h, C.byref(coo)))
        out = {"nnz": int(coo.nnz)}
        for k in ("row", "col", "alt", "ref", "unk", "value", "ref_value"):
            out[k] = C.cast(getattr(coo, k), C.c_void_p).value or 0
        return out

    def comm_init(self, ident: bytes, rank: int, world: int):
        """Join the RCCL communicator of the sharded run (collective; vtx_comm_init)."""
        assert len(ident) == COMM_ID_BYTES
        self._check(self._L.vtx_comm_init(self._h, C.c_char_p(ident), rank, world))

    def write_mtx(self, path: str, n_rows: int, n_cols: int, which: int = 0) -> float:
        """The last run's triplets as Matrix-Market text, formatted on the device and streamed into ``path``; returns the sum of the
        values.  Raises VTX_E_UNSUPPORTED for non-integral values (alt_frac): use fetch_coo + hostlib.write_mtx."""
        s = C.c_double(0.0)
        self._check(self._L.vtx_write_mtx(self._h, path.encode(), n_rows, n_cols, which, C.byref(s)))
        return float(s.value)

    def comm_ranks(self) -> int:
        n = C.c_int(0)
        self._check(self._L.vtx_comm_ranks(self._h, C.byref(n)))
        return int(n.value)

    def gather_coo(self, dst: int = 0) -> dict:
        """Collective: every rank's triplets to rank ``dst`` over RCCL (vtx_gather_coo).  Returns the device addresses of
        the gathered arrays + nnz on ``dst`` (nnz = 0 elsewhere), same form as ``device_coo``."""
        coo = abi.VtxCoo()
        self._check(self._L.vtx_gather_coo(self._h, dst, C.byref(coo)))
        out = {"nnz": int(coo.nnz)}
        for k in ("row", "col", "alt", "ref", "unk", "value", "ref_value"):
            out[k] = C.cast(getattr(coo, k), C.c_void_p).value or 0
        return out

    def gather_abort(self):
        """Collective: takes part in the status round of ``gather_coo`` with an error flag, so that the other ranks leave
        their ``gather_coo`` with VTX_E_PEER (vtx_gather_abort)."""
        self._check(self._L.vtx_gather_abort(self._h))

    def fetch_gathered(self) -> dict:
        coo = abi.VtxCoo()
        self._check(self._L.vtx_fetch_gathered(self._h, C.byref(coo)))
        n = int(coo.nnz)
        out = {}
        for k, dt in (("row", np.uint32), ("col", np.uint32), ("alt", np.uint32), ("ref", np.uint32),
                      ("unk", np.uint32), ("value", np.float64), ("ref_value", np.float64)):
            out[k] = np.ctypeslib.as_array(getattr(coo, k), shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
        return out

    # ---- audit / test hooks (include/vtx.h: vtx_set_debug) ----
    def set_stage_trace(self, on: bool = True):
        self._check(self._L.vtx_set_debug(self._h, abi.DEBUG_STAGE_TRACE, int(bool(on))))

    def set_poison(self, value=None):
        """Every later run first fills the score arrays with ``value`` (None: off)."""
        if value is not None:
            self._check(self._L.vtx_set_debug(self._h, abi.DEBUG_POISON_VALUE, int(value)))
        self._check(self._L.vtx_set_debug(self._h, abi.DEBUG_POISON_SCORES, 0 if value is None else 1))

    def fetch_stage(self) -> np.ndarray:
        """One byte per task (2 * record + haplotype): the stage that decided its score (abi.STAGE_*)."""
        st = np.zeros(2 * self.n_records, np.uint8)
        self._check(self._L.vtx_fetch_stage(self._h, st.ctypes.data))
        return st

    def debug_bands(self, tasks, stride: int):
        """(lo, hi, status) of band_sweep_kernel for the given tasks of the resident batch (vtx_debug_bands)."""
        t = np.ascontiguousarray(tasks, np.uint32)
        lo = np.zeros((len(t), stride), np.uint16)
        hi = np.zeros((len(t), stride), np.uint16)
        status = np.zeros(len(t), np.uint8)
        self._check(self._L.vtx_debug_bands(self._h, t.ctypes.data, len(t), stride, lo.ctypes.data, hi.ctypes.data, status.ctypes.data))
        return lo, hi, status

    def debug_tables(self) -> np.ndarray:
        """The haplotype k-mer tables the last banded run left in global memory, as bytes (vtx_debug_tables); empty: tables in LDS."""
        n = C.c_uint64(0)
        self._check(self._L.vtx_debug_tables(self._h, None, 0, C.byref(n)))
        out = np.zeros(int(n.value), np.uint8)
        if n.value:
            self._check(self._L.vtx_debug_tables(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def timing(self) -> abi.VtxTiming:
        t = abi.VtxTiming()
        self._check(self._L.vtx_last_timing(self._h, C.byref(t)))
        return t

    def cells(self) -> int:
        n = C.c_uint64(0)
        self._check(self._L.vtx_last_cells(self._h, C.byref(n)))
        return int(n.value)
