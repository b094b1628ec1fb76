"""C-ABI surface checks that run without a GPU: the library loads, exports every
symbol include/vtx.h declares, struct layouts agree with the ctypes mirror, and
compute entry points fail loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vartrix_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return lib.load()


def test_exports_every_declared_symbol(L):
    header = open(os.path.join(ROOT, "include", "vtx.h")).read()
    declared = set(re.findall(r"\b(vtx_[a-z_]+)\s*\(", header))
    declared -= {"vtx_ctx"}
    assert declared == set(lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_sizes_match(L):
    out = (C.c_uint32 * 13)()
    assert L.vtx_abi_sizes(out, 13) == abi.VTX_ABI_VERSION
    assert list(out) == [C.sizeof(abi.VtxConfig), abi.LOCUS_DTYPE.itemsize, abi.RECORD_DTYPE.itemsize,
                         C.sizeof(abi.VtxBatch), C.sizeof(abi.VtxCoo), C.sizeof(abi.VtxTiming),
                         abi.RAW_RECORD_DTYPE.itemsize, C.sizeof(abi.VtxRawBatch), C.sizeof(abi.VtxRawStats),
                         abi.BGZF_BLOCK_DTYPE.itemsize, abi.BAM_INTERVAL_DTYPE.itemsize, C.sizeof(abi.VtxBamIngest),
                         C.sizeof(abi.VtxIngestStats)]


def test_config_default_is_reference_constants(L):
    cfg = abi.VtxConfig()
    L.vtx_config_default(C.byref(cfg))
    want = abi.default_config()
    for f, _ in abi.VtxConfig._fields_:
        assert getattr(cfg, f) == getattr(want, f), f
    # src/main.rs:30-38
    assert (cfg.match_score, cfg.mismatch_score, cfg.gap_open, cfg.gap_extend) == (1, -5, -5, -1)
    assert (cfg.min_score, cfg.kmer_k, cfg.band_w) == (25, 6, 20)


def test_create_rejects_bad_config(L):
    h = C.c_void_p()
    cfg = abi.default_config(abi_version=99)
    assert L.vtx_create(C.byref(cfg), C.byref(h)) == abi.VTX_E_INVAL
    cfg = abi.default_config(mismatch_score=-4)
    assert L.vtx_create(C.byref(cfg), C.byref(h)) == abi.VTX_E_UNSUPPORTED
    assert b"src/main.rs:35-38" in L.vtx_strerror(None)


def test_no_cpu_fallback_without_device(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.VtxError) as ei:
        lib.Context(abi.default_config(aligner="full", n_barcodes=4))
    assert ei.value.status == abi.VTX_E_NODEVICE


def test_status_names(L):
    assert lib.status_name(0) == "VTX_OK"
    assert lib.status_name(abi.VTX_E_NODEVICE) == "VTX_E_NODEVICE"


def test_production_library_has_no_experiment_hooks(L):
    """libvtx.so as shipped reads ONE environment variable, VTX_DEBUG (stderr diagnostics).  Stage switches, ablations ("results
    wrong by design"), buffer caps and the socket transport that stands in for RCCL are compiled into libvtx_dev.so only
    (-DVTX_DEVTOOLS): a stray variable in a production environment cannot change a matrix."""
    csrc = os.path.join(ROOT, "vartrix_amd", "csrc")
    hooks = set()
    for f in ("vtx_api.hip", "vtx_band.hip", "vtx_sweep.hip", "vtx_kernels.hip", "vtx_prep.hip", "vtx_comm_test.hip"):
        src = open(os.path.join(csrc, f)).read()
        hooks |= set(re.findall(r'VTX_DEV_ENV\("([A-Z0-9_]+)"\)', src))
        # the only raw getenv of the library sources is VTX_DEBUG (the test transport, dev-only, reads its own directory variable)
        assert set(re.findall(r'[^_]getenv\("([A-Z0-9_]+)"\)', src)) <= ({"VTX_DEBUG"} if f != "vtx_comm_test.hip" else {"VTX_COMM_TEST_TRANSPORT"}), f
    assert len(hooks) >= 30 and {"VTX_DIAG_ABLATE", "VTX_SWEEP_ABLATE", "VTX_BAND_LEGACY", "VTX_COMM_TEST_TRANSPORT"} <= hooks

    def present(path):
        blob = open(path, "rb").read()
        return {h for h in hooks if h.encode() in blob}
    here = os.path.dirname(lib.lib_path(""))
    assert present(os.path.join(here, "libvtx.so")) == set()
    assert b"VTX_DEBUG" in open(os.path.join(here, "libvtx.so"), "rb").read()
    assert present(os.path.join(here, "libvtx_dev.so")) == hooks
    # the variants of the band-semantics tests are production builds with one constant changed: no hooks either
    for v in ("lazy0", "anchor5", "noseed0"):
        assert present(lib.lib_path(v)) == set(), v
    # the developer library exports the same C-ABI
    D = lib.load("dev")
    for name in lib.SYMBOLS:
        assert hasattr(D, name), name
