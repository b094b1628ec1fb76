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
    out = (C.c_uint32 * 9)()
    assert L.vtx_abi_sizes(out, 9) == abi.VTX_ABI_VERSION
    assert list(out) == [C.sizeof(abi.VtxConfig), abi.LOCUS_DTYPE.itemsize, abi.RECORD_DTYPE.itemsize,
                         C.sizeof(abi.VtxBatch), C.sizeof(abi.VtxCoo), C.sizeof(abi.VtxTiming),
                         abi.RAW_RECORD_DTYPE.itemsize, C.sizeof(abi.VtxRawBatch), C.sizeof(abi.VtxRawStats)]


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
