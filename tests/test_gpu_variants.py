"""The device follows include/vtx_band_semantics.h: libvtx_lazy0.so is the same source compiled with the band's lazy extension
set to 0 (`make -C vartrix_amd/csrc variants`), and must reproduce the oracle run with the same override — on the real reads of
test.bam and on batches where the extension changes scores.  (A maintainer who holds bio-0.30.0/src/alignment/pairwise/banded.rs
corrects the header; this test is what says the device and the oracle both follow.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import oracle
from vartrix_amd import lib
from vartrix_amd.abi import default_config
import stress_batches as SB
from test_band_variants import _all_reads_batch
assert lib.LIB_PATH.endswith("libvtx_lazy0.so")
L = oracle.lib(); L.vtxo_set_variant.argtypes = [C.c_int, C.c_int]
b, _, n_cb = _all_reads_batch()
moved = 0
for label, batch, nb in [("test.bam", b, n_cb)] + list(SB.synthetic_batches(per_model=1))[:4] + list(SB.real_shape_batches(trials=1)):
    cfg = default_config(aligner="banded", scoring_mode="coverage", n_barcodes=nb)
    with lib.Context(cfg) as ctx:
        ctx.submit(batch); ctx.run(); r, a = ctx.fetch_scores()
    base = oracle.batch_scores(batch, cfg, threads=1)
    L.vtxo_set_variant(0, 0)
    try:
        o = oracle.batch_scores(batch, cfg, threads=1)
    finally:
        L.vtxo_set_variant(0, -1)
    assert np.array_equal(r, o[0]) and np.array_equal(a, o[1]), label
    moved += int((o[0] != base[0]).sum() + (o[1] != base[1]).sum())
assert moved > 0, "the override must change some scores, or the test shows nothing"
print("variant-ok", moved)
''' % (ROOT, os.path.join(ROOT, "tests"))


def test_device_follows_the_header_when_the_lazy_extension_is_recompiled_to_zero():
    so = os.path.join(ROOT, "vartrix_amd", "libvtx_lazy0.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "vartrix_amd", "csrc"), "variants"])
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, VTX_LIB_VARIANT="lazy0"))
    assert r.returncode == 0 and "variant-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
