"""The device follows include/vtx_band_semantics.h: libvtx_lazy0.so / libvtx_anchor5.so / libvtx_noseed0.so are the same sources
compiled with ONE recollected detail of the crate's band at its alternative (`make -C vartrix_amd/csrc variants`), and must
reproduce the oracle run with the same override — on the real reads of test.bam, on tests/golden/band_kat.json and on batches
where the override changes scores.  (A maintainer who holds bio-0.30.0/src/alignment/pairwise/banded.rs
corrects the header; this test is what says the device and the oracle both follow.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from oracle import oracle
from vartrix_amd import lib
from vartrix_amd.abi import default_config
import stress_batches as SB
from test_band_variants import _all_reads_batch
variant, which, value, must_move = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
assert lib.LIB_PATH.endswith("libvtx_%%s.so" %% variant)
L = oracle.lib(); L.vtxo_set_variant.argtypes = [C.c_int, C.c_int]
b, _, n_cb = _all_reads_batch()
# the known-answer vectors too: the ones tagged with this detail are where the override shows (no common 6-mer: only there)
kat = json.load(open(os.path.join(%r, "tests", "golden", "band_kat.json")))["vectors"]
haps = [(v["hap"].encode("latin-1"), v["hap"].encode("latin-1")) for v in kat]
reads = [[(0, 0, v["read"].encode("latin-1"))] for v in kat]
moved = 0
for label, batch, nb in [("test.bam", b, n_cb), ("band_kat.json", SB.manual_batch(haps, reads, 4), 4)] + list(SB.synthetic_batches(per_model=1))[:4] + list(SB.real_shape_batches(trials=1)):
    cfg = default_config(aligner="banded", scoring_mode="coverage", n_barcodes=nb)
    with lib.Context(cfg) as ctx:
        ctx.submit(batch); ctx.run(); r, a = ctx.fetch_scores()
    base = oracle.batch_scores(batch, cfg, threads=1)
    L.vtxo_set_variant(which, value)
    try:
        o = oracle.batch_scores(batch, cfg, threads=1)
    finally:
        L.vtxo_set_variant(which, -1)
    bad = np.nonzero((r != o[0]) | (a != o[1]))[0]
    assert bad.size == 0, (label, int(bad[0]), int(r[bad[0]]), int(o[0][bad[0]]), int(a[bad[0]]), int(o[1][bad[0]]))
    moved += int((o[0] != base[0]).sum() + (o[1] != base[1]).sum())
assert (moved > 0) == bool(must_move), "the override must change some scores (lazy extension, no-seed band) / none (last anchor): %%d" %% moved
print("variant-ok", moved)
''' % (ROOT, os.path.join(ROOT, "tests"), ROOT)


# variant, oracle hook (vtxo_set_variant which / value), must the override move scores?
@pytest.mark.parametrize("variant,which,value,must_move", [("lazy0", 0, 0, 1), ("lazy40", 0, 40, 1), ("anchor5", 1, 5, 0), ("noseed0", 2, 0, 1)])
def test_device_follows_the_header_when_a_recollected_detail_is_recompiled(variant, which, value, must_move):
    """libvtx_lazy0.so / libvtx_lazy40.so (round 6: 2 w, the other plausible reading of set_boundaries) / libvtx_anchor5.so / libvtx_noseed0.so: the production sources with ONE constant of
    include/vtx_band_semantics.h at its alternative.  Each must reproduce the oracle run with the same override on test.bam's real
    reads, on the known-answer vectors and on the stress batches.  (anchor5 moves nothing — tests/test_band_kat.py shows why — and the
    device agrees; the fourth detail, sdpkpp's tie rule, is the order of the packed words the kernels maximise, not a constant:
    band_kat.json carries 28 vectors whose score tells the two rules apart, for the crate holder.)"""
    so = os.path.join(ROOT, "vartrix_amd", "libvtx_%s.so" % variant)
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "vartrix_amd", "csrc"), "variants"])
    r = subprocess.run([sys.executable, "-c", CODE, variant, str(which), str(value), str(must_move)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, VTX_LIB_VARIANT=variant))
    assert r.returncode == 0 and "variant-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
