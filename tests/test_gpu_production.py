"""The production library under a hostile environment: every experiment knob of the developer build set to a value that would
change results there (ablations are "wrong by design", VTX_BAND_HARD_CAP=1 starves the buffers, the test transport replaces RCCL)
— libvtx.so, which has none of them compiled in, must still give the oracle's scores and matrix on a config-2-shaped batch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CODE = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
from oracle import oracle
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
assert lib.lib_path().endswith("libvtx.so")
# BASELINE configs[1] shape (SNV loci x 5 k barcodes, 150 bp reads, coverage mode) at a size the oracle finishes in seconds, plus a
# repeat-rich genome (the sweep runs) and noisy reads (the refinement and the one-diagonal DP run)
for kw in (dict(n_loci=300, reads_per_locus=64), dict(n_loci=120, reads_per_locus=32, sub_error=0.05),
           dict(n_loci=150, reads_per_locus=32, genome_fasta=os.path.join(%r, "tests", "golden", "test_dna.fa"))):
    spec = synth.SynthSpec(n_barcodes=5000, seed=11, **kw)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="banded", scoring_mode="coverage", n_barcodes=spec.n_barcodes)
    with lib.Context(cfg) as ctx:
        ctx.submit(batch); ctx.run()
        ref, alt = ctx.fetch_scores(); coo = ctx.fetch_coo()
    oref, oalt = oracle.batch_scores(batch, cfg, threads=8)
    assert np.array_equal(ref, oref) and np.array_equal(alt, oalt), kw
    ocoo = oracle.batch_reduce(batch, cfg, oref, oalt)
    for k in ("row", "col", "alt", "ref", "unk", "value", "ref_value"):
        assert np.array_equal(coo[k], ocoo[k]), (kw, k)
print("production-ok")
''' % (ROOT, ROOT)


def test_production_library_ignores_every_experiment_knob(tmp_path):
    hostile = dict(VTX_DIAG_ABLATE="5", VTX_SWEEP_ABLATE="2", VTX_COOP_ABLATE="1", VTX_BAND_ABLATE="1", VTX_BAND_HARD_CAP="1",
                   VTX_BAND_SLOTS="1", VTX_BAND_NO_DIAG="1", VTX_BAND_LEGACY="1", VTX_BAND_NO_TIGHT="1", VTX_BAND_CHUNK="256",
                   VTX_BAND_TABLES_V1="1", VTX_BAND_GT_BYTES="1", VTX_DP_KERNEL="lut", VTX_COMM_TEST_TRANSPORT=str(tmp_path))
    env = dict(os.environ, **hostile)
    env.pop("VTX_LIB_VARIANT", None)
    r = subprocess.run([sys.executable, "-c", _CODE], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "production-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_CODE2 = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
from oracle import oracle
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
assert lib.lib_path().endswith("libvtx.so")
spec = synth.SynthSpec(n_loci=15000, n_barcodes=5000, reads_per_locus=256, seed=20260926, genome_fasta=os.path.join(%r, "tests", "golden", "test_dna.fa"))
batch = synth.make_batch(spec)
cfg = default_config(aligner="banded", scoring_mode="consensus", n_barcodes=spec.n_barcodes)
with lib.Context(cfg) as ctx:
    ctx.submit(batch); ctx.run()
    ref, alt = ctx.fetch_scores()
    t = ctx.timing()
assert t.diag2_tasks >= 700000 and t.swept_tasks > 0 and t.diag2_scored > 0, (t.diag2_tasks, t.swept_tasks, t.diag2_scored)
oref, oalt = oracle.batch_scores(batch, cfg, threads=os.cpu_count() or 8)
bad = np.nonzero((ref != oref) | (alt != oalt))[0]
assert bad.size == 0, (int(bad[0]), int(ref[bad[0]]), int(oref[bad[0]]), int(alt[bad[0]]), int(oalt[bad[0]]))
print("production-second-stage-ok", int(t.diag2_tasks), int(t.diag2_streamed), int(t.swept_tasks))
''' % (ROOT, ROOT)


def test_production_library_second_stage_on_its_own_threshold():
    """the second single-diagonal stage (band_diag2_kernel, band_stream_kernel, the full-matrix check, the one-diagonal
    DP behind it) runs in libvtx.so only on lists above its threshold (700 000 tasks since round 6) — the other GPU tests reach it
    through libvtx_dev.so with the threshold forced to 1.  Here the PRODUCTION library on 15 000 loci drawn from real sequence
    (7.3 M alignments, 1 M of them repeat tasks): the stage runs on its own, and every score is the oracle's."""
    env = dict(os.environ)
    env.pop("VTX_LIB_VARIANT", None)
    r = subprocess.run([sys.executable, "-c", _CODE2], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0 and "production-second-stage-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
