"""The certificate's inequalities  cert <= banded <= full <= ub_exact <= ub  (oracle/vtx_certify.c) on the distributions of
tools/certify_stress.py, size-capped so that the driver's CPU run includes them (the tool runs all 322 k alignments)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import certify_stats as cs  # noqa: E402
import stress_batches as SB  # noqa: E402
from vartrix_amd.abi import default_config  # noqa: E402


def _violations(r, exact):
    has = r["cert"] >= 0
    v = int((r["ub"] < r["full"]).sum()) + int((has & (r["cert"] > r["banded"])).sum()) + int((r["banded"] > r["full"]).sum())
    if exact:
        v += int((r["ub_exact"] < r["full"]).sum()) + int((r["ub"] < r["ub_exact"]).sum())
    return v


def test_error_models():
    cfg = default_config(aligner="banded", n_barcodes=500)
    tot = 0
    for label, batch, nb in SB.synthetic_batches(per_model=1, n_loci=60, reads=32):
        r = cs.certify(batch, cfg, os.cpu_count() or 8, exact=False)
        assert _violations(r, False) == 0, label
        tot += len(r["full"])
    assert tot > 20000


def test_repeat_rich_with_the_exact_bound():
    cfg = default_config(aligner="banded", n_barcodes=30)
    tot = 0
    for label, batch, nb in SB.repeat_rich_batches(trials=4, loci=20, reads=12):
        r = cs.certify(batch, cfg, os.cpu_count() or 8, exact=True)
        assert _violations(r, True) == 0, label
        tot += len(r["full"])
    assert tot > 1000


def test_real_read_shapes():
    cfg = default_config(aligner="banded", n_barcodes=30)
    for label, batch, nb in SB.real_shape_batches(trials=2, loci=40, reads=24):
        r = cs.certify(batch, cfg, os.cpu_count() or 8, exact=False)
        assert _violations(r, False) == 0, label
