"""What is at stake in every recollected detail of bio 0.30's banded aligner (include/vtx_band_semantics.h, oracle/vtx_oracle.c
VTXO_VAR_*): alignments whose banded score changes and per-read calls that change when ONE detail takes its alternative, on
  (a) every read of the reference's test/test.bam that reaches the aligner with the barcode list ignored (576 real reads),
  (b) a config-5-shaped synthetic batch (30 % indel loci <= 20 bp, UMIs),
  (c) real-read shapes (soft clips, adapters, spliced reads; tests/stress_batches.py).
CPU only (the oracle).  Writes profiles/r06_band_semantics_sensitivity.json and prints a markdown table.
    python tools/band_semantics_table.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import stress_batches as SB  # noqa: E402
from oracle import oracle  # noqa: E402
from test_band_variants import _all_reads_batch  # noqa: E402
from vartrix_amd import synth  # noqa: E402
from vartrix_amd.abi import default_config  # noqa: E402

EDGE = 0x7fffffff
VARIANTS = [  # (label, which, value)
    ("default (the recollection)", None, None),
    ("lazy extension 0 instead of 2k", 0, 0),
    ("lazy extension k instead of 2k", 0, 6),
    ("lazy extension 2w = 40 instead of 2k", 0, 40),
    ("lazy extension to the matrix edge", 0, EDGE),
    ("add_kmer anchors 0..k-1 instead of 0..k", 1, 5),
    ("no k-mer match: empty band instead of the whole matrix", 2, 0),
    ("sdpkpp ties to the smaller match index", 3, 0),
]


def calls(r, a, ms=25):
    return np.where((r < ms) & (a < ms), 0, np.where(r > a, 1, np.where(a > r, 2, 3)))


def scores(batch, cfg, which, value):
    L = oracle.lib()
    L.vtxo_set_variant.argtypes = [C.c_int, C.c_int]
    if which is not None:
        L.vtxo_set_variant(which, value)
    try:
        return oracle.batch_scores(batch, cfg, threads=1)          # the hooks are process globals: one thread
    finally:
        if which is not None:
            L.vtxo_set_variant(which, -1)


def table(batches):
    rows = []
    for name, batch, nb in batches:
        cfg = default_config(aligner="banded", n_barcodes=nb)
        base = scores(batch, cfg, None, None)
        full = oracle.batch_scores(batch, default_config(aligner="full", n_barcodes=nb), threads=8)
        for label, which, value in VARIANTS:
            r, a = scores(batch, cfg, which, value)
            rows.append({"workload": name, "variant": label, "alignments": 2 * batch.n_records,
                         "alignments_changed_vs_default": int((r != base[0]).sum() + (a != base[1]).sum()),
                         "calls_changed_vs_default": int((calls(r, a) != calls(*base)).sum()),
                         "alignments_ne_full_matrix": int((r != full[0]).sum() + (a != full[1]).sum()),
                         "calls_ne_full_matrix": int((calls(r, a) != calls(*full)).sum())})
    return rows


def workloads(small=False):
    b, _, n_cb = _all_reads_batch()
    yield ("test.bam, all 576 real reads", b, n_cb)
    spec = synth.SynthSpec(n_loci=60 if small else 400, n_barcodes=500, reads_per_locus=32 if small else 64, indel_frac=0.30, use_umi=True)
    yield ("config-5 shape (30 % indel loci, UMIs)", synth.make_batch(spec), 500)
    for label, batch, nb in SB.real_shape_batches(trials=1, loci=20 if small else 80, reads=16 if small else 40):
        yield ("real-read shapes (clips, adapters, splices)", batch, nb)


if __name__ == "__main__":
    rows = table(workloads())
    out = os.path.join(ROOT, "profiles", "r06_band_semantics_sensitivity.json")
    json.dump({"method": "oracle (CPU) with one recollected detail switched at a time: tools/band_semantics_table.py", "rows": rows}, open(out, "w"), indent=1)
    print("| workload | variant | alignments | changed vs default | calls changed | != full matrix | calls != full |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %d | %d | %d | %d | %d |" % (r["workload"], r["variant"], r["alignments"], r["alignments_changed_vs_default"],
                                                        r["calls_changed_vs_default"], r["alignments_ne_full_matrix"], r["calls_ne_full_matrix"]))
