"""How much do the three unpinned recollections of bio 0.30's band geometry matter on REAL reads?

include/vtx_band_semantics.h names them; this test varies the most parity-sensitive one — the lazy extension of
Band::set_boundaries — over {0, 2K (the recollection), to the matrix edge} on every read of the
reference's own test/test.bam that reaches the aligner when the barcode list is ignored (all CB values accepted), and
counts the alignments whose banded score leaves the full-matrix score and the per-read calls that change
(SURVEY.md §8c asks for exactly these two numbers).  The reference's fixtures only hold 15 of these reads, all of them
insensitive — so the numbers below are the distance between "what we restated" and "what we can prove".
"""
import ctypes as C
import os

import numpy as np

from oracle import oracle, refpipe
from vartrix_amd.abi import default_config

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EDGE = 0x7fffffff


def _all_reads_batch():
    bam = refpipe.read_bam(os.path.join(G, "test.bam"))
    cbs = {}
    for r in bam.recs:
        v = refpipe.aux_string(r.aux, b"CB")
        if v is not None and v not in cbs:
            cbs[v] = len(cbs)
    vcf = refpipe.read_vcf(os.path.join(G, "test.vcf"))
    batch, metrics = refpipe.pack(vcf, refpipe.read_fasta(os.path.join(G, "test.fa")), bam, cbs, refpipe.Args())
    return batch, metrics, len(cbs)


def _scores(batch, cfg, ext):
    L = oracle.lib()
    L.vtxo_set_lazy_extension.argtypes = [C.c_int]
    L.vtxo_set_lazy_extension(ext)
    try:
        return oracle.batch_scores(batch, cfg, threads=1)      # the hook is a process global: one thread
    finally:
        L.vtxo_set_lazy_extension(-1)


def _calls(r, a, ms=25):
    return np.where((r < ms) & (a < ms), 0, np.where(r > a, 1, np.where(a > r, 2, 3)))


def test_lazy_extension_sensitivity_on_test_bam():
    batch, metrics, n_cb = _all_reads_batch()
    assert batch.n_loci == 4 and batch.n_records > 400          # hundreds of real, soft-clipped reads, not the 15 of the fixtures
    full = oracle.batch_scores(batch, default_config(aligner="full", n_barcodes=n_cb), threads=4)
    cfg = default_config(aligner="banded", n_barcodes=n_cb)
    report = {}
    for name, ext in (("0", 0), ("2K", -1), ("edge", EDGE)):
        r, a = _scores(batch, cfg, ext)
        n_aln = int((r != full[0]).sum() + (a != full[1]).sum())
        n_call = int((_calls(r, a) != _calls(*full)).sum())
        assert np.all(r <= full[0]) and np.all(a <= full[1])    # a band only removes paths
        report[name] = (n_aln, n_call)
    print("test.bam, %d reads reaching the aligner (%d alignments): banded != full / calls changed for extension " % (
        batch.n_records, 2 * batch.n_records) + ", ".join("%s: %d / %d" % (k, *v) for k, v in report.items()))
    # monotone in the extension: a longer extension only widens the band
    assert report["0"][0] >= report["2K"][0] >= report["edge"][0]
    assert report["edge"] == (0, 0)          # with the band run out to the matrix corners nothing differs from full SW here
    assert 0 < report["2K"][0] <= 8          # the recollected 2K: a handful of alignments move (SURVEY §8c saw 2 of 1 152) ...
    assert report["2K"][1] <= 2              # ... and at most a couple of calls


def test_default_variant_is_the_named_constant():
    """vtxo_set_lazy_extension(-1) == VTX_BAND_LAZY_EXT(6) == 12: explicit 12 gives the same scores."""
    batch, _, n_cb = _all_reads_batch()
    cfg = default_config(aligner="banded", n_barcodes=n_cb)
    a = _scores(batch, cfg, -1)
    b = _scores(batch, cfg, 12)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_every_recollected_detail_is_switchable_and_bounded():
    """tools/band_semantics_table.py at reduced size: every recollected detail of the crate (lazy extension, add_kmer's last
    anchor, no-seed band, sdpkpp tie direction) switched one at a time in the oracle.  The full-size table is
    profiles/r03_band_semantics_sensitivity.json; here: the hooks work, restore, and nothing but the lazy
    extension moves more than a handful of alignments."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import band_semantics_table as T
    rows = T.table(T.workloads(small=True))
    by = {}
    for r in rows:
        by.setdefault(r["workload"], {})[r["variant"]] = r
    for wl, v in by.items():
        assert v["default (the recollection)"]["alignments_changed_vs_default"] == 0
        n = v["default (the recollection)"]["alignments"]
        assert v["lazy extension to the matrix edge"]["alignments_ne_full_matrix"] == 0       # band out to the corners == full SW here
        for name in ("add_kmer anchors 0..k-1 instead of 0..k", "no k-mer match: empty band instead of the whole matrix",
                     "sdpkpp ties to the smaller match index"):
            assert v[name]["alignments_changed_vs_default"] <= max(2, n // 2000), (wl, name, v[name])
        assert v["lazy extension 0 instead of 2k"]["alignments_changed_vs_default"] <= max(6, n // 200)
    # the hooks restore: a default run afterwards equals the first
    batch, _, n_cb = _all_reads_batch()
    cfg = default_config(aligner="banded", n_barcodes=n_cb)
    a = oracle.batch_scores(batch, cfg, threads=1)
    b = _scores(batch, cfg, -1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
