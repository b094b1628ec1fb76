"""Pin the CPU oracle against the reference's own fixtures.

The reference's runnable tests (src/main.rs:1208-1390) run `_main` on
test/test.{vcf,bam,fa} + barcodes.tsv and compare the produced .mtx, as CSR,
with stored matrices.  Here the same inputs (copied as data fixtures into
tests/golden/) go through oracle/refpipe.py (ingest+filters) and the C oracle
(alignment, calls, UMI collapse, matrix modes), and must reproduce every
fixture — for both aligner flavours.
"""
import os

import numpy as np
import pytest

from oracle import oracle, refpipe
from vartrix_amd.abi import default_config

CASES = [
    # (reference test,                          mode,        umi,   barcodes,          main fixture,            ref fixture)
    ("test_consensus_matrix :1208",             "consensus", False, "barcodes.tsv",    "test_consensus.mtx",    None),
    ("test_frac_matrix :1236",                  "alt_frac",  False, "barcodes.tsv",    "test_frac.mtx",         None),
    ("test_coverage_matrices :1266",            "coverage",  False, "barcodes.tsv",    "test_coverage.mtx",     "test_coverage_ref.mtx"),
    ("test_coverage_matrices_umi :1303",        "coverage",  True,  "barcodes.tsv",    "test_coverage_umi.mtx", "test_coverage_ref_umi.mtx"),
    ("test_coverage_matrices_umi_gzipped :1342","coverage",  True,  "barcodes.tsv.gz", "test_coverage_umi.mtx", "test_coverage_ref_umi.mtx"),
]


@pytest.fixture(scope="module")
def inputs(golden_dir):
    g = golden_dir
    return dict(vcf=refpipe.read_vcf(os.path.join(g, "test.vcf")),
                fasta=refpipe.read_fasta(os.path.join(g, "test.fa")),
                bam=refpipe.read_bam(os.path.join(g, "test.bam")))


def run_oracle(inputs, golden_dir, mode, umi, bcfile, aligner):
    bcs = refpipe.load_barcodes(os.path.join(golden_dir, bcfile))
    args = refpipe.Args(use_umi=umi)
    batch, metrics = refpipe.pack(inputs["vcf"], inputs["fasta"], inputs["bam"], bcs, args)
    cfg = default_config(aligner=aligner, scoring_mode=mode, use_umi=int(umi), n_barcodes=len(bcs))
    ref, alt = oracle.batch_scores(batch, cfg)
    coo = oracle.batch_reduce(batch, cfg, ref, alt)
    return batch, metrics, ref, alt, coo, len(inputs["vcf"]), len(bcs)


@pytest.mark.parametrize("aligner", ["banded", "full"])
@pytest.mark.parametrize("case", CASES, ids=[c[0].split()[0] for c in CASES])
def test_reference_fixture(inputs, golden_dir, case, aligner):
    _, mode, umi, bcfile, main_fx, ref_fx = case
    batch, metrics, ref, alt, coo, nv, nb = run_oracle(inputs, golden_dir, mode, umi, bcfile, aligner)
    shape, want = refpipe.read_mtx(os.path.join(golden_dir, main_fx))
    assert shape == (nv, nb)
    got = {(int(r), int(c)): float(v) for r, c, v in zip(coo["row"], coo["col"], coo["value"])}
    assert got == want
    if ref_fx:
        shape, want = refpipe.read_mtx(os.path.join(golden_dir, ref_fx))
        got = {(int(r), int(c)): float(v) for r, c, v in zip(coo["row"], coo["col"], coo["ref_value"])}
        assert got == want


def test_fixture_facts(inputs, golden_dir):
    """reads fetched per locus 12/398/97/69; 15 reach the aligner."""
    bcs = refpipe.load_barcodes(os.path.join(golden_dir, "barcodes.tsv"))
    assert len(bcs) == 20
    batch, metrics = refpipe.pack(inputs["vcf"], inputs["fasta"], inputs["bam"], bcs, refpipe.Args())
    assert metrics["num_reads"] == 12 + 398 + 97 + 69
    assert batch.n_records == 15
    assert list(batch.loci["rec_count"]) == [1, 7, 7, 0]
    assert metrics["num_invalid_recs"] == 0 and metrics["num_multiallelic_recs"] == 0


def test_scores_behind_fixtures(inputs, golden_dir):
    """per-read (ref, alt) scores with full SW; banded equal on these."""
    bcs = refpipe.load_barcodes(os.path.join(golden_dir, "barcodes.tsv"))
    batch, _ = refpipe.pack(inputs["vcf"], inputs["fasta"], inputs["bam"], bcs, refpipe.Args())
    cfg = default_config(aligner="full", n_barcodes=len(bcs))
    ref, alt = oracle.batch_scores(batch, cfg)
    pairs = sorted(zip(ref.tolist(), alt.tolist()))
    want = sorted([(110, 104), (106, 100), (85, 79), (85, 79), (120, 114), (135, 129), (135, 129), (134, 128),
                   (144, 150), (144, 150), (131, 137), (125, 131), (125, 131), (119, 125), (105, 111)])
    assert pairs == want
    cfgb = default_config(aligner="banded", n_barcodes=len(bcs))
    refb, altb = oracle.batch_scores(batch, cfgb)
    assert np.array_equal(ref, refb) and np.array_equal(alt, altb)


def test_mtx_text_matches_fixture_bytes(inputs, golden_dir):
    """write_matrix_market text (src/main.rs:381): same bytes as test_consensus.mtx / test_frac.mtx
    (those two fixtures are in the merge-loop order; the coverage fixtures predate the sort at :932)."""
    for mode, fx in (("consensus", "test_consensus.mtx"), ("alt_frac", "test_frac.mtx")):
        _, _, _, _, coo, nv, nb = run_oracle(inputs, golden_dir, mode, False, "barcodes.tsv", "banded")
        text = refpipe.mtx_text(nv, nb, coo["row"], coo["col"], coo["value"])
        assert text == open(os.path.join(golden_dir, fx)).read()
