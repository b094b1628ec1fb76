"""Shape per task (round-5 VERDICT item 4): a batch that mixes haplotypes of up to 255 bases with a FEW longer ones (a long deletion, a
long insertion) is scored in two passes — every ordinary locus through band_diag_kernel<., uint16_t> / band_sweep_kernel as if the long
ones were not there, the long ones through round 3's kernels — instead of the whole batch on round 3's path.  Call site of both:
src/main.rs:898-901; the haplotype lengths come from construct_haplotypes, src/main.rs:958-994 (2 x padding + allele)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from vartrix_amd import abi, lib, synth
from vartrix_amd.abi import PackedBatch, default_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

ROUND3_STAGES = (abi.STAGE_UNKNOWN, abi.STAGE_RUN_DP, abi.STAGE_GENERAL_DP)      # (0 = decided by band_run_kernel's certificate)


def long_loci_batch(n, seed, reads=64, max_indel=90):
    """n loci whose REF or ALT haplotype exceeds 255 bases: indels of 56 .. max_indel bases at padding 100."""
    spec = synth.SynthSpec(n_loci=4 * n + 8, n_barcodes=5000, reads_per_locus=reads, indel_frac=1.0, max_indel=max_indel, seed=seed)
    b = synth.make_batch(spec)
    long_ = np.nonzero(np.maximum(b.loci["ref_len"], b.loci["alt_len"]) > 255)[0][:n]
    assert len(long_) == n
    return [b.slice_loci(int(l), int(l) + 1) for l in long_]


def mixed_batch(n_loci, positions, reads=64, seed=11):
    base = synth.make_batch(synth.SynthSpec(n_loci=n_loci, n_barcodes=5000, reads_per_locus=reads, seed=seed))
    longs = long_loci_batch(len(positions), seed + 1, reads)
    parts, prev = [], 0
    for pos, lb in zip(positions, longs):
        parts += [base.slice_loci(prev, pos), lb]
        prev = pos
    parts.append(base.slice_loci(prev, n_loci))
    out = PackedBatch.concat(parts)
    is_long = np.maximum(out.loci["ref_len"], out.loci["alt_len"]) > 255
    assert int(is_long.sum()) == len(positions)
    return out, is_long


def run_banded(batch, n_barcodes=5000):
    with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=n_barcodes)) as ctx:
        ctx.set_stage_trace(True)
        ctx.set_poison(-4242)
        ctx.submit(batch)
        ctx.run()
        r, a = ctx.fetch_scores()
        return r, a, ctx.fetch_stage(), ctx.timing()


@pytest.mark.parametrize("positions", [(5000,), (0,), (9999,), (17, 18, 19, 4000, 9000)], ids=["middle", "first", "last", "five"])
def test_a_few_long_haplotypes_do_not_move_the_batch_to_the_round3_path(positions):
    """BASELINE.json configs[1] shape (10 k SNV loci x 5 k barcodes; 64 reads per locus to keep the oracle's share short) with one /
    five long-haplotype loci in it: fewer than 1 % of the alignments take round 3's kernels (all of them took them before), every
    alignment of the long loci and of their neighbours plus a sample of the rest equals the oracle, no score is left poisoned."""
    from audit_util import oracle_scores_of, stage_report
    n_loci = 10_000
    batch, is_long = mixed_batch(n_loci, [p for p in positions], seed=11 + len(positions))
    r, a, stage, t = run_banded(batch)
    assert not (r == -4242).any() and not (a == -4242).any()
    round3 = np.isin(stage, ROUND3_STAGES)
    print("%d alignments, round-3 path %d (%.3f %%); stages %s" % (len(stage), int(round3.sum()), 100 * round3.mean(), stage_report(stage)))
    assert round3.mean() < 0.01
    rec_locus = np.repeat(np.arange(batch.n_loci), batch.loci["rec_count"])
    near = np.zeros(batch.n_loci, bool)
    for l in np.nonzero(is_long)[0]:
        near[max(0, l - 2):l + 3] = True
    # the long loci were NOT decided by the first pass, and the short ones not by the second: a short locus' alignments carry the
    # stages of the sweep path (1, 2, 10, 3, 4, 7 — and 0 / 6 for the few two-diagonal chains)
    long_tasks = np.repeat(is_long[rec_locus], 2)
    assert np.isin(stage[long_tasks], ROUND3_STAGES + (abi.STAGE_DIAG_CERT, abi.STAGE_REFINE_CERT)).all()
    rng = np.random.default_rng(5)
    recs = np.unique(np.concatenate([np.nonzero(near[rec_locus])[0], rng.choice(batch.n_records, 4000, replace=False)]))
    ids, oref, oalt = oracle_scores_of(batch, recs, "banded", 5000)
    bad = np.nonzero((r[ids] != oref) | (a[ids] != oalt))[0]
    assert bad.size == 0, "record %d: device (%d, %d) oracle (%d, %d)" % (ids[bad[0]], r[ids[bad[0]]], a[ids[bad[0]]], oref[bad[0]], oalt[bad[0]])


def test_two_passes_equal_the_single_pass():
    """libvtx_dev.so under VTX_BAND_NO_SPLIT=1 scores the mixed batch in one pass on round 3's path, as rounds 3 - 5 did: the same
    scores for every alignment, and the matrices that follow from them."""
    code = r'''
import sys, os, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_shape as T
from vartrix_amd import lib
from vartrix_amd.abi import default_config
batch, is_long = T.mixed_batch(3000, [0, 1500, 1501, 2999], seed=23)
with lib.Context(default_config(aligner="banded", scoring_mode="consensus", n_barcodes=5000)) as ctx:
    ctx.submit(batch); ctx.run(); r, a = ctx.fetch_scores(); coo = ctx.fetch_coo()
np.savez(sys.argv[1], r=r, a=a, rows=coo['row'], cols=coo['col'], vals=coo['value'])
''' % (ROOT, os.path.join(ROOT, "tests"))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        res = []
        for env_extra in ({}, {"VTX_BAND_NO_SPLIT": "1"}):
            path = os.path.join(td, "s%d.npz" % len(res))
            p = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, VTX_LIB_VARIANT="dev", **env_extra))
            assert p.returncode == 0, p.stderr[-3000:]
            res.append(np.load(path))
        for k in ("r", "a", "rows", "cols", "vals"):
            assert np.array_equal(res[0][k], res[1][k]), k
        assert res[0]["r"].size > 150_000


def test_many_long_haplotypes_keep_the_single_pass():
    """More than an eighth of the loci long (a batch at a larger --padding): one pass, as before — and still the oracle's scores."""
    from audit_util import oracle_scores_of
    base = synth.make_batch(synth.SynthSpec(n_loci=30, n_barcodes=5000, reads_per_locus=32, seed=4))
    longs = long_loci_batch(10, 9, reads=32)
    batch = PackedBatch.concat([base.slice_loci(0, 15)] + longs + [base.slice_loci(15, 30)])
    r, a, stage, t = run_banded(batch)
    ids, oref, oalt = oracle_scores_of(batch, np.arange(batch.n_records), "banded", 5000)
    assert np.array_equal(r[ids], oref) and np.array_equal(a[ids], oalt)
    assert not np.isin(stage, (abi.STAGE_SWEEP_DP, abi.STAGE_BAND_CERT)).any()            # (the sweep path was not taken)


@pytest.mark.parametrize("read_len", [250, 200, 256])
def test_reads_above_192_bases_stay_in_the_first_stage(read_len):
    """A fourth mask word (round 6): band_diag_kernel takes reads up to 256 bases (192 before: longer ones all went to band_run_kernel).
    250-base reads at padding 100 hang over the window on both sides; at padding 150 (haplotypes above 255 bases) the batch is round
    3's anyway.  Every alignment against the oracle."""
    from audit_util import oracle_scores_of, stage_report
    spec = synth.SynthSpec(n_loci=300, n_barcodes=2000, reads_per_locus=48, read_len=read_len, padding=100, seed=31 + read_len)
    batch = synth.make_batch(spec)
    r, a, stage, t = run_banded(batch, 2000)
    assert not (r == -4242).any() and not (a == -4242).any()
    round3 = np.isin(stage, ROUND3_STAGES)
    print("%d-base reads: %d alignments, round-3 path %.2f %%; stages %s" % (read_len, len(stage), 100 * round3.mean(), stage_report(stage)))
    assert round3.mean() < 0.02
    ids, oref, oalt = oracle_scores_of(batch, np.arange(batch.n_records), "banded", 2000)
    bad = np.nonzero((r[ids] != oref) | (a[ids] != oalt))[0]
    assert bad.size == 0, "record %d: device (%d, %d) oracle (%d, %d)" % (ids[bad[0]], r[ids[bad[0]]], a[ids[bad[0]]], oref[bad[0]], oalt[bad[0]])


def test_read_lengths_around_the_mask_edges():
    """Reads of 186 .. 262 bases (every length) with errors, in one batch of mixed lengths: 192 / 193 (the third word's edge), 255 / 256
    (the capacity), 257 and more (declined by the first stage: band_run_kernel), against haplotypes of 201 .. 255 bases."""
    from audit_util import oracle_scores_of
    rng = np.random.default_rng(77)
    g = bytes(rng.choice(list(b"ACGT"), 40_000).tolist())
    haps, reads = [], []
    import stress_batches as SB
    for l in range(60):
        p0 = 500 + 600 * l
        pad = int(rng.integers(100, 128))
        alt_base = b"ACGT".replace(g[p0:p0 + 1], b"")[int(rng.integers(0, 3)):][:1]
        haps.append((g[p0 - pad:p0 + pad + 1], g[p0 - pad:p0] + alt_base + g[p0 + 1:p0 + pad + 1]))
        rs = []
        for c, ln in enumerate(range(186, 263)):
            s = p0 - int(rng.integers(0, ln))
            x = bytearray(g[s:s + ln])
            if rng.random() < 0.5:
                x[p0 - s] = alt_base[0]
            for e in np.nonzero(rng.random(ln) < 0.01)[0]:
                x[e] = b"ACGT"[int(rng.integers(0, 4))]
            rs.append((c, 0, bytes(x)))
        reads.append(rs)
    batch = SB.manual_batch(haps, reads, 100)
    r, a, stage, t = run_banded(batch, 100)
    ids, oref, oalt = oracle_scores_of(batch, np.arange(batch.n_records), "banded", 100)
    bad = np.nonzero((r[ids] != oref) | (a[ids] != oalt))[0]
    assert bad.size == 0, "record %d (length %d): device (%d, %d) oracle (%d, %d)" % (
        ids[bad[0]], batch.records["read_len"][ids[bad[0]]], r[ids[bad[0]]], a[ids[bad[0]]], oref[bad[0]], oalt[bad[0]])
    lens = np.repeat(batch.records["read_len"], 2)
    first = np.isin(stage, (abi.STAGE_DIAG_CERT, abi.STAGE_REFINE_CERT, abi.STAGE_BAND_CERT, abi.STAGE_DIAG_DP))
    assert first[lens <= 256].mean() > 0.9 and not first[lens > 256].any()
