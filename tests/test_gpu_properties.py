"""Full-size runs (BASELINE.json configs[1]: 10k SNV loci x 5k barcodes, coverage) checked through
size-independent properties, plus an oracle spot-check on a random sample of records."""
import numpy as np
import pytest

from oracle import oracle
from vartrix_amd import lib, synth
from vartrix_amd.abi import PackedBatch, default_config

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    spec = synth.config2()
    return spec, synth.make_batch(spec)


def run(batch, cfg):
    with lib.Context(cfg) as ctx:
        ctx.submit(batch)
        ctx.run()
        return ctx.fetch_scores() + (ctx.fetch_coo(),)


@pytest.mark.parametrize("aligner", ["banded", "full"])
def test_config2_properties(big, aligner):
    spec, batch = big
    check_properties(spec, batch, aligner, "coverage")


def test_config3_full_size_properties():
    """BASELINE.json configs[2], the configuration the metric is quoted on: 100 k SNV loci x 10 k barcodes, consensus
    mode, banded aligner — 24.3 M scored reads through the same size-independent properties + the oracle spot check."""
    spec = synth.config3()
    check_properties(spec, synth.make_batch(spec), "banded", "consensus", sample=2000)


def check_properties(spec, batch, aligner, mode, sample=3000):
    cfg = default_config(aligner=aligner, scoring_mode=mode, n_barcodes=spec.n_barcodes)
    ref, alt, coo = run(batch, cfg)
    n = batch.n_records
    assert ref.shape == (n,) and ref.min() >= 0 and alt.min() >= 0
    assert ref.max() <= spec.read_len and alt.max() <= spec.read_len          # local score <= read length
    # SNV haplotypes differ in one base: full-matrix scores differ by at most match - mismatch = 6
    # (not an invariant of the banded flavour: the two haplotypes get their own seed chains and bands)
    if aligner == "full":
        assert np.abs(ref - alt).max() <= 6
    # evaluate_scores (:1019-1030) recomputed on the host from the device scores == the histogram on the device
    none = (ref < 25) & (alt < 25)
    calls_ref = int(((ref > alt) & ~none).sum())
    calls_alt = int(((alt > ref) & ~none).sum())
    calls_unk = int(((alt == ref) & ~none).sum())
    assert (int(coo["ref"].sum()), int(coo["alt"].sum())) == (calls_ref, calls_alt)
    if mode == "coverage":      # consensus drops the groups that hold only UNKNOWN calls, their counts go with them (below)
        assert int(coo["unk"].sum()) == calls_unk
    # one triplet per (locus, cell) group, in merge-loop order (row asc, then col asc)
    key = coo["row"].astype(np.int64) * spec.n_barcodes + coo["col"]
    assert np.all(np.diff(key) > 0)
    rec_key = np.repeat(batch.loci["row"].astype(np.int64), batch.loci["rec_count"]) * spec.n_barcodes + batch.records["cell_index"]
    groups, inv = np.unique(rec_key, return_inverse=True)
    if mode == "coverage":      # every (locus, cell) group is emitted, explicit zeros included (:1147-1164)
        assert np.array_equal(key, groups)
    else:                       # consensus keeps the groups that hold a REF or an ALT call (:1111-1129)
        g_ref = np.bincount(inv, weights=((ref > alt) & ~none), minlength=len(groups))
        g_alt = np.bincount(inv, weights=((alt > ref) & ~none), minlength=len(groups))
        g_unk = np.bincount(inv, weights=((alt == ref) & ~none), minlength=len(groups))
        kept = (g_ref > 0) | (g_alt > 0)
        assert np.array_equal(key, groups[kept])
        assert np.array_equal(coo["ref"], g_ref[kept].astype(np.uint32)) and np.array_equal(coo["alt"], g_alt[kept].astype(np.uint32))
        assert np.array_equal(coo["unk"], g_unk[kept].astype(np.uint32))
    if mode == "coverage":      # values are the counts (:1160-1161)
        assert np.array_equal(coo["value"], coo["alt"].astype(np.float64)) and np.array_equal(coo["ref_value"], coo["ref"].astype(np.float64))
    else:                       # consensus (:1111-1129): 1 ref only, 2 alt only, 3 both; groups without ref/alt calls are dropped
        want = np.where((coo["ref"] > 0) & (coo["alt"] > 0), 3.0, np.where(coo["alt"] > 0, 2.0, 1.0))
        assert np.array_equal(coo["value"], want) and np.all((coo["ref"] > 0) | (coo["alt"] > 0))
    # oracle spot check: random records, bit-exact
    rng = np.random.default_rng(1)
    pick = np.sort(rng.choice(n, sample, replace=False))
    rec_locus = np.repeat(np.arange(batch.n_loci), batch.loci["rec_count"])
    for r in pick:
        rec = batch.records[r]
        loc = batch.loci[rec_locus[r]]
        read = bytes(batch.read_arena[rec["read_off"]:rec["read_off"] + rec["read_len"]])
        rh = bytes(batch.hap_arena[loc["ref_off"]:loc["ref_off"] + loc["ref_len"]])
        ah = bytes(batch.hap_arena[loc["alt_off"]:loc["alt_off"] + loc["alt_len"]])
        f = oracle.sw_banded if aligner == "banded" else oracle.sw_full
        assert (f(read, rh), f(read, ah)) == (int(ref[r]), int(alt[r])), r


def test_shard_invariance_and_determinism(big):
    """Loci shard independently (src/main.rs:284-291): 3 uneven shards run separately give the same triplets."""
    spec, batch = big
    cfg = default_config(aligner="banded", scoring_mode="consensus", n_barcodes=spec.n_barcodes)
    ref, alt, coo = run(batch, cfg)
    ref2, alt2, coo2 = run(batch, cfg)
    assert np.array_equal(ref, ref2) and np.array_equal(alt, alt2)
    assert all(np.array_equal(coo[k].view(np.uint8), coo2[k].view(np.uint8)) for k in coo)
    cuts = [0, 1234, 7001, batch.n_loci]
    parts = [run(batch.slice_loci(a, b), cfg)[2] for a, b in zip(cuts[:-1], cuts[1:])]
    for k in coo:
        assert np.array_equal(np.concatenate([p[k] for p in parts]).view(np.uint8), coo[k].view(np.uint8)), k


def test_ref_alt_swap_symmetry(big):
    """Swapping the REF and ALT haplotypes of every locus swaps the two scores of every record."""
    spec, batch = big
    sub = batch.slice_loci(0, 1500)
    loci = sub.loci.copy()
    loci["ref_off"], loci["alt_off"] = sub.loci["alt_off"], sub.loci["ref_off"]
    loci["ref_len"], loci["alt_len"] = sub.loci["alt_len"], sub.loci["ref_len"]
    swapped = PackedBatch(loci, sub.records, sub.hap_arena, sub.read_arena)
    for aligner in ("banded", "full"):
        cfg = default_config(aligner=aligner, scoring_mode="coverage", n_barcodes=spec.n_barcodes)
        ref, alt, coo = run(sub, cfg)
        sref, salt, scoo = run(swapped, cfg)
        assert np.array_equal(ref, salt) and np.array_equal(alt, sref)
        assert np.array_equal(coo["alt"], scoo["ref"]) and np.array_equal(coo["ref"], scoo["alt"])


@pytest.mark.parametrize("aligner", ["banded", "full"])
def test_config5_shape_against_oracle(aligner):
    """BASELINE.json configs[4] at reduced size: mixed SNV + indel (<= 20 bp) loci, alt_frac with UMI collapse —
    every score and every matrix value against the oracle (alt_frac within the stated 1e-6; it is in fact
    bit-identical), then the same batch as two shards (the multi-GPU path's partition) gives the same matrix."""
    import os
    from vartrix_amd import shard
    spec = synth.config5(2500)
    batch = synth.make_batch(spec)
    assert len(np.unique(batch.loci["ref_len"] - batch.loci["alt_len"].astype(np.int64))) > 10      # indels both ways
    cfg = default_config(aligner=aligner, scoring_mode="alt_frac", use_umi=1, n_barcodes=spec.n_barcodes)
    ref, alt, coo = run(batch, cfg)
    oref, oalt = oracle.batch_scores(batch, cfg, threads=os.cpu_count() or 8)
    assert np.array_equal(ref, oref) and np.array_equal(alt, oalt)
    ocoo = oracle.batch_reduce(batch, cfg, oref, oalt)
    for k in ("row", "col", "alt", "ref", "unk"):
        assert np.array_equal(coo[k], ocoo[k]), k
    both = ~(np.isnan(coo["value"]) | np.isnan(ocoo["value"]))
    assert np.array_equal(np.isnan(coo["value"]), np.isnan(ocoo["value"]))
    assert np.max(np.abs(coo["value"][both] - ocoo["value"][both]), initial=0.0) <= 1e-6
    assert np.array_equal(coo["value"].view(np.uint64), ocoo["value"].view(np.uint64))
    # UMI collapse really collapsed something, and some UMIs were decided by the 0.75 rule
    assert coo["alt"].sum() + coo["ref"].sum() + coo["unk"].sum() < ((oref >= 25) | (oalt >= 25)).sum()
    parts = []
    for lo, hi in shard.partition_loci(batch, 2):
        sub = batch.slice_loci(lo, hi)
        parts.append(run(sub, cfg)[2])
    for k in ("row", "col", "alt", "ref", "unk"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), coo[k]), k


def test_config4_shape_many_barcodes():
    """BASELINE.json configs[3] at reduced size: 50 k barcodes (matrix columns up to 49 999), rows sharded in 8 and
    concatenated — through the host-prepared and the device-prepared submit paths, against the oracle."""
    import os
    from oracle import prep
    from vartrix_amd import shard
    spec = synth.SynthSpec(n_loci=1600, n_barcodes=50_000, reads_per_locus=64, seed=44)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="banded", scoring_mode="consensus", n_barcodes=spec.n_barcodes)
    ref, alt, coo = run(batch, cfg)
    oref, oalt = oracle.batch_scores(batch, cfg, threads=os.cpu_count() or 8)
    ocoo = oracle.batch_reduce(batch, cfg, oref, oalt)
    assert np.array_equal(ref, oref) and np.array_equal(alt, oalt)
    for k in ("row", "col", "alt", "ref", "unk", "value"):
        assert np.array_equal(coo[k], ocoo[k]), k
    assert coo["col"].max() > 49_000
    parts = [run(batch.slice_loci(lo, hi), cfg)[2] for lo, hi in shard.partition_loci(batch, 8)]
    for k in ("row", "col", "value"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), coo[k]), k
    raw, barcodes = synth.make_raw(batch, spec.n_barcodes, False, seed=3)
    with lib.Context(default_config(aligner="banded", scoring_mode="consensus", n_barcodes=len(barcodes))) as ctx:
        ctx.set_barcodes(barcodes)
        st = ctx.submit_raw(raw)
        ctx.run()
        rcoo = ctx.fetch_coo()
    want, wst = prep.prep_raw(raw, barcodes, False)
    assert int(st.kept) == want.n_records and int(st.num_not_cell_bc) == wst["num_not_cell_bc"]
    wcoo = run(want, cfg)[2]
    for k in ("row", "col", "value"):
        assert np.array_equal(rcoo[k], wcoo[k]), k


def test_config4_full_size_properties():
    """BASELINE.json configs[3] at FULL size on one GPU (100 k SNV loci x 50 k barcodes, consensus; the 8-GPU run cuts exactly
    this workload over the ranks): the size-independent properties + the oracle spot check, and the matrix summary that
    `bench.py --gpus N` compares its gathered result with (profiles/expected_results.json)."""
    import json
    import os
    spec = synth.config4()
    batch = synth.make_batch(spec)
    check_properties(spec, batch, "banded", "consensus", sample=1000)
    exp_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "expected_results.json")
    exp = json.load(open(exp_path)).get(spec.name + ", consensus mode, banded aligner")
    assert exp is not None
    cfg = default_config(aligner="banded", scoring_mode="consensus", n_barcodes=spec.n_barcodes)
    coo = run(batch, cfg)[2]
    assert len(coo["row"]) == exp["nnz"]


def test_config5_full_size_properties():
    """BASELINE.json configs[4] at FULL size on one GPU: 100 k loci, 70 % SNV / 15 % insertions / 15 % deletions (<= 20 bp),
    alt_frac with UMI collapse.  The whole reduction is recomputed on the host with numpy from the device's per-read scores
    (evaluate_scores :1019-1030, the 0.75 rule of parse_scores :1070-1081 as 4a >= 3t, alt_frac :1131-1145 incl. NaN for 0 / 0)
    and must equal the device's triplets bit for bit; a sample of records is scored by the oracle."""
    spec = synth.config5()
    batch = synth.make_batch(spec)
    assert len(np.unique(batch.loci["ref_len"].astype(np.int64) - batch.loci["alt_len"])) > 30
    cfg = default_config(aligner="banded", scoring_mode="alt_frac", use_umi=1, n_barcodes=spec.n_barcodes)
    ref, alt, coo = run(batch, cfg)
    n = batch.n_records
    assert ref.min() >= 0 and alt.min() >= 0 and max(ref.max(), alt.max()) <= spec.read_len
    none = (ref < 25) & (alt < 25)
    is_r, is_a, is_u = (ref > alt) & ~none, (alt > ref) & ~none, (alt == ref) & ~none
    row = np.repeat(batch.loci["row"].astype(np.int64), batch.loci["rec_count"])
    cell_key = row * spec.n_barcodes + batch.records["cell_index"]
    # records are sorted by (locus, cell, umi): groups are runs
    umi = batch.records["umi_id"].astype(np.int64)
    head_u = np.ones(n, bool)
    head_u[1:] = (cell_key[1:] != cell_key[:-1]) | (umi[1:] != umi[:-1])
    gid = np.cumsum(head_u) - 1
    ng = int(gid[-1]) + 1
    r = np.bincount(gid, weights=is_r, minlength=ng).astype(np.int64)
    a = np.bincount(gid, weights=is_a, minlength=ng).astype(np.int64)
    u = np.bincount(gid, weights=is_u, minlength=ng).astype(np.int64)
    t = r + a + u
    call_a = (t > 0) & (4 * a >= 3 * t)
    call_r = (t > 0) & ~call_a & (4 * r >= 3 * t)
    call_u = (t > 0) & ~call_a & ~call_r
    g_cell = cell_key[head_u]
    cells, inv = np.unique(g_cell, return_inverse=True)
    ca = np.bincount(inv, weights=call_a, minlength=len(cells)).astype(np.uint32)
    cr = np.bincount(inv, weights=call_r, minlength=len(cells)).astype(np.uint32)
    cu = np.bincount(inv, weights=call_u, minlength=len(cells)).astype(np.uint32)
    key = coo["row"].astype(np.int64) * spec.n_barcodes + coo["col"]
    assert np.array_equal(key, cells)                      # alt_frac emits every (locus, cell) group, in merge-loop order
    assert np.array_equal(coo["alt"], ca) and np.array_equal(coo["ref"], cr) and np.array_equal(coo["unk"], cu)
    with np.errstate(invalid="ignore", divide="ignore"):
        want = ca.astype(np.float64) / (ca.astype(np.float64) + cr + cu)
    assert np.array_equal(coo["value"].view(np.uint64) & 0x7fffffffffffffff, want.view(np.uint64) & 0x7fffffffffffffff)   # NaN where 0 / 0 (sign aside)
    assert np.isnan(coo["value"]).sum() == int(((ca + cr + cu) == 0).sum())
    assert (t > 1).sum() > 100000 and (call_u.sum() > 0)    # UMI families really collapsed, some by the 0.75 rule into UNKNOWN
    rng = np.random.default_rng(2)
    rec_locus = np.repeat(np.arange(batch.n_loci), batch.loci["rec_count"])
    for k in np.sort(rng.choice(n, 1500, replace=False)):
        rec, loc = batch.records[k], batch.loci[rec_locus[k]]
        read = bytes(batch.read_arena[rec["read_off"]:rec["read_off"] + rec["read_len"]])
        rh = bytes(batch.hap_arena[loc["ref_off"]:loc["ref_off"] + loc["ref_len"]])
        ah = bytes(batch.hap_arena[loc["alt_off"]:loc["alt_off"] + loc["alt_len"]])
        assert (oracle.sw_banded(read, rh), oracle.sw_banded(read, ah)) == (int(ref[k]), int(alt[k])), k


def test_config3_full_size_stage_audit():
    """BASELINE.json configs[2] at full size, every alignment audited: the banded and the full flavour run on the
    device, the stage byte of every one of the 48.6 M banded alignments is fetched, and
      * every alignment whose banded score differs from its full-matrix score was decided by a DP stage — never by a certificate
        (cert <= banded <= full: a certificate that equals the upper bound of the FULL score cannot coexist with banded < full);
      * ALL of those alignments, ALL alignments the certificate stages did not decide, and a 1 % random sample of the records
        are compared with the oracle (reference call site src/main.rs:898-901; the scores of Scores.ref_score / alt_score, :926-927)."""
    from vartrix_amd import abi
    from audit_util import assert_stage_invariant, oracle_scores_of, stage_report
    spec = synth.config3()
    batch = synth.make_batch(spec)
    out = {}
    for aligner in ("banded", "full"):
        with lib.Context(default_config(aligner=aligner, scoring_mode="consensus", n_barcodes=spec.n_barcodes)) as ctx:
            ctx.submit(batch)
            if aligner == "banded":
                ctx.set_stage_trace(True)
                ctx.set_poison(-4242)
            ctx.run()
            out[aligner] = ctx.fetch_scores() + ((ctx.fetch_stage(),) if aligner == "banded" else ())
    rb, ab, stage = out["banded"]
    rf, af = out["full"]
    assert not (rb == -4242).any() and not (ab == -4242).any()
    differ = assert_stage_invariant(stage, (rb, ab), (rf, af), "config 3")
    by_cert = np.isin(stage, (abi.STAGE_DIAG_CERT, abi.STAGE_REFINE_CERT, abi.STAGE_UNKNOWN))
    print("config 3: %d alignments, banded != full on %d; stages %s" % (len(stage), int(differ.sum()), stage_report(stage)))
    assert by_cert.mean() > 0.99                                                  # the headline's premise
    rng = np.random.default_rng(7)
    recs = np.unique(np.concatenate([np.nonzero(differ)[0] >> 1, np.nonzero(~by_cert)[0] >> 1,
                                     rng.choice(batch.n_records, batch.n_records // 100, replace=False)]))
    ids, oref, oalt = oracle_scores_of(batch, recs, "banded", spec.n_barcodes)
    bad = np.nonzero((rb[ids] != oref) | (ab[ids] != oalt))[0]
    assert bad.size == 0, "record %d: device (%d, %d) oracle (%d, %d), stages %s" % (
        ids[bad[0]], rb[ids[bad[0]]], ab[ids[bad[0]]], oref[bad[0]], oalt[bad[0]], stage[2 * ids[bad[0]]:2 * ids[bad[0]] + 2])
    print("config 3: %d records (%d alignments) compared with the oracle: all equal" % (len(ids), 2 * len(ids)))
