"""GPU parity of ``vtx_submit_raw`` (device-side barcode lookup, UMI grouping, sort — SURVEY §8f rank 2)
against the CPU restatement in ``oracle/prep.py`` of reference ``src/main.rs:697-718, :737-750, :867-894,
:932, :1047-1057``: same filter counters, same prepared records (up to the order inside a UMI group, which
no result depends on), same matrix as the host-prepared path, bit-exact."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle, prep
from vartrix_amd import lib, synth
from vartrix_amd.abi import (LOCUS_DTYPE, RAW_RECORD_DTYPE, TAG_MISSING, VTX_E_INVAL, VTX_E_STATE, RawBatch,
                             default_config)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_prepared(raw, barcodes, cfg, run=True):
    with lib.Context(cfg) as ctx:
        ctx.set_barcodes(barcodes)
        stats = ctx.submit_raw(raw)
        recs, begin, count = ctx.fetch_records()
        coo = None
        if run:
            ctx.run()
            coo = ctx.fetch_coo()
    return stats, recs, begin, count, coo


def host_prepared_coo(batch, cfg):
    with lib.Context(cfg) as ctx:
        ctx.submit(batch)
        ctx.run()
        return ctx.fetch_coo()


def check(raw, barcodes, cfg):
    stats, recs, begin, count, coo = device_prepared(raw, barcodes, cfg)
    want, wstats = prep.prep_raw(raw, barcodes, bool(cfg.use_umi))
    assert int(stats.num_not_cell_bc) == wstats["num_not_cell_bc"]
    assert int(stats.num_non_umi) == wstats["num_non_umi"]
    assert int(stats.kept) == want.n_records
    assert np.array_equal(begin, want.loci["rec_begin"]) and np.array_equal(count, want.loci["rec_count"])
    assert prep.canonical_records(recs, begin, count) == prep.canonical_records(
        want.records, want.loci["rec_begin"], want.loci["rec_count"])
    # device order: (cell, umi group) non-decreasing inside every locus — what vtx_submit demands of a host
    for b0, c in zip(begin, count):
        r = recs[int(b0):int(b0) + int(c)]
        key = r["cell_index"].astype(np.int64) << 32 | r["umi_id"].astype(np.int64)
        assert np.all(np.diff(key) >= 0)
    wcoo = host_prepared_coo(want, cfg)
    for k in ("row", "col", "alt", "ref", "unk"):
        assert np.array_equal(coo[k], wcoo[k]), k
    assert np.array_equal(coo["value"].view(np.uint64), wcoo["value"].view(np.uint64))
    assert np.array_equal(coo["ref_value"], wcoo["ref_value"])
    return stats


@pytest.mark.parametrize("aligner", ["full", "banded"])
@pytest.mark.parametrize("umi,mode", [(0, "consensus"), (1, "alt_frac"), (1, "coverage"), (0, "coverage")])
def test_raw_batch_matches_oracle_prep(umi, mode, aligner):
    spec = synth.SynthSpec(n_loci=120, n_barcodes=200, reads_per_locus=60, read_len=120, padding=80, indel_frac=0.3,
                           use_umi=bool(umi), read_len_jitter=40, seed=31 + umi)
    batch = synth.make_batch(spec)
    raw, barcodes = synth.make_raw(batch, spec.n_barcodes, bool(umi), seed=9)
    cfg = default_config(aligner=aligner, scoring_mode=mode, use_umi=umi, n_barcodes=len(barcodes))
    stats = check(raw, barcodes, cfg)
    assert stats.hash_rounds == 1 and stats.num_not_cell_bc > 0


def test_duplicated_barcodes_keep_first_index():
    """load_barcodes keeps the first index of a repeated line (src/main.rs:704-710)."""
    spec = synth.SynthSpec(n_loci=40, n_barcodes=50, reads_per_locus=40, use_umi=True, seed=3)
    batch = synth.make_batch(spec)
    raw, barcodes = synth.make_raw(batch, spec.n_barcodes, True, seed=4, dup_barcodes=20)
    assert len(barcodes) == 70 and barcodes[50:] == barcodes[:20]
    cfg = default_config(aligner="full", scoring_mode="coverage", use_umi=1, n_barcodes=len(barcodes))
    check(raw, barcodes, cfg)


def test_read_length_mix_builds_every_work_list():
    """Reads from 1 to 1000 bases: every kernel shape gets a work list built on the device."""
    spec = synth.SynthSpec(n_loci=24, n_barcodes=30, reads_per_locus=40, read_len=520, read_len_jitter=480, padding=520,
                           seed=12)
    batch = synth.make_batch(spec)
    raw, barcodes = synth.make_raw(batch, spec.n_barcodes, False, seed=5)
    cfg = default_config(aligner="banded", scoring_mode="consensus", n_barcodes=len(barcodes))
    check(raw, barcodes, cfg)


def test_umi_hash_collisions_are_detected_and_reseeded():
    """With the test hook the first two rounds hash every UMI to 0: distinct UMIs of a cell then share a hash,
    the byte comparison must notice, and the third round (real hash) must give the exact grouping."""
    code = r"""
import numpy as np
from oracle import prep
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
spec = synth.SynthSpec(n_loci=30, n_barcodes=10, reads_per_locus=60, use_umi=True, seed=8)
batch = synth.make_batch(spec)
raw, barcodes = synth.make_raw(batch, 10, True, seed=2)
cfg = default_config(aligner="full", scoring_mode="alt_frac", use_umi=1, n_barcodes=10)
with lib.Context(cfg) as ctx:
    ctx.set_barcodes(barcodes)
    st = ctx.submit_raw(raw)
    recs, begin, count = ctx.fetch_records()
want, _ = prep.prep_raw(raw, barcodes, True)
assert st.hash_rounds == 3, st.hash_rounds
assert prep.canonical_records(recs, begin, count) == prep.canonical_records(want.records, want.loci["rec_begin"], want.loci["rec_count"])
print("ROUNDS", st.hash_rounds)
"""
    env = dict(os.environ, VTX_LIB_VARIANT="dev", VTX_PREP_WEAK_ROUNDS="2", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ROUNDS 3" in out.stdout


def test_edge_cases():
    cfg = default_config(aligner="full", scoring_mode="coverage", use_umi=1, n_barcodes=3)
    barcodes = [b"AAAA-1", b"", b"CCCC-1"]           # an empty byte string is a legal list entry
    hap = np.frombuffer(b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT" * 2, np.uint8)
    reads = np.frombuffer(b"ACGTACGTACGTACGTACGTACGTACGTACGT", np.uint8)
    tags = np.frombuffer(b"AAAA-1CCCC-1GGGG-1UMI1UMI2", np.uint8)
    loci = np.zeros(3, LOCUS_DTYPE)
    loci["row"] = [0, 1, 2]
    loci["ref_off"], loci["ref_len"], loci["alt_off"], loci["alt_len"] = 0, 40, 40, 40
    loci["rec_begin"] = [0, 0, 5]
    loci["rec_count"] = [0, 5, 2]                    # locus 0 has no reads at all
    raw = np.zeros(7, RAW_RECORD_DTYPE)
    raw["read_off"], raw["read_len"] = 0, 32
    #            AAAA/UMI1  ""/UMI1   GGGG(unlisted) CCCC/no-UB  AAAA/UMI2 | locus 2: both dropped
    raw["bc_off"] = [0, 0, 12, 6, 0, 12, 6]
    raw["bc_len"] = [6, 0, 6, 6, 6, 6, 6]
    raw["umi_off"] = [18, 18, 18, 0, 22, 18, 0]
    raw["umi_len"] = [4, 4, 4, TAG_MISSING, 4, 4, TAG_MISSING]
    rb = RawBatch(loci, raw, hap, reads, tags)
    stats, recs, begin, count, coo = device_prepared(rb, barcodes, cfg)
    assert (int(stats.num_not_cell_bc), int(stats.num_non_umi), int(stats.kept)) == (2, 2, 3)
    assert list(count) == [0, 3, 0] and list(begin) == [0, 0, 3]
    assert list(recs["cell_index"]) == [0, 0, 1]
    assert recs["umi_id"][0] != recs["umi_id"][1]
    want, _ = prep.prep_raw(rb, barcodes, True)
    wcoo = host_prepared_coo(want, cfg)
    assert np.array_equal(coo["row"], wcoo["row"]) and np.array_equal(coo["col"], wcoo["col"])
    # empty batch, and a batch whose every record is dropped
    empty = RawBatch(np.zeros(0, LOCUS_DTYPE), np.zeros(0, RAW_RECORD_DTYPE), np.zeros(0, np.uint8), np.zeros(0, np.uint8),
                     np.zeros(0, np.uint8))
    stats, recs, begin, count, coo = device_prepared(empty, barcodes, cfg)
    assert int(stats.kept) == 0 and coo["row"].size == 0
    raw2 = raw.copy()
    raw2["bc_off"], raw2["bc_len"] = 12, 6
    stats, recs, begin, count, coo = device_prepared(RawBatch(loci, raw2, hap, reads, tags), barcodes, cfg)
    assert int(stats.kept) == 0 and int(stats.num_not_cell_bc) == 7 and coo["row"].size == 0


def test_errors():
    cfg = default_config(aligner="full", n_barcodes=2)
    spec = synth.SynthSpec(n_loci=4, n_barcodes=2, reads_per_locus=8, seed=1)
    raw, barcodes = synth.make_raw(synth.make_batch(spec), 2, False, seed=1)
    with lib.Context(cfg) as ctx:
        with pytest.raises(lib.VtxError) as e:
            ctx.submit_raw(raw)                                   # no barcode list yet
        assert e.value.status == VTX_E_STATE
        with pytest.raises(lib.VtxError) as e:
            ctx.set_barcodes(barcodes + [b"X"])                   # count differs from cfg.n_barcodes
        assert e.value.status == VTX_E_INVAL
        ctx.set_barcodes(barcodes)
        bad = RawBatch(raw.loci, raw.records.copy(), raw.hap_arena, raw.read_arena, raw.tag_arena)
        bad.records["bc_off"][3] = raw.tag_arena.size             # tag bytes outside the arena
        with pytest.raises(lib.VtxError) as e:
            ctx.submit_raw(bad)
        assert e.value.status == VTX_E_INVAL
        with pytest.raises(lib.VtxError):
            ctx.run()                                             # the failed submit left nothing resident
        ctx.submit_raw(raw)
        ctx.run()


def test_large_raw_batch_properties():
    """2.4 M records: counters, group structure and matrix equal the host-prepared path (no per-record oracle)."""
    spec = synth.SynthSpec(n_loci=10_000, n_barcodes=5_000, reads_per_locus=240, use_umi=True, seed=20260926)
    batch = synth.make_batch(spec)
    raw, barcodes = synth.make_raw(batch, spec.n_barcodes, True, seed=6)
    cfg = default_config(aligner="full", scoring_mode="alt_frac", use_umi=1, n_barcodes=len(barcodes))
    stats, recs, begin, count, coo = device_prepared(raw, barcodes, cfg)
    assert int(stats.kept) == batch.n_records
    assert int(stats.kept) + int(stats.num_not_cell_bc) + int(stats.num_non_umi) == raw.n_records
    assert np.array_equal(count, batch.loci["rec_count"])
    # same multiset of (locus, cell, read_off) and the same number of UMI groups as the generator's batch
    loc = np.repeat(np.arange(batch.n_loci), batch.loci["rec_count"])
    a = np.stack([loc, batch.records["cell_index"], batch.records["read_off"]], 1)
    b = np.stack([np.repeat(np.arange(batch.n_loci), count), recs["cell_index"], recs["read_off"]], 1)
    assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])])
    groups = lambda r, l: np.unique(np.stack([l, r["cell_index"], r["umi_id"]], 1), axis=0).shape[0]
    assert groups(recs, b[:, 0]) == groups(batch.records, loc)
    wcoo = host_prepared_coo(batch, cfg)
    for k in ("row", "col", "alt", "ref", "unk"):
        assert np.array_equal(coo[k], wcoo[k]), k
    assert np.array_equal(coo["value"].view(np.uint64), wcoo["value"].view(np.uint64))
    print("prep_ms %.2f for %d raw records" % (stats.prep_ms, raw.n_records))
