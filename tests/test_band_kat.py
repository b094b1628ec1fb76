"""tests/golden/band_kat.json — known-answer vectors of the banded aligner as restated by the oracle (tools/make_band_kat.py wrote
them; INTEGRATION.md holds the Rust test a maintainer with bio 0.30.0 replays them with).  CPU: the oracle still reproduces every
vector (score, full-matrix score, band column by column) and so does the scalar model of band_sweep_kernel; GPU: the device's
banded and full scores of the same pairs, through the C-ABI.  Reference call site: src/main.rs:898-901."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    return json.load(open(os.path.join(HERE, "golden", "band_kat.json")))


def unrle(pairs):
    return np.concatenate([np.full(c, v, np.int64) for v, c in pairs])


def test_every_recollected_detail_has_discriminating_vectors(kat):
    """A crate holder who replays the file must be able to tell WHICH recollected detail a failure means: every
    vector is tagged with the details whose alternative changes its banded score or its band.  At least ten score-discriminating
    vectors per detail that CAN be discriminated; `kmer_last_anchor -> k - 1` changes no band on any vector (nor on 2 000 stress
    pairs, below): it is not a parity risk.  The tags are recomputed here (a stale file fails)."""
    import ctypes as C2
    L = oracle.lib()
    L.vtxo_set_variant.argtypes = [C2.c_int, C2.c_int]
    alt = {("lazy_extension", "0"): (0, 0), ("lazy_extension", "k"): (0, 6), ("lazy_extension", "to the matrix edge"): (0, 0x7fffffff),
           ("kmer_last_anchor", "k - 1"): (1, 5), ("no_seed", "empty band"): (2, 0), ("sdpkpp_ties", "smaller match index"): (3, 0)}
    score_hits = {}
    for v in kat["vectors"]:
        x, y = v["read"].encode("latin-1"), v["hap"].encode("latin-1")
        tagged = {(d["detail"], d["alternative"]): d for d in v["discriminates"]}
        assert ("kmer_last_anchor", "k - 1") not in tagged
        for key, (which, value) in alt.items():
            L.vtxo_set_variant(which, value)
            try:
                b = oracle.sw_banded(x, y)
                lo, hi, _ = oracle.band_create(x, y)
            finally:
                L.vtxo_set_variant(which, -1)
            moved = not (np.array_equal(lo, unrle(v["lo_rle"])) and np.array_equal(hi, unrle(v["hi_rle"])))
            assert (key in tagged) == (b != v["banded_score"] or moved), (key, v["read"][:20])
            if key in tagged:
                assert tagged[key]["banded_score_then"] == b and tagged[key]["band_changes"] == moved
                score_hits[key[0]] = score_hits.get(key[0], 0) + (b != v["banded_score"])
    assert score_hits["lazy_extension"] >= 100 and score_hits["sdpkpp_ties"] >= 10 and score_hits["no_seed"] >= 10, score_hits


def test_last_anchor_alternative_never_moves_a_band():
    """add_kmer with anchors 0 .. k - 1 instead of 0 .. k (oracle hook VTXO_VAR_LAST_ANCHOR): the cell after a chained k-mer's last
    base is anchored by add_gap's origin, by the next k-mer or by the lazy extension either way — identical bands on the stress
    distributions, tandem repeats included.  The one recollected detail that cannot be told apart, and does not need to be."""
    import ctypes as C2
    import stress_batches as SB
    from test_sweep_model import tasks_of
    L = oracle.lib()
    L.vtxo_set_variant.argtypes = [C2.c_int, C2.c_int]
    n = 0
    for gen in (SB.synthetic_batches(per_model=1, n_loci=12, reads=8), SB.repeat_rich_batches(trials=3, loci=10, reads=8),
                SB.real_sequence_batches(trials=1), SB.real_shape_batches(trials=1)):
        for label, batch, _nb in gen:
            for x, y in tasks_of(batch, 250):
                lo0, hi0, _ = oracle.band_create(x, y)
                L.vtxo_set_variant(1, 5)
                try:
                    lo, hi, _ = oracle.band_create(x, y)
                finally:
                    L.vtxo_set_variant(1, -1)
                assert np.array_equal(lo, lo0) and np.array_equal(hi, hi0), (label, x, y)
                n += 1
    assert n > 2000


def test_file_shape(kat):
    vs = kat["vectors"]
    assert len(vs) >= 300 and kat["k"] == 6 and kat["w"] == 20
    assert sum(v["banded_score"] != v["full_score"] for v in vs) >= 150          # the band matters in most of them
    assert sum(v["chain_diagonals"] > 1 for v in vs) >= 100
    for v in vs:
        assert len(unrle(v["lo_rle"])) == len(v["hap"]) + 1 == len(unrle(v["hi_rle"]))
        assert v["banded_score"] <= v["full_score"]


def test_oracle_reproduces_every_vector(kat):
    for v in kat["vectors"]:
        x, y = v["read"].encode("latin-1"), v["hap"].encode("latin-1")
        assert oracle.sw_banded(x, y) == v["banded_score"] and oracle.sw_full(x, y) == v["full_score"]
        lo, hi, cells = oracle.band_create(x, y)
        assert np.array_equal(lo, unrle(v["lo_rle"])) and np.array_equal(hi, unrle(v["hi_rle"])) and cells == v["band_cells"]
        assert oracle.sw_ranges(x, y, lo, hi) == v["banded_score"]


def test_sweep_model_reproduces_every_band(kat):
    subprocess.check_call(["make", "-C", os.path.join(HERE, "sweepmodel"), "-s"])
    L = C.CDLL(os.path.join(HERE, "sweepmodel", "libsweep_model.so"))
    L.vtxs_band.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vtxs_band.restype = C.c_int
    L.vtxs_band2.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vtxs_band2.restype = C.c_int
    done = 0
    for v in kat["vectors"]:
        x, y = v["read"].encode("latin-1"), v["hap"].encode("latin-1")
        for model in (1, 2):                                   # round 4's formulation (dp ring) and round 5's (sections per diagonal)
            lo = np.zeros(len(y) + 1, np.int32)
            hi = np.zeros(len(y) + 1, np.int32)
            if model == 1:
                rc = L.vtxs_band(x, len(x), y, len(y), 1 << 20, 1 << 20, lo.ctypes.data, hi.ctypes.data, None)
            else:
                rc = L.vtxs_band2(x, len(x), y, len(y), 1 << 20, 1 << 20, 1 << 20, lo.ctypes.data, hi.ctypes.data, None)
            if rc == 1:
                continue                                       # bytes outside ACGTN / more than 255 bases: the kernel declines those
            assert rc == 0
            assert np.array_equal(lo, unrle(v["lo_rle"])) and np.array_equal(hi, unrle(v["hi_rle"]))
            done += 1
    assert done >= 300


@pytest.mark.gpu
def test_device_scores_every_vector(kat):
    from vartrix_amd import lib
    from vartrix_amd.abi import default_config
    from stress_batches import manual_batch
    vs = kat["vectors"]
    haps = [(v["hap"].encode("latin-1"), v["hap"].encode("latin-1")) for v in vs]       # REF = ALT = the vector's haplotype
    reads = [[(0, 0, v["read"].encode("latin-1"))] for v in vs]
    batch = manual_batch(haps, reads, 4)
    for aligner, key in (("banded", "banded_score"), ("full", "full_score")):
        with lib.Context(default_config(aligner=aligner, scoring_mode="coverage", n_barcodes=4)) as ctx:
            ctx.submit(batch)
            ctx.run()
            r, a = ctx.fetch_scores()
        want = np.array([v[key] for v in vs], np.int32)
        assert np.array_equal(r, want) and np.array_equal(a, want), aligner
