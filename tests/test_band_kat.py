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
    done = 0
    for v in kat["vectors"]:
        x, y = v["read"].encode("latin-1"), v["hap"].encode("latin-1")
        lo = np.zeros(len(y) + 1, np.int32)
        hi = np.zeros(len(y) + 1, np.int32)
        rc = L.vtxs_band(x, len(x), y, len(y), 1 << 20, 1 << 20, lo.ctypes.data, hi.ctypes.data, None)
        if rc == 1:
            continue                                           # bytes outside ACGTN / more than 255 bases: the kernel declines those
        assert rc == 0
        assert np.array_equal(lo, unrle(v["lo_rle"])) and np.array_equal(hi, unrle(v["hi_rle"]))
        done += 1
    assert done >= 150


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
