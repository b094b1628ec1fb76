"""CPU check of the ALGORITHM of band_sweep_kernel (vartrix_amd/csrc/vtx_sweep.hip) — the row sweep that replaces sdpkpp's sorted
event list, the section log that replaces its per-match predecessor links, the closed-form band — restated lane-free in
tests/sweepmodel/sweep_model.cpp, against the oracle's literal restatement of bio 0.30.0's Band::create (oracle/vtx_oracle.c:
vtxo_band_create; reference call site src/main.rs:898-901).  For every task: the model's [lo, hi) per column must be the
oracle's, or the model must decline for one of its documented reasons (bytes outside ACGTN, more than 255 bases, capacities).
The device kernel is checked the same way through the C-ABI in tests/test_gpu_sweep.py (vtx_debug_bands)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle

import stress_batches as SB

HERE = os.path.dirname(os.path.abspath(__file__))
LOGCAP, SECCAP = 256, 28          # vtx_sweep_v1.hip (round 4's kernel, libvtx_dev.so)
LOGCAP2, STASH2 = 1024, 7         # vtx_sweep.hip (round 5: sections per diagonal, log in global memory, stash buckets)


@pytest.fixture(scope="module")
def model():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "sweepmodel"), "-s"])
    L = C.CDLL(os.path.join(HERE, "sweepmodel", "libsweep_model.so"))
    L.vtxs_band.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vtxs_band.restype = C.c_int
    L.vtxs_band2.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vtxs_band2.restype = C.c_int
    return L


def tasks_of(batch, limit):
    hb, rb = batch.hap_arena.tobytes(), batch.read_arena.tobytes()
    n = 0
    for loc in batch.loci:
        for ri in range(int(loc["rec_begin"]), int(loc["rec_begin"]) + int(loc["rec_count"])):
            r = batch.records[ri]
            x = rb[int(r["read_off"]):int(r["read_off"]) + int(r["read_len"])]
            for off, ln in ((int(loc["ref_off"]), int(loc["ref_len"])), (int(loc["alt_off"]), int(loc["alt_len"]))):
                if len(x) and ln:
                    yield x, hb[off:off + ln]
                    n += 1
                    if n >= limit:
                        return


def check(L, batch, label, limit, log_cap=LOGCAP, sec_cap=SECCAP):
    status = {}
    for x, y in tasks_of(batch, limit):
        lo = np.zeros(len(y) + 1, np.int32)
        hi = np.zeros(len(y) + 1, np.int32)
        st = np.zeros(3, np.int32)
        rc = L.vtxs_band(x, len(x), y, len(y), log_cap, sec_cap, lo.ctypes.data, hi.ctypes.data, st.ctypes.data)
        status[rc] = status.get(rc, 0) + 1
        assert rc in (0, 1, 2, 3), "%s: status %d" % (label, rc)
        if rc == 1:
            assert len(x) > 255 or len(y) > 255 or set(x) - set(b"ACGTN") or set(y) - set(b"ACGTN"), label
        if rc == 0:
            olo, ohi, _ = oracle.band_create(x, y)
            assert np.array_equal(lo, olo) and np.array_equal(hi, ohi), "%s: band differs (read %r, haplotype %r)" % (label, x, y)
    return status


def test_error_models_and_indels(model):
    done = 0
    for label, batch, _nb in SB.synthetic_batches(per_model=1, n_loci=24, reads=12):
        st = check(model, batch, label, 600)
        done += st.get(0, 0)
    assert done > 2000


def test_repeat_rich_genomes_beyond_the_log_capacity(model):
    """Tandem repeats over 2- to 4-letter alphabets: thousands of k-mer matches, hundreds of sections per task.  With the kernel's
    capacities some tasks are declined (status 2); without them every band must still be the oracle's."""
    seen = {}
    for label, batch, _nb in SB.repeat_rich_batches(trials=4, loci=16, reads=10):
        for k, v in check(model, batch, label, 320).items():
            seen[k] = seen.get(k, 0) + v
        check(model, batch, label + " (no capacity)", 160, log_cap=1 << 20, sec_cap=1 << 20)
    assert seen.get(0, 0) > 300 and seen.get(2, 0) > 0, seen


def test_near_repeats_real_sequence_and_real_read_shapes(model):
    done = 0
    for gen in (SB.near_repeat_batches(trials=3), SB.real_sequence_batches(trials=2), SB.real_shape_batches(trials=2)):
        for label, batch, _nb in gen:
            st = check(model, batch, label, 500)
            done += st.get(0, 0)
            assert st.get(2, 0) + st.get(3, 0) <= 0.02 * sum(st.values()), (label, st)       # the capacities fit real sequence
    assert done > 2500


def test_edge_shapes(model):
    rng = np.random.default_rng(5)
    g = bytes(rng.choice(list(b"ACGT"), 400).tolist())
    cases = [(b"ACGTA", g[:50]), (g[:150], b"ACG"), (g[10:16], g[:40]), (g[:255], g[:255]), (g[3:153], g[:201]),
             (b"A" * 150, b"A" * 201), (b"AC" * 75, b"CA" * 100), (g[:100] + b"N" * 8 + g[108:150], g[:201]),
             (g[:150], g[:100] + b"NNNNNNNN" + g[108:201]), (g[:256], g[:100]), (g[:100], g[:256]), (g[:60].lower(), g[:100])]
    for x, y in cases:
        lo = np.zeros(len(y) + 1, np.int32)
        hi = np.zeros(len(y) + 1, np.int32)
        rc = model.vtxs_band(x, len(x), y, len(y), 1 << 20, 1 << 20, lo.ctypes.data, hi.ctypes.data, None)
        if len(x) > 255 or len(y) > 255 or set(x) - set(b"ACGTN") or set(y) - set(b"ACGTN"):
            assert rc == 1
            continue
        assert rc == 0
        olo, ohi, _ = oracle.band_create(x, y)
        assert np.array_equal(lo, olo) and np.array_equal(hi, ohi), (x, y)


# ---- round 5: dp = x + 6 - o per SECTION, the latest section per diagonal, stashed END events (vtxs_band2) ----
def check2(L, batch, label, limit, log_cap=LOGCAP2, stash=STASH2, sec_cap=SECCAP):
    status, jumps, fullest = {}, 0, 0
    for x, y in tasks_of(batch, limit):
        lo = np.zeros(len(y) + 1, np.int32)
        hi = np.zeros(len(y) + 1, np.int32)
        st = np.zeros(5, np.int32)
        rc = L.vtxs_band2(x, len(x), y, len(y), log_cap, sec_cap, stash, lo.ctypes.data, hi.ctypes.data, st.ctypes.data)
        status[rc] = status.get(rc, 0) + 1
        assert rc in (0, 1, 2, 3, 5), "%s: status %d" % (label, rc)
        jumps += int(st[3]); fullest = max(fullest, int(st[4]))
        if rc == 0:
            olo, ohi, _ = oracle.band_create(x, y)
            assert np.array_equal(lo, olo) and np.array_equal(hi, ohi), "%s: band differs (read %r, haplotype %r)" % (label, x, y)
    return status, jumps, fullest


def test_section_store_on_every_distribution(model):
    """The same distributions as above through the section-store formulation: identical bands wherever it does not decline, and it
    declines (full stash bucket: status 5) only on tandem repeats.  Jumps that beat a continuation DO occur (the stash is exercised)."""
    done = jumps = 0
    for label, batch, _nb in SB.synthetic_batches(per_model=1, n_loci=24, reads=12):
        st, j, _ = check2(model, batch, label, 600)
        assert st.get(5, 0) == 0 and st.get(2, 0) == 0, (label, st)
        done += st.get(0, 0); jumps += j
    for gen in (SB.near_repeat_batches(trials=3), SB.real_sequence_batches(trials=2), SB.real_shape_batches(trials=2)):
        for label, batch, _nb in gen:
            st, j, full = check2(model, batch, label, 500)
            assert st.get(5, 0) == 0 and st.get(2, 0) + st.get(3, 0) <= 0.02 * sum(st.values()) and full <= 5, (label, st, full)
            done += st.get(0, 0); jumps += j
    rep = {}
    for label, batch, _nb in SB.repeat_rich_batches(trials=4, loci=16, reads=10):
        st, j, _ = check2(model, batch, label, 320)
        jumps += j
        for k, v in st.items():
            rep[k] = rep.get(k, 0) + v
        check2(model, batch, label + " (no capacity)", 160, log_cap=1 << 20, stash=1 << 20, sec_cap=1 << 20)
    assert done > 5000 and jumps > 50 and rep.get(0, 0) > 700 and 0 < rep.get(5, 0) < 0.05 * sum(rep.values()), (done, jumps, rep)


def test_section_store_edge_shapes(model):
    rng = np.random.default_rng(5)
    g = bytes(rng.choice(list(b"ACGT"), 400).tolist())
    cases = [(b"ACGTA", g[:50]), (g[:150], b"ACG"), (g[10:16], g[:40]), (g[:255], g[:255]), (g[3:153], g[:201]),
             (b"A" * 150, b"A" * 201), (b"AC" * 75, b"CA" * 100), (g[:100] + b"N" * 8 + g[108:150], g[:201]),
             (g[:150], g[:100] + b"NNNNNNNN" + g[108:201]), (g[:256], g[:100]), (g[:100], g[:256]), (g[:60].lower(), g[:100]),
             (b"ACG" * 50, b"ACG" * 67), (g[:60] + g[:60] + g[:30], g[:80] + g[20:80] + g[:61])]
    for x, y in cases:
        lo = np.zeros(len(y) + 1, np.int32)
        hi = np.zeros(len(y) + 1, np.int32)
        rc = model.vtxs_band2(x, len(x), y, len(y), 1 << 20, 1 << 20, 1 << 20, lo.ctypes.data, hi.ctypes.data, None)
        if len(x) > 255 or len(y) > 255 or set(x) - set(b"ACGTN") or set(y) - set(b"ACGTN"):
            assert rc == 1
            continue
        assert rc == 0
        olo, ohi, _ = oracle.band_create(x, y)
        assert np.array_equal(lo, olo) and np.array_equal(hi, ohi), (x, y)
