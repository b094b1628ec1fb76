"""GPU tests of round 4's robust path of the banded flavour (reference call site src/main.rs:898-901):

  * band_sweep_kernel's BAND, column by column, against the oracle's Band::create (vtx_debug_bands): the kernel is checked on
    what it computes, not only on the score the masked DP derives from it;
  * the per-task stage byte (vtx_fetch_stage) and the invariant it audits: banded != full  =>  decided by a DP stage, never by a
    certificate or by the full-matrix check;
  * poisoned score arrays: every stage writes every score it is responsible for, run after run;
  * the round-3 path (VTX_BAND_LEGACY=1) and the hooks of the new one give identical scores.
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import oracle
from vartrix_amd import abi, lib, synth
from vartrix_amd.abi import default_config

import stress_batches as SB
from audit_util import assert_stage_invariant, stage_report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def model_lib():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "sweepmodel"), "-s"])
    L = C.CDLL(os.path.join(HERE, "sweepmodel", "libsweep_model.so"))
    L.vtxs_band.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vtxs_band.restype = C.c_int
    L.vtxs_band2.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vtxs_band2.restype = C.c_int
    return L


def bands_vs_oracle(batch, nb, label, max_tasks=6000, tier=None):
    """Every task of the batch through vtx_debug_bands; the band of every accepted task equals the oracle's, and a task is declined
    exactly when the CPU model of the kernel (same capacities) declines it.  tier None: band_sweep_kernel (round 5: the section
    store per diagonal, 1 024 log entries in global memory, seven stash entries per END row — model vtxs_band2); tier 0 / 1: round 4's
    kernel in libvtx_dev.so (VTX_SWEEP_V1=1; 256 / 1 024 log entries — model vtxs_band)."""
    M = model_lib()
    n_tasks = min(2 * batch.n_records, max_tasks)
    tasks = np.arange(n_tasks, dtype=np.uint32)
    stride = int(max(batch.loci["ref_len"].max(), batch.loci["alt_len"].max())) + 1
    if tier is not None:
        os.environ["VTX_SWEEP_V1"] = "1"                           # (both read at every vtx_debug_bands call of libvtx_dev.so)
        os.environ["VTX_SWEEP_TIER"] = str(tier)
    try:
        with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=nb), variant=None if tier is None else "dev") as ctx:
            ctx.submit(batch)
            lo, hi, status = ctx.debug_bands(tasks, stride)
    finally:
        os.environ.pop("VTX_SWEEP_TIER", None)
        os.environ.pop("VTX_SWEEP_V1", None)
    log_cap = 1024 if tier != 0 else 256
    rec_locus = np.repeat(np.arange(batch.n_loci), batch.loci["rec_count"])
    hb, rb = batch.hap_arena.tobytes(), batch.read_arena.tobytes()
    declined = 0
    for t in range(n_tasks):
        rec = batch.records[t >> 1]
        loc = batch.loci[rec_locus[t >> 1]]
        x = rb[int(rec["read_off"]):int(rec["read_off"]) + int(rec["read_len"])]
        off, ln = (int(loc["alt_off"]), int(loc["alt_len"])) if t & 1 else (int(loc["ref_off"]), int(loc["ref_len"]))
        y = hb[off:off + ln]
        mlo = np.zeros(len(y) + 1, np.int32)
        mhi = np.zeros(len(y) + 1, np.int32)
        if not len(x) or not len(y):
            rc = 0
        elif tier is None:
            rc = M.vtxs_band2(x, len(x), y, len(y), log_cap, 28, 7, mlo.ctypes.data, mhi.ctypes.data, None)
        else:
            rc = M.vtxs_band(x, len(x), y, len(y), log_cap, 28, mlo.ctypes.data, mhi.ctypes.data, None)
        assert (status[t] != 0) == (rc != 0), "%s: task %d device status %d, model %d" % (label, t, status[t], rc)
        if status[t]:
            declined += 1
            continue
        if not len(x) or not len(y):
            continue
        olo, ohi, _ = oracle.band_create(x, y)
        dlo = lo[t, :len(y) + 1].astype(np.int64)
        dhi = hi[t, :len(y) + 1].astype(np.int64)
        empty = ohi <= olo
        assert np.array_equal(dhi[empty], np.zeros(int(empty.sum()), np.int64)), "%s: task %d: column outside the band" % (label, t)
        assert np.array_equal(dlo[~empty], olo[~empty]) and np.array_equal(dhi[~empty], ohi[~empty]), \
            "%s: task %d (read %d bases, haplotype %d): band differs from the oracle's" % (label, t, len(x), len(y))
    return n_tasks, declined


def test_bands_of_clean_noisy_and_indel_batches():
    tot = dec = 0
    for label, batch, nb in SB.synthetic_batches(per_model=1, n_loci=40, reads=16):
        if max(int(batch.loci["ref_len"].max()), int(batch.loci["alt_len"].max())) > 255:
            continue                                          # (the run takes the round-3 path for such a batch)
        n, d = bands_vs_oracle(batch, nb, label, 1500)
        tot += n
        dec += d
    assert tot > 5000 and dec == 0


def test_bands_of_repeats_real_sequence_and_read_shapes():
    tot = dec = 0
    for gen in (SB.repeat_rich_batches(trials=4, loci=20, reads=12, pad_range=(30, 120)), SB.near_repeat_batches(trials=2),
                SB.real_sequence_batches(trials=2), SB.real_shape_batches(trials=2)):
        for label, batch, nb in gen:
            n, d = bands_vs_oracle(batch, nb, label, 1200)
            tot += n
            dec += d
    print("band parity: %d tasks, %d declined (capacities, bytes outside ACGTN)" % (tot, dec))
    assert tot > 8000 and 0 < dec < 0.25 * tot


def test_bands_of_round4s_kernel():
    """Round 4's kernel (libvtx_dev.so, VTX_SWEEP_V1=1: the A/B reference of the timing campaigns), both log capacities, on tandem
    repeats: the same bands."""
    for tier in (0, 1):
        tot = dec = 0
        for label, batch, nb in SB.repeat_rich_batches(trials=3, loci=20, reads=12, pad_range=(30, 120)):
            n, d = bands_vs_oracle(batch, nb, label, 600, tier=tier)
            tot += n
            dec += d
        print("round-4 kernel, tier %d: %d tasks, %d declined" % (tier, tot, dec))
        assert tot > 1200 and (dec < 0.05 * tot if tier else dec > 0)


def run_both(batch, nb, trace=True, poison=None, runs=1):
    out = {}
    for aligner in ("banded", "full"):
        cfg = default_config(aligner=aligner, scoring_mode="coverage", n_barcodes=nb)
        with lib.Context(cfg) as ctx:
            ctx.submit(batch)
            if trace:
                ctx.set_stage_trace(True)
            if poison is not None:
                ctx.set_poison(poison)
            for _ in range(runs):
                ctx.run()
                r, a = ctx.fetch_scores()
                if poison is not None:
                    assert not (r == poison).any() and not (a == poison).any(), "%s: a score was never written" % aligner
            out[aligner] = (r, a, ctx.fetch_stage() if trace else None, ctx.timing())
    return out


@pytest.mark.parametrize("kind", ["clean", "noisy", "indels", "real-sequence", "repeats", "read-shapes"])
def test_stage_invariant_and_oracle(kind):
    if kind == "clean":
        batches = [("config-3 shape", synth.make_batch(synth.SynthSpec(n_loci=300, n_barcodes=500, reads_per_locus=64, seed=5)), 500)]
    elif kind == "noisy":
        batches = [("%.0f %% errors" % (100 * e), synth.make_batch(synth.SynthSpec(n_loci=200, n_barcodes=500, reads_per_locus=48, sub_error=e, seed=6)), 500)
                   for e in (0.03, 0.08)]
    elif kind == "indels":
        batches = [("config-5 shape", synth.make_batch(synth.SynthSpec(n_loci=256, n_barcodes=200, reads_per_locus=48, indel_frac=0.6,
                                                                         read_len_jitter=60, seed=23, sub_error=0.02, use_umi=True)), 200)]
    elif kind == "real-sequence":
        batches = list(SB.real_sequence_batches(trials=2))
    elif kind == "repeats":
        batches = list(SB.repeat_rich_batches(trials=4, loci=30, reads=16, pad_range=(30, 120))) + list(SB.near_repeat_batches(trials=2))
    else:
        batches = list(SB.real_shape_batches(trials=2))
    for label, batch, nb in batches:
        out = run_both(batch, nb, poison=-31337)
        rb, ab, stage, t = out["banded"]
        rf, af, stage_f, _ = out["full"]
        assert np.all(stage_f == abi.STAGE_FULL_DP)
        differ = assert_stage_invariant(stage, (rb, ab), (rf, af), label)
        oref, oalt = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=nb), threads=os.cpu_count() or 8)
        bad = np.nonzero((rb != oref) | (ab != oalt))[0]
        assert bad.size == 0, "%s: record %d device (%d, %d) oracle (%d, %d), stages %s" % (
            label, bad[0], rb[bad[0]], ab[bad[0]], oref[bad[0]], oalt[bad[0]], stage[2 * bad[0]:2 * bad[0] + 2])
        print("%s: %d alignments, banded != full on %d; decided by %s; checked %d, swept %d, declined %d" % (
            label, len(stage), int(differ.sum()), stage_report(stage), t.checked_tasks, t.swept_tasks, t.overflow_tasks))
        assert int(t.swept_tasks) >= int(np.isin(stage, (abi.STAGE_SWEEP_DP, abi.STAGE_GENERAL_DP)).sum())


def test_poisoned_reruns_write_every_score():
    """Three runs of one context with the score arrays poisoned before each (banded: the certificate stages, the check, the sweep
    and the masked DP each own a disjoint set of tasks; a task nobody writes would keep the poison)."""
    for label, batch, nb in list(SB.real_sequence_batches(trials=1)) + list(SB.synthetic_batches(per_model=1, n_loci=60, reads=24))[:3]:
        if max(int(batch.loci["ref_len"].max()), int(batch.loci["alt_len"].max())) > 255:
            continue
        out = run_both(batch, nb, trace=False, poison=-7, runs=3)
        oref, oalt = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=nb), threads=os.cpu_count() or 8)
        assert np.array_equal(out["banded"][0], oref) and np.array_equal(out["banded"][1], oalt), label


CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import stress_batches as SB
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
out = []
batches = list(SB.real_sequence_batches(trials=1)) + list(SB.repeat_rich_batches(trials=2, loci=30, reads=16)) + list(SB.real_shape_batches(trials=1))
batches += [("noisy", synth.make_batch(synth.SynthSpec(n_loci=200, n_barcodes=300, reads_per_locus=48, sub_error=0.05, indel_frac=0.3, seed=3)), 300)]
for label, batch, nb in batches:
    with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=nb)) as ctx:
        ctx.submit(batch); ctx.run()
        r, a = ctx.fetch_scores()
        t = ctx.timing()
        out.append(r); out.append(a)
        print(label, "left", t.diag_left, "checked", t.checked_tasks, "swept", t.swept_tasks, "hard", t.hard_tasks, "declined", t.overflow_tasks, file=sys.stderr)
np.save(sys.argv[1], np.concatenate(out))
''' % (ROOT, HERE)


@pytest.mark.parametrize("hook", ["VTX_BAND_LEGACY", "VTX_BAND_CHECK", "VTX_BAND_NO_TIGHT", "VTX_BAND_SLOTS", "VTX_SWEEP_V1", "VTX_BAND_DIAG2_MIN",
                                  "VTX_BAND_NO_CORRIDOR", "VTX_DIAG_FOUR_WORDS", "VTX_DIAG_NO_TWINS", "VTX_DIAG_T3", "VTX_DIAG_NO_T3"])
def test_hooks_give_the_same_scores(hook):
    """VTX_BAND_DIAG2_MIN=1: the second single-diagonal stage (band_diag2_kernel) on every list, however short — by default lists
    below 700 k tasks skip it, i.e. every batch of this test suite but the full-size ones; VTX_SWEEP_V1=1: round 4's band_sweep_kernel (two passes) instead of round 5's; VTX_BAND_LEGACY=1: round 3's band_run_kernel / pending / general kernels instead of the sweep; VTX_BAND_CHECK=1: the full-matrix
    check in front of the DP of the tasks that left with a certificate; VTX_BAND_NO_TIGHT=1: those tasks go to the sweep like the
    others (the sweep's band and the certificate's one-diagonal band must give the same scores); VTX_BAND_SLOTS=5: the sweep + masked
    DP in slices of five band slots; VTX_BAND_NO_CORRIDOR=1 (round 6): round 5's routing of the tasks that hold a certificate —
    band_refine_kernel for the ones with main pieces only, the masked DP for the rest — instead of band_corridor_kernel;
    VTX_DIAG_FOUR_WORDS=1: band_diag_kernel's build for reads up to 256 bases on these batches of short reads (it is chosen by the
    batch's longest read); VTX_DIAG_NO_TWINS=1: band_diag_kernel probes every row that is not intact and unique instead of taking the
    matches of the intact rows from the haplotype's twin list; VTX_DIAG_T3=1: its pooled probes take blocks of three rows and the tables'
    three-row sets whatever the depth, VTX_DIAG_NO_T3=1: never (by default from 16 tasks per locus: vtx_band.hip, band_use_t3) instead of a queue
    entry and a presence-bitmap word per row.  Identical scores (separate processes: the hooks are read once)."""
    res = []
    with tempfile.TemporaryDirectory() as td:
        for on in (0, 1):
            env = dict(os.environ, VTX_LIB_VARIANT="dev")           # (the hooks exist in libvtx_dev.so only)
            env.pop(hook, None)
            if on:
                env[hook] = "5" if hook == "VTX_BAND_SLOTS" else "1"           # (VTX_BAND_DIAG2_MIN=1: every list)
            path = os.path.join(td, "h%d.npy" % on)
            p = subprocess.run([sys.executable, "-c", CODE, path], env=env, capture_output=True, text=True, timeout=1200)
            assert p.returncode == 0, p.stderr[-3000:]
            print(hook, on, p.stderr.strip().replace("\n", " | "))
            res.append(np.load(path))
    assert np.array_equal(res[0], res[1])


def _tables_and_scores(batch, nb, variant=None):
    with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=nb), variant=variant) as ctx:
        ctx.submit(batch)
        ctx.run()
        return ctx.debug_tables(), ctx.fetch_scores()


def _table_layout(max_hap, n_heads):
    """vtx_fast_core.h: tab_bytes_off .. tab_stride."""
    bytes_off = max_hap * 8 + n_heads * 2
    fb_off = bytes_off + max_hap + 8
    uq_off = (fb_off + max_hap + 8 + 3) & ~3
    pb_off = uq_off + 4 * (8 + (max_hap + 31) // 32 + 8)
    return bytes_off, fb_off, uq_off, pb_off, (pb_off + 512 + 128 + 2048 + 15) & ~15        # (round 6: tw[128], the twin list, and t3[2048], the three-row sets, behind pb[])


def _defined_bytes(tables, batch, n_heads=1024):
    """The bytes of the tables that MEAN something (entries of existing k-mers, head words, haplotype and flag bytes, the two
    bitmaps), table by table; the gaps between them are whatever the LDS held."""
    nl = batch.n_loci
    stride = len(tables) // (2 * nl)
    longest = int(max(batch.loci["ref_len"].max(), batch.loci["alt_len"].max()))
    max_hap = next(m for m in range(longest, longest + 64) if _table_layout(m, n_heads)[4] == stride)
    bytes_off, fb_off, uq_off, pb_off, _ = _table_layout(max_hap, n_heads)
    out = []
    for t in range(2 * nl):
        tb = tables[t * stride:(t + 1) * stride]
        hn = int(batch.loci["alt_len" if t & 1 else "ref_len"][t >> 1])
        nk = max(hn - 5, 0)
        out += [tb[:8 * nk], tb[max_hap * 8:bytes_off], tb[bytes_off:bytes_off + hn], tb[fb_off:fb_off + hn], tb[uq_off:pb_off + 512]]
        tw = tb[pb_off + 512:pb_off + 640]
        out += [tw[:1], tw[8:8 + 2 * (0 if tw[0] == 0xff else int(tw[0]))]]          # the twin list: its length, its pairs
        if _uses_t3(batch):
            out += [tb[pb_off + 640:pb_off + 640 + 2048]]                             # the three-row sets
    return np.concatenate(out)


def _uses_t3(batch):
    """vtx_band.hip, band_use_t3: t3[] is built for batches of at least 16 tasks (8 reads) per locus on average."""
    return (2 * batch.n_records) // max(batch.n_loci, 1) >= 16


def _expected_t3_and_twins(hap):
    """vtx_fast_core.h, Tab: t3[] (three 16-bit sets per 8-bit code of four bases) and the twin list of one haplotype, from its bytes."""
    code = [(b >> 1) & 3 for b in hap]
    t3 = np.zeros(512, np.uint32)
    nk = max(len(hap) - 5, 0)
    for y in range(nk):
        c = sum(code[y + i] << (2 * i) for i in range(6))
        t3[2 * (c >> 4)] |= np.uint32(1 << (c & 15))
        t3[2 * ((c >> 2) & 0xff)] |= np.uint32(1 << (16 + ((c & 3) | ((c >> 10) << 2))))
        t3[2 * (c & 0xff) + 1] |= np.uint32(1 << (c >> 8))
    pairs = [(y, z) for y in range(nk) for z in range(nk) if z != y and hap[y:y + 6] == hap[z:z + 6]]
    ok = len(pairs) <= 60 and len(hap) <= 256 and max(hap, default=0) < 0x80
    return t3, (pairs if ok else None)


def test_table_kernel_against_round3s():
    """band_tables_kernel (a table per wavefront, the hash chains linked by all 64 lanes: rounds of 64 positions, ds_max per bucket
    slot) must leave the bytes round 3's kernel left (VTX_BAND_TABLES_V1=1: a locus per wavefront, serial insertion) — on random
    sequence (one trip per round), on repeats (a bucket with many k-mers inside one round: as many trips), on real sequence, on
    haplotypes of one repeated unit."""
    cases = list(SB.synthetic_batches(per_model=1, n_loci=120, reads=8))[:3]
    cases += list(SB.repeat_rich_batches(trials=4, loci=60, reads=6, pad_range=(60, 120)))
    cases += list(SB.real_sequence_batches(trials=1, n_loci=200, reads=6))
    haps = [(b"A" * 201, b"A" * 100 + b"C" + b"A" * 100), (b"AC" * 100 + b"A", b"AC" * 50 + b"T" + b"AC" * 50),
            (b"ACG" * 67, b"ACG" * 33 + b"T" + b"ACG" * 33), (b"ACGTN" * 5 + b"\x90" + b"ACGGT" * 20, b"ACGTTGCA" * 12)]
    rds = [[(0, 0, h[0][20:170])] * 4 for h in haps]
    cases.append(("haplotypes of one repeated unit, a byte above 0x7f", SB.manual_batch(haps, rds, 10), 10))
    cases.append(("twelve reads per locus (t3[] on)", synth.make_batch(synth.SynthSpec(n_loci=80, n_barcodes=100, reads_per_locus=12, seed=11)), 100))
    checked = with_t3 = 0
    for label, batch, nb in cases:
        hl = np.maximum(batch.loci["ref_len"], batch.loci["alt_len"])
        if (hl > 255).any() and (hl <= 255).any() and int((hl > 255).sum()) * 8 <= batch.n_loci:
            continue                                     # (vtx_run scores such a batch in two passes, tests/test_gpu_shape.py: the buffer holds the second pass' few tables)
        os.environ.pop("VTX_BAND_TABLES_V1", None)
        tp, sp = _tables_and_scores(batch, nb)                   # the production library
        os.environ["VTX_BAND_TABLES_V1"] = "1"
        try:
            ts, ss = _tables_and_scores(batch, nb, variant="dev")   # round 3's kernel lives on in libvtx_dev.so
        finally:
            os.environ.pop("VTX_BAND_TABLES_V1", None)
        assert len(tp) == len(ts)
        if len(tp) == 0:
            continue                                     # (this shape keeps its tables in LDS: nothing to compare)
        dp, ds = _defined_bytes(tp, batch), _defined_bytes(ts, batch)
        assert np.array_equal(dp, ds), "%s: tables differ (%d of %d defined bytes)" % (label, int((dp != ds).sum()), len(dp))
        assert np.array_equal(sp[0], ss[0]) and np.array_equal(sp[1], ss[1]), label
        # round 6's arrays against a plain restatement (a few tables per batch)
        stride = len(tp) // (2 * batch.n_loci)
        longest = int(max(batch.loci["ref_len"].max(), batch.loci["alt_len"].max()))
        max_hap = next(m for m in range(longest, longest + 64) if _table_layout(m, 1024)[4] == stride)
        pb_off = _table_layout(max_hap, 1024)[3]
        for t in list(range(0, 2 * batch.n_loci, max(1, batch.n_loci // 4)))[:8]:
            L = batch.loci[t >> 1]
            off, hn = (int(L["alt_off"]), int(L["alt_len"])) if t & 1 else (int(L["ref_off"]), int(L["ref_len"]))
            t3, pairs = _expected_t3_and_twins(bytes(batch.hap_arena[off:off + hn]))
            tb = tp[t * stride:(t + 1) * stride]
            if _uses_t3(batch):
                assert np.array_equal(tb[pb_off + 640:pb_off + 640 + 2048].view(np.uint32), t3), (label, t)
                with_t3 += 1
            tw = tb[pb_off + 512:pb_off + 640]
            if pairs is None:
                assert tw[0] == 0xff, (label, t)
            else:
                assert tw[0] == len(pairs) and [(int(tw[8 + 2 * i]), int(tw[9 + 2 * i])) for i in range(len(pairs))] == pairs, (label, t)
        checked += 1
    assert checked >= 5 and with_t3 >= 4, (checked, with_t3)


CODE2 = r'''
import os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import stress_batches as SB
from oracle import oracle
from vartrix_amd import abi, lib, synth
from vartrix_amd.abi import default_config
from audit_util import assert_stage_invariant
batches = list(SB.real_sequence_batches(trials=2)) + list(SB.repeat_rich_batches(trials=3, loci=30, reads=16, pad_range=(60, 120)))
batches += list(SB.near_repeat_batches(trials=2)) + list(SB.real_shape_batches(trials=1))
batches += [("real sequence, deep", synth.make_batch(synth.SynthSpec(n_loci=400, n_barcodes=500, reads_per_locus=48, seed=4,
             genome_fasta=os.path.join(%r, "tests", "golden", "test_dna.fa"))), 500)]
looked = scored = streamed = 0
for label, batch, nb in batches:
    out = {}
    for aligner in ("banded", "full"):
        with lib.Context(default_config(aligner=aligner, scoring_mode="coverage", n_barcodes=nb)) as ctx:
            ctx.submit(batch)
            if aligner == "banded":
                ctx.set_stage_trace(True); ctx.set_poison(-999)
            ctx.run()
            out[aligner] = ctx.fetch_scores() + ((ctx.fetch_stage(), ctx.timing()) if aligner == "banded" else ())
    rb, ab, stage, t = out["banded"]
    assert_stage_invariant(stage, (rb, ab), out["full"], label)
    oref, oalt = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=nb), threads=os.cpu_count() or 8)
    bad = np.nonzero((rb != oref) | (ab != oalt))[0]
    assert bad.size == 0, (label, int(bad[0]), int(rb[bad[0]]), int(oref[bad[0]]), int(ab[bad[0]]), int(oalt[bad[0]]), stage[2 * bad[0]:2 * bad[0] + 2].tolist())
    looked += t.diag2_tasks; scored += t.diag2_scored; streamed += t.diag2_streamed
    print(label, "second stage looked at", t.diag2_tasks, "scored", t.diag2_scored, "streamed", t.diag2_streamed, "one-diagonal bands", t.checked_tasks, "swept", t.swept_tasks, file=sys.stderr)
    np.save(os.path.join(sys.argv[1], "s%%d.npy" %% len(os.listdir(sys.argv[1]))), np.concatenate([rb, ab]))
if os.environ.get("VTXT_OVERFLOW_PATH"):
    # (nothing reached the stage from band_diag_kernel's repeat list — VTX_BAND_DENSE_MASK=0 keeps it empty: every task it looked at came
    # from band_run_kernel's overflow list, the last-chunk call)
    assert looked > 300, looked
    print("second-stage-ok", looked, scored, streamed)
    sys.exit(0)
assert looked > 3000 and scored > 300, (looked, scored)
if not os.environ.get("VTX_BAND_NO_STREAM"):
    assert streamed > 300, streamed
else:
    assert streamed == 0
print("second-stage-ok", looked, scored, streamed)
''' % (ROOT, HERE, ROOT)


def test_second_stage_against_the_oracle():
    """band_diag2_kernel forced on every list (libvtx_dev.so, VTX_BAND_DIAG2_MIN=1) on real-sequence loci, tandem repeats, near repeats
    and real-read shapes: every score is the oracle's, banded != full only on DP stages, no score is left unwritten (poisoned
    arrays), and the stage really takes tasks (thousands looked at, hundreds scored outright, hundreds through band_stream_kernel).
    A second run with VTX_BAND_NO_STREAM=1 (what exceeds the list goes to the sweep, as before band_stream_kernel existed): the same."""
    with tempfile.TemporaryDirectory() as td:
        for no_stream in (0, 1):
            env = dict(os.environ, VTX_LIB_VARIANT="dev", VTX_BAND_DIAG2_MIN="1")
            env.pop("VTX_BAND_NO_STREAM", None)
            if no_stream:
                env["VTX_BAND_NO_STREAM"] = "1"
            d = os.path.join(td, str(no_stream)); os.mkdir(d)
            p = subprocess.run([sys.executable, "-c", CODE2, d], env=env, capture_output=True, text=True, timeout=1500)
            assert p.returncode == 0 and "second-stage-ok" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
            print(p.stderr.strip().replace("\n", " | "))
        files = sorted(os.listdir(os.path.join(td, "0")))
        assert files and files == sorted(os.listdir(os.path.join(td, "1")))
        for f in files:
            assert np.array_equal(np.load(os.path.join(td, "0", f)), np.load(os.path.join(td, "1", f))), f


def test_second_stage_on_the_overflow_list_of_band_run_kernel():
    """Round-5 ADVICE: the LAST-CHUNK call of the second stage — on what overflowed band_run_kernel's piece lists (vtx_run: `nA`, tables
    of every locus still resident, d_dense reused as its output) — had no test of its own: small batches never send anything to
    band_run_kernel (a short fail list joins the repeats).  libvtx_dev.so with VTX_BAND_RUN_MIN=1 (band_run_kernel takes every list)
    and VTX_BAND_DENSE_MASK=0 (band_diag_kernel sends it the repeats too) makes its lists overflow on repeat-rich loci; with
    VTX_BAND_DIAG2_MIN=1 the overflow list then takes the second stage.  Every score against the oracle, stage invariant, poison."""
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, VTX_LIB_VARIANT="dev", VTX_BAND_DIAG2_MIN="1", VTX_BAND_RUN_MIN="1", VTX_BAND_DENSE_MASK="0",
                   VTXT_OVERFLOW_PATH="1")
        p = subprocess.run([sys.executable, "-c", CODE2, td], env=env, capture_output=True, text=True, timeout=1500)
        assert p.returncode == 0 and "second-stage-ok" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
        print(p.stderr.strip().replace("\n", " | "))
