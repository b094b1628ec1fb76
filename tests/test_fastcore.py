"""CPU unit test of the per-task logic of band_diag_kernel (vartrix_amd/csrc/vtx_fast_core.h) — the SAME source the device
kernel is compiled from, built for the host by tests/fastcore/Makefile — against the oracle.

band_diag_kernel may score a task only when its certificate equals its upper bound, and must leave every other task to
band_run_kernel.  So for every task: either the host build declines (score -1, with a reason), or its score IS the oracle's
banded score (bio 0.30 banded::Aligner::local as restated, reference call site src/main.rs:898-901).  The device runs the same
checks through the C-ABI in tests/test_gpu_*.py; this file is where the logic meets adversarial shapes without a GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle
from vartrix_amd import abi, synth
from vartrix_amd.abi import VtxBatch, default_config

import stress_batches as SB

HERE = os.path.dirname(os.path.abspath(__file__))
WHY = ["ok", "shape", "no-diagonal", "pieces", "matches", "not-harmless", "-", "generic", "not-tight", "no-main"]


@pytest.fixture(scope="module")
def core():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "fastcore"), "-s"])
    L = C.CDLL(os.path.join(HERE, "fastcore", "libfastcore_host.so"))
    L.vtxt_fastcore_batch.argtypes = [C.POINTER(VtxBatch), C.c_uint32, C.c_void_p, C.c_void_p]
    L.vtxt_fastcore_batch.restype = C.c_int
    return L


def run_core(L, batch, n_heads=1024):
    st = batch.as_struct()
    sc = np.zeros(2 * batch.n_records, np.int32)
    why = np.zeros(2 * batch.n_records, np.uint32)
    assert L.vtxt_fastcore_batch(C.byref(st), n_heads, sc.ctypes.data, why.ctypes.data) == 0
    return sc, why


def check(L, batch, n_barcodes, label, n_heads=1024):
    sc, why = run_core(L, batch, n_heads)
    r, a = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=n_barcodes), threads=8)
    want = np.empty(2 * batch.n_records, np.int32)
    want[0::2], want[1::2] = r, a
    decided = sc >= 0
    bad = np.nonzero(decided & (sc != want))[0]
    assert bad.size == 0, "%s: task %d scored %d, oracle %d" % (label, bad[0], sc[bad[0]], want[bad[0]])
    assert np.all((why == 0) == decided)
    # cert == ub decides a task, and ub bounds the FULL-matrix score: a decided score is the full score too (cert <= banded <= full <= ub).
    # (Round 6: the refined join was priced above what an excursion over a far piece costs — ub 36 < full 37 on a task whose banded
    #  score happened to be 36 as well; comparing with the banded score alone never saw it.)
    if n_heads & (1 << 29):                  # (the corridor certificate bounds the BANDED score: it decides tasks with banded < full too)
        return float(decided.mean()), {WHY[k]: int(v) for k, v in zip(*np.unique(why, return_counts=True))}
    rf_, af_ = oracle.batch_scores(batch, default_config(aligner="full", n_barcodes=n_barcodes), threads=8)
    full = np.empty(2 * batch.n_records, np.int32)
    full[0::2], full[1::2] = rf_, af_
    bad = np.nonzero(decided & (sc != full))[0]
    assert bad.size == 0, "%s: task %d decided at %d by a bound of the full score, which is %d" % (label, bad[0], sc[bad[0]], full[bad[0]])
    return float(decided.mean()), {WHY[k]: int(v) for k, v in zip(*np.unique(why, return_counts=True))}


def test_config3_shape_is_decided_and_exact(core):
    spec = synth.SynthSpec(n_loci=300, n_barcodes=500, reads_per_locus=64)
    frac, why = check(core, synth.make_batch(spec), 500, "config-3 shape")
    assert frac > 0.95, why          # the point of the stage: nearly every task of the headline workload ends here


@pytest.mark.parametrize("n_heads", [256, 1024])
def test_error_models_and_indels(core, n_heads):
    tot = 0
    for label, batch, nb in SB.synthetic_batches(per_model=1):
        frac, why = check(core, batch, nb, label, n_heads)
        tot += 2 * batch.n_records
    assert tot > 80000


def test_repeat_rich_and_small_alphabets(core):
    """Hundreds of k-mer matches per alignment, ties everywhere: nearly everything must be declined, nothing may be wrong."""
    for label, batch, nb in SB.repeat_rich_batches(trials=8):
        check(core, batch, nb, label)


@pytest.mark.parametrize("wide", [0, 1])
def test_near_repeats_and_real_sequence(core, wide):
    """The far-piece condition of back() (vtx_fast_core.h, condition (*)): chance-like off-diagonal matches planted 1-40 diagonals
    from the main one, and loci drawn from 181 kb of real sequence (20-40 off-diagonal matches per task are ordinary there).
    wide = 1: the four-byte match entries (20 per task) the device uses for haplotypes above 255 bases."""
    heads = 1024 | (wide << 31)
    fr = []
    for label, batch, nb in SB.near_repeat_batches(trials=8):
        fr.append(check(core, batch, nb, label, heads)[0])
    assert min(fr) > (0.2 if wide else 0.6) and max(fr) > (0.9 if wide else 0.95), fr
    fr = [check(core, batch, nb, label, heads)[0] for label, batch, nb in SB.real_sequence_batches(trials=3)]
    print("decided on real sequence (%s entries): %s" % ("4-byte" if wide else "2-byte", ", ".join("%.3f" % f for f in fr)))
    assert fr[0] > (0.5 if wide else 0.75), fr


def test_corridor_refinement_is_exact_and_decides_more(core):
    """Bit 30 of the harness' flags: back() prices same-diagonal joins of >= 3 close errors by the corridor DP over the real
    neighbour diagonals (what band_refine_kernel runs on the records band_diag_kernel leaves).  Every decided score is still the
    oracle's, and at 3 % substitution errors clearly more tasks are decided."""
    dec = {}
    for err in (0.01, 0.03, 0.08):
        for rl, pad in ((150, 100), (100, 60), (250, 100), (200, 120)):            # (above 192 bases: the fourth mask word, round 6)
            batch = synth.make_batch(synth.SynthSpec(n_loci=120 if rl <= 150 else 40, n_barcodes=500, reads_per_locus=48, sub_error=err, read_len=rl, padding=pad, seed=77))
            for rf in (0, 1):
                frac, why = check(core, batch, 500, "err %g len %d, refine %d" % (err, rl, rf), 1024 | (rf << 30))
                dec[(err, rl, rf)] = frac
    assert dec[(0.03, 150, 1)] > dec[(0.03, 150, 0)] + 0.02, dec
    for label, batch, nb in list(SB.near_repeat_batches(trials=4)) + list(SB.repeat_rich_batches(trials=3)):
        check(core, batch, nb, label, 1024 | (1 << 30))


def test_refined_join_stays_below_an_excursion_over_a_far_piece(core):
    """Round 6, found by the first full audit of the 8 percent workload (tools/full_audit.py e8, task 25 715 285): seven errors within
    16 bases between two main runs, and a 7-base exact run four diagonals out.  The excursion D4 - 7 matches - I4 joins the runs for 12;
    join_gap3(16) = 13 assumed no run above 5 bases outside the corridor, so the refined bound said 36 where the full-matrix score is
    37 (the banded score is 36: the band cuts the read's start off — the output was right by accident).  join_gap3_far prices the
    excursion with the task's far k-mer matches: the task must be left undecided (or decided at the full score)."""
    read = b'AGTGTCATGCTCAGAGCTTTCGTTACAGACAAGTCGCTCGCCCCGTCATGTGTTGGTGACTTGTTCAATAAGGGCCCGCGACAGTGCGTATCTTCACAGGTAGGGCCACTCGATTGGCTTGTCAAATTGTACCCACGCGGTTGGAGTTCT'
    ref = b'CGAGTGCGCTGCATGACGACTAGATCAACATCACCGAACAACAAAATGTCTCACATCATGGCGTGGGCACTATGATCCGGACAGTGTCACGCTCAGAGCTCCCGTTAGACACAGACAGCTCGCCCCGTCAAGTATTGGTGACTTGTTCAATAACGGCCCGCGACGGTGCGTATCTTCTCGGGTAGGGGCATTCTGTTGGCT'
    alt = b'CGAGTGCGCTGCATGACGACTAGATCAACATCACCGAACAACAAAATGTCTCACATCATGGCGTGGGCACTATGATCCGGACAGTGTCACGCTCAGAGCTTCCGTTAGACACAGACAGCTCGCCCCGTCAAGTATTGGTGACTTGTTCAATAACGGCCCGCGACGGTGCGTATCTTCTCGGGTAGGGGCATTCTGTTGGCT'
    loci = np.zeros(1, abi.LOCUS_DTYPE)
    loci["rec_count"] = 1
    loci["ref_len"], loci["alt_off"], loci["alt_len"] = len(ref), len(ref), len(alt)
    recs = np.zeros(1, abi.RECORD_DTYPE)
    recs["read_len"] = len(read)
    batch = abi.PackedBatch(loci, recs, np.frombuffer(ref + alt, np.uint8).copy(), np.frombuffer(read + bytes(16), np.uint8).copy())
    rf_, af_ = oracle.batch_scores(batch, default_config(aligner="full", n_barcodes=4), threads=1)
    rb_, ab_ = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=4), threads=1)
    assert (int(rf_[0]), int(af_[0]), int(rb_[0]), int(ab_[0])) == (36, 37, 36, 36)
    for flags in (1024, 1024 | (1 << 30)):
        sc, why = run_core(core, batch, flags)
        assert sc[1] in (-1, 37), (flags, sc, why)
        assert sc[0] in (-1, 36), (flags, sc, why)


def test_mask_pieces_and_mismatch_counts_of_a_diagonal(core):
    """diag_mask and front_rest AS COMPILED from vtx_fast_core.h on single (read, haplotype, diagonal) triples against the
    definitions: bit i of the mask = (x[i] == y[i + d]) where both exist; the main pieces = the runs of >= 6 ones, in order;
    nibble i of zc = the zeros between piece i - 1 and piece i (capped at 15).  Reads up to 256 bases (four mask words since round 6), diagonals from far left of
    the haplotype to far right of it, bytes above 0x7f, lower case, N."""
    core.vtxt_front_of_diagonal.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(99)
    out = np.zeros(16, np.uint64)
    n_pieces = 0
    for trial in range(1500):
        n = int(rng.integers(8, 256))
        alpha = [b"ACGT", b"ACGTN", b"AC", bytes([65, 67, 71, 84, 0x80, 0xff, 97, 110])][trial % 4]
        y = bytes(rng.choice(list(alpha), n).tolist())
        m = int(rng.integers(6, 257))
        d = int(rng.integers(-m + 1, n))
        x = bytearray(rng.choice(list(alpha), m).tolist())
        for i in range(m):                                     # mostly the haplotype's own bases on that diagonal
            if 0 <= i + d < n and rng.random() < 0.93:
                x[i] = y[i + d]
        assert core.vtxt_front_of_diagonal(bytes(x), m, y, n, d, out.ctypes.data) == 0
        want = 0
        for i in range(m):
            if 0 <= i + d < n and x[i] == y[i + d]:
                want |= 1 << i
        have = int(out[0]) | (int(out[1]) << 64) | (int(out[2]) << 128) | (int(out[14]) << 192)
        assert have == want, (trial, m, n, d)
        # pieces and nibbles from the definition (front_rest declines above RM pieces or without any: skip those)
        lo, hi = max(0, -d), min(m, n - d)
        runs, zeros_before, z = [], [], 0
        i = lo
        while i < hi:
            if (want >> i) & 1:
                j = i
                while j < hi and (want >> j) & 1:
                    j += 1
                if j - i >= 6:
                    runs.append((i, j - 1)); zeros_before.append(z); z = 0
                i = j
            else:
                z += 1; i += 1
        why = int(out[3]) >> 32
        if why == 0:
            r = int(out[3]) & 0xffffffff
            assert r == len(runs) and r <= 8, (trial, r, runs)
            for k, (a, b) in enumerate(runs):
                w = int(out[6 + k])
                assert (w & 0xff, (w >> 8) & 0xff) == (a, b), (trial, k)
                if k:
                    assert (int(out[4]) >> (4 * k)) & 15 == min(zeros_before[k], 15), (trial, k)
            n_pieces += r
    assert n_pieces > 2000


def test_probe_phase_finds_every_off_diagonal_match(core):
    """The claim behind probing only SOME rows (vtx_fast_core.h, front_rest): a row whose main-diagonal k-mer is intact and unique in
    the haplotype holds no other match.  For single (read, haplotype, diagonal) triples: every off-diagonal k-mer match the oracle
    lists sits in a row front_rest() asks for, and probe_rows() (presence bitmap, tagged heads, bucket walk: AS COMPILED) returns
    exactly the off-diagonal matches of those rows — on iid, two-letter, tandem-repeat and poly-A haplotypes."""
    core.vtxt_probe_of_diagonal.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    rng = np.random.default_rng(123)
    need = np.zeros(4, np.uint64)
    sbuf = np.zeros(20000, np.uint32)
    n_matches = n_done = 0
    for trial in range(600):
        kind = trial % 4
        n = int(rng.integers(60, 250))
        if kind == 0:
            y = bytes(rng.choice(list(b"ACGT"), n).tolist())
        elif kind == 1:
            y = bytes(rng.choice(list(b"AC"), n).tolist())
        elif kind == 2:
            unit = [b"AC", b"AAT", b"CAG", b"ACACAT"][int(rng.integers(0, 4))]
            y = (bytes(rng.choice(list(b"ACGT"), 30).tolist()) + unit * 40)[:n]
            n = len(y)
        else:
            yb = bytearray(rng.choice(list(b"ACGT"), n).tolist())
            a0 = int(rng.integers(10, n - 40))
            yb[a0:a0 + 25] = b"A" * 25
            y = bytes(yb)
        m = int(rng.integers(30, 160)) if trial % 3 else int(rng.integers(160, 257))
        d = int(rng.integers(-20, max(n - m + 20, -19)))
        x = bytearray(rng.choice(list(b"ACGT"), m).tolist())
        for i in range(m):
            if 0 <= i + d < n and rng.random() < 0.97:
                x[i] = y[i + d]
        x = bytes(x)
        got = core.vtxt_probe_of_diagonal(x, m, y, n, d, need.ctypes.data, sbuf.ctypes.data, len(sbuf))
        if got < 0:
            continue
        nmask = int(need[0]) | (int(need[1]) << 64) | (int(need[2]) << 128) | (int(need[3]) << 192)
        mt = oracle.kmer_matches(x, y)
        off = {(int(a), int(b)) for a, b in mt if int(b) - int(a) != d}
        for (a, b) in off:
            assert (nmask >> a) & 1, (trial, a, b, d)          # no off-diagonal match outside the rows asked for
        have = {(int(v) >> 16, int(v) & 0xffff) for v in sbuf[:got]}
        assert got == len(have)
        assert have == off, (trial, sorted(have ^ off)[:5])
        n_matches += len(off)
        n_done += 1
    assert n_done > 400 and n_matches > 20000, (n_done, n_matches)


def test_harmless_verdict_means_the_chain_stays_on_the_diagonal(core):
    """What the harmless test promises (vtx_fast_core.h): if every off-diagonal k-mer match is harmless, the reference's sdpkpp
    chain — computed here by the oracle over ALL matches — consists of main-diagonal matches only.  Reads with indels against
    the other allele, planted near repeats and tandem repeats make chains that do leave the diagonal: the verdict must then be
    'not harmless' (or the logic must have declined earlier).  With a harmless verdict the closed-form certificate equals the
    oracle's walk of the chain's staircase."""
    core.vtxt_harmless.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    core.vtxt_last_band_pack.restype = C.c_uint32
    oracle.lib().vtxo_chain_cert.restype = C.c_int32
    rng = np.random.default_rng(2718)
    said_yes = off_chain_seen = 0
    for trial in range(3000):
        kind = trial % 5
        n = 201
        yb = bytearray(rng.choice(list(b"ACGT"), n).tolist())
        if kind == 1:                                        # planted short repeats near the diagonal
            for _ in range(int(rng.integers(1, 6))):
                ln_, per = int(rng.integers(6, 13)), int(rng.integers(4, 30))
                y0 = int(rng.integers(10, n - ln_ - per - 1))
                yb[y0 + per:y0 + per + ln_] = yb[y0:y0 + ln_]
        elif kind == 2:                                      # a tandem repeat inside
            unit = [b"AC", b"AAT", b"CAG", b"ACACAT"][int(rng.integers(0, 4))]
            a0 = int(rng.integers(20, 120))
            rep = (unit * 30)[:int(rng.integers(12, 50))]
            yb[a0:a0 + len(rep)] = rep
        y = bytes(yb[:n])
        s0 = int(rng.integers(0, 50))
        x = bytearray(y[s0:s0 + 150])
        if kind == 3:                                        # deletion in the read (a chain over two diagonals)
            cut, gap = int(rng.integers(30, 100)), int(rng.integers(1, 21))
            x = bytearray(y[s0:s0 + cut] + y[s0 + cut + gap:s0 + cut + gap + 90])
        elif kind == 4:                                      # insertion in the read
            cut = int(rng.integers(30, 100))
            x = bytearray(y[s0:s0 + cut] + bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 21))).tolist()) + y[s0 + cut:s0 + cut + 90])
        for e in rng.integers(0, len(x), size=int(rng.integers(0, 5))):
            x[int(e)] = b"ACGT"[int(rng.integers(0, 4))]
        x = bytes(x)
        d, cert = C.c_int(0), C.c_int(0)
        verdict = core.vtxt_harmless(x, len(x), y, len(y), C.byref(d), C.byref(cert))
        mt = oracle.kmer_matches(x, y)
        if len(mt) == 0:
            continue
        chain, _ = oracle.sdpkpp(mt)
        off = [int(p) for p in chain if int(mt[p, 1]) - int(mt[p, 0]) != d.value]
        if verdict == 1:
            said_yes += 1
            assert not off, (trial, d.value, [(int(mt[p, 0]), int(mt[p, 1])) for p in off][:4])
            # ... and then the certificate is the oracle's walk of that chain's staircase (oracle/vtx_certify.c: vtxo_chain_cert)
            assert cert.value == oracle.lib().vtxo_chain_cert(x, len(x), y, len(y), 6), trial
            # ... and the band the masked DP expands from vtxf::band_pack (sw_banded_kernel<.., 2>: one diagonal stretch widened by
            # the (2w + 1)-squares) is the oracle's Band::create, column by column
            pk = core.vtxt_last_band_pack()
            dd, ca, cb = (pk >> 16) - 256, (pk >> 8) & 0xff, pk & 0xff
            assert dd == d.value
            olo, ohi, _ = oracle.band_create(x, y)
            for j in range(len(y) + 1):
                if ca - 20 <= j <= cb + 20:
                    lo_j = max(0, max(j - 20, ca) - dd - 20)
                    hi_j = min(len(x) + 1, min(j + 20, cb) - dd + 21)
                    assert (lo_j, hi_j) == (int(olo[j]), int(ohi[j])), (trial, j)
                else:
                    assert ohi[j] <= olo[j], (trial, j)
        elif off:
            off_chain_seen += 1
    assert said_yes > 700 and off_chain_seen > 300, (said_yes, off_chain_seen)


def test_pieces_far_apart_on_one_diagonal(core):
    """Same-diagonal joins of 100 - 170 bases (tests/stress_batches.py: far_apart_batches), with and without the refinement."""
    for rf in (0, 1):
        for label, batch, nb in SB.far_apart_batches(trials=4):
            check(core, batch, nb, label, 1024 | (rf << 30))


def test_join_closed_forms_of_the_kernel_header(core):
    """join_free / join_same / join_gap3 AS COMPILED from vtx_fast_core.h against brute force over (gap events g, gap length G per
    direction, mismatches mm): a stretch of D bases between two runs on one diagonal has D - G diagonal columns, mm of them
    mismatches, the g + mm events separate at most g + mm - 1 short runs of <= 5 matches; cost 5 g + 2 G + 5 mm - matches."""
    def brute(D, g0):
        best = 10 ** 6
        for g in range(2, 2 * D + 4):
            for G in range(max((g + 1) // 2, g0), D + 1):
                for mm in range(0, D - G + 1):
                    if D - G - mm <= 5 * (g + mm - 1):
                        best = min(best, 5 * g + 2 * G + 5 * mm - (D - G - mm))
                        break
        return best
    for D in range(1, 190):
        free = min(6 * e - D for e in range(1, D + 1) if D - e <= 5 * (e - 1))
        assert core.vtxt_join_free(D) == free, D
        gap = brute(D, 1)
        for e in range(1, min(D, 14) + 1):
            if D >= 2 or e == 1:
                want = min(6 * e - D, gap)
                assert core.vtxt_join_same(D, e) == want, (D, e)
        if D >= 3:
            g3 = brute(D, 3)
            have = core.vtxt_join_gap3(D)
            assert have == g3 if D <= 22 else 11 <= have <= g3, (D, have, g3)


def test_corridor_cost_is_the_optimum_of_the_corridor(core):
    """corridor_cost AS COMPILED from vtx_fast_core.h against a plain dictionary DP written here: the cheapest path (match -1,
    mismatch +5, gap of length L +5 + L, no floor) from a base of the first run, up to mu_a bases before its end, to a base of the
    second run, up to mu_b bases behind its start, at 1 per base given up, over the diagonals d - 2 .. d + 2."""
    import itertools
    rng = np.random.default_rng(12)
    NEG = -10 ** 6

    def reference(x, y, xb, d, D, mu_a, mu_b):
        r0, r1 = xb + 1 - mu_a, xb + D + 2 + mu_b
        H, E, F = {}, {}, {}
        H[(r0, r0 + d)] = -mu_a
        for i in range(r0, r1 + 1):
            for k in range(-2, 3):
                j = i + d + k
                if not (0 <= j <= len(y) and 0 <= i <= len(x)) or (i, j) == (r0, r0 + d):
                    continue
                if i == r0 and k < 0:
                    continue
                h = NEG
                if i > r0 and (i - 1, j - 1) in H and H[(i - 1, j - 1)] > NEG // 2 and i >= 1 and j >= 1:
                    h = H[(i - 1, j - 1)] + (1 if x[i - 1] == y[j - 1] else -5)
                f = max(F.get((i - 1, j), NEG) - 1, H.get((i - 1, j), NEG) - 6) if k + 1 <= 2 and i > r0 else NEG
                e = max(E.get((i, j - 1), NEG) - 1, H.get((i, j - 1), NEG) - 6) if k - 1 >= -2 else NEG
                f = f if f > NEG // 2 else NEG
                e = e if e > NEG // 2 else NEG
                F[(i, j)], E[(i, j)] = f, e
                H[(i, j)] = max(h, f, e)
        end = H.get((r1, r1 + d), NEG)
        return (1 << 20) if end <= NEG // 2 else mu_b + 1 - end

    core.vtxt_corridor_cost.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    n_checked = 0
    for trial in range(400):
        alpha = b"ACGT" if trial % 3 else b"AC"
        y = bytes(rng.choice(list(alpha), 120).tolist())
        d = int(rng.integers(-3, 8))
        xs = int(rng.integers(10, 30))
        x = bytearray(y[xs + d:xs + d + 80]) if xs + d >= 0 else bytearray(y[:80])
        d = d if xs + d >= 0 else 0
        lo = 20
        D = int(rng.integers(3, 14))
        for p in rng.choice(np.arange(lo + 1, lo + D + 1), size=min(D, int(rng.integers(2, 6))), replace=False):
            x[int(p)] = b"ACGT"[(b"ACGT".index(bytes([x[int(p)]])) + 1 + int(rng.integers(0, 3))) % 4]
        # diagonal of x vs y: x[i] faces y[i + xs + d]... the runs' diagonal in haplotype coordinates
        dd = xs + d
        mu_a, mu_b = int(rng.integers(0, 9)), int(rng.integers(0, 9))
        have = core.vtxt_corridor_cost(bytes(x), len(x), y, len(y), lo, dd, D, mu_a, mu_b)
        want = reference(bytes(x), y, lo, dd, D, mu_a, mu_b)
        assert have == want, (trial, have, want)
        n_checked += 1
    assert n_checked == 400


def test_real_read_shapes(core):
    """Soft clips, adapter tails, spliced reads, poly-A, N bases (tests/stress_batches.py)."""
    fr = []
    for label, batch, nb in SB.real_shape_batches(trials=3):
        frac, why = check(core, batch, nb, label)
        fr.append(frac)
    print("decided on real-read shapes: %s" % ", ".join("%.3f" % f for f in fr))


def test_edge_shapes(core):
    """Reads at the capacity edge (256 bases; 192 until round 6), beyond it, shorter than a k-mer; haplotypes shorter than the read;
    lower-case and N bytes; identical haplotypes; a read that is a pure repeat."""
    rng = np.random.default_rng(5)
    g = bytes(rng.choice(list(b"ACGT"), 3000).tolist())
    haps = [(g[100:301], g[100:200] + b"T" + g[201:301]), (g[500:520], g[500:510] + g[512:520]),
            (g[800:1001], g[800:900] + b"n" + g[901:1001]), (b"AC" * 100, b"AC" * 50 + b"G" + b"AC" * 50),
            (g[1500:1900], g[1500:1700] + b"ACGTACGTAC" + g[1700:1900]),
            (g[2100:2355], g[2100:2227] + b"T" + g[2228:2355])]
    reads = [
        [(0, 0, g[80:272]), (1, 0, g[60:253]), (2, 0, g[150:155]), (3, 0, g[150:156]), (4, 0, g[100:292]), (5, 0, b"")],
        [(0, 0, g[480:560]), (1, 0, g[500:520]), (2, 0, g[505:511])],
        [(0, 0, g[820:970]), (1, 0, g[820:900] + b"n" + g[901:970]), (2, 0, g[820:900] + b"N" + g[901:970])],
        [(0, 0, b"AC" * 75), (1, 0, b"CA" * 60), (2, 0, b"AC" * 40 + b"G" + b"AC" * 30)],
        [(0, 0, g[1600:1750]), (1, 0, g[1620:1700] + b"ACGTACGTAC" + g[1700:1760]), (2, 0, g[1400:1550])],
        [(0, 0, g[2100:2355]), (1, 0, g[2050:2306]), (2, 0, g[2050:2307]), (3, 0, g[2090:2283]), (4, 0, g[2200:2450]),
         (5, 0, g[2100:2200] + b"A" + g[2201:2300] + b"C" + g[2301:2355])],
    ]
    check(core, SB.manual_batch(haps, reads, 8), 8, "edge shapes")


# ---- the second stage (vtxf::fast_task2 = band_diag2_kernel + band_stream_kernel per task): 64 list entries, the harmless bound from the
#      matches that can really precede a match ----
def _band_from_pack(pk, m, n):
    dd, ca, cb = (pk >> 16) - 256, (pk >> 8) & 0xff, pk & 0xff
    lo = np.full(n + 1, m + 1, np.int64)
    hi = np.zeros(n + 1, np.int64)
    for j in range(n + 1):
        if ca - 20 <= j <= cb + 20:
            lo[j] = max(0, max(j - 20, ca) - dd - 20)
            hi[j] = min(m + 1, min(j + 20, cb) - dd + 21)
    return lo, hi


def _check_second_stage(core, batch, nb, label, band_checks=400):
    """Every task of the batch through fast_task2: a SCORE verdict is the oracle's banded score; a TIGHT verdict promises that the
    reference's band is the one diagonal stretch of band_pack — checked column by column against the oracle's Band::create on a
    sample, and the masked DP over that band (oracle.sw_ranges) gives the oracle's score on every one."""
    core.vtxt_fastcore2_batch.argtypes = [C.POINTER(VtxBatch), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    st = batch.as_struct()
    n = 2 * batch.n_records
    verdict, score, pack, why = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    assert core.vtxt_fastcore2_batch(C.byref(st), 1024, verdict.ctypes.data, score.ctypes.data, pack.ctypes.data, why.ctypes.data) == 0
    r, a = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=nb), threads=8)
    want = np.empty(n, np.int32)
    want[0::2], want[1::2] = r, a
    dec = verdict == 0
    bad = np.nonzero(dec & (score != want))[0]
    assert bad.size == 0, "%s: task %d scored %d, oracle %d" % (label, bad[0], score[bad[0]], want[bad[0]])
    tight = np.nonzero(verdict == 1)[0]
    assert np.all(score[tight] <= want[tight]), label                  # the certificate is a lower bound of the banded score
    rec_locus = np.repeat(np.arange(batch.n_loci), batch.loci["rec_count"])
    hb, rb = batch.hap_arena.tobytes(), batch.read_arena.tobytes()
    rng = np.random.default_rng(1)
    for t in (tight if len(tight) <= band_checks else rng.choice(tight, band_checks, replace=False)):
        rec, loc = batch.records[t >> 1], batch.loci[rec_locus[t >> 1]]
        x = rb[int(rec["read_off"]):int(rec["read_off"]) + int(rec["read_len"])]
        off, ln = (int(loc["alt_off"]), int(loc["alt_len"])) if t & 1 else (int(loc["ref_off"]), int(loc["ref_len"]))
        y = hb[off:off + ln]
        lo, hi = _band_from_pack(int(pack[t]), len(x), len(y))
        olo, ohi, _ = oracle.band_create(x, y)
        empty = ohi <= olo
        assert np.array_equal(lo[~empty], olo[~empty]) and np.array_equal(hi[~empty], ohi[~empty]) and not hi[empty].any(), (label, int(t))
    return float(dec.mean()), float((verdict == 1).mean()), float((verdict == 2).mean())


def test_second_stage_on_real_sequence_repeats_and_noise(core):
    tot = {}
    cases = [("real sequence", synth.make_batch(synth.SynthSpec(n_loci=300, n_barcodes=500, reads_per_locus=12, seed=3,
                                                                genome_fasta=os.path.join(HERE, "golden", "test_dna.fa"))), 500)]
    cases += list(SB.real_sequence_batches(trials=1)) + list(SB.near_repeat_batches(trials=3)) + list(SB.real_shape_batches(trials=1))
    cases += list(SB.repeat_rich_batches(trials=3, loci=16, reads=10, pad_range=(60, 120)))
    cases += [(lbl, b, nb) for lbl, b, nb in SB.synthetic_batches(per_model=1, n_loci=24, reads=12)]
    for label, batch, nb in cases:
        if max(int(batch.loci["ref_len"].max()), int(batch.loci["alt_len"].max())) > 255:
            continue
        tot[label] = _check_second_stage(core, batch, nb, label)
    d, t, s = tot["real sequence"]
    print("second stage, real sequence: decided %.1f %%, one-diagonal band %.1f %%, left to the sweep %.1f %%" % (100 * d, 100 * t, 100 * s))
    assert s < 0.05 and d > 0.80, tot["real sequence"]           # (the first stage alone leaves 15 % of these tasks)


def test_streaming_harmless_test_equals_the_list_version(core):
    """probe_harmless_stream (a window of the last six rows, for tasks whose matches do not fit the list) must give the verdict of
    back_harmless over the whole list wherever the list holds every match."""
    core.vtxt_harmless_stream_vs_list.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
    core.vtxt_harmless_stream_vs_list.restype = C.c_uint32
    from test_sweep_model import tasks_of
    n = yes = no = 0
    gens = (SB.real_sequence_batches(trials=1), SB.repeat_rich_batches(trials=3, loci=12, reads=8, pad_range=(60, 120)),
            SB.near_repeat_batches(trials=2), SB.synthetic_batches(per_model=1, n_loci=12, reads=8))
    for gen in gens:
        for label, batch, _nb in gen:
            if max(int(batch.loci["ref_len"].max()), int(batch.loci["alt_len"].max())) > 255:
                continue
            for x, y in tasks_of(batch, 300):
                r = core.vtxt_harmless_stream_vs_list(x, len(x), y, len(y), 0)
                if r == 0xffffffff or not (r & 0x10000):
                    continue
                lv, sv = r & 0xff, (r >> 8) & 0xff
                assert sv != 2 and lv == sv, (label, lv, sv, x, y)
                n += 1; yes += lv; no += 1 - lv
    assert n > 1500 and yes > 1000 and no > 50, (n, yes, no)


def test_whole_read_is_decided_whatever_else_matches(core):
    """vtxf::whole_read: a read that matches its haplotype base for base on one diagonal is scored m without a probe — also where the
    same read fits several diagonals just as perfectly (tandem repeats, a duplicated segment: the tie rule picks the reference's
    chain), where hundreds of off-diagonal matches surround it, and where it ends at the haplotype's edge.  Every such task has to
    be decided, with the oracle's banded score (= m = the full-matrix score)."""
    rng = np.random.default_rng(77)
    g = bytes(rng.choice(list(b"ACGT"), 4000).tolist())
    haps, reads = [], []
    for unit_len in (2, 3, 7, 13, 23, 41):
        unit = bytes(rng.choice(list(b"ACGT"), unit_len).tolist())
        hap = (unit * (220 // unit_len + 2))[:220]
        alt = hap[:110] + bytes([hap[110] ^ 6]) + hap[111:]
        haps.append((hap, alt))
        reads.append([(c, 0, hap[o:o + ln]) for c, (o, ln) in enumerate([(0, 150), (3, 150), (40, 120), (70, 150), (100, 60), (214, 6), (0, 192)])])
    seg = g[100:250]
    haps.append((g[0:40] + seg + g[300:305] + seg[:50], g[0:40] + seg + b"T" + g[300:305] + seg[:50]))      # a duplicated segment
    reads.append([(0, 0, seg), (1, 0, seg[:50]), (2, 0, seg[10:140]), (3, 0, (g[0:40] + seg)[20:170])])
    haps.append((g[1000:1201], g[1000:1100] + b"A" + g[1101:1201]))                                         # plain sequence, reads at the edges
    reads.append([(0, 0, g[1000:1150]), (1, 0, g[1051:1201]), (2, 0, g[1000:1192]), (3, 0, g[1100:1106])])
    batch = SB.manual_batch(haps, reads, 8)
    sc, why = run_core(core, batch)
    r, a = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=8), threads=8)
    rf, af = oracle.batch_scores(batch, default_config(aligner="full", n_barcodes=8), threads=8)
    want = np.empty(2 * batch.n_records, np.int32); want[0::2], want[1::2] = r, a
    full = np.empty(2 * batch.n_records, np.int32); full[0::2], full[1::2] = rf, af
    lens = np.repeat(batch.records["read_len"].astype(np.int32), 2)
    decided = sc >= 0
    assert np.all(sc[decided] == want[decided])
    whole = full == lens                                     # the full-matrix score is the read's length: the read fits somewhere base for base
    assert whole.sum() >= 45 and np.all(want[whole] == lens[whole])
    # (a whole read the stage does not decide found no diagonal: every 6-mer of a short-unit repeat is repeated, no sampled row gives a
    # candidate — those go on to band_sweep_kernel)
    assert decided[whole].sum() >= 8 and set(why[whole & ~decided].tolist()) <= {2}, {WHY[k]: int(v) for k, v in zip(*np.unique(why[whole], return_counts=True))}


def test_band_trimmed_bound_decides_the_banded_score(core):
    """vtx_band_trim.h (not in any kernel yet): the run bound over the main pieces trimmed to the one-diagonal band bounds the
    BANDED score.  Behind the corridor refinement, on substitution-error models, indels, real-read shapes, repeats and real sequence:
    every score it decides is the oracle's banded score — also (and mostly) where banded < full, which the bound of the full score can
    never decide — and it decides a third to a half of what the refinement leaves on noisy reads."""
    core.vtxt_fastcore_trim_batch.argtypes = [C.POINTER(VtxBatch), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    core.vtxt_fastcore_trim_batch.restype = C.c_int
    cases = [("sub %.3f" % e, synth.make_batch(synth.SynthSpec(n_loci=250, n_barcodes=500, reads_per_locus=24, sub_error=e, seed=300 + i)), 500)
             for i, e in enumerate((0.01, 0.03, 0.08, 0.08, 0.15))]
    cases += list(SB.synthetic_batches(per_model=1, n_loci=40, reads=16))
    cases += list(SB.real_shape_batches(trials=1)) + list(SB.real_sequence_batches(trials=1))
    cases += list(SB.repeat_rich_batches(trials=2, loci=12, reads=8, pad_range=(60, 120))) + list(SB.near_repeat_batches(trials=1))
    n_trim = n_below = 0
    left = {}
    for label, batch, nb in cases:
        if max(int(batch.loci["ref_len"].max()), int(batch.loci["alt_len"].max())) > 255:
            continue
        st = batch.as_struct()
        n = 2 * batch.n_records
        sc, why, tr = np.zeros(n, np.int32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        assert core.vtxt_fastcore_trim_batch(C.byref(st), 1024, sc.ctypes.data, why.ctypes.data, tr.ctypes.data) == 0
        s0, w0 = run_core(core, batch, 1024 | (1 << 30))                       # the same logic without the trimmed bound
        r, a = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=nb), threads=8)
        rf, af = oracle.batch_scores(batch, default_config(aligner="full", n_barcodes=nb), threads=8)
        want, full = np.empty(n, np.int32), np.empty(n, np.int32)
        want[0::2], want[1::2] = r, a
        full[0::2], full[1::2] = rf, af
        dec = sc >= 0
        bad = np.nonzero(dec & (sc != want))[0]
        assert bad.size == 0, "%s: task %d scored %d, oracle %d (trimmed %d)" % (label, bad[0], sc[bad[0]], want[bad[0]], tr[bad[0]])
        assert np.array_equal(sc[tr == 0], s0[tr == 0]) and np.all(s0[tr == 1] < 0) and np.all(w0[tr == 1] == 8)   # it only adds verdicts
        n_trim += int((tr == 1).sum())
        n_below += int(((tr == 1) & (want < full)).sum())
        if label.startswith("sub 0.08"):
            left[label + str(len(left))] = (int((w0 == 8).sum()), int((why == 8).sum()))
    assert n_trim > 2000 and n_below > 1500, (n_trim, n_below)
    for before, after in left.values():
        assert after < 0.6 * before, left


def test_twin_list_gives_the_matches_the_probes_gave(core):
    """Round 6, the haplotype's twin list (vtx_fast_core.h: Tab tw[], twin_matches): band_diag_kernel takes the off-diagonal matches of
    the rows whose main-diagonal k-mer is intact from the list and probes only the other rows.  Single (read, haplotype) pairs, AS
    COMPILED: the match set equals the one the probes of every row that is not (intact and unique) give — on iid sequence, a
    two-letter alphabet, tandem repeats next to iid sequence, a planted duplication, reads with errors, overhangs and indels; and
    against the oracle's list of k-mer matches."""
    core.vtxt_twin_vs_probe.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    core.vtxt_twin_vs_probe.restype = C.c_uint32
    rng = np.random.default_rng(77)
    sa, sb = np.zeros(64, np.uint32), np.zeros(64, np.uint32)
    n_done = n_twin_matches = n_nolist = 0
    for trial in range(1500):
        kind = trial % 5
        n = int(rng.integers(60, 256))
        yb = bytearray(rng.choice(list(b"ACGT"), n).tolist())
        if kind == 1:
            yb = bytearray(rng.choice(list(b"AC"), n).tolist())
        elif kind == 2:
            unit = [b"AC", b"AAT", b"CAG", b"ACACAT"][int(rng.integers(0, 4))]
            reps = unit * int(rng.integers(3, 9))
            a0 = int(rng.integers(0, n - len(reps)))
            yb[a0:a0 + len(reps)] = reps
        elif kind == 3:
            ln = int(rng.integers(7, 20))
            a0, b0 = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
            yb[b0:b0 + ln] = yb[a0:a0 + ln]
        elif kind == 4 and trial % 10 == 4:
            yb[int(rng.integers(0, n))] = ord("N")
        y = bytes(yb)
        m = int(rng.integers(40, 160)) if trial % 3 else int(rng.integers(160, 257))
        d = int(rng.integers(-40, max(n - m + 40, -39)))
        x = bytearray(rng.choice(list(b"ACGT"), m).tolist())
        err = [0.0, 0.005, 0.03, 0.08][trial % 4]
        for i in range(m):
            if 0 <= i + d < n and rng.random() >= err:
                x[i] = y[i + d]
        if trial % 7 == 0 and m > 80:                      # a deletion in the read: its second half lies on another diagonal
            cut = int(rng.integers(30, m - 30))
            x = x[:cut] + x[cut + 3:]
            m = len(x)
        x = bytes(x)
        got = core.vtxt_twin_vs_probe(x, m, y, n, sa.ctypes.data, sb.ctypes.data, 64)
        if got == 0xfffffffe:
            n_nolist += 1
            continue
        if got == 0xffffffff:
            continue
        assert got != 0xfffffffd, trial
        na, nb = got & 0xffff, got >> 16
        assert na == nb, (trial, na, nb)
        if na == 0xffff:
            continue
        assert np.array_equal(sa[:na], sb[:nb]), (trial, sa[:na].tolist(), sb[:nb].tolist())
        n_done += 1
        n_twin_matches += na
    assert n_done > 700 and n_twin_matches > 3000 and n_nolist > 50, (n_done, n_twin_matches, n_nolist)


def test_twin_list_mode_decides_what_the_probes_decided(core):
    """Whole batches through the three phases + the corridor certificate with (bit 28) and without the twin list: the same verdicts and
    scores, task for task (the list changes where the matches come from, not what they are), and exact against the oracle."""
    cases = list(SB.synthetic_batches(per_model=1, n_loci=60, reads=24)) + list(SB.real_sequence_batches(trials=1))
    cases += list(SB.near_repeat_batches(trials=2)) + list(SB.repeat_rich_batches(trials=2))
    tot = 0
    for label, batch, nb in cases:
        if int(max(batch.loci["ref_len"].max(), batch.loci["alt_len"].max())) > 255:
            continue
        s0, w0 = run_core(core, batch, 1024 | (1 << 29))
        s1, w1 = run_core(core, batch, 1024 | (1 << 29) | (1 << 28))
        assert np.array_equal(s0, s1) and np.array_equal(w0, w1), (label, int(np.nonzero((s0 != s1) | (w0 != w1))[0][0]))
        tot += len(s0)
    check(core, cases[0][1], cases[0][2], cases[0][0], 1024 | (1 << 29) | (1 << 28))
    assert tot > 30000, tot
