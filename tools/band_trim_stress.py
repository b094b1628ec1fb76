"""vtx_band_trim.h (the run bound restricted to the one-diagonal band: an upper bound of the BANDED score; host build under
tests/fastcore, not in any kernel yet) against the oracle: every score the logic decides — with the trimmed bound behind the corridor
refinement — must be the oracle's banded score.  Prints, per error rate, what the bound adds.  CPU only, test infrastructure.
    python tools/band_trim_stress.py            # 2 M tasks, ~4 min on 32 threads;  output of record: profiles/r05_band_trim_cpu.txt
"""
import ctypes as C, os, subprocess, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
subprocess.check_call(['make', '-C', 'tests/fastcore', '-s'])
from vartrix_amd import synth
from vartrix_amd.abi import VtxBatch, default_config
from oracle import oracle
import stress_batches as SB
L = C.CDLL('tests/fastcore/libfastcore_host.so')
L.vtxt_fastcore_trim_batch.argtypes = [C.POINTER(VtxBatch), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
L.vtxt_fastcore_batch.argtypes = [C.POINTER(VtxBatch), C.c_uint32, C.c_void_p, C.c_void_p]
THREADS = min(32, os.cpu_count() or 8)


def run(label, b, nb, with_full=False):
    """-> (tasks, decided without, decided with, undecided with certificate before, after, decided by the trimmed bound, of which banded < full)"""
    if max(int(b.loci["ref_len"].max()), int(b.loci["alt_len"].max())) > 255:
        return np.zeros(7, np.int64)
    st = b.as_struct(); n = 2 * b.n_records
    sc, why, tr = np.zeros(n, np.int32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
    assert L.vtxt_fastcore_trim_batch(C.byref(st), 1024, sc.ctypes.data, why.ctypes.data, tr.ctypes.data) == 0
    s0, w0 = np.zeros(n, np.int32), np.zeros(n, np.uint32)
    L.vtxt_fastcore_batch(C.byref(st), 1024 | (1 << 30), s0.ctypes.data, w0.ctypes.data)
    rb, ab = oracle.batch_scores(b, default_config(aligner="banded", n_barcodes=nb), threads=THREADS)
    band = np.empty(n, np.int32); band[0::2], band[1::2] = rb, ab
    bad = np.nonzero((sc >= 0) & (sc != band))[0]
    assert bad.size == 0, (label, bad[:5], sc[bad[:5]], band[bad[:5]], tr[bad[:5]])
    below = 0
    if with_full:
        rf, af = oracle.batch_scores(b, default_config(aligner="full", n_barcodes=nb), threads=THREADS)
        full = np.empty(n, np.int32); full[0::2], full[1::2] = rf, af
        below = int(((tr == 1) & (band < full)).sum())
    return np.array([n, (s0 >= 0).sum(), (sc >= 0).sum(), (w0 == 8).sum(), (why == 8).sum(), tr.sum(), below], np.int64)


print("per error rate (1 000 loci x 16 reads x 2 haplotypes each; refinement on): tasks | decided without -> with the trimmed bound | "
      "certificate but bounds apart: before -> after | decided by the trimmed bound (of which banded < full)")
for err, seed in ((0.005, 1), (0.01, 2), (0.03, 3), (0.08, 4), (0.15, 6)):
    r = run("err %g" % err, synth.make_batch(synth.SynthSpec(n_loci=1000, n_barcodes=1000, reads_per_locus=16, seed=seed, sub_error=err)), 1000, True)
    print("  %4.1f %% errors: %d | %.1f %% -> %.1f %% | %d -> %d | %d (%d)" % (100 * err, r[0], 100 * r[1] / r[0], 100 * r[2] / r[0], r[3], r[4], r[5], r[6]), flush=True)
tot = np.zeros(7, np.int64)
seed = 9000
for rnd in range(6):
    for err in (0.02, 0.04, 0.06, 0.08, 0.1, 0.12, 0.2):
        for rl, pad in ((150, 100), (100, 60), (190, 30), (60, 100), (150, 20)):
            seed += 1
            b = synth.make_batch(synth.SynthSpec(n_loci=300, n_barcodes=500, reads_per_locus=16, sub_error=err, read_len=rl, padding=pad, seed=seed,
                                                 indel_frac=0.3 if seed % 3 == 0 else 0.0, read_len_jitter=20 if seed % 2 else 0))
            tot += run('err %g rl %d pad %d seed %d' % (err, rl, pad, seed), b, 500)
for label, b, nb in (list(SB.real_shape_batches(trials=3)) + list(SB.real_sequence_batches(trials=3)) +
                     list(SB.repeat_rich_batches(trials=6, loci=20, reads=12, pad_range=(40, 120))) + list(SB.near_repeat_batches(trials=3))):
    tot += run(label, b, nb)
print("stress (error rates 2 - 20 %%, read lengths 60 - 190, paddings 20 - 100, indel loci, real-read shapes, real sequence, repeats): "
      "%d tasks, %d decided by the trimmed bound, every decided score the oracle's banded score" % (tot[0], tot[5]))

# The same under the alternative of the one recollected detail that moves bands (include/vtx_band_semantics.h: the lazy extension
# recompiled to 0, the oracle run with the same override): the trimmed rows follow band_pack, whatever built it.
import tempfile
with tempfile.TemporaryDirectory() as td:
    so = os.path.join(td, "libfastcore_lazy0.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DVTX_BAND_LAZY_EXT(k)=0", "-o", so, "tests/fastcore/fastcore_host.cpp"])
    L = C.CDLL(so)
    L.vtxt_fastcore_trim_batch.argtypes = [C.POINTER(VtxBatch), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vtxt_fastcore_batch.argtypes = [C.POINTER(VtxBatch), C.c_uint32, C.c_void_p, C.c_void_p]
    O = oracle.lib(); O.vtxo_set_variant.argtypes = [C.c_int, C.c_int]
    O.vtxo_set_variant(0, 0)
    THREADS = 1                          # (the oracle's variant switches are process globals: single-threaded use)
    try:
        tot = np.zeros(7, np.int64)
        seed = 500
        for err in (0.03, 0.08, 0.12):
            for rl, pad in ((150, 100), (100, 40)):
                seed += 1
                tot += run("lazy0 err %g" % err, synth.make_batch(synth.SynthSpec(n_loci=120, n_barcodes=500, reads_per_locus=12, sub_error=err,
                                                                                    read_len=rl, padding=pad, seed=seed)), 500)
    finally:
        O.vtxo_set_variant(0, -1)
    print("lazy extension 0 (harness and oracle): %d tasks, %d decided by the trimmed bound, every decided score the oracle's banded score" % (tot[0], tot[5]))
