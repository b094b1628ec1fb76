"""The per-task logic of band_diag_kernel / band_refine_kernel (vartrix_amd/csrc/vtx_fast_core.h, host build under tests/fastcore)
against the oracle at a few million tasks: every score the logic decides must be the oracle's banded score.  CPU only, test
infrastructure.  Seeds alternate the entry width (two-byte / four-byte match entries) and the corridor refinement.
    python tools/fastcore_stress.py 0 24          # seeds 0 .. 23: 8.5 M tasks, ~2.5 min on 16 cores
"""
import os
import sys, numpy as np, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import subprocess
subprocess.check_call(['make', '-C', 'tests/fastcore', '-s'])
import test_fastcore as TF
import stress_batches as SB
from vartrix_amd import synth
from vartrix_amd.abi import VtxBatch
L = C.CDLL('tests/fastcore/libfastcore_host.so')
L.vtxt_fastcore_batch.argtypes = [C.POINTER(VtxBatch), C.c_uint32, C.c_void_p, C.c_void_p]
tot=0; t0=time.time()
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    flags = 1024 | ((seed & 1) << 30) | (((seed >> 1) & 1) << 31)
    gens = []
    for err in (0.002, 0.01, 0.03, 0.06):
        gens.append(('genome err %g'%err, synth.make_batch(synth.SynthSpec(n_loci=800, n_barcodes=500, reads_per_locus=24, sub_error=err, genome_fasta='tests/golden/test_dna.fa', seed=7000+seed)), 500))
        gens.append(('iid err %g'%err, synth.make_batch(synth.SynthSpec(n_loci=400, n_barcodes=500, reads_per_locus=48, sub_error=err, seed=8000+seed)), 500))
    gens += list(SB.near_repeat_batches(trials=8, loci=80, reads=32, seed=9000+seed))
    gens += list(SB.far_apart_batches(trials=4, seed=9500+seed))
    gens += list(SB.repeat_rich_batches(trials=4, seed=9700+seed))
    for label, b, nb in gens:
        TF.check(L, b, nb, label, flags); tot += 2*b.n_records
    print('seed', seed, 'flags', hex(flags), 'total', tot, '%.0f s'%(time.time()-t0), flush=True)
print('TOTAL', tot, 'tasks, all decided scores exact')
