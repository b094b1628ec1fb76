"""Census of band_diag_kernel's probe phase on a synthetic batch, from the CPU build of its per-task logic (tools/probe_census.cpp on top of
tests/fastcore/fastcore_host.cpp): rows asked for per task (overhang / around errors / intact but not unique), presence-bitmap hits, matches,
blocks of three rows.  TEST / MEASUREMENT INFRASTRUCTURE (the product never loads it).  Counted with the twin lists OFF (front(.., tw = false)):
it is the census that motivated them (profiles/r06_probe_census.txt).
    g++ -O2 -std=c++17 -fPIC -shared -o /tmp/libprobe_census.so tools/probe_census.cpp && python tools/probe_census.py [n_loci] [key=value ...]"""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
from vartrix_amd import synth
from vartrix_amd.abi import VtxBatch
L = C.CDLL('/tmp/libprobe_census.so')
L.probe_stats.argtypes = [C.POINTER(VtxBatch), C.c_void_p]
kw = dict(n_loci=int(sys.argv[1]) if len(sys.argv) > 1 else 300, n_barcodes=10000, reads_per_locus=256, seed=3)
for a in sys.argv[2:]:
    k, v = a.split('='); kw[k] = float(v) if '.' in v else int(v)
b = synth.make_batch(synth.SynthSpec(**kw))
out = np.zeros(80, np.uint64)
st = b.as_struct()
L.probe_stats(C.byref(st), out.ctypes.data)
o = out.astype(float)
names = ['tasks','whole','live','need rows','pair-union rows','presence hits own','presence hits pair entries','ns','triples own','triples pair','-','r pieces','overhang rows','nonunique intact rows','rest rows']
for i, n in enumerate(names): print('%-28s %12d  per task %.2f  per live %.2f' % (n, out[i], o[i] / o[0], o[i] / max(o[2], 1)))
print('ns hist', out[20:62].tolist())
