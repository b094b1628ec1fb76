"""Where a wavefront of band_diag_kernel spends its cycles (libvtx_dev.so, VTX_DIAG_PHASES=1: s_memtime between the kernel's phases,
summed over the wavefronts).  Run on the GPU box:  VTX_LIB_VARIANT=dev VTX_DIAG_PHASES=1 python tools/diag_phases.py [bench-like args]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("VTX_LIB_VARIANT", "dev"); os.environ.setdefault("VTX_DIAG_PHASES", "1")
import numpy as np, torch
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config

kw = dict(n_loci=100000, n_barcodes=10000, reads_per_locus=256, seed=20260926)
for a in sys.argv[1:]:
    k, v = a.split("="); kw[k] = (float(v) if "." in v else int(v)) if v.replace(".", "").isdigit() else v
b = synth.make_batch(synth.SynthSpec(**kw))
L = C.CDLL(os.path.join(os.path.dirname(lib.__file__), "libvtx_dev.so"))
out = (C.c_ulonglong * 16)()
NAMES = ["set-up", "read words", "diagonal search", "mask", "pieces/cert/rows/twins", "queue fill", "pass 1", "walks", "sort", "harmless", "closure + bound",
         "lists"]
with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=kw["n_barcodes"])) as ctx:
    ctx.submit(b); ctx.run(); torch.cuda.synchronize()
    L.vtxk_diag_phases(out)
    ctx.run(); torch.cuda.synchronize()
    L.vtxk_diag_phases(out)
    tot = sum(out[i] for i in range(12))
    print("wavefronts %d, cycles per wavefront %.0f, band_diag_kernel %.2f ms" % (out[15], tot / max(out[15], 1), getattr(ctx.timing(), "diag_ms", float("nan"))))
    for i, n in enumerate(NAMES):
        print("  %-26s %5.1f %%   %8.0f cycles / wavefront" % (n, 100.0 * out[i] / max(tot, 1), out[i] / max(out[15], 1)))
