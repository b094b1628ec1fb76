import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import test_gpu_shape as T
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config, PackedBatch
# config-3 shape with k long loci spread over it: time per step, split on (prod) — run under VTX_LIB_VARIANT=dev VTX_BAND_NO_SPLIT=1 for the old path
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
base = synth.make_batch(synth.SynthSpec(n_loci=nl, n_barcodes=10000))
if k:
    longs = T.long_loci_batch(k, 77, reads=256)
    pos = sorted(set(np.linspace(nl // (2 * k), nl - 1 - nl // (2 * k), k).astype(int).tolist()))
    parts, prev = [], 0
    for p, lb in zip(pos, longs):
        parts += [base.slice_loci(prev, p), lb]; prev = p
    parts.append(base.slice_loci(prev, nl))
    batch = PackedBatch.concat(parts)
else:
    batch = base
with lib.Context(default_config(aligner="banded", scoring_mode="consensus", n_barcodes=10000)) as ctx:
    ctx.submit(batch)
    for i in range(5):
        ctx.run(); t = ctx.timing()
        print("long loci %d: step %.2f ms (sw %.2f, diag %.2f, launches %d)" % (k, t.total_ms, t.sw_ms, t.diag_ms, t.sw_launches), flush=True)
