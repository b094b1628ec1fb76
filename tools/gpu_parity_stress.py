"""The batches of tools/certify_stress.py through the device (banded and full flavours) against the oracle's scores:
every alignment must agree.  Needs an MI355X:      python tools/gpu_parity_stress.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import certify_stress as S  # noqa: E402
import stress_batches as SB  # noqa: E402
from oracle import oracle  # noqa: E402
from vartrix_amd import lib  # noqa: E402
from vartrix_amd.abi import default_config  # noqa: E402

total = bad = 0
def _more():
    # round 3: planted near-diagonal repeats, loci drawn from real sequence (band_diag_kernel's far-piece condition,
    # band_coop_kernel's rows), at a few times the size the pytest versions use
    for seed in range(4):
        yield from SB.near_repeat_batches(trials=8, loci=120, reads=32, seed=500 + seed)
    yield from SB.real_sequence_batches(trials=6, n_loci=1500, reads=24, seed=9100)


for gen in (S.synthetic_batches, S.repeat_rich_batches, _more):
    for label, batch, nb in gen():
        for aligner in (("banded", "full") if gen is not _more else ("banded",)):
            cfg = default_config(aligner=aligner, scoring_mode="coverage", n_barcodes=nb)
            with lib.Context(cfg) as ctx:
                ctx.submit(batch)
                ctx.run()
                r, a = ctx.fetch_scores()
                t = ctx.timing()
            oref, oalt = oracle.batch_scores(batch, cfg, threads=os.cpu_count() or 8)
            n_bad = int((r != oref).sum() + (a != oalt).sum())
            total += 2 * batch.n_records
            bad += n_bad
            print("%-50s %-6s %6d alignments, %d mismatches, %d hard, %d overflow" % (label, aligner, 2 * batch.n_records, n_bad,
                                                                                     t.hard_tasks, t.overflow_tasks), flush=True)
print("total %d alignments, %d mismatches" % (total, bad))
sys.exit(1 if bad else 0)
