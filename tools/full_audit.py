#!/usr/bin/env python3
"""One-off full audit: EVERY alignment of a full-size workload, device against the oracle.

    python tools/full_audit.py [config3|real|e3|e8] ...      (GPU box; ~2.5 min of 16 CPU threads per workload)

The banded flavour runs on the device with the stage trace on and the score arrays poisoned; oracle.batch_scores (the restated
reference CPU path: bio 0.30.0's banded::Aligner::local per read and haplotype, src/main.rs:898-901) scores the whole batch on the
host's cores; every one of the 2 x records scores is compared, and the mismatches — none are expected — are broken down by the
stage that decided them.  The full flavour runs too: banded != full must imply a DP stage or the band-restricted certificate."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle                               # noqa: E402
from vartrix_amd import abi, lib, synth                 # noqa: E402
from vartrix_amd.abi import default_config              # noqa: E402
from audit_util import assert_stage_invariant, stage_report   # noqa: E402

WORKLOADS = {
    "config3": dict(),
    "real": dict(genome_fasta=os.path.join(ROOT, "tests", "golden", "test_dna.fa")),
    "e3": dict(sub_error=0.03),
    "e8": dict(sub_error=0.08),
}


def main():
    for name in (sys.argv[1:] or ["config3", "real"]):
        spec = synth.SynthSpec(n_loci=100_000, n_barcodes=10_000, reads_per_locus=256, seed=20260926, **WORKLOADS[name])
        batch = synth.make_batch(spec)
        out = {}
        for aligner in ("banded", "full"):
            with lib.Context(default_config(aligner=aligner, scoring_mode="consensus", n_barcodes=spec.n_barcodes)) as ctx:
                ctx.submit(batch)
                if aligner == "banded":
                    ctx.set_stage_trace(True)
                    ctx.set_poison(-4242)
                ctx.run()
                ctx.run()
                out[aligner] = ctx.fetch_scores() + ((ctx.fetch_stage(), ctx.timing()) if aligner == "banded" else ())
        rb, ab, stage, t = out["banded"]
        rf, af = out["full"]
        differ = assert_stage_invariant(stage, (rb, ab), (rf, af), name)
        t0 = time.time()
        threads = len(os.sched_getaffinity(0))
        oref, oalt = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=spec.n_barcodes), threads=threads)
        dt = time.time() - t0
        b = np.empty(2 * batch.n_records, np.int32)
        o = np.empty_like(b)
        b[0::2], b[1::2] = rb, ab
        o[0::2], o[1::2] = oref, oalt
        bad = b != o
        print("%s: %s" % (name, spec.name))
        print("  %d alignments on the device (second run of the context, scores poisoned before it); decided by %s" % (len(b), stage_report(stage)))
        print("  banded != full on %d alignments, every one decided by a DP stage or the band-restricted certificate" % int(differ.sum()))
        print("  oracle: all %d alignments in %.0f s on %d threads (%.3g alignments/s)" % (len(o), dt, threads, len(o) / dt))
        print("  MISMATCHES device vs oracle: %d%s" % (int(bad.sum()), "" if not bad.any() else
              "  by stage %s, first at task %d: device %d oracle %d" % (stage_report(stage[bad]), int(np.nonzero(bad)[0][0]),
                                                                        int(b[bad][0]), int(o[bad][0]))))
        print("  left by the first certificate stage %d, second stage looked at %d (scored %d, %d through band_stream_kernel), one-diagonal bands %d, swept %d, general kernel %d" % (
            t.diag_left, t.diag2_tasks, t.diag2_scored, t.diag2_streamed, t.checked_tasks, t.swept_tasks, t.overflow_tasks), flush=True)
        assert not bad.any()


if __name__ == "__main__":
    main()
