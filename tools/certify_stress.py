"""Stress of the certificate's inequalities on the CPU (test infrastructure): cert <= banded <= full <= ub_exact <= ub on
~320 k alignments — synthetic batches from clean to 30 % substitution errors with indels, ragged read lengths and paddings,
and repeat-rich / small-alphabet genomes with hundreds of pieces per alignment.  Prints one line per batch and the
total number of violations (must be 0).      python tools/certify_stress.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import certify_stats as cs  # noqa: E402
import test_gpu_parity as TP  # noqa: E402  (only _manual_batch: no GPU needed)
from vartrix_amd import synth  # noqa: E402
from vartrix_amd.abi import default_config  # noqa: E402


from stress_batches import repeat_rich_batches, synthetic_batches  # noqa: E402,F401  (tests/stress_batches.py)


def synthetic():
    tot = bad = 0
    cfg = default_config(aligner="banded", n_barcodes=500)
    seed = 1000
    for (sub, indel, jitter, rl, pad) in [(0.005, 0, 0, 150, 100), (0.02, 0.3, 40, 150, 100), (0.05, 0.5, 60, 120, 60),
                                          (0.1, 0.2, 30, 100, 150), (0.15, 0.6, 50, 80, 40), (0.01, 0.8, 70, 150, 200),
                                          (0.3, 0.1, 0, 150, 100)]:
        for _ in range(3):
            seed += 1
            spec = synth.SynthSpec(n_loci=150, n_barcodes=500, reads_per_locus=48, indel_frac=indel, sub_error=sub,
                                   read_len_jitter=jitter, read_len=rl, padding=pad, seed=seed)
            r = cs.certify(synth.make_batch(spec), cfg, os.cpu_count() or 8, exact=False)
            has = r["cert"] >= 0
            v = int((r["ub"] < r["full"]).sum()) + int((has & (r["cert"] > r["banded"])).sum()) + int((r["banded"] > r["full"]).sum())
            tot += len(r["full"]); bad += v
            print("sub %.3f indel %.1f jitter %d len %d pad %d: %d tasks, %d violations, cert == ub %.3f"
                  % (sub, indel, jitter, rl, pad, len(r["full"]), v, float((r["cert"] == r["ub"]).mean())), flush=True)
    return tot, bad


def repeat_rich():
    rng = np.random.default_rng(2024)
    cfg = default_config(aligner="banded", n_barcodes=30)
    tot = bad = 0
    for trial in range(12):
        alpha = [b"ACGT", b"AC", b"AT", b"ACG"][trial % 4]
        units = [b"A", b"AC", b"AAT", b"ACGT", b"AAAAC", b"AG", b"T", b"CAG", b"ACACAT", b"GATTACA"]
        g = bytearray()
        while len(g) < 40000:
            g += units[int(rng.integers(0, len(units)))] * int(rng.integers(2, 40))
            g += bytes(rng.choice(list(alpha), int(rng.integers(0, 30))).tolist())
        g = bytes(g)
        haps, reads = [], []
        for _ in range(60):
            p = int(rng.integers(300, len(g) - 500))
            pad = int(rng.integers(30, 160))
            ref = g[p - pad:p + pad + 1]
            kind = rng.random()
            if kind < 0.5:
                alt = ref[:pad] + bytes([b"ACGT"[(b"ACGT".index(ref[pad:pad + 1]) + 1) % 4]]) + ref[pad + 1:]
            elif kind < 0.75:
                alt = ref[:pad + 1] + bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 21))).tolist()) + ref[pad + 1:]
            else:
                alt = ref[:pad + 1] + ref[pad + 1 + int(rng.integers(1, min(20, pad - 1))):]
            haps.append((ref, alt))
            rl = []
            for _k in range(24):
                ln = int(rng.integers(40, 200))
                s = max(p - int(rng.integers(0, ln)), 0)
                rd = bytearray(g[s:s + ln])
                for e in np.nonzero(rng.random(len(rd)) < rng.choice([0.0, 0.02, 0.08]))[0]:
                    rd[e] = b"ACGT"[int(rng.integers(0, 4))]
                if len(rd) >= 10:
                    rl.append((int(rng.integers(0, 30)), 0, bytes(rd)))
            reads.append(rl)
        r = cs.certify(TP._manual_batch(haps, reads, 30), cfg, os.cpu_count() or 8, exact=True)
        has = r["cert"] >= 0
        v = (int((r["ub"] < r["full"]).sum()) + int((has & (r["cert"] > r["banded"])).sum()) + int((r["banded"] > r["full"]).sum())
             + int((r["ub_exact"] < r["full"]).sum()) + int((r["ub"] < r["ub_exact"]).sum()))
        tot += len(r["full"]); bad += v
        print("repeat-rich, alphabet %s: %d tasks, %d violations, cert == ub %.3f, up to %d pieces"
              % (alpha.decode(), len(r["full"]), v, float((r["cert"] == r["ub"]).mean()), int(r["pieces"].max())), flush=True)
    return tot, bad


if __name__ == "__main__":
    t0 = time.time()
    a, b = synthetic()
    c, d = repeat_rich()
    print("total %d alignments, %d violations, %.0f s" % (a + c, b + d, time.time() - t0))
    sys.exit(1 if b + d else 0)
