"""How often does the DP-free certificate decide an alignment?  (CPU, test infrastructure.)

For every (record, haplotype) task of a synthetic batch: full score, banded score, the chain
certificate (lower bound) and the exact-match-run upper bound (oracle/vtx_certify.c).  Reports
validity (ub >= full always, cert <= banded always) and how many tasks have cert == ub.
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from vartrix_amd import synth  # noqa: E402
from vartrix_amd.abi import default_config  # noqa: E402


def certify(batch, cfg, threads, exact=True):
    L = oracle.lib()
    n = 2 * batch.n_records
    arrs = {k: np.zeros(n, np.int32) for k in ("full", "banded", "cert", "ub_exact", "ub", "passes", "pieces")}
    st = batch.as_struct()
    L.vtxo_batch_certify.restype = C.c_int
    L.vtxo_batch_certify(C.byref(st), C.byref(cfg), *[C.c_void_p(arrs[k].ctypes.data) if (exact or k != "ub_exact") else C.c_void_p(0)
                                                      for k in ("full", "banded", "cert", "ub_exact", "ub", "passes", "pieces")],
                         C.c_int(threads))
    return arrs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loci", type=int, default=200)
    ap.add_argument("--reads", type=int, default=64)
    ap.add_argument("--indel-frac", type=float, default=0.0)
    ap.add_argument("--sub-error", type=float, default=0.005)
    ap.add_argument("--jitter", type=int, default=0)
    ap.add_argument("--seed", type=int, default=20260926)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--no-exact", action="store_true")
    a = ap.parse_args()
    spec = synth.SynthSpec(n_loci=a.loci, n_barcodes=500, reads_per_locus=a.reads, indel_frac=a.indel_frac,
                           sub_error=a.sub_error, read_len_jitter=a.jitter, seed=a.seed)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="banded", n_barcodes=500)
    r = certify(batch, cfg, a.threads, exact=not a.no_exact)
    n = len(r["full"])
    has = r["cert"] >= 0
    out = {
        "tasks": int(n), "with_kmer": int(has.sum()),
        "ub_lt_full (must be 0)": int((r["ub"] < r["full"]).sum()),
        "cert_gt_banded (must be 0)": int((has & (r["cert"] > r["banded"])).sum()),
        "banded_ne_full": int((r["banded"] != r["full"]).sum()),
        "cert_eq_full": float((has & (r["cert"] == r["full"])).mean()),
        "cert_eq_ub": float((has & (r["cert"] == r["ub"])).mean()),
        "mean_pieces": float(r["pieces"].mean()), "max_pieces": int(r["pieces"].max()),
        "passes_hist": np.bincount(r["passes"]).tolist(),
    }
    if not a.no_exact:
        out["ub_exact_lt_full (must be 0)"] = int((r["ub_exact"] < r["full"]).sum())
        out["ub_lt_ub_exact (must be 0)"] = int((r["ub"] < r["ub_exact"]).sum())
        out["cert_eq_ub_exact"] = float((has & (r["cert"] == r["ub_exact"])).mean())
    gap = (r["ub"] - r["cert"])[has]
    out["ub_minus_cert_hist"] = np.bincount(np.clip(gap, 0, 12)).tolist()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
