#!/usr/bin/env python3
"""Writes tests/golden/band_kat.json: known-answer vectors of the banded aligner AS THE ORACLE RESTATES IT (oracle/vtx_oracle.c),
for a maintainer who holds bio 0.30.0's source (this repository does not: SURVEY.md §8c) to replay against the crate:

    banded::Aligner::new(-5, -1, |a, b| if a == b { 1 } else { -5 }, 6, 20).local(read, hap).score     (src/main.rs:898-901)

Every vector: read, haplotype, the banded score, the full-matrix score, and the band — per column j = 0 .. len(hap) of the DP matrix
the row range [lo[j], hi[j]) (rows 0 .. len(read)), run-length encoded.  The pairs are ADVERSARIAL for the recollected details
(include/vtx_band_semantics.h): chosen from the stress generators (tests/stress_batches.py) where the band matters — banded != full,
chains over several diagonals, reads hanging over the window, repeats — plus a share of ordinary ones.  INTEGRATION.md holds the
20-line Rust test that replays the file.  Deterministic: fixed seeds, no GPU.

    python tools/make_band_kat.py            # rewrites tests/golden/band_kat.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle           # noqa: E402
import stress_batches as SB         # noqa: E402


def rle(a):
    out, i = [], 0
    while i < len(a):
        j = i
        while j + 1 < len(a) and a[j + 1] == a[i]:
            j += 1
        out.append([int(a[i]), j - i + 1])
        i = j + 1
    return out


def pairs(batch, limit):
    hb, rb = batch.hap_arena.tobytes(), batch.read_arena.tobytes()
    n = 0
    for loc in batch.loci:
        for ri in range(int(loc["rec_begin"]), int(loc["rec_begin"]) + int(loc["rec_count"])):
            r = batch.records[ri]
            x = rb[int(r["read_off"]):int(r["read_off"]) + int(r["read_len"])]
            for off, ln in ((int(loc["ref_off"]), int(loc["ref_len"])), (int(loc["alt_off"]), int(loc["alt_len"]))):
                if len(x) >= 6 and ln >= 6:
                    yield x, hb[off:off + ln]
                    n += 1
                    if n >= limit:
                        return


# The four recollected details and the alternative each is switched to (oracle/vtx_oracle.c: vtxo_set_variant).  A vector
# DISCRIMINATES a detail when the oracle run with that alternative gives another banded score ("score") or another band ("band"):
# a maintainer replaying the file against the crate who sees exactly the vectors tagged with one detail fail knows which
# constant of include/vtx_band_semantics.h to change, and to what.
EDGE = 0x7fffffff
ALTERNATIVES = [("lazy_extension", "0", 0, 0), ("lazy_extension", "k", 0, 6), ("lazy_extension", "2 * w", 0, 40), ("lazy_extension", "to the matrix edge", 0, EDGE),
                ("kmer_last_anchor", "k - 1", 1, 5), ("no_seed", "empty band", 2, 0), ("sdpkpp_ties", "smaller match index", 3, 0)]


def discriminates(x, y, b0, lo0, hi0):
    L = oracle.lib()
    L.vtxo_set_variant.argtypes = [C.c_int, C.c_int]
    out = []
    for detail, alt, which, value in ALTERNATIVES:
        L.vtxo_set_variant(which, value)
        try:
            b = oracle.sw_banded(x, y)
            lo, hi, _ = oracle.band_create(x, y)
        finally:
            L.vtxo_set_variant(which, -1)
        band_moved = not (np.array_equal(lo, lo0) and np.array_equal(hi, hi0))
        if b != b0 or band_moved:
            out.append({"detail": detail, "alternative": alt, "banded_score_then": int(b), "band_changes": bool(band_moved)})
    return out


def no_seed_pairs(n, seed=11):
    """Reads that share no 6-mer with their haplotype (bio: Band::full_matrix): every fifth base of a window of the haplotype
    replaced, so that the full-matrix alignment still scores (runs of four matches) and an EMPTY band would not."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        hap = bytes(rng.choice(list(b"ACGT"), int(rng.integers(60, 200))).tolist())
        a = int(rng.integers(0, len(hap) - 40))
        read = bytearray(hap[a:a + int(rng.integers(30, min(120, len(hap) - a)))])
        for i in range(int(rng.integers(0, 5)), len(read), 5):
            read[i] = (b"ACGT".replace(bytes([read[i]]), b""))[int(rng.integers(0, 3))]
        read = bytes(read)
        if len(oracle.kmer_matches(read, hap)) == 0 and oracle.sw_full(read, hap) > 0:
            out.append((read, hap))
    return out


def whole_read_pairs(n, seed=29):
    """Reads that match their haplotype base for base on one diagonal — in random sequence, inside tandem repeats (the read then matches
    on SEVERAL diagonals) and hanging against either end of the window.  The device scores such a task len(read) before any k-mer
    probe (`whole_read`, vtx_fast_core.h); that rests on sdpkpp as recalled (include/vtx_band_semantics.h, last paragraph)."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        kind = len(out) % 3
        if kind == 1:                                            # a tandem repeat in the middle of the window
            unit = bytes(rng.choice(list(b"ACGT"), int(rng.integers(2, 13))).tolist())
            core = (unit * 40)[:int(rng.integers(30, 120))]
            hap = bytes(rng.choice(list(b"ACGT"), int(rng.integers(20, 80))).tolist()) + core + bytes(rng.choice(list(b"ACGT"), int(rng.integers(20, 80))).tolist())
        else:
            hap = bytes(rng.choice(list(b"ACGT"), int(rng.integers(80, 230))).tolist())
        m = int(rng.integers(20, min(150, len(hap)) + 1))
        a = 0 if kind == 2 and len(out) % 2 == 0 else (len(hap) - m if kind == 2 else int(rng.integers(0, len(hap) - m + 1)))
        out.append((hap[a:a + m], hap))
    return out


def main():
    vectors, seen = [], set()
    sources = [("error models and indels", SB.synthetic_batches(per_model=1, n_loci=30, reads=12), 500, 10),
               ("near repeats", SB.near_repeat_batches(trials=4), 400, 8),
               ("loci from real sequence", SB.real_sequence_batches(trials=2), 600, 12),
               ("real-read shapes", SB.real_shape_batches(trials=2), 600, 12),
               ("tandem repeats", SB.repeat_rich_batches(trials=4, loci=12, reads=8, pad_range=(30, 120)), 200, 10)]
    for kind, gen, per_batch, keep_plain in sources:
        for label, batch, _nb in gen:
            plain = 0
            for x, y in pairs(batch, per_batch):
                if (x, y) in seen:
                    continue
                b, f = oracle.sw_banded(x, y), oracle.sw_full(x, y)
                mt = oracle.kmer_matches(x, y)
                chain, _ = oracle.sdpkpp(mt) if len(mt) else ([], 0)
                diags = {int(mt[p, 1]) - int(mt[p, 0]) for p in chain}
                interesting = b != f or len(diags) > 1
                if not interesting:
                    if plain >= keep_plain:
                        continue
                    plain += 1
                seen.add((x, y))
                lo, hi, cells = oracle.band_create(x, y)
                vectors.append({"kind": kind, "from": label, "read": x.decode("latin-1"), "hap": y.decode("latin-1"),
                                "banded_score": int(b), "full_score": int(f), "band_cells": int(cells),
                                "chain_diagonals": len(diags), "kmer_matches": int(len(mt)), "lo_rle": rle(lo), "hi_rle": rle(hi)})
    # keep the file small: every vector where the band changes the score, then the others up to 320 in all
    vectors.sort(key=lambda v: (v["banded_score"] == v["full_score"], v["chain_diagonals"] <= 1))
    vectors = vectors[:320]
    # pairs without any common 6-mer: the only vectors that can tell "whole matrix" from "empty band"
    for x, y in no_seed_pairs(16):
        lo, hi, cells = oracle.band_create(x, y)
        vectors.append({"kind": "no common 6-mer", "from": "tools/make_band_kat.py: no_seed_pairs", "read": x.decode("latin-1"), "hap": y.decode("latin-1"),
                        "banded_score": int(oracle.sw_banded(x, y)), "full_score": int(oracle.sw_full(x, y)), "band_cells": int(cells),
                        "chain_diagonals": 0, "kmer_matches": 0, "lo_rle": rle(lo), "hi_rle": rle(hi)})
    for x, y in whole_read_pairs(24):
        lo, hi, cells = oracle.band_create(x, y)
        mt = oracle.kmer_matches(x, y)
        chain, _ = oracle.sdpkpp(mt) if len(mt) else ([], 0)
        vectors.append({"kind": "whole read", "from": "tools/make_band_kat.py: whole_read_pairs", "read": x.decode("latin-1"), "hap": y.decode("latin-1"),
                        "banded_score": int(oracle.sw_banded(x, y)), "full_score": int(oracle.sw_full(x, y)), "band_cells": int(cells),
                        "chain_diagonals": len({int(mt[p, 1]) - int(mt[p, 0]) for p in chain}), "kmer_matches": int(len(mt)),
                        "lo_rle": rle(lo), "hi_rle": rle(hi)})
    n_whole = 0
    for v in vectors:                                            # (any vector of the file, not only the ones made for it)
        v["whole_read"] = v["read"] in v["hap"]
        n_whole += v["whole_read"]
        assert not v["whole_read"] or v["banded_score"] == len(v["read"]) == v["full_score"], v["read"]
    per_detail = {}
    for v in vectors:
        x, y = v["read"].encode("latin-1"), v["hap"].encode("latin-1")
        v["discriminates"] = discriminates(x, y, v["banded_score"], np.concatenate([np.full(c, a) for a, c in v["lo_rle"]]),
                                           np.concatenate([np.full(c, a) for a, c in v["hi_rle"]]))
        for d in v["discriminates"]:
            e = per_detail.setdefault("%s -> %s" % (d["detail"], d["alternative"]), {"score": 0, "band": 0})
            e["score"] += d["banded_score_then"] != v["banded_score"]
            e["band"] += d["band_changes"]
    out = {"what": "known-answer vectors of banded::Aligner::new(-5, -1, score(1, -5), 6, 20).local(read, hap) as restated by oracle/vtx_oracle.c "
                   "(reference call site src/main.rs:898-901); band = per column j of the DP matrix the row range [lo, hi), run-length encoded as [value, count]",
           "k": 6, "w": 20, "scoring": {"match": 1, "mismatch": -5, "gap_open": -5, "gap_extend": -1},
           "recollected_details": {"lazy_extension": "2 * k", "kmer_last_anchor": "k", "no_seed": "full matrix", "sdpkpp_ties": "larger match index"},
           "discriminating_vectors": per_detail,
           "whole_read": "%d vectors carry whole_read = true: the read occurs in the haplotype base for base.  The device scores such a task len(read) "
                         "before any k-mer probe (vtx_fast_core.h: whole_read); the argument rests on sdpkpp as recalled (a jump always costs gap_open "
                         "+ gap_extend per base, a match's dp never exceeds x + k) and on the band's end event at + k — include/vtx_band_semantics.h, "
                         "last paragraph.  If the crate gives another score on exactly these vectors, rebuild with the shortcut off (libvtx_dev.so, "
                         "VTX_DIAG_ABLATE=10) and compare" % n_whole,
           "how_to_read_a_failure": "every vector carries `discriminates`: the recollected details whose ALTERNATIVE would change its banded score "
                                    "(banded_score_then) or its band.  If the crate disagrees with banded_score exactly on the vectors that list "
                                    "one detail, and agrees with their banded_score_then, that detail's constant in include/vtx_band_semantics.h "
                                    "takes the alternative.  kmer_last_anchor -> k - 1 changes no band at all (the cell after a chained k-mer's "
                                    "last base is anchored by add_gap's origin or by the next k-mer either way): it cannot be told apart and does not matter",
           "generator": "tools/make_band_kat.py", "vectors": vectors}
    path = os.path.join(ROOT, "tests", "golden", "band_kat.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    ne = sum(v["banded_score"] != v["full_score"] for v in vectors)
    print("%s: %d vectors, %d with banded != full, %d with a chain over several diagonals, %.0f KB" % (
        path, len(vectors), ne, sum(v["chain_diagonals"] > 1 for v in vectors), os.path.getsize(path) / 1024))
    for k, e in sorted(per_detail.items()):
        print("  %-44s score changes on %3d vectors, band on %3d" % (k, e["score"], e["band"]))


if __name__ == "__main__":
    main()
