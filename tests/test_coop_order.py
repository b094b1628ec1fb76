"""CPU restatement (numpy / plain Python, test infrastructure) of the ORDER band_coop_kernel (vartrix_amd/csrc/vtx_band.hip)
processes sdpkpp's events in, against the oracle's event-ordered sdpkpp (bio 0.30.0 sparse::sdpkpp as restated in
oracle/vtx_oracle.c).

The reference sorts START events (x, y) and END events (x + k, y + k) by (x, y), END before START at equal coordinates, and
keeps a max-Fenwick tree over columns.  The kernel goes row by row: iteration X handles the START events of row X - 1 and
then the END events of row X - k — START(X - 1) therefore sees every END up to row X - 1 and none of row X — with

  * the tree as packed words  v << 16 | match index  (v = dp + xe + ye): max of packed words == the reference's (value, index) order;
  * a dense "last match of the previous row at column c" array for the continuation partner (x - 1, y - 1) instead of a search;
  * best = max over END events of  dp << 16 | index.

This file states that procedure again with Python integers and checks chain and score against the oracle on random, repeat-rich
and low-entropy inputs: if the ordering argument or a tie rule were wrong it would show here, without a GPU.
"""
import numpy as np

from oracle import oracle

K = 6
NONE = 0xFFFF


def rowwise_sdpkpp(mt, m, n):
    """mt: matches (x, y) sorted by (x, y).  Returns (chain as match indices, score) the way band_coop_kernel computes them."""
    M = len(mt)
    rows = m - K + 1
    row_off = np.searchsorted(mt[:, 0], np.arange(rows + 1), side="left")
    tn = n + K + 2
    tree = [0] * (tn + 1)
    lastm = [[0] * (n + K + 4), [0] * (n + K + 4)]
    dp = [0] * M
    prev = [NONE] * M
    cont = [NONE] * M
    best = 0
    for X in range(0, m + 1):
        sr, er = X - 1, X - K
        if 0 <= sr < rows:                                   # START events of row X - 1 (they read the tree BEFORE this iteration's ENDs)
            for p in range(row_off[sr], row_off[sr + 1]):
                yv = int(mt[p, 1])
                bq = 0
                i = yv + 1
                while i > 0:
                    bq = max(bq, tree[i]); i -= i & (-i)
                dv, pr = K, NONE
                if bq:
                    cand = (bq >> 16) - 5 - (sr + yv) + K
                    if cand >= dv:
                        dv, pr = cand, bq & 0xFFFF
                lm = lastm[(sr + 1) & 1][yv - 1] if yv > 0 else 0
                cont[p] = (lm & 0xFFFF) if (sr > 0 and (lm >> 16) == sr) else NONE
                dp[p], prev[p] = dv, pr
            for p in range(row_off[sr], row_off[sr + 1]):    # (written after the row's reads: the kernel's lanes read before they write)
                lastm[sr & 1][int(mt[p, 1])] = ((sr + 1) << 16) | p
        if 0 <= er < rows:                                   # END events of row X - k
            for p in range(row_off[er], row_off[er + 1]):
                yv = int(mt[p, 1])
                c = cont[p]
                if c != NONE:
                    cand = dp[c] + 1
                    if cand > dp[p] or (cand == dp[p] and (prev[p] == NONE or c > prev[p])):
                        dp[p], prev[p] = cand, c
                v = dp[p] + X + yv + K
                packed = (v << 16) | p
                i = yv + K + 1
                while i <= tn:
                    tree[i] = max(tree[i], packed); i += i & (-i)
                best = max(best, (dp[p] << 16) | p)
    chain = []
    cur = best & 0xFFFF
    while cur != NONE:
        chain.append(cur); cur = prev[cur]
    return chain[::-1], best >> 16


def check(x, y):
    mt = oracle.kmer_matches(x, y)
    if len(mt) == 0 or len(mt) >= 0xFFFF:
        return 0
    want_path, want_score = oracle.sdpkpp(mt)
    have_path, have_score = rowwise_sdpkpp(mt.astype(np.int64), len(x), len(y))
    assert have_score == want_score, (x, y)
    assert list(have_path) == [int(v) for v in want_path], (x, y)
    return len(mt)


def test_rowwise_order_gives_the_reference_chain():
    rng = np.random.default_rng(31)
    total = 0
    for trial in range(600):
        kind = trial % 5
        if kind == 0:                                        # iid read against its window, a few errors
            y = bytes(rng.choice(list(b"ACGT"), 201).tolist())
            s0 = int(rng.integers(0, 60))
            x = bytearray(y[s0:s0 + 150])
            for e in rng.integers(0, len(x), size=int(rng.integers(0, 6))):
                x[int(e)] = b"ACGT"[int(rng.integers(0, 4))]
        elif kind == 1:                                      # tandem repeats: many matches per row, ties everywhere
            unit = [b"A", b"AC", b"AAT", b"ACGT", b"CAG", b"ACACAT"][int(rng.integers(0, 6))]
            y = (unit * 60)[:int(rng.integers(60, 220))]
            y = bytes(y[:40]) + bytes(rng.choice(list(b"ACGT"), 20).tolist()) + bytes(y[40:])
            s0 = int(rng.integers(0, 30))
            x = bytearray(y[s0:s0 + int(rng.integers(30, 150))])
            for e in rng.integers(0, len(x), size=int(rng.integers(0, 4))):
                x[int(e)] = b"ACGT"[int(rng.integers(0, 4))]
        elif kind == 2:                                      # two-letter alphabet
            y = bytes(rng.choice(list(b"AC"), int(rng.integers(30, 120))).tolist())
            x = bytearray(rng.choice(list(b"AC"), int(rng.integers(12, 80))).tolist())
        elif kind == 3:                                      # read with a deletion / insertion against the window (two diagonals)
            y = bytes(rng.choice(list(b"ACGT"), 221).tolist())
            s0 = int(rng.integers(0, 40))
            cut = int(rng.integers(30, 100))
            gap = int(rng.integers(1, 21))
            x = bytearray(y[s0:s0 + cut] + y[s0 + cut + gap:s0 + cut + gap + 100])
        else:                                                # poly-A stretch inside
            y = bytearray(rng.choice(list(b"ACGT"), 201).tolist())
            a0 = int(rng.integers(40, 120))
            y[a0:a0 + int(rng.integers(12, 40))] = b"A" * 40
            y = bytes(y[:201])
            s0 = int(rng.integers(0, 50))
            x = bytearray(y[s0:s0 + 150])
        total += check(bytes(x), bytes(y))
    assert total > 100000
