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


W = 20
LAZY = 2 * K            # VTX_BAND_LAZY_EXT(k)
LAST = K                # VTX_BAND_KMER_LAST_ANCHOR(k)


def staircase_ranges(mt, chain, m, n):
    """band_coop_kernel's phase C: a 'lane' per chain link writes the anchor row of its own columns (pass 1), vertical remainders
    raise rmax of the column they happen in (pass 2, an atomic max on the device), then every column takes its row range from
    rmin / rmax of the columns within W (band_ranges).  Returns (lo, hi) as oracle.band_create gives them."""
    L = len(chain)
    link = [(int(mt[p, 0]), int(mt[p, 1])) for p in chain]
    fx, fy = link[0]
    d0 = min(fx, fy, LAZY)
    cA = fy - d0
    lx, ly = link[-1][0] + LAST, link[-1][1] + LAST
    d1 = min(m - lx, n - ly, LAZY)
    cB = ly + d1
    rmin, rmax = {}, {}

    def start_of(t):
        if t == 0:
            return fx - d0, fy - d0
        (qx, qy), (px, py) = link[t - 1], link[t]
        sq = 1 if (px == qx + 1 and py == qy + 1) else LAST
        return qx + sq, qy + sq

    for t in range(L + 1):                                   # pass 1 (any order: the links own disjoint columns)
        if t == L:
            for i in range(1, d1 + 1):
                rmin[ly + i] = rmax[ly + i] = lx + i
            continue
        px, py = link[t]
        ax, ay = start_of(t)
        if t == 0:
            rmin[ay] = rmax[ay] = ax
        dr, dc = px - ax, py - ay
        dg = min(dr, dc)
        for i in range(1, dg + 1):
            rmin[ay + i] = rmax[ay + i] = ax + i
        for c in range(ay + dg + 1, py + 1):
            rmin[c] = rmax[c] = px
        st = LAST
        if t + 1 < L and link[t + 1] == (px + 1, py + 1):
            st = 1
        for i in range(1, st + 1):
            rmin[py + i] = rmax[py + i] = px + i
    for t in range(L):                                       # pass 2
        px, py = link[t]
        ax, ay = start_of(t)
        dr, dc = px - ax, py - ay
        if dr > dc:
            rmax[ay + dc] = max(rmax[ay + dc], px)
    lo = np.zeros(n + 1, np.int32)
    hi = np.zeros(n + 1, np.int32)
    for j in range(n + 1):
        if j < cA - W or j > cB + W:
            lo[j], hi[j] = 0x7FFF, 0
            continue
        c0, c1 = max(j - W, cA), min(j + W, cB)
        lo[j] = max(rmin[c0] - W, 0)
        hi[j] = min(rmax[c1] + W + 1, m + 1)
    return lo, hi


def test_parallel_staircase_gives_the_reference_band():
    rng = np.random.default_rng(32)
    n_checked = 0
    for trial in range(400):
        kind = trial % 4
        y = bytes(rng.choice(list(b"ACGT"), 221).tolist())
        s0 = int(rng.integers(0, 40))
        if kind == 0:                                        # substitutions only
            x = bytearray(y[s0:s0 + 150])
            for e in rng.integers(0, len(x), size=int(rng.integers(0, 8))):
                x[int(e)] = b"ACGT"[int(rng.integers(0, 4))]
        elif kind == 1:                                      # deletion in the read: a vertical / horizontal remainder in the staircase
            cut, gap = int(rng.integers(30, 100)), int(rng.integers(1, 25))
            x = bytearray(y[s0:s0 + cut] + y[s0 + cut + gap:s0 + cut + gap + 90])
        elif kind == 2:                                      # insertion in the read
            cut = int(rng.integers(30, 100))
            x = bytearray(y[s0:s0 + cut] + bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 25))).tolist()) + y[s0 + cut:s0 + cut + 80])
        else:                                                # repeats: chains that hop
            unit = [b"AC", b"AAT", b"CAG", b"ACACAT"][int(rng.integers(0, 4))]
            y = (bytes(rng.choice(list(b"ACGT"), 40).tolist()) + unit * 30 + bytes(rng.choice(list(b"ACGT"), 60).tolist()))[:221]
            x = bytearray(y[s0:s0 + int(rng.integers(60, 150))])
            for e in rng.integers(0, len(x), size=int(rng.integers(0, 4))):
                x[int(e)] = b"ACGT"[int(rng.integers(0, 4))]
        x = bytes(x)
        mt = oracle.kmer_matches(x, y)
        if len(mt) == 0:
            continue
        chain, _ = oracle.sdpkpp(mt)
        lo, hi = staircase_ranges(mt.astype(np.int64), [int(p) for p in chain], len(x), len(y))
        want_lo, want_hi, _cells = oracle.band_create(x, y)
        # the oracle leaves columns outside the band as lo = hi (empty); compare the in-band columns and the emptiness of the others
        for j in range(len(y) + 1):
            if want_hi[j] > want_lo[j]:
                assert (lo[j], hi[j]) == (want_lo[j], want_hi[j]), (trial, j, lo[j], hi[j], want_lo[j], want_hi[j])
            else:
                assert hi[j] <= lo[j], (trial, j)
        n_checked += 1
    assert n_checked > 350
