import sys, os, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import oracle
from vartrix_amd import lib
from vartrix_amd.abi import default_config
import stress_batches as SB
label, batch, nb = next(iter(SB.synthetic_batches(per_model=1, n_loci=40, reads=16)))
n_tasks = min(2 * batch.n_records, 1500)
tasks = np.arange(n_tasks, dtype=np.uint32)
stride = int(max(batch.loci["ref_len"].max(), batch.loci["alt_len"].max())) + 1
with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=nb)) as ctx:
    ctx.submit(batch)
    lo, hi, status = ctx.debug_bands(tasks, stride)
    # one task at a time as well: does a task fail alone?
rec_locus = np.repeat(np.arange(batch.n_loci), batch.loci["rec_count"])
hb, rb = batch.hap_arena.tobytes(), batch.read_arena.tobytes()
bad = []
for t in range(n_tasks):
    rec = batch.records[t >> 1]; loc = batch.loci[rec_locus[t >> 1]]
    x = rb[int(rec["read_off"]):int(rec["read_off"]) + int(rec["read_len"])]
    off, ln = (int(loc["alt_off"]), int(loc["alt_len"])) if t & 1 else (int(loc["ref_off"]), int(loc["ref_len"]))
    y = hb[off:off + ln]
    olo, ohi, _ = oracle.band_create(x, y)
    dlo = lo[t, :len(y) + 1].astype(np.int64); dhi = hi[t, :len(y) + 1].astype(np.int64)
    e = ohi <= olo
    ok = status[t] == 0 and np.array_equal(dlo[~e], olo[~e]) and np.array_equal(dhi[~e], ohi[~e]) and not dhi[e].any()
    if not ok:
        bad.append(t)
        if len(bad) <= 4:
            mt = oracle.kmer_matches(x, y); path, sc = oracle.sdpkpp(mt)
            print("TASK", t, "slot-in-wave", t % 8, "status", status[t], "m", len(x), "n", len(y), "matches", len(mt), "chain score", sc)
            ch = mt[path]
            # sections of the oracle chain
            secs = []; s0 = 0
            for i in range(1, len(ch) + 1):
                if i == len(ch) or not (ch[i][0] == ch[i-1][0] + 1 and ch[i][1] == ch[i-1][1] + 1):
                    secs.append((int(ch[s0][0]), int(ch[s0][1]), i - s0)); s0 = i
            print(" oracle sections", secs)
            print(" read", x.decode()); print(" hap ", y.decode())
            print(" dev lo", dlo[:40].tolist()); print(" ora lo", olo[:40].tolist())
            print(" dev hi", dhi[-40:].tolist()); print(" ora hi", ohi[-40:].tolist())
print("bad", len(bad), "of", n_tasks, bad[:40])
with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=nb)) as ctx:
    ctx.submit(batch)
    for t in bad[:6]:
        l1, h1, s1 = ctx.debug_bands(np.array([t], np.uint32), stride)
        print("alone: task", t, "status", s1[0], "same as in batch:", np.array_equal(l1[0], lo[t]) and np.array_equal(h1[0], hi[t]), "lo[:8]", l1[0][:8].tolist())
os.environ["VTX_SWEEP_DBG"] = "1"
with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=nb)) as ctx:
    ctx.submit(batch)
    ctx.debug_bands(np.array(bad[:2] + [0], np.uint32), stride)
