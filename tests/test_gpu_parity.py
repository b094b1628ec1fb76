"""GPU parity: the HIP path behind the C-ABI vs the CPU oracle, bit-exact.

Small/medium seeded batches are compared record-by-record (scores) and
triplet-by-triplet (COO) with the oracle; the golden BAM fixtures go through
the device too.  Full-size runs are covered by size-independent properties in
test_gpu_properties.py.
"""
import os

import numpy as np
import pytest

from oracle import oracle, refpipe
from vartrix_amd import lib, synth
from vartrix_amd.abi import LOCUS_DTYPE, RECORD_DTYPE, PackedBatch, default_config

pytestmark = pytest.mark.gpu


def run_device(batch, cfg):
    with lib.Context(cfg) as ctx:
        ctx.submit(batch)
        ctx.run()
        ref, alt = ctx.fetch_scores()
        coo = ctx.fetch_coo()
        cells = ctx.cells()
    return ref, alt, coo, cells


def assert_same(batch, cfg, threads=8):
    ref, alt, coo, cells = run_device(batch, cfg)
    oref, oalt = oracle.batch_scores(batch, cfg, threads=threads)
    bad = np.nonzero((ref != oref) | (alt != oalt))[0]
    assert bad.size == 0, "first mismatch at record %d: device (%d,%d) oracle (%d,%d)" % (
        bad[0], ref[bad[0]], alt[bad[0]], oref[bad[0]], oalt[bad[0]])
    ocoo = oracle.batch_reduce(batch, cfg, oref, oalt)
    for k in ("row", "col", "alt", "ref", "unk"):
        assert np.array_equal(coo[k], ocoo[k]), k
    # alt_frac may hold NaN (0/0, src/main.rs:1140): compare bit patterns
    assert np.array_equal(coo["value"].view(np.uint64), ocoo["value"].view(np.uint64))
    assert np.array_equal(coo["ref_value"], ocoo["ref_value"])
    if cfg.aligner == 1:
        assert cells == oracle.batch_cells(batch, cfg)
    return ref, alt, coo


ALIGNERS = ["full", "banded"]


@pytest.mark.parametrize("aligner", ALIGNERS)
@pytest.mark.parametrize("mode", ["consensus", "alt_frac", "coverage"])
@pytest.mark.parametrize("umi", [0, 1])
def test_snv_batch(mode, umi, aligner):
    spec = synth.SynthSpec(n_loci=96, n_barcodes=300, reads_per_locus=48, use_umi=bool(umi), seed=7 + umi)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner=aligner, scoring_mode=mode, use_umi=umi, n_barcodes=spec.n_barcodes)
    assert_same(batch, cfg)


@pytest.mark.parametrize("aligner", ALIGNERS)
@pytest.mark.parametrize("mode", ["consensus", "alt_frac", "coverage"])
def test_indel_umi_ragged_batch(mode, aligner):
    """Config-5 shape: SNV + indels <= 20bp, UMIs with disagreeing members, ragged read lengths."""
    spec = synth.SynthSpec(n_loci=128, n_barcodes=150, reads_per_locus=40, indel_frac=0.5, use_umi=True,
                           read_len_jitter=120, umi_flip=0.2, seed=11)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner=aligner, scoring_mode=mode, use_umi=1, n_barcodes=spec.n_barcodes)
    assert_same(batch, cfg)


def test_banded_differs_from_full_and_device_follows_the_band():
    """On indel batches the band changes some scores; the device must track the banded oracle, and report
    how many alignments needed the band-masked DP (the certificate handles the rest)."""
    spec = synth.SynthSpec(n_loci=256, n_barcodes=200, reads_per_locus=48, indel_frac=0.6, read_len_jitter=60, seed=23,
                           sub_error=0.02)
    batch = synth.make_batch(spec)
    cfgb = default_config(aligner="banded", scoring_mode="coverage", n_barcodes=spec.n_barcodes)
    with lib.Context(cfgb) as ctx:
        ctx.submit(batch)
        ctx.run()
        ref, alt = ctx.fetch_scores()
        hard = ctx.timing().hard_tasks
    oref, oalt = oracle.batch_scores(batch, cfgb, threads=8)
    assert np.array_equal(ref, oref) and np.array_equal(alt, oalt)
    fref, falt = oracle.batch_scores(batch, default_config(aligner="full", n_barcodes=spec.n_barcodes), threads=8)
    n_diff = int((fref != oref).sum() + (falt != oalt).sum())
    assert n_diff > 0, "workload too clean to exercise the band"
    assert n_diff <= hard <= 2 * batch.n_records
    print("banded != full on %d of %d alignments; %d took the masked DP" % (n_diff, 2 * batch.n_records, hard))


def test_banded_low_complexity_overflow_slabs():
    """Poly-A / tandem-repeat reads against repeat-rich haplotypes: thousands of k-mer matches per
    alignment, which overflow the first scratch slab and take the larger-slab rerun."""
    rng = np.random.default_rng(99)
    haps, reads = [], []
    for i in range(6):
        flank = bytes(rng.choice(list(b"ACGT"), 80).tolist())
        rep = [b"A" * 60, b"AC" * 30, b"AAAAAT" * 10, b"A" * 25 + b"G" + b"A" * 34, b"ACG" * 20, b"T" * 60][i]
        ref = flank + rep + flank[::-1]
        alt = flank + rep[:30] + b"C" + rep[31:] + flank[::-1]
        haps.append((ref, alt))
        rl = []
        for k in range(10):
            o = int(rng.integers(0, 60))
            rd = bytearray((flank + rep + flank[::-1])[o:o + 150])
            if k % 3 == 0:
                rd = bytearray(b"A" * 150) if i % 2 == 0 else bytearray((b"AC" * 75))
            rl.append((k % 5, 0, bytes(rd)))
        reads.append(rl)
    batch = _manual_batch(haps, reads, 8)
    cfg = default_config(aligner="banded", scoring_mode="coverage", n_barcodes=8)
    assert_same(batch, cfg, threads=4)


@pytest.mark.parametrize("read_len", [1, 5, 17, 31, 32, 33, 64, 65, 100, 151, 160, 161, 200, 250, 256, 257, 400, 1000])
def test_read_length_buckets(read_len):
    """Every kernel shape (rows-per-lane x lanes-per-record) incl. bucket edges."""
    spec = synth.SynthSpec(n_loci=12, n_barcodes=40, reads_per_locus=12, read_len=read_len,
                           padding=max(100, read_len // 2 + 20), indel_frac=0.4, seed=read_len)
    batch = synth.make_batch(spec)
    for aligner in ALIGNERS:
        cfg = default_config(aligner=aligner, scoring_mode="coverage", n_barcodes=spec.n_barcodes)
        assert_same(batch, cfg, threads=8)


def _manual_batch(haps, reads_per_locus, n_barcodes):
    loci, recs, hb, rb = [], [], bytearray(), bytearray()
    for i, ((ref, alt), reads) in enumerate(zip(haps, reads_per_locus)):
        begin = len(recs)
        for cell, umi, seq in sorted(reads, key=lambda t: (t[0], t[1])):
            recs.append((len(rb), len(seq), cell, umi))
            rb += seq
        loci.append((i, begin, len(recs) - begin, len(hb), len(ref), len(hb) + len(ref), len(alt), 0))
        hb += ref + alt
    return PackedBatch(np.array(loci, LOCUS_DTYPE).reshape(-1), np.array(recs, RECORD_DTYPE).reshape(-1),
                       np.frombuffer(bytes(hb), np.uint8), np.frombuffer(bytes(rb), np.uint8))


def test_edge_cases():
    """Empty loci, empty reads, empty ALT haplotype flank, N / lower-case bytes (byte equality,
    src/main.rs:898), ties (UNKNOWN), sub-threshold reads (None), a cell with only None calls."""
    rng = np.random.default_rng(3)
    g = bytes(rng.choice(list(b"ACGT"), 400).tolist())
    ref = g[100:301]
    alt = g[100:200] + b"t" + g[201:301]          # lower-case ALT allele never matches upper-case reads
    altn = g[100:200] + b"N" + g[201:301]
    haps = [(ref, alt), (ref, altn), (ref, ref), (g[0:50], g[0:20] + g[30:50]), (ref, alt)]
    reads = [
        [(0, 0, g[120:270]), (0, 0, g[150:300]), (1, 0, b""), (2, 5, b"ACGT"), (3, 1, g[190:215])],
        [(0, 0, g[120:200] + b"N" + g[201:270]), (0, 1, g[120:270]), (7, 0, b"N" * 60)],
        [(4, 0, g[130:280]), (4, 0, g[131:281])],                       # identical haps -> ties -> UNKNOWN
        [(1, 0, g[0:50]), (2, 0, g[0:20] + g[30:50]), (5, 0, g[5:45])],
        [],                                                              # locus with no reads
    ]
    batch = _manual_batch(haps, reads, 8)
    for aligner in ALIGNERS:
        for mode in ("consensus", "alt_frac", "coverage"):
            for umi in (0, 1):
                cfg = default_config(aligner=aligner, scoring_mode=mode, use_umi=umi, n_barcodes=8)
                ref_s, alt_s, coo = assert_same(batch, cfg, threads=1)
    # identical haplotypes: every read ties
    l2 = batch.loci[2]
    sl = slice(int(l2["rec_begin"]), int(l2["rec_begin"] + l2["rec_count"]))
    assert np.array_equal(ref_s[sl], alt_s[sl])


def test_empty_batch():
    batch = PackedBatch(np.zeros(0, LOCUS_DTYPE), np.zeros(0, RECORD_DTYPE), np.zeros(0, np.uint8), np.zeros(0, np.uint8))
    cfg = default_config(aligner="full", n_barcodes=4)
    ref, alt, coo, cells = run_device(batch, cfg)
    assert ref.size == 0 and coo["row"].size == 0 and cells == 0


def test_submit_validation():
    spec = synth.SynthSpec(n_loci=4, n_barcodes=10, reads_per_locus=8)
    batch = synth.make_batch(spec)
    with lib.Context(default_config(aligner="full", n_barcodes=10)) as ctx:
        bad = PackedBatch(batch.loci.copy(), batch.records.copy(), batch.hap_arena, batch.read_arena)
        bad.records["cell_index"][0] = 99
        with pytest.raises(lib.VtxError):
            ctx.submit(bad)
        bad = PackedBatch(batch.loci.copy(), batch.records[::-1].copy(), batch.hap_arena, batch.read_arena)
        with pytest.raises(lib.VtxError):
            ctx.submit(bad)
        with pytest.raises(lib.VtxError):
            ctx.run()          # nothing resident after a failed submit
        ctx.submit(batch)
        ctx.run()


def test_rerun_is_idempotent():
    spec = synth.SynthSpec(n_loci=32, n_barcodes=64, reads_per_locus=32, use_umi=True)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="full", scoring_mode="alt_frac", use_umi=1, n_barcodes=64)
    with lib.Context(cfg) as ctx:
        ctx.submit(batch)
        ctx.run()
        a = ctx.fetch_coo()
        ctx.run()
        b = ctx.fetch_coo()
    for k in a:
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))


GOLDEN = [
    ("consensus", False, "barcodes.tsv", "test_consensus.mtx", None),
    ("alt_frac", False, "barcodes.tsv", "test_frac.mtx", None),
    ("coverage", False, "barcodes.tsv", "test_coverage.mtx", "test_coverage_ref.mtx"),
    ("coverage", True, "barcodes.tsv", "test_coverage_umi.mtx", "test_coverage_ref_umi.mtx"),
    ("coverage", True, "barcodes.tsv.gz", "test_coverage_umi.mtx", "test_coverage_ref_umi.mtx"),
]


@pytest.mark.parametrize("aligner", ALIGNERS)
@pytest.mark.parametrize("case", GOLDEN, ids=["consensus", "alt_frac", "coverage", "coverage_umi", "coverage_umi_gz"])
def test_reference_fixtures_on_device(golden_dir, case, aligner):
    """The reference's own regression tests (src/main.rs:1208-1390), device path: CSR-equal .mtx."""
    mode, umi, bcfile, main_fx, ref_fx = case
    g = golden_dir
    bcs = refpipe.load_barcodes(os.path.join(g, bcfile))
    vcf = refpipe.read_vcf(os.path.join(g, "test.vcf"))
    batch, _ = refpipe.pack(vcf, refpipe.read_fasta(os.path.join(g, "test.fa")),
                            refpipe.read_bam(os.path.join(g, "test.bam")), bcs, refpipe.Args(use_umi=umi))
    cfg = default_config(aligner=aligner, scoring_mode=mode, use_umi=int(umi), n_barcodes=len(bcs))
    ref, alt, coo = assert_same(batch, cfg, threads=1)
    shape, want = refpipe.read_mtx(os.path.join(g, main_fx))
    assert shape == (len(vcf), len(bcs))
    assert {(int(r), int(c)): float(v) for r, c, v in zip(coo["row"], coo["col"], coo["value"])} == want
    if ref_fx:
        _, want = refpipe.read_mtx(os.path.join(g, ref_fx))
        assert {(int(r), int(c)): float(v) for r, c, v in zip(coo["row"], coo["col"], coo["ref_value"])} == want


def test_band_kernel_table_sizes_agree(tmp_path):
    """The k-mer table geometry of band_run_kernel (VTX_BAND_HEADS: 256 / 512 / 2048-entry head arrays, hence
    different chain lengths, loci per pass and probe orders inside a bucket) must not change any score
    (separate processes: the knob is read per launch from the environment of the process)."""
    import subprocess
    import sys
    code = '''
import numpy as np, sys
sys.path.insert(0, %r)
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
spec = synth.SynthSpec(n_loci=200, n_barcodes=100, reads_per_locus=40, indel_frac=0.5, read_len_jitter=50, seed=5, sub_error=0.02)
b = synth.make_batch(spec)
with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=100)) as ctx:
    ctx.submit(b); ctx.run(); r, a = ctx.fetch_scores()
np.save(sys.argv[1], np.stack([r, a]))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for heads in ("256", "512", "2048"):
        out = str(tmp_path / (heads + ".npy"))
        env = dict(os.environ, VTX_LIB_VARIANT="dev", VTX_BAND_HEADS=heads)
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=300)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    spec = synth.SynthSpec(n_loci=200, n_barcodes=100, reads_per_locus=40, indel_frac=0.5, read_len_jitter=50, seed=5, sub_error=0.02)
    batch = synth.make_batch(spec)
    oref, oalt = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=100), threads=8)
    assert np.array_equal(outs[0][0], oref) and np.array_equal(outs[0][1], oalt)


def test_certificate_decides_most_clean_alignments():
    """The DP-free certificate (cert == ub, vtx_band.hip) must leave only a small residue of a clean SNV workload
    to the band-masked DP — and whatever it decides must equal the oracle's banded score."""
    spec = synth.SynthSpec(n_loci=400, n_barcodes=300, reads_per_locus=64, seed=31)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="banded", scoring_mode="consensus", n_barcodes=spec.n_barcodes)
    with lib.Context(cfg) as ctx:
        ctx.submit(batch)
        ctx.run()
        ref, alt = ctx.fetch_scores()
        hard = ctx.timing().hard_tasks
    oref, oalt = oracle.batch_scores(batch, cfg, threads=os.cpu_count() or 8)
    assert np.array_equal(ref, oref) and np.array_equal(alt, oalt)
    frac = hard / (2.0 * batch.n_records)
    print("hard fraction %.4f" % frac)
    assert frac < 0.08


def test_records_beyond_the_fast_limits_take_the_exact_slow_path():
    """A 3 000-base read, a haplotype pair of 2 600 / 5 400 bases (a long sequence-resolved insertion), and ordinary
    records around them in one batch: the reference aligns any length (src/main.rs:898-901), so nothing may be rejected
    and every score must match the oracle — both flavours, all modes going through the same reduction."""
    rng = np.random.default_rng(41)
    g = bytes(rng.choice(list(b"ACGT"), 12000).tolist())

    def mutate(seq, n):
        b = bytearray(seq)
        for _ in range(n):
            b[int(rng.integers(0, len(b)))] = b"ACGT"[int(rng.integers(0, 4))]
        return bytes(b)
    ins = bytes(rng.choice(list(b"ACGT"), 2800).tolist())
    haps = [
        (g[100:301], g[100:200] + b"T" + g[201:301]),                         # ordinary SNV locus
        (g[1000:3600], g[1000:2300] + ins + g[2300:3600]),                    # haplotypes beyond the LDS tables
        (g[5000:5201], g[5000:5100] + b"G" + g[5101:5201]),                   # ordinary locus with one very long read
        (g[7000:7201], g[7000:7100] + g[7108:7201]),
    ]
    reads = [
        [(0, 0, g[120:270]), (1, 0, mutate(g[130:280], 2)), (2, 0, g[100:200] + b"T" + g[201:260])],
        [(0, 0, g[2200:2350]), (1, 0, g[2250:2300] + ins[:100]), (2, 0, mutate(g[1000:2300] + ins[:700], 12)),
         (3, 0, ins[2700:] + g[2300:2400]), (4, 0, b"ACG")],
        [(0, 0, g[5050:5200]), (1, 0, mutate(g[4000:5100] + b"G" + g[5101:7000], 25)), (2, 0, g[5090:5101] + b"G" + g[5101:5160])],
        [(0, 0, g[7010:7160]), (0, 1, g[7020:7100] + g[7108:7180])],
    ]
    batch = _manual_batch(haps, reads, 8)
    assert int(batch.records["read_len"].max()) == 3000 and int(batch.loci["alt_len"].max()) == 5400
    for aligner in ALIGNERS:
        for mode, umi in (("coverage", 0), ("alt_frac", 1)):
            cfg = default_config(aligner=aligner, scoring_mode=mode, use_umi=umi, n_barcodes=8)
            ref_s, alt_s, coo = assert_same(batch, cfg, threads=4)
    assert ref_s.max() > 1000          # the long read really aligned end to end


@pytest.mark.parametrize("raw", [False, True])
def test_slow_records_with_ordinary_reads_only(raw):
    """--padding 1500: every haplotype (3 001 bases) exceeds the fast kernels' tables, so EVERY record is on the slow
    list — with reads of 150 bases only.  The slow path's DP columns are sized by the longest read among the slow
    records (round-2 ADVICE: they were sized by the longest read above 1 024 bases, i.e. by nothing here, and the
    columns of neighbouring lanes overlapped).  Both submit paths compute that maximum."""
    spec = synth.SynthSpec(n_loci=5, n_barcodes=24, reads_per_locus=14, padding=1500, seed=77, read_len_jitter=30)
    batch = synth.make_batch(spec)
    assert int(batch.loci["ref_len"].min()) == 3001 and int(batch.records["read_len"].max()) <= 150
    for aligner in ALIGNERS:
        cfg = default_config(aligner=aligner, scoring_mode="coverage", use_umi=0, n_barcodes=spec.n_barcodes)
        if not raw:
            ref_s, _, _ = assert_same(batch, cfg, threads=4)
            assert ref_s.max() > 100
            continue
        rb, barcodes = synth.make_raw(batch, spec.n_barcodes, use_umi=False, seed=3)
        with lib.Context(cfg) as ctx:
            ctx.set_barcodes(barcodes)
            ctx.submit_raw(rb)
            ctx.run()
            coo = ctx.fetch_coo()
        oref, oalt = oracle.batch_scores(batch, cfg, threads=4)
        ocoo = oracle.batch_reduce(batch, cfg, oref, oalt)
        for k in ("row", "col", "alt", "ref", "unk"):
            assert np.array_equal(coo[k], ocoo[k]), k


@pytest.mark.parametrize("hook", ["VTX_BAND_HARD_CAP", "VTX_BAND_SLOTS"])
def test_band_buffer_caps_spill_into_the_general_kernel(tmp_path, hook):
    """Polyline records, pending records and band slots are sized for a fraction of the tasks.  VTX_BAND_HARD_CAP=3 leaves
    three of each kind: whatever exceeds them must take the general kernel's route and still come out exact.
    VTX_BAND_SLOTS=3 leaves three band slots only: the masked DP then runs in many slices of three hard tasks
    (separate process: the hooks are read from the environment)."""
    import subprocess
    import sys
    code = '''
import numpy as np, sys
sys.path.insert(0, %r)
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
spec = synth.SynthSpec(n_loci=150, n_barcodes=100, reads_per_locus=48, indel_frac=0.5, read_len_jitter=50, seed=15, sub_error=0.03)
b = synth.make_batch(spec)
with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=100)) as ctx:
    ctx.submit(b); ctx.run(); r, a = ctx.fetch_scores(); t = ctx.timing()
np.save(sys.argv[1], np.stack([r, a])); print("hard", t.hard_tasks, "overflow", t.overflow_tasks)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "capped.npy")
    # (VTX_BAND_HARD_CAP bounds the polyline / pending records of the round-3 path: VTX_BAND_LEGACY=1; the band slots bound both paths)
    extra = {hook: "3", "VTX_BAND_LEGACY": "1"} if hook == "VTX_BAND_HARD_CAP" else {hook: "3"}
    r = subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, VTX_LIB_VARIANT="dev", **extra), timeout=300,
                       capture_output=True, text=True)
    got = np.load(out)
    spec = synth.SynthSpec(n_loci=150, n_barcodes=100, reads_per_locus=48, indel_frac=0.5, read_len_jitter=50, seed=15, sub_error=0.03)
    batch = synth.make_batch(spec)
    oref, oalt = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=100), threads=8)
    assert np.array_equal(got[0], oref) and np.array_equal(got[1], oalt)
    overflow = int(r.stdout.split("overflow")[1])
    hard = int(r.stdout.split("hard")[1].split()[0])
    if hook == "VTX_BAND_HARD_CAP":
        assert overflow > 100, r.stdout      # the caps really were exceeded
    else:
        assert hard > 100, r.stdout          # many slices of three


@pytest.mark.parametrize("reads_per_locus", [5, 40])
def test_lds_table_variants_at_shallow_depth(tmp_path, reads_per_locus):
    """Below ~208 tasks per locus the k-mer tables live in global memory (band_tables_kernel); the LDS-table variants of
    band_run_kernel (one-wavefront workgroups below 64 tasks per locus, 256-lane workgroups above) remain for batches
    whose tables would not fit the buffer.  VTX_BAND_GT_MAX_TPL=0 forces them (separate process: read from the
    environment); both must equal the oracle."""
    import subprocess
    import sys
    code = '''
import numpy as np, sys
sys.path.insert(0, %r)
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
spec = synth.SynthSpec(n_loci=400, n_barcodes=100, reads_per_locus=int(sys.argv[2]), indel_frac=0.3, read_len_jitter=40, seed=33, sub_error=0.02)
b = synth.make_batch(spec)
with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=100)) as ctx:
    ctx.submit(b); ctx.run(); r, a = ctx.fetch_scores()
np.save(sys.argv[1], np.stack([r, a]))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = synth.SynthSpec(n_loci=400, n_barcodes=100, reads_per_locus=reads_per_locus, indel_frac=0.3, read_len_jitter=40, seed=33, sub_error=0.02)
    batch = synth.make_batch(spec)
    oref, oalt = oracle.batch_scores(batch, default_config(aligner="banded", n_barcodes=100), threads=8)
    # (the third run: a table buffer for ~45 of the 400 loci — the stage then runs in chunks of tasks whose loci fit)
    for i, env in enumerate((dict(VTX_BAND_GT_MAX_TPL="0"), dict(VTX_BAND_GT_MAX_TPL="100000"), dict(VTX_BAND_GT_BYTES="400000"))):
        out = str(tmp_path / ("t%d.npy" % i))
        subprocess.run([sys.executable, "-c", code, out, str(reads_per_locus)], check=True,
                       env=dict(os.environ, VTX_LIB_VARIANT="dev", **env), timeout=300, capture_output=True, text=True)
        got = np.load(out)
        assert np.array_equal(got[0], oref) and np.array_equal(got[1], oalt), env


@pytest.mark.parametrize("kw", [dict(reads_per_locus=4), dict(reads_per_locus=16), dict(reads_per_locus=8, depth_sigma=1.0),
                                dict(reads_per_locus=3, indel_frac=0.4, read_len_jitter=60), dict(reads_per_locus=70)])
def test_shallow_and_mixed_depth_loci(kw):
    """Real single-cell loci are shallow: a handful of reads per locus, log-normal over loci.  The band kernel then runs
    one wavefront per workgroup with all of its loci resident (tables of up to 32 haplotypes), the DP kernels fall back
    from the per-locus LUTs when a workgroup spans too many loci: every score and triplet vs the oracle, both flavours."""
    spec = synth.SynthSpec(n_loci=1500, n_barcodes=400, seed=77, **kw)
    batch = synth.make_batch(spec)
    for aligner in ALIGNERS:
        cfg = default_config(aligner=aligner, scoring_mode="coverage", n_barcodes=spec.n_barcodes)
        assert_same(batch, cfg, threads=os.cpu_count() or 8)


@pytest.mark.parametrize("seed,sub_error,indel_frac,read_len,padding", [
    (101, 0.03, 0.5, 150, 100), (102, 0.08, 0.3, 100, 60), (103, 0.01, 0.7, 151, 120), (104, 0.15, 0.5, 80, 100)])
def test_banded_large_random_vs_oracle(seed, sub_error, indel_frac, read_len, padding):
    """Stress of the band kernels' rare paths (breakpoints, open-piece splits, list compaction, fallback):
    tens of thousands of noisy, ragged reads over indel loci, every score checked against the oracle."""
    spec = synth.SynthSpec(n_loci=600, n_barcodes=500, reads_per_locus=80, read_len=read_len, padding=padding,
                           indel_frac=indel_frac, sub_error=sub_error, read_len_jitter=read_len // 3, seed=seed)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="banded", scoring_mode="coverage", n_barcodes=spec.n_barcodes)
    with lib.Context(cfg) as ctx:
        ctx.submit(batch)
        ctx.run()
        ref, alt = ctx.fetch_scores()
        hard = ctx.timing().hard_tasks
    oref, oalt = oracle.batch_scores(batch, cfg, threads=os.cpu_count() or 8)
    bad = np.nonzero((ref != oref) | (alt != oalt))[0]
    assert bad.size == 0, "%d mismatches, first at record %d: device (%d,%d) oracle (%d,%d)" % (
        bad.size, bad[0], ref[bad[0]], alt[bad[0]], oref[bad[0]], oalt[bad[0]])
    print("seed %d: %d records ok, %d hard tasks" % (seed, batch.n_records, hard))


def test_banded_repeat_rich_genome_vs_oracle():
    """Haplotypes and reads drawn from a low-entropy genome (tandem repeats, homopolymers): hundreds of
    k-mer matches per alignment, many concurrent diagonals — exercises piece-list compaction and the
    general fallback kernel."""
    rng = np.random.default_rng(77)
    units = [b"A", b"AC", b"AAT", b"ACGT", b"AAAAC", b"AG", b"T", b"CAG"]
    g = bytearray()
    while len(g) < 60000:
        u = units[int(rng.integers(0, len(units)))]
        g += u * int(rng.integers(3, 30))
        g += bytes(rng.choice(list(b"ACGT"), int(rng.integers(5, 40))).tolist())
    g = bytes(g)
    haps, reads = [], []
    for i in range(120):
        p = int(rng.integers(200, len(g) - 400))
        ref = g[p - 100:p + 101]
        alt = ref[:100] + bytes([b"ACGT"[(b"ACGT".index(ref[100:101]) + 1) % 4]]) + ref[101:]
        haps.append((ref, alt))
        rl = []
        for k in range(40):
            s = p - int(rng.integers(0, 150))
            rd = bytearray(g[s:s + 150])
            if k % 2:
                rd[p - s] = alt[100]
            for e in np.nonzero(rng.random(len(rd)) < 0.02)[0]:
                rd[e] = b"ACGT"[int(rng.integers(0, 4))]
            rl.append((int(rng.integers(0, 30)), 0, bytes(rd)))
        reads.append(rl)
    batch = _manual_batch(haps, reads, 30)
    for aligner in ALIGNERS:
        cfg = default_config(aligner=aligner, scoring_mode="coverage", n_barcodes=30)
        assert_same(batch, cfg, threads=os.cpu_count() or 8)


@pytest.mark.parametrize("aligner", ALIGNERS)
def test_shared_prefix_kernel_corner_cases(aligner):
    """sw_full_duo_kernel shares the REF == ALT prefix between two reads of a 16-lane row.  Corner cases of
    its phase arithmetic: prefixes shorter than 4 columns (variant at the haplotype start), haplotypes shorter
    than a 16-step window, neighbouring loci with different prefix lengths paired in one row, an odd number
    of records, identical REF and ALT (prefix = the whole haplotype), variant in the last columns, reads of
    mixed length inside a row, and a byte outside ACGTN in one read of a pair (both are re-scored)."""
    rng = np.random.default_rng(2024)
    g = bytes(rng.choice(list(b"ACGT"), 3000).tolist())

    def mut(c):
        return bytes([b"ACGT"[(b"ACGT".index(bytes([c])) + 1) % 4]])

    haps, reads = [], []
    specs = [  # (left flank, right flank, variant kind)
        (0, 100, "snv"), (1, 100, "snv"), (3, 60, "ins"), (5, 100, "del"), (100, 100, "snv"), (100, 0, "snv"),
        (100, 2, "ins"), (7, 6, "snv"), (2, 3, "snv"), (60, 100, "same"), (100, 100, "del"), (33, 100, "ins"),
        (12, 9, "del"), (100, 100, "snv")]
    for k, (lf, rf, kind) in enumerate(specs):
        p = 200 + 190 * k
        left, right = g[p - lf:p], g[p + 1:p + 1 + rf]
        refa = g[p:p + 1]
        if kind == "snv":
            ref, alt = left + refa + right, left + mut(refa[0]) + right
        elif kind == "ins":
            ref, alt = left + refa + right, left + refa + b"ACGTTGCA"[:1 + k % 7] + right
        elif kind == "del":
            ref, alt = left + g[p:p + 6] + g[p + 6:p + 6 + rf], left + refa + g[p + 6:p + 6 + rf]
        else:
            ref = alt = left + refa + right
        haps.append((ref, alt))
        rl = []
        n_reads = 1 + (k * 5) % 9           # 1..9 reads: rows pair reads of neighbouring loci
        for q in range(n_reads):
            src = alt if q % 2 else ref
            ln = int(rng.integers(1, max(2, min(150, len(src)) + 1)))
            s0 = int(rng.integers(0, len(src) - ln + 1))
            rd = bytearray(src[s0:s0 + ln])
            if rng.random() < 0.3 and ln > 4:
                rd[int(rng.integers(0, ln))] = mut(rd[0])[0]
            if k == 4 and q == 1:
                rd[len(rd) // 2] = ord("R")          # IUPAC byte: not in the LUT alphabet
            if k == 6 and q == 0:
                rd = bytearray(rd.lower())            # lower-case read: byte equality says mismatch everywhere
            rl.append((int(rng.integers(0, 6)), int(rng.integers(0, 3)), bytes(rd)))
        reads.append(rl)
    if sum(len(r) for r in reads) % 2 == 0:
        reads[-1].append((0, 0, haps[-1][0][20:170]))
    batch = _manual_batch(haps, reads, 6)
    assert batch.n_records % 2 == 1
    for mode, umi in (("coverage", 0), ("alt_frac", 1)):
        cfg = default_config(aligner=aligner, scoring_mode=mode, use_umi=umi, n_barcodes=6)
        assert_same(batch, cfg, threads=4)


def test_shared_prefix_kernel_equals_single_read_kernel():
    """The same batches through both DP kernels (VTX_DP_KERNEL=lut disables the shared-prefix kernel)."""
    import subprocess
    import sys
    code = r"""
import numpy as np
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
out = []
for seed, indel, jitter, rl, pad in ((1, 0.0, 0, 150, 100), (2, 0.5, 50, 120, 70), (3, 1.0, 30, 90, 10), (4, 0.3, 90, 100, 2)):
    spec = synth.SynthSpec(n_loci=300, n_barcodes=50, reads_per_locus=37, read_len=rl, padding=pad, indel_frac=indel,
                           read_len_jitter=jitter, seed=seed)
    batch = synth.make_batch(spec)
    with lib.Context(default_config(aligner="full", scoring_mode="coverage", n_barcodes=50)) as ctx:
        ctx.submit(batch); ctx.run()
        r, a = ctx.fetch_scores()
    out.append(np.concatenate([r, a]))
np.save(OUT, np.concatenate(out))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for kern in ("duo", "lut"):
        path = "/tmp/vtx_dp_%s_%d.npy" % (kern, os.getpid())
        env = dict(os.environ, VTX_LIB_VARIANT="dev", VTX_DP_KERNEL=kern, PYTHONPATH=root)
        p = subprocess.run([sys.executable, "-c", code.replace("OUT", repr(path))], env=env, cwd=root, capture_output=True,
                           text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        res[kern] = np.load(path)
        os.remove(path)
    assert res["duo"].size > 80000 and np.array_equal(res["duo"], res["lut"])


@pytest.mark.parametrize("aligner", ALIGNERS)
@pytest.mark.parametrize("padding,reads_per_locus", [(260, 64), (120, 64), (40, 5)])
def test_dp_kernel_modes_by_haplotype_length_and_depth(aligner, padding, reads_per_locus):
    """The shared-prefix DP kernel picks its lookup mode from the haplotype length and the depth: pair table (deep,
    default padding), two lookups (long haplotypes: the pair table would not hold the prefix; or shallow loci, up to
    8 tables per workgroup).  Same scores as the oracle in every mode."""
    spec = synth.SynthSpec(n_loci=60 if reads_per_locus > 8 else 400, n_barcodes=100, reads_per_locus=reads_per_locus,
                           read_len=150, padding=padding, indel_frac=0.3, read_len_jitter=20, seed=padding)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner=aligner, scoring_mode="coverage", n_barcodes=spec.n_barcodes)
    assert_same(batch, cfg, threads=os.cpu_count() or 8)


def test_banded_stage_in_many_chunks():
    """The band kernels process the tasks in chunks sized from the free HBM (one chunk for the benchmark); with the chunk
    forced down to 3000 tasks the banded stage runs dozens of chunks — scores and hard-task count must not change."""
    import subprocess
    import sys
    code = r"""
import numpy as np
from vartrix_amd import lib, synth
from vartrix_amd.abi import default_config
spec = synth.SynthSpec(n_loci=150, n_barcodes=60, reads_per_locus=120, read_len=130, padding=90, indel_frac=0.4, sub_error=0.04,
                       read_len_jitter=30, seed=77)
batch = synth.make_batch(spec)
with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=60)) as ctx:
    ctx.submit(batch); ctx.run()
    r, a = ctx.fetch_scores()
    hard = ctx.timing().hard_tasks
np.save(OUT, np.concatenate([r, a, [hard]]))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for name, env in (("one", {}), ("many", {"VTX_BAND_CHUNK": "3000"})):
        path = "/tmp/vtx_chunk_%s_%d.npy" % (name, os.getpid())
        p = subprocess.run([sys.executable, "-c", code.replace("OUT", repr(path))], env=dict(os.environ, VTX_LIB_VARIANT="dev", PYTHONPATH=root, **env),
                           cwd=root, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        res[name] = np.load(path)
        os.remove(path)
    assert res["one"][-1] > 100 and np.array_equal(res["one"], res["many"])


def test_nibble_read_arena_gives_the_same_result():
    """vtx_set_read_format(VTX_READS_NIBBLES): the read arena arrives two bases per byte (what the packer copies out of a BAM
    record, half the bytes over PCIe) and unpack_nibbles_kernel writes the arena every other kernel reads — scores and triplets
    must be those of the byte submit, through vtx_submit and through vtx_submit_raw, and the next byte submit must work again."""
    from vartrix_amd import abi
    spec = synth.SynthSpec(n_loci=300, n_barcodes=200, reads_per_locus=24, indel_frac=0.2, sub_error=0.02, read_len_jitter=0, seed=77)
    batch = synth.make_batch(spec)
    nib = batch.to_nibbles()
    assert nib.read_arena.size * 2 >= batch.read_arena.size
    with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=200)) as ctx:
        ctx.submit(batch); ctx.run()
        want_scores, want = ctx.fetch_scores(), ctx.fetch_coo()
        ctx.submit(nib); ctx.run()
        got_scores, got = ctx.fetch_scores(), ctx.fetch_coo()
        assert np.array_equal(want_scores[0], got_scores[0]) and np.array_equal(want_scores[1], got_scores[1])
        for k in want:
            assert np.array_equal(want[k], got[k]), k
        ctx.submit(batch); ctx.run()                                   # (the format is per submit: back to bytes)
        again = ctx.fetch_scores()
        assert np.array_equal(want_scores[0], again[0]) and np.array_equal(want_scores[1], again[1])
    # the raw path: tags as bytes, reads as nibbles
    raw, barcodes = synth.make_raw(batch, 200, use_umi=False)
    assert not (raw.records["read_off"] & 1).any()
    raw_nib = abi.RawBatch(raw.loci, raw.records, raw.hap_arena, abi.pack_nibbles(raw.read_arena), raw.tag_arena, abi.READS_NIBBLES)
    outs = []
    for rb in (raw, raw_nib):
        with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=200)) as ctx:
            ctx.set_barcodes(barcodes)
            ctx.submit_raw(rb); ctx.run()
            outs.append(ctx.fetch_coo())
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_nibble_arena_rejects_a_read_at_an_odd_base():
    """include/vtx.h: with VTX_READS_NIBBLES every read starts at an even base.  A record that does not is an error of the caller
    (it would be aligned shifted by one base): vtx_submit and vtx_submit_raw return VTX_E_INVAL and name it."""
    from vartrix_amd import abi
    spec = synth.SynthSpec(n_loci=20, n_barcodes=50, reads_per_locus=8, seed=3)
    batch = synth.make_batch(spec)
    nib = batch.to_nibbles()
    recs = nib.records.copy()
    recs["read_off"][5] += 1
    bad = abi.PackedBatch(nib.loci, recs, nib.hap_arena, nib.read_arena, abi.READS_NIBBLES)
    with lib.Context(default_config(aligner="banded", scoring_mode="coverage", n_barcodes=50)) as ctx:
        with pytest.raises(lib.VtxError) as e:
            ctx.submit(bad)
        assert e.value.status == abi.VTX_E_INVAL and "record 5" in str(e.value) and "odd" in str(e.value)
        ctx.submit(nib); ctx.run()                       # the context is still usable
        raw, barcodes = synth.make_raw(batch, 50, use_umi=False)
        rr = raw.records.copy()
        rr["read_off"][3] += 1
        ctx.set_barcodes(barcodes)
        with pytest.raises(lib.VtxError) as e:
            ctx.submit_raw(abi.RawBatch(raw.loci, rr, raw.hap_arena, abi.pack_nibbles(raw.read_arena), raw.tag_arena, abi.READS_NIBBLES))
        assert e.value.status == abi.VTX_E_INVAL and "odd" in str(e.value)
