"""Host logic (C++ packer behind include/vtx_host.h) on CPU: same packed batch as the Python
restatement on the reference's fixtures, oracle on top reproduces the .mtx fixtures, MTX text,
CLI argument / output-path behaviour of the reference (src/main.rs:475-542)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle, refpipe
from vartrix_amd import hostlib
from vartrix_amd.abi import default_config

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def same_batch(a, b):
    """Equal as packed batches up to the placement of reads inside the arena."""
    if not (np.array_equal(a.loci, b.loci) and np.array_equal(a.hap_arena, b.hap_arena)):
        return False
    if a.n_records != b.n_records:
        return False
    for k in ("read_len", "cell_index", "umi_id"):
        if not np.array_equal(a.records[k], b.records[k]):
            return False
    for ra, rb in zip(a.records, b.records):
        sa = a.read_arena[int(ra["read_off"]):int(ra["read_off"]) + int(ra["read_len"])]
        sb = b.read_arena[int(rb["read_off"]):int(rb["read_off"]) + int(rb["read_len"])]
        if not np.array_equal(sa, sb):
            return False
    return True


@pytest.fixture(scope="module", autouse=True)
def built():
    # the developer build of the host library and the CLI: this module's hooks (VTXH_BATCH_BYTES, VTXH_CHUNK_BLOCKS, VTXH_NO_INDEX,
    # VTXH_ZLIB_INFLATE, VTXH_POOL_MIN) do not exist in the production library
    hostlib.use_variant("dev")
    if not (os.path.exists(hostlib.LIB_PATH) and os.path.exists(hostlib.CLI_PATH)):
        import __graft_entry__
        __graft_entry__.build()
    yield
    hostlib.use_variant("dev" if os.environ.get("VTX_LIB_VARIANT") == "dev" else "")


def test_production_host_library_has_no_test_hooks():
    """libvtxhost.so / bin/vartrix as shipped: VTXH_PROFILE is the only environment variable they know."""
    import re
    here = os.path.dirname(hostlib.LIB_PATH)
    for path in (os.path.join(here, "libvtxhost.so"), os.path.join(here, "bin", "vartrix")):
        names = set(re.findall(rb"VTXH?_[A-Z][A-Z0-9_]{3,}", open(path, "rb").read()))
        names = {n for n in names if not n.startswith((b"VTX_E_", b"VTX_OK"))}
        assert names <= {b"VTXH_PROFILE"}, (path, names)


def _inputs(bc="barcodes.tsv"):
    return dict(vcf=os.path.join(G, "test.vcf"), bam=os.path.join(G, "test.bam"), fasta=os.path.join(G, "test.fa"),
                cell_barcodes=os.path.join(G, bc))


@pytest.mark.parametrize("umi", [False, True])
@pytest.mark.parametrize("bc", ["barcodes.tsv", "barcodes.tsv.gz"])
def test_packer_equals_python_restatement(umi, bc):
    batch, metrics, nv, barcodes, variants = hostlib.pack_files(use_umi=umi, threads=3, **_inputs(bc))
    bcs = refpipe.load_barcodes(os.path.join(G, bc))
    want, wmetrics = refpipe.pack(refpipe.read_vcf(os.path.join(G, "test.vcf")), refpipe.read_fasta(os.path.join(G, "test.fa")),
                                  refpipe.read_bam(os.path.join(G, "test.bam")), bcs, refpipe.Args(use_umi=umi))
    assert metrics == wmetrics
    assert nv == 4 and barcodes == list(bcs.keys())
    assert variants == ["1_199", "17_199", "2_199", "7_199"]          # write_variants prints 0-based pos (:1174)
    assert same_batch(batch, want)


def test_packer_filters():
    base, m0, *_ = hostlib.pack_files(**_inputs())
    _, m1, *_ = hostlib.pack_files(mapq=255, **_inputs())
    assert m1["num_low_mapq"] > 0 and m1["num_reads"] == m0["num_reads"]
    _, m2, *_ = hostlib.pack_files(primary_only=True, **_inputs())
    assert m2["num_non_primary"] >= 0
    b3, m3, *_ = hostlib.pack_files(valid_chars="ATC", **_inputs())   # G not allowed -> every ALT haplotype invalid
    assert m3["num_invalid_recs"] == 4 and b3.n_loci == 0 and m3["num_reads"] == 0
    b4, m4, *_ = hostlib.pack_files(bam_tag="XX", **_inputs())
    assert b4.n_records == 0 and m4["num_not_cell_bc"] > 0
    # python restatement agrees on each variant
    for kw, args in (({"mapq": 255}, refpipe.Args(mapq=255)), ({"primary_only": True}, refpipe.Args(primary=True)),
                     ({"no_duplicates": True}, refpipe.Args(duplicates=True)), ({"padding": 30}, refpipe.Args(padding=30))):
        b, m, *_ = hostlib.pack_files(**kw, **_inputs())
        wb, wm = refpipe.pack(refpipe.read_vcf(os.path.join(G, "test.vcf")), refpipe.read_fasta(os.path.join(G, "test.fa")),
                              refpipe.read_bam(os.path.join(G, "test.bam")), refpipe.load_barcodes(os.path.join(G, "barcodes.tsv")), args)
        assert m == wm and same_batch(b, wb)


def test_validate_inputs_errors():
    """validate_inputs (:574-591): contig must be in FASTA and BAM; REF end <= chromosome length."""
    with pytest.raises(hostlib.HostError, match="larger than the chromosome length"):
        hostlib.pack_files(os.path.join(G, "test_dna.vcf"), os.path.join(G, "test.bam"), os.path.join(G, "test.fa"),
                           os.path.join(G, "dna_barcodes.tsv"))


def test_validate_inputs_unknown_contig(tmp_path):
    vcf = tmp_path / "x.vcf"
    vcf.write_text("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\nzz\t10\t.\tA\tC\t.\t.\t.\n")
    with pytest.raises(hostlib.HostError, match="Sequence zz not seen in FASTA"):
        hostlib.pack_files(str(vcf), os.path.join(G, "test.bam"), os.path.join(G, "test.fa"), os.path.join(G, "barcodes.tsv"))


def make_dna_bam(tmp_path, seed=1, n_reads=600, block=20000, index="linear"):
    """Author a coordinate-sorted BAM over test_dna.fa covering the loci of test_dna.vcf (SNV, INS,
    DEL, one multi-allelic record) with assorted CIGARs, flags, tags and soft clips."""
    from oracle import bamwriter
    rng = np.random.default_rng(seed)
    fa = refpipe.read_fasta(os.path.join(G, "test_dna.fa"))["1"].upper()
    vcf = refpipe.read_vcf(os.path.join(G, "test_dna.vcf"))
    bcs = list(refpipe.load_barcodes(os.path.join(G, "dna_barcodes.tsv")).keys())
    recs = []
    for k in range(n_reads):
        v = vcf[int(rng.integers(0, len(vcf)))]
        start = max(0, v.pos - int(rng.integers(0, 140)))
        carry_alt = len(v.alleles) >= 2 and rng.random() < 0.5
        ln = int(rng.integers(60, 151))
        if carry_alt and start <= v.pos:
            refa, alta = v.alleles[0], v.alleles[1]
            hapseq = fa[start:v.pos] + alta.upper() + fa[v.pos + len(refa):v.pos + len(refa) + 200]
            seq = hapseq[:ln].decode()
            left = v.pos - start
            d = len(alta) - len(refa)
            if d == 0 or left + 1 >= ln:
                cigar = "%dM" % len(seq)
            elif d > 0:
                cigar = "%dM%dI%dM" % (left + 1, min(d, ln - left - 1), max(ln - left - 1 - d, 0)) if ln - left - 1 - d > 0 else "%dM%dS" % (left + 1, ln - left - 1)
            else:
                cigar = "%dM%dD%dM" % (left + 1, -d, ln - left - 1)
        else:
            seq = fa[start:start + ln].decode()
            cigar = "%dM" % len(seq)
        flag = 0
        u = rng.random()
        if u < 0.05:
            flag |= 0x400
        elif u < 0.10:
            flag |= 0x100
        elif u < 0.13:
            flag |= 0x800
        if rng.random() < 0.1 and cigar.endswith("M") and cigar.count("M") == 1 and len(seq) > 20:
            clip = int(rng.integers(1, 15))
            cigar = "%dS%dM" % (clip, len(seq) - clip)
        tags = []
        if rng.random() < 0.3:      # fixed-size and array fields BEFORE the barcode: the scan must step over them
            tags.append(("XD", "d", 0.5)) if rng.random() < 0.5 else tags.append(("XB", "B", ("S", [1, 2, 3])))
            tags.append(("XF", "f", 1.5))
        if rng.random() < 0.9:
            tags.append(("CB", "Z", bcs[int(rng.integers(0, 40))] if rng.random() < 0.9 else b"NOTLISTED-1"))
        if rng.random() < 0.85:
            tags.append(("UB", "Z", "UMI%02d" % int(rng.integers(0, 12))))
        if rng.random() < 0.05:
            tags.append(("CB", "i", 7)) if not any(t[0] == "CB" for t in tags) else None
        tags.append(("NM", "i", 0))
        mapq = int(rng.choice([0, 3, 30, 60, 255]))
        recs.append((start, bamwriter.record(0, start, "r%04d" % k, seq, cigar, flag=flag, mapq=mapq, tags=tags)))
    recs.sort(key=lambda t: t[0])
    bam = str(tmp_path / "dna.bam")
    bamwriter.write_bam(bam, [("1", len(fa))], [r for _, r in recs], block=block, index=index)
    return bam


@pytest.mark.parametrize("window_blocks", [0, 1, 2])
@pytest.mark.parametrize("kw", [dict(), dict(use_umi=True), dict(mapq=30), dict(primary_only=True, no_duplicates=True),
                                dict(padding=20), dict(use_umi=True, padding=150)])
def test_packer_on_authored_indel_bam(tmp_path, kw, monkeypatch, window_blocks):
    """C++ packer == Python restatement on a BAM with indel reads over test_dna.vcf (46 records: 37 SNV,
    5 DEL, 3 INS, 1 multi-allelic -> skipped with its row kept, :646-653).  window_blocks: the sweep inflates that
    many BGZF blocks per window (VTXH_CHUNK_BLOCKS) — with 1 or 2 the file is many windows, records straddle them, and
    the parse of one window runs beside the indexing of the next (the waiting window moves to the second buffer)."""
    if window_blocks:
        monkeypatch.setenv("VTXH_CHUNK_BLOCKS", str(window_blocks))
    bam = make_dna_bam(tmp_path)
    vcfp, fap, bcp = (os.path.join(G, n) for n in ("test_dna.vcf", "test_dna.fa", "dna_barcodes.tsv"))
    batch, metrics, nv, barcodes, variants = hostlib.pack_files(vcfp, bam, fap, bcp, threads=2, **kw)
    args = refpipe.Args(mapq=kw.get("mapq", 0), primary=kw.get("primary_only", False), duplicates=kw.get("no_duplicates", False),
                        use_umi=kw.get("use_umi", False), padding=kw.get("padding", 100))
    want, wm = refpipe.pack(refpipe.read_vcf(vcfp), refpipe.read_fasta(fap), refpipe.read_bam(bam),
                            refpipe.load_barcodes(bcp), args)
    assert metrics == wm and metrics["num_multiallelic_recs"] == 1
    assert nv == 46 and len(barcodes) == 1331 and batch.n_loci == 45
    assert same_batch(batch, want)
    assert batch.n_records > 100


def test_mtx_writer_bytes(tmp_path):
    for v, s in [(1.0, "1"), (0.0, "0"), (0.5, "0.5"), (1 / 3, "0.3333333333333333"), (float("nan"), "NaN"),
                 (2 / 3, "0.6666666666666666"), (1 / 30000, "0.000033333333333333335"), (7.0, "7")]:
        assert hostlib.format_f64(v) == s == oracle.format_f64(v)
    p = str(tmp_path / "m.mtx")
    hostlib.write_mtx(p, 4, 20, [0, 1, 1, 2], [19, 14, 19, 17], [1.0, 1.0, 1.0, 2.0])
    assert open(p).read() == open(os.path.join(G, "test_consensus.mtx")).read()


def test_mtx_writer_many_threads(tmp_path):
    """Above 65 536 triplets the lines are formatted by several threads and every thread writes its buffer at its own
    file offset: same text as the sequential restatement (integers, fractions, NaN)."""
    rng = np.random.default_rng(3)
    n = 150_000
    row = np.sort(rng.integers(0, 5000, n)).astype(np.uint32)
    col = rng.integers(0, 700, n).astype(np.uint32)
    val = rng.integers(0, 4, n).astype(np.float64)
    val[::7] = rng.integers(1, 9, len(val[::7])) / rng.integers(1, 9, len(val[::7]))
    val[::1001] = np.nan
    p = str(tmp_path / "big.mtx")
    hostlib.write_mtx(p, 5000, 700, row, col, val)
    assert open(p).read() == refpipe.mtx_text(5000, 700, row, col, val)


def test_oracle_on_cpp_packed_batch_reproduces_fixtures():
    for mode, umi, fx, rfx in (("consensus", False, "test_consensus.mtx", None), ("alt_frac", False, "test_frac.mtx", None),
                               ("coverage", True, "test_coverage_umi.mtx", "test_coverage_ref_umi.mtx")):
        batch, _, nv, barcodes, _ = hostlib.pack_files(use_umi=umi, **_inputs())
        cfg = default_config(aligner="banded", scoring_mode=mode, use_umi=int(umi), n_barcodes=len(barcodes))
        ref, alt = oracle.batch_scores(batch, cfg)
        coo = oracle.batch_reduce(batch, cfg, ref, alt)
        _, want = refpipe.read_mtx(os.path.join(G, fx))
        assert {(int(r), int(c)): float(v) for r, c, v in zip(coo["row"], coo["col"], coo["value"])} == want
        if rfx:
            _, want = refpipe.read_mtx(os.path.join(G, rfx))
            assert {(int(r), int(c)): float(v) for r, c, v in zip(coo["row"], coo["col"], coo["ref_value"])} == want


def _cli(args, cwd):
    return subprocess.run([hostlib.CLI_PATH] + args, cwd=cwd, capture_output=True, text=True, timeout=120)


def test_cli_refuses_existing_output_and_missing_input(tmp_path):
    i = _inputs()
    base = ["-v", i["vcf"], "-b", i["bam"], "-f", i["fasta"], "-c", i["cell_barcodes"]]
    out = tmp_path / "o.mtx"
    out.write_text("x")
    r = _cli(base + ["-o", str(out)], tmp_path)                       # validate_output_path :477-480
    assert r.returncode == 1 and "Output path already exists" in r.stderr
    (tmp_path / "ref_matrix.mtx").write_text("x")                      # default ref matrix is checked too (:509-511)
    r = _cli(base + ["-o", str(tmp_path / "new.mtx")], tmp_path)
    assert r.returncode == 1 and "Output path already exists" in r.stderr
    os.remove(tmp_path / "ref_matrix.mtx")
    r = _cli(["-v", "/nonexistent.vcf", "-b", i["bam"], "-f", i["fasta"], "-c", i["cell_barcodes"]], tmp_path)
    assert r.returncode == 1 and "does not exist" in r.stderr
    r = _cli(base + ["-s", "bogus"], tmp_path)
    assert r.returncode == 1
    r = _cli(["-v", i["vcf"]], tmp_path)
    assert r.returncode == 1 and "required" in r.stderr


def _raw_equals_cooked(inputs, **kw):
    """Raw packer output, resolved / filtered / sorted by the oracle restatement (oracle/prep.py), must be the
    cooked packer's batch; the counters the raw packer leaves to the device are the difference of the metrics."""
    from oracle import prep
    cooked, m, nv, barcodes, variants = hostlib.pack_files(**kw, **inputs)
    raw, mr, nvr, barcodes_r, variants_r = hostlib.pack_files(raw=True, **kw, **inputs)
    assert (nv, barcodes, variants) == (nvr, barcodes_r, variants_r)
    got, st = prep.prep_raw(raw, barcodes, bool(kw.get("use_umi", False)))
    for k in m:
        if k == "num_not_cell_bc":
            assert mr[k] + st["num_not_cell_bc"] == m[k]
        elif k == "num_non_umi":
            assert mr[k] == 0 and st["num_non_umi"] == m[k]
        else:
            assert mr[k] == m[k], k
    assert same_batch(got, cooked)
    return raw, cooked


@pytest.mark.parametrize("umi", [False, True])
def test_raw_packer_equals_cooked_packer_on_reference_bam(umi):
    raw, cooked = _raw_equals_cooked(_inputs(), use_umi=umi, threads=2)
    assert raw.n_records >= cooked.n_records and raw.tag_arena.size > 0


@pytest.mark.parametrize("umi", [False, True])
def test_raw_packer_equals_cooked_packer_on_authored_bam(tmp_path, umi):
    bam = make_dna_bam(tmp_path, seed=5, n_reads=900)
    inputs = dict(vcf=os.path.join(G, "test_dna.vcf"), bam=bam, fasta=os.path.join(G, "test_dna.fa"),
                  cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    for kw in ({}, {"mapq": 30, "no_duplicates": True}, {"primary_only": True, "padding": 40}):
        raw, cooked = _raw_equals_cooked(inputs, use_umi=umi, **kw)
    assert raw.n_records > cooked.n_records          # unlisted barcodes / missing UB are still in the raw batch


def _concat_batches(batches):
    """Loci keep their global row; records / reads of the batches are compared through the canonical form."""
    rows, recs = [], []
    for b in batches:
        for loc in b.loci:
            rows.append(int(loc["row"]))
            r = b.records[int(loc["rec_begin"]):int(loc["rec_begin"]) + int(loc["rec_count"])]
            recs.append([(int(x["cell_index"]) if "cell_index" in r.dtype.names else bytes(b.tag_arena[int(x["bc_off"]):int(x["bc_off"]) + int(x["bc_len"])]),
                          int(x["umi_id"]) if "umi_id" in r.dtype.names else (None if int(x["umi_len"]) == 0xFFFF else bytes(b.tag_arena[int(x["umi_off"]):int(x["umi_off"]) + int(x["umi_len"])])),
                          bytes(b.read_arena[int(x["read_off"]):int(x["read_off"]) + int(x["read_len"])])) for x in r])
    return rows, recs


@pytest.mark.parametrize("raw", [False, True])
@pytest.mark.parametrize("umi", [False, True])
def test_packer_splits_large_inputs_into_batches(tmp_path, monkeypatch, raw, umi):
    """Reads spanning more than the 32-bit arena limit come as several batches (windows of the same arenas with
    rebased offsets).  With the limit forced down to a few KB the concatenation equals the single-batch pack."""
    bam = make_dna_bam(tmp_path, seed=9, n_reads=1200)
    inputs = dict(vcf=os.path.join(G, "test_dna.vcf"), bam=bam, fasta=os.path.join(G, "test_dna.fa"),
                  cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    one, m1, *_ = hostlib.pack_files(use_umi=umi, raw=raw, threads=2, all_batches=True, **inputs)
    assert len(one) == 1
    monkeypatch.setenv("VTXH_BATCH_BYTES", "9000")
    many, m2, *_ = hostlib.pack_files(use_umi=umi, raw=raw, threads=2, all_batches=True, **inputs)
    assert len(many) > 3 and m1 == m2
    assert _concat_batches(many) == _concat_batches(one)
    for b in many:                                   # every batch is self-consistent under the 32-bit layout
        assert b.read_arena.size <= 9000 + 200
        ends = b.loci["rec_begin"].astype(np.int64) + b.loci["rec_count"]
        assert ends.max(initial=0) == b.n_records
        assert np.all(b.records["read_off"].astype(np.int64) + b.records["read_len"] <= b.read_arena.size)
    with pytest.raises(hostlib.HostError, match="batches"):
        hostlib.pack_files(use_umi=umi, raw=raw, **inputs)          # the single-batch call refuses a multi-batch pack
    monkeypatch.setenv("VTXH_BATCH_BYTES", "100")
    with pytest.raises(hostlib.HostError, match="alone needs more"):
        hostlib.pack_files(use_umi=umi, raw=raw, all_batches=True, **inputs)


@pytest.mark.parametrize("raw", [False, True])
def test_packs_of_row_ranges_add_up_to_the_whole_pack(tmp_path, raw):
    """Streaming (vtxh_pack_files_range): the packs of consecutive ranges of VCF records hold, together, exactly the loci,
    records and reads of the whole pack, in order, and their metrics sum to its metrics — so a host can keep ONE range in
    memory at a time (the reference holds one locus' reads at a time, src/main.rs:822-830).  Every range pack holds only
    the reads of ITS loci: its size follows the range, not the BAM."""
    bam = make_dna_bam(tmp_path, seed=11, n_reads=1500)
    inputs = dict(vcf=os.path.join(G, "test_dna.vcf"), bam=bam, fasta=os.path.join(G, "test_dna.fa"),
                  cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    whole, m, nv, bcs, names = hostlib.pack_files(use_umi=True, raw=raw, threads=2, all_batches=True, **inputs)
    assert nv == 46 and len(whole) == 1
    for cuts in ([0, 46], [0, 10, 20, 46], [0, 1, 2, 45, 46, 46], list(range(0, 47, 5)) + [46]):
        parts, msum = [], None
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            b, mm, nv2, bcs2, names2 = hostlib.pack_files(use_umi=True, raw=raw, threads=2, all_batches=True, rows=(lo, hi), **inputs)
            assert nv2 == nv and bcs2 == bcs and names2 == names          # the matrix keeps its shape in every range
            assert all(lo <= r < hi for bb in b for r in bb.loci["row"])
            parts += b
            msum = dict(mm) if msum is None else {k: msum[k] + mm[k] for k in mm}
        assert msum == m
        assert _concat_batches(parts) == _concat_batches(whole)
    # a range holds its own reads only
    b, *_ = hostlib.pack_files(use_umi=True, raw=raw, threads=2, all_batches=True, rows=(0, 5), **inputs)
    assert sum(bb.n_records for bb in b) < 0.5 * whole[0].n_records


@pytest.mark.parametrize("index", ["linear", "csi"])
def test_index_guided_skipping_on_a_sparse_vcf(tmp_path, monkeypatch, index):
    """A few loci over a BAM that covers the whole contig: with the .bai — or, round 6, a .csi (src/main.rs:520-529 accepts either:
    the window table is rebuilt from the leaf bins' loffsets) — the packer inflates only the stretches that can
    hold their reads (the reference does an indexed fetch per locus, src/main.rs:822-826) — and packs exactly what the
    full sweep packs.  Reads with long reference skips (spliced, N) start far before the locus they overlap: the linear
    index accounts for them."""
    from oracle import bamwriter
    rng = np.random.default_rng(12)
    fa = refpipe.read_fasta(os.path.join(G, "test_dna.fa"))["1"].upper()
    bcs = list(refpipe.load_barcodes(os.path.join(G, "dna_barcodes.tsv")).keys())
    def clean(p):                                                       # first position >= p whose +-100 window is pure ACGT
        while any(c not in b"ACGT" for c in fa[p - 100:p + 101]):
            p += 50
        return p
    loci_pos = [clean(p) for p in (5000, 5300, 60000, 150000, 150090, 230000)]   # two pairs close together, the rest far apart
    vcf = tmp_path / "sparse.vcf"
    with open(vcf, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##contig=<ID=1,length=%d>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" % len(fa))
        for p in loci_pos:
            ref = chr(fa[p])
            fh.write("1\t%d\t.\t%s\t%s\t.\t.\t.\n" % (p + 1, ref, "ACGT"[("ACGT".index(ref) + 1) % 4]))
    recs = []
    for k in range(24000):                                              # reads all over the contig
        start = int(rng.integers(0, len(fa) - 200))
        ln = int(rng.integers(60, 151))
        recs.append((start, bamwriter.record(0, start, "r%05d" % k, fa[start:start + ln].decode(), "%dM" % ln, mapq=60,
                                             tags=[("CB", "Z", bcs[int(rng.integers(0, 40))]), ("UB", "Z", "U%03d" % int(rng.integers(0, 99)))])))
    for p in loci_pos:                                                  # reads at the loci, some spliced across 30 kb
        for k in range(30):
            start = p - int(rng.integers(0, 100))
            recs.append((start, bamwriter.record(0, start, "l%d_%d" % (p, k), fa[start:start + 120].decode(), "120M", mapq=60,
                                                 tags=[("CB", "Z", bcs[int(rng.integers(0, 40))]), ("UB", "Z", "U%03d" % k)])))
        if p > 40000:
            s0 = p - 30050
            seq = fa[s0:s0 + 40] + fa[p - 10:p + 70]
            recs.append((s0, bamwriter.record(0, s0, "sp%d" % p, seq.decode(), "40M%dN80M" % (p - 10 - (s0 + 40)), mapq=60,
                                              tags=[("CB", "Z", bcs[3]), ("UB", "Z", "USP")])))
    recs.sort(key=lambda t: t[0])
    bam = str(tmp_path / "wide.bam")
    bamwriter.write_bam(bam, [("1", len(fa))], [r for _, r in recs], block=4000, index=index)
    assert os.path.exists(bam + (".csi" if index == "csi" else ".bai")) and not os.path.exists(bam + (".bai" if index == "csi" else ".csi"))
    fap, bcp = os.path.join(G, "test_dna.fa"), os.path.join(G, "dna_barcodes.tsv")
    got, gm, nv, _, _ = hostlib.pack_files(str(vcf), bam, fap, bcp, threads=3, use_umi=True)
    st = dict(hostlib.last_ingest_stats)
    with hostlib.plan_ingest(str(vcf), bam, fap, bcp, use_umi=True) as plan:      # (the device's plan takes its record starts from the same table)
        assert plan.reason is None or "sparse" in plan.reason
    monkeypatch.setenv("VTXH_NO_INDEX", "1")
    full, fm, _, _, _ = hostlib.pack_files(str(vcf), bam, fap, bcp, threads=3, use_umi=True)
    st_full = dict(hostlib.last_ingest_stats)
    monkeypatch.delenv("VTXH_NO_INDEX")
    want, wm = refpipe.pack(refpipe.read_vcf(str(vcf)), refpipe.read_fasta(fap), refpipe.read_bam(bam),
                            refpipe.load_barcodes(bcp), refpipe.Args(use_umi=True))
    assert gm == fm == wm
    assert same_batch(got, want) and same_batch(full, want)
    assert got.n_records > 150 and any(int(c) > 30 for c in got.loci["rec_count"])     # the spliced reads were found
    assert st_full["blocks_inflated"] == st_full["blocks_total"] and st_full["index_jumps"] == 0
    # (each far locus drags in the 30 kb its spliced read spans, and a restart over-reads up to ~100 of these tiny 4 kB
    # blocks: a third of the file stays untouched here; with 64 kB blocks and a 50 GB BAM it is nearly all of it)
    assert st["index_jumps"] >= 2 and st["blocks_inflated"] < 0.7 * st["blocks_total"], st


@pytest.mark.parametrize("fixture", ["test", "test_dna"])
def test_bcf_input_equals_vcf(tmp_path, fixture):
    """bcf::Reader::from_path (src/main.rs:220) reads BCF as well as text VCF; so does the packer since round 6 (by content: "BCF\\2\\2"
    behind the BGZF layer).  The reference's two VCFs converted by the test's own BCF writer (oracle/bamwriter.py: vcf_to_bcf — typed
    strings of every length class, a multi-allelic record, IDs): the same pack, the same names, the same skipped-record counters."""
    from oracle import bamwriter
    vcfp = os.path.join(G, fixture + ".vcf")
    bcfp = str(tmp_path / (fixture + ".bcf"))
    bamwriter.vcf_to_bcf(vcfp, bcfp)
    if fixture == "test":
        rest = dict(bam=os.path.join(G, "test.bam"), fasta=os.path.join(G, "test.fa"), cell_barcodes=os.path.join(G, "barcodes.tsv"))
    else:
        rest = dict(bam=make_dna_bam(tmp_path, seed=2, n_reads=800), fasta=os.path.join(G, "test_dna.fa"), cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    a = hostlib.pack_files(vcf=vcfp, threads=2, use_umi=True, **rest)
    b = hostlib.pack_files(vcf=bcfp, threads=2, use_umi=True, **rest)
    assert same_batch(a[0], b[0]) and a[1:] == b[1:]
    assert a[2] == (4 if fixture == "test" else 46) and (fixture == "test" or a[1]["num_multiallelic_recs"] == 1)
    # a long allele (> 127 bytes: the 16-bit length class of a typed string) and a truncated file
    long_vcf = str(tmp_path / "long.vcf")
    fa = refpipe.read_fasta(os.path.join(G, "test_dna.fa"))["1"].upper()

    def clean(p):                                                       # first position >= p whose +-130 window is pure ACGT
        while any(c not in b"ACGT" for c in fa[p - 130:p + 131]):
            p += 50
        return p
    p1 = clean(60000)
    p2 = clean(p1 + 2000)
    with open(long_vcf, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##contig=<ID=1,length=%d>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" % len(fa))
        fh.write("1\t%d\trs1\t%s\t%s\t.\t.\t.\n" % (p1 + 1, fa[p1:p1 + 1].decode(), fa[p1:p1 + 1].decode() + "ACGT" * 40))
        fh.write("1\t%d\t.\t%s\t%s\t.\t.\t.\n" % (p2 + 1, fa[p2:p2 + 20].decode(), fa[p2:p2 + 1].decode()))
    long_bcf = str(tmp_path / "long.bcf")
    bamwriter.vcf_to_bcf(long_vcf, long_bcf)
    dna = dict(bam=make_dna_bam(tmp_path, seed=2, n_reads=300), fasta=os.path.join(G, "test_dna.fa"), cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    a = hostlib.pack_files(vcf=long_vcf, **dna)
    b = hostlib.pack_files(vcf=long_bcf, **dna)
    assert same_batch(a[0], b[0]) and a[1:] == b[1:] and a[0].loci["alt_len"][0] == 100 + 161 + 100
    import gzip
    raw = gzip.open(long_bcf, "rb").read()
    cut = str(tmp_path / "cut.bcf")
    with gzip.open(cut, "wb") as fh:
        fh.write(raw[:-7])
    with pytest.raises(hostlib.HostError) as ei:
        hostlib.pack_files(vcf=cut, **dna)
    assert "BCF" in str(ei.value)


def test_packer_survives_corrupted_bam(tmp_path):
    """Truncated payloads, flipped bytes and random 32-bit words inside an authored BAM: the packer either packs
    what is still a consistent file or returns an error — it never reads outside its buffers (the cases run in one child
    process, so a crash shows up as its exit code)."""
    import struct
    import subprocess
    import sys
    import zlib
    from oracle import bamwriter
    bam = make_dna_bam(tmp_path)
    raw = open(bam, "rb").read()
    payload, o = b"", 0
    while o + 18 <= len(raw):
        bs = int.from_bytes(raw[o + 16:o + 18], "little") + 1
        payload += zlib.decompress(raw[o + 18:o + bs - 8], -15)
        o += bs
    rng = np.random.default_rng(5)
    cases = [payload[:int(c)] for c in rng.integers(50, len(payload), 12)]
    for _ in range(12):
        b = bytearray(payload)
        b[int(rng.integers(300, len(payload)))] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    for _ in range(8):
        b = bytearray(payload)
        pos = int(rng.integers(300, len(payload) - 4))
        b[pos:pos + 4] = struct.pack("<I", int(rng.integers(0, 2 ** 32)))
        cases.append(bytes(b))
    paths = []
    for k, pay in enumerate(cases):
        p = str(tmp_path / ("c%02d.bam" % k))
        with open(p, "wb") as fh:
            for q in range(0, len(pay), 20000):
                fh.write(bamwriter._bgzf_block(pay[q:q + 20000]))
            fh.write(bamwriter._bgzf_block(b""))
        if k % 2:                                       # half of them with the (now stale) index of the intact file
            with open(p + ".bai", "wb") as fh:
                fh.write(open(bam + ".bai", "rb").read())
        paths.append(p)
    code = '''
import sys
sys.path.insert(0, %r)
from vartrix_amd import hostlib
ok = err = 0
for p in sys.argv[4:]:
    try:
        hostlib.pack_files(sys.argv[1], p, sys.argv[2], sys.argv[3], threads=2)
        ok += 1
    except Exception:
        err += 1
print("packed", ok, "refused", err)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    vcfp, fap, bcp = (os.path.join(G, n) for n in ("test_dna.vcf", "test_dna.fa", "dna_barcodes.tsv"))
    r = subprocess.run([sys.executable, "-c", code, vcfp, fap, bcp] + paths, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-400:])
    assert "packed" in r.stdout and int(r.stdout.split()[3]) > 0, r.stdout     # the truncations inside a record are refused


# ---- the packer's own DEFLATE decoder (vtx_inflate.h) against zlib ----
def _own_inflate(raw, n):
    import ctypes as C
    L = hostlib.load()
    out = (C.c_uint8 * (n + 64))()
    C.memset(C.addressof(out) + n, 0xAB, 64)
    rc = L.vtxh_test_inflate(raw, len(raw), C.addressof(out), n)
    assert bytes(out[n:n + 64]) == b"\xab" * 64, "the decoder wrote beyond its output"
    return rc, bytes(out[:n])


def test_own_inflate_equals_zlib_on_every_block_kind():
    """Stored, fixed and dynamic blocks, literal-only and match-heavy data, every size from empty to a full BGZF block: accepted
    and byte-identical; a wrong output size or a truncated stream is declined."""
    import random
    import zlib
    rng = random.Random(1)
    accepted = 0
    for trial in range(60):
        n = rng.choice([0, 1, 5, 100, 1000, 20000, 65280])
        kind = trial % 5
        if kind == 0:
            data = bytes(rng.getrandbits(8) for _ in range(n))
        elif kind == 1:
            data = bytes(rng.choice(b"ACGT") for _ in range(n))
        elif kind == 2:
            data = (b"ACGTTGCA" * (n // 8 + 1))[:n]
        elif kind == 3:
            data = bytes(rng.choice(b"AB") for _ in range(n))
        else:
            data = bytes([rng.randrange(4)]) * n
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strat)
                raw = co.compress(data) + co.flush()
                rc, out = _own_inflate(raw, len(data))
                assert rc == 1 and out == data, (trial, level, strat, n)
                accepted += 1
                if data:
                    assert _own_inflate(raw, len(data) - 1)[0] == 0
                assert _own_inflate(raw, len(data) + 1)[0] == 0
                if len(raw) > 2 and data:
                    assert _own_inflate(raw[:len(raw) // 2], len(data))[0] == 0
    assert accepted == 60 * 16


def test_own_inflate_never_accepts_what_zlib_rejects():
    """Bit flips in valid streams: the decoder either declines (the packer then asks zlib) or returns exactly what zlib returns."""
    import random
    import zlib
    rng = random.Random(7)
    accepted = declined = 0
    for trial in range(1500):
        n = rng.choice([50, 500, 5000, 30000])
        data = bytes(rng.choice(b"ACGTN") for _ in range(n)) if trial % 2 else bytes(rng.getrandbits(8) & 0x3f for _ in range(n))
        co = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, -15)
        raw = bytearray(co.compress(data) + co.flush())
        for _ in range(rng.randint(1, 4)):
            raw[rng.randrange(len(raw))] ^= 1 << rng.randrange(8)
        raw = bytes(raw)
        rc, out = _own_inflate(raw, n)
        try:
            d = zlib.decompressobj(-15)
            z = d.decompress(raw) + d.flush()
            zok = d.eof and len(z) == n
        except zlib.error:
            zok, z = False, None
        if rc:
            accepted += 1
            assert zok and out == z, trial
        else:
            declined += 1
    assert accepted > 100 and declined > 100


def test_own_inflate_on_the_reference_bam_and_pack_equality(monkeypatch):
    """Every BGZF block of the reference's test BAM through the decoder, and the whole pack with the decoder against the pack with
    zlib only (VTXH_ZLIB_INFLATE=1)."""
    import struct
    import zlib
    f = open(os.path.join(G, "test.bam"), "rb").read()
    o = blocks = 0
    while o + 18 <= len(f):
        xlen = struct.unpack_from("<H", f, o + 10)[0]
        bsize = struct.unpack_from("<H", f, o + 16)[0] + 1
        raw = f[o + 12 + xlen:o + bsize - 8]
        isize = struct.unpack_from("<I", f, o + bsize - 4)[0]
        if isize:
            rc, out = _own_inflate(raw, isize)
            assert rc == 1 and out == zlib.decompress(raw, -15) and zlib.crc32(out) == struct.unpack_from("<I", f, o + bsize - 8)[0]
            blocks += 1
        o += bsize
    assert blocks >= 1
    a = hostlib.pack_files(**_inputs())
    monkeypatch.setenv("VTXH_ZLIB_INFLATE", "1")
    b = hostlib.pack_files(**_inputs())
    assert same_batch(a[0], b[0]) and a[1:] == b[1:]


# ---- read arenas as the BAM holds bases: two per byte (vtxh_args.read_format = VTX_READS_NIBBLES) ----
def _unpacked(b):
    """A nibble batch with its arena expanded the way unpack_nibbles_kernel does on the device."""
    from vartrix_amd import abi
    if getattr(b, "read_format", 0) != abi.READS_NIBBLES:
        return b
    lut = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)
    out = np.empty(2 * b.read_arena.size, np.uint8)
    out[0::2] = lut[b.read_arena >> 4]
    out[1::2] = lut[b.read_arena & 15]
    import copy
    c = copy.copy(b)
    c.read_arena, c.read_format = out, abi.READS_BYTES
    return c


@pytest.mark.parametrize("raw", [False, True])
@pytest.mark.parametrize("umi", [False, True])
def test_nibble_pack_holds_the_same_reads(tmp_path, monkeypatch, raw, umi):
    """The packer can leave the bases two per byte (half the arena to write and to ship; the device unpacks).  Offsets and
    lengths still count bases; every read starts at an even one.  Same records, same reads, same metrics as the byte pack —
    over one window and many, one batch and many."""
    from vartrix_amd import abi
    bam = make_dna_bam(tmp_path, seed=11, n_reads=1500)
    inputs = dict(vcf=os.path.join(G, "test_dna.vcf"), bam=bam, fasta=os.path.join(G, "test_dna.fa"),
                  cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    for env in ({}, {"VTXH_CHUNK_BLOCKS": "1"}, {"VTXH_BATCH_BYTES": "9000"}):
        for k in ("VTXH_CHUNK_BLOCKS", "VTXH_BATCH_BYTES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        want, m1, *_ = hostlib.pack_files(use_umi=umi, raw=raw, threads=3, all_batches=True, **inputs)
        got, m2, *_ = hostlib.pack_files(use_umi=umi, raw=raw, threads=3, all_batches=True, nibbles=True, **inputs)
        assert m1 == m2 and len(got) == len(want) and (len(got) > 3) == ("VTXH_BATCH_BYTES" in env)
        for b in got:
            assert b.read_format == abi.READS_NIBBLES
            assert not (b.records["read_off"] & 1).any()
            assert b.as_struct().read_bytes == 2 * b.read_arena.size
            assert np.all(b.records["read_off"].astype(np.int64) + b.records["read_len"] <= 2 * b.read_arena.size)
        assert _concat_batches([_unpacked(b) for b in got]) == _concat_batches(want)
        assert sum(b.read_arena.size for b in got) <= sum(b.read_arena.size for b in want) // 2 + 2 * sum(b.n_records for b in want)


def test_nibble_helpers_round_trip():
    from vartrix_amd import abi, synth
    b = synth.make_batch(synth.SynthSpec(n_loci=30, n_barcodes=10, reads_per_locus=6, seed=5))
    n = b.to_nibbles()
    assert n.read_format == abi.READS_NIBBLES and n.read_arena.size == (b.read_arena.size + 1) // 2
    r = n.to_bytes()
    assert np.array_equal(r.read_arena[:b.read_arena.size], b.read_arena)
    s = n.slice_loci(7, 19)
    t = b.slice_loci(7, 19)
    assert np.array_equal(s.records, t.records) and np.array_equal(s.to_bytes().read_arena[:t.read_arena.size], t.read_arena)


def test_own_inflate_on_long_huffman_codes():
    """Geometrically distributed symbols: zlib's length-limited codes reach 15 bits, beyond the decoder's 11-bit primary table —
    the subtable path."""
    import random
    import zlib
    rng = random.Random(3)
    for trial in range(6):
        syms = list(range(256))
        rng.shuffle(syms)
        data = bytearray()
        while len(data) < 60000:
            k = 0
            while k < 40 and rng.random() < 0.62:
                k += 1
            data.append(syms[k * 6 % 256 if k < 40 else rng.randrange(256)])
        data = bytes(data)
        for strat in (zlib.Z_HUFFMAN_ONLY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED):
            co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, strat)
            raw = co.compress(data) + co.flush()
            rc, out = _own_inflate(raw, len(data))
            assert rc == 1 and out == data, (trial, strat)


def test_packs_reuse_each_others_memory():
    """The packer keeps released buffers of a megabyte and more and hands them out again (touched pages are what its time is made
    of).  A buffer then arrives with another pack's bytes in it: with the threshold at 64 bytes every buffer of these small packs
    is a reused one, and every result must still be what it is with fresh memory."""
    import subprocess
    import sys
    env = dict(os.environ, VTXH_POOL_MIN="64")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "packer_equals or raw_packer or nibble_pack or splits_large or row_ranges or authored_indel"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_trim_returns_the_kept_buffers():
    """vtxh_trim(): the buffers a freed pack leaves in the process-wide pool (touched pages for the next pack of a streamed run) go
    back to the allocator; packing afterwards still works and gives the same batch."""
    import ctypes as C
    L = hostlib.load()
    L.vtxh_trim.restype = None
    a, m1, *_ = hostlib.pack_files(**_inputs())
    L.vtxh_trim()
    L.vtxh_trim()                                   # (idempotent)
    b, m2, *_ = hostlib.pack_files(**_inputs())
    assert m1 == m2 and same_batch(a, b)
