"""End to end on the GPU through the drop-in command line: the reference's own regression tests
(src/main.rs:1208-1390) re-run against `vartrix_amd/bin/vartrix`, plus an authored indel BAM."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle, refpipe
from vartrix_amd import hostlib
from vartrix_amd.abi import default_config

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True)
def built():
    """The CLI and both libraries are build artefacts (git-ignored): build them when the tree does not have them."""
    from vartrix_amd import lib
    if not (os.path.exists(hostlib.CLI_PATH) and os.path.exists(hostlib.LIB_PATH) and os.path.exists(lib.LIB_PATH)):
        import __graft_entry__
        __graft_entry__.build()


def run_cli(args, cwd):
    r = subprocess.run([hostlib.CLI_PATH] + args, cwd=cwd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r


def csr(path):
    return refpipe.read_mtx(path)


def base_args(bc="barcodes.tsv"):
    return ["-v", os.path.join(G, "test.vcf"), "-b", os.path.join(G, "test.bam"), "-f", os.path.join(G, "test.fa"),
            "-c", os.path.join(G, bc)]


def test_consensus_matrix(tmp_path):          # src/main.rs:1208-1233
    out = str(tmp_path / "out.mtx")
    run_cli(base_args() + ["-o", out], tmp_path)
    assert csr(out) == csr(os.path.join(G, "test_consensus.mtx"))
    assert open(out).read() == open(os.path.join(G, "test_consensus.mtx")).read()   # byte-identical too


def test_frac_matrix(tmp_path):               # :1236-1263
    out = str(tmp_path / "out.mtx")
    run_cli(base_args() + ["-o", out, "-s", "alt_frac"], tmp_path)
    assert open(out).read() == open(os.path.join(G, "test_frac.mtx")).read()


def test_coverage_matrices(tmp_path):         # :1266-1300
    out, ref = str(tmp_path / "out.mtx"), str(tmp_path / "ref.mtx")
    run_cli(base_args() + ["-o", out, "-s", "coverage", "--ref-matrix", ref], tmp_path)
    assert csr(out) == csr(os.path.join(G, "test_coverage.mtx"))
    assert csr(ref) == csr(os.path.join(G, "test_coverage_ref.mtx"))


def test_coverage_matrices_umi(tmp_path):     # :1303-1339
    out, ref = str(tmp_path / "out.mtx"), str(tmp_path / "ref.mtx")
    run_cli(base_args() + ["-o", out, "-s", "coverage", "--ref-matrix", ref, "--umi"], tmp_path)
    assert csr(out) == csr(os.path.join(G, "test_coverage_umi.mtx"))
    assert csr(ref) == csr(os.path.join(G, "test_coverage_ref_umi.mtx"))


def test_coverage_matrices_umi_gzipped_bcs(tmp_path):   # :1342-1390
    out, ref, obc, ovar = (str(tmp_path / n) for n in ("out.mtx", "ref.mtx", "bcs.tsv", "vars.txt"))
    run_cli(base_args("barcodes.tsv.gz") + ["-o", out, "-s", "coverage", "--ref-matrix", ref, "--umi",
                                              "--out-barcodes", obc, "--out-variants", ovar], tmp_path)
    assert csr(out) == csr(os.path.join(G, "test_coverage_umi.mtx"))
    assert csr(ref) == csr(os.path.join(G, "test_coverage_ref_umi.mtx"))
    assert open(obc).read() == open(os.path.join(G, "barcodes.tsv")).read()          # barcode round trip :1387-1389
    assert open(ovar).read() == "1_199\n17_199\n2_199\n7_199\n"                        # 0-based pos (:1174)


def test_coverage_mode_without_ref_matrix_flag_writes_the_default_ref_matrix(tmp_path):
    """src/main.rs:385 tests `args.is_present("ref_matrix")`, and clap 2.33 reports an argument that has a
    default_value (:100) as present: coverage mode ALWAYS writes the REF-count matrix, to ./ref_matrix.mtx when the
    flag is absent (which is also why the default path goes through validate_output_path, :509-511)."""
    out = str(tmp_path / "out.mtx")
    run_cli(base_args() + ["-o", out, "-s", "coverage"], tmp_path)                    # cwd = tmp_path
    assert os.path.exists(out) and os.path.exists(tmp_path / "ref_matrix.mtx")
    assert csr(str(tmp_path / "ref_matrix.mtx")) == csr(os.path.join(G, "test_coverage_ref.mtx"))
    # consensus mode does not write it
    out2 = str(tmp_path / "out2.mtx")
    os.remove(tmp_path / "ref_matrix.mtx")
    run_cli(base_args() + ["-o", out2], tmp_path)
    assert os.path.exists(out2) and not os.path.exists(tmp_path / "ref_matrix.mtx")


# where the reads are prepared: "host" = the packer threads do everything (--ingest host --prep host); "device" = the packer reads the
# BAM, the device does barcode lookup / UMI grouping / sort (--ingest host --prep device); "ingest" = the device reads the BAM itself
# (--ingest device: BGZF inflate, record split, filters, tags — vtx_submit_bam; no fallback: a declined range is an error)
PATHS = {"host": ["--ingest", "host", "--prep", "host"], "device": ["--ingest", "host", "--prep", "device"], "ingest": ["--ingest", "device"]}


@pytest.mark.parametrize("prep", ["host", "device", "ingest"])
@pytest.mark.parametrize("aligner", ["banded", "full"])
@pytest.mark.parametrize("mode", ["consensus", "alt_frac", "coverage"])
@pytest.mark.parametrize("umi", [False, True])
def test_authored_indel_bam_end_to_end(tmp_path, mode, umi, aligner, prep):
    """test_dna.vcf (SNV + INS + DEL + multi-allelic) over an authored BAM: CLI output is byte-identical to
    the oracle pipeline (Python ingest restatement + C oracle, same aligner flavour) rendered as .mtx."""
    from tests.test_host import make_dna_bam
    bam = make_dna_bam(tmp_path, seed=3, n_reads=1500)
    vcfp, fap, bcp = (os.path.join(G, n) for n in ("test_dna.vcf", "test_dna.fa", "dna_barcodes.tsv"))
    out, ref = str(tmp_path / "out.mtx"), str(tmp_path / "ref.mtx")
    args = ["-v", vcfp, "-b", bam, "-f", fap, "-c", bcp, "-o", out, "-s", mode, "--ref-matrix", ref, "--threads", "4",
            "--aligner", aligner, "--log-level", "info"] + PATHS[prep]
    r = run_cli(args + (["--umi"] if umi else []), tmp_path)
    assert ("ingest on the device" in r.stderr) == (prep == "ingest")
    bcs = refpipe.load_barcodes(bcp)
    vcf = refpipe.read_vcf(vcfp)
    batch, wm = refpipe.pack(vcf, refpipe.read_fasta(fap), refpipe.read_bam(bam), bcs, refpipe.Args(use_umi=umi))
    # the counters of the log (:350-379) are the same whichever side did the barcode / UMI tests
    log = r.stdout + r.stderr
    # all nine Metrics counters (src/main.rs:449-459) as the info log prints them (:350-379)
    want_lines = [
        "Number of alignments evaluated: %d" % wm["num_reads"],
        "Number of alignments skipped due to low mapping quality: %d" % wm["num_low_mapq"],
        "Number of alignments skipped due to not being primary: %d" % wm["num_non_primary"],
        "Number of alignments skipped due to being duplicates: %d" % wm["num_duplicates"],
        "Number of alignments skipped due to not being associated with a cell barcode: %d" % wm["num_not_cell_bc"],
        "Number of alignments skipped due to not intersecting variant: %d" % wm["num_not_useful"],
        "Number of alignments skipped due to not having a UMI: %d" % wm["num_non_umi"],
        "Number of VCF records skipped due to having invalid characters in the alternative haplotype: %d" % wm["num_invalid_recs"],
        "Number of VCF records skipped due to being multi-allelic: %d" % wm["num_multiallelic_recs"],
    ]
    for ln in want_lines:
        assert ln + "\n" in log, ln
    assert wm["num_reads"] > wm["num_not_cell_bc"] > 0 and wm["num_multiallelic_recs"] == 1
    cfg = default_config(aligner=aligner, scoring_mode=mode, use_umi=int(umi), n_barcodes=len(bcs))
    r, a = oracle.batch_scores(batch, cfg, threads=8)
    coo = oracle.batch_reduce(batch, cfg, r, a)
    assert open(out).read() == refpipe.mtx_text(len(vcf), len(bcs), coo["row"], coo["col"], coo["value"])
    if mode == "coverage":
        assert open(ref).read() == refpipe.mtx_text(len(vcf), len(bcs), coo["row"], coo["col"], coo["ref_value"])
    assert len(coo["row"]) > 50


@pytest.mark.parametrize("umi", [False, True])
def test_reference_fixtures_with_device_prep(tmp_path, umi):
    """The reference's coverage fixtures (src/main.rs:1266-1339) with barcode lookup / UMI grouping / sort on the GPU."""
    out, ref = str(tmp_path / "out.mtx"), str(tmp_path / "ref.mtx")
    run_cli(base_args() + ["-o", out, "-s", "coverage", "--ref-matrix", ref, "--prep", "device", "--ingest", "host"] + (["--umi"] if umi else []), tmp_path)
    sfx = "_umi" if umi else ""
    assert csr(out) == csr(os.path.join(G, "test_coverage%s.mtx" % sfx))         # the reference compares CSR (:1296-1299)
    assert csr(ref) == csr(os.path.join(G, "test_coverage_ref%s.mtx" % sfx))
    out2 = str(tmp_path / "c.mtx")
    run_cli(base_args() + ["-o", out2, "--prep", "device", "--devices", "1", "--ingest", "host"], tmp_path)
    assert open(out2).read() == open(os.path.join(G, "test_consensus.mtx")).read()


@pytest.mark.parametrize("umi", [False, True])
def test_reference_fixtures_with_device_ingest(tmp_path, umi):
    """The reference's fixtures (src/main.rs:1208-1339) with the BAM read ON THE DEVICE (--ingest device: vtx_submit_bam — BGZF inflate,
    record split, filters, tags; the default with one GPU, forced here so that a silent fallback to the host packer would fail)."""
    out, ref = str(tmp_path / "out.mtx"), str(tmp_path / "ref.mtx")
    r = run_cli(base_args() + ["-o", out, "-s", "coverage", "--ref-matrix", ref, "--ingest", "device", "--log-level", "info"] + (["--umi"] if umi else []), tmp_path)
    assert "ingest on the device" in r.stderr and "packing on the host" not in r.stderr
    sfx = "_umi" if umi else ""
    assert csr(out) == csr(os.path.join(G, "test_coverage%s.mtx" % sfx))
    assert csr(ref) == csr(os.path.join(G, "test_coverage_ref%s.mtx" % sfx))
    if not umi:
        for mode, fixture in (("consensus", "test_consensus.mtx"), ("alt_frac", "test_frac.mtx")):
            o2 = str(tmp_path / (mode + ".mtx"))
            if os.path.exists(ref):
                os.remove(ref)
            run_cli(base_args() + ["-o", o2, "-s", mode, "--ref-matrix", ref, "--ingest", "device"], tmp_path)
            assert open(o2).read() == open(os.path.join(G, fixture)).read()


@pytest.mark.parametrize("prep", ["host", "device"])
def test_multi_batch_run_equals_single_batch(tmp_path, prep):
    """Inputs whose reads exceed the 32-bit arena limit are processed as several batches (forced here with a tiny
    limit): same bytes in the .mtx, same counters in the log."""
    from tests.test_host import make_dna_bam
    bam = make_dna_bam(tmp_path, seed=4, n_reads=1500)
    vcfp, fap, bcp = (os.path.join(G, n) for n in ("test_dna.vcf", "test_dna.fa", "dna_barcodes.tsv"))
    outs = {}
    for name, env in (("one", {}), ("many", {"VTXH_BATCH_BYTES": "12000"})):
        out = str(tmp_path / (name + ".mtx"))
        args = ["-v", vcfp, "-b", bam, "-f", fap, "-c", bcp, "-o", out, "-s", "alt_frac", "--umi", "--threads", "3", "--prep", prep, "--ingest", "host",
                "--log-level", "info", "--ref-matrix", str(tmp_path / (name + "_ref.mtx")), "--stream-loci", "0"]
        # (the tiny batch limit is a hook of the developer build: bin/vartrix_dev, libvtxhost_dev.so; the single-batch run is the product)
        r = subprocess.run([hostlib.cli_path("dev") if env else hostlib.CLI_PATH] + args, cwd=tmp_path, capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        log = r.stdout + r.stderr
        outs[name] = (open(out).read(), sorted(ln.split("] ", 1)[1] for ln in log.splitlines() if "Number of" in ln),
                      int(__import__("re").search(r"pack of range 0: [\d.]+ s \((\d+) batch", log).group(1)))
    assert outs["one"][2] == 1 and outs["many"][2] > 3
    assert outs["one"][0] == outs["many"][0] and outs["one"][1] == outs["many"][1]


def test_cli_row_gather_through_the_library(tmp_path):
    """--devices N > 1 merges the shards with vtx_gather_coo (RCCL) instead of host-side concatenation; `--gather library` takes
    that path with one device: same bytes as the plain run."""
    out1, out2 = str(tmp_path / "a.mtx"), str(tmp_path / "b.mtx")
    run_cli(base_args() + ["-o", out1, "-s", "alt_frac"], tmp_path)
    os.remove(tmp_path / "ref_matrix.mtx") if os.path.exists(tmp_path / "ref_matrix.mtx") else None
    r = subprocess.run([hostlib.CLI_PATH] + base_args() + ["-o", out2, "-s", "alt_frac", "--gather", "library"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(out1).read() == open(out2).read() == open(os.path.join(G, "test_frac.mtx")).read()


@pytest.mark.parametrize("prep", ["host", "device", "ingest"])
def test_streamed_ranges_give_the_same_files_and_counters(tmp_path, prep):
    """--stream-loci: the VCF taken in ranges of 7 records (7 ranges for test_dna.vcf, packed by a producer thread while the
    device works on the range before) writes byte-identical matrices and logs the same nine counters as the whole input at once
    (--stream-loci 0) — the reference has no such knob because it streams per locus anyway (src/main.rs:822-830)."""
    import re
    from tests.test_host import make_dna_bam
    bam = make_dna_bam(tmp_path, seed=5, n_reads=2500)
    vcfp, fap, bcp = (os.path.join(G, n) for n in ("test_dna.vcf", "test_dna.fa", "dna_barcodes.tsv"))
    outs = {}
    for name, sl in (("whole", "0"), ("ranges", "7"), ("single", "1")):
        out, ref, ov = (str(tmp_path / ("%s_%s" % (name, n))) for n in ("out.mtx", "ref.mtx", "vars.txt"))
        r = run_cli(["-v", vcfp, "-b", bam, "-f", fap, "-c", bcp, "-o", out, "-s", "coverage", "--ref-matrix", ref, "--umi",
                     "--threads", "4", "--log-level", "info", "--stream-loci", sl, "--out-variants", ov] + PATHS[prep], tmp_path)
        log = r.stdout + r.stderr
        counters = re.findall(r"Number of [^:]+: (\d+)", log)
        assert len(counters) == 9
        outs[name] = (open(out).read(), open(ref).read(), open(ov).read(), counters)
        n_ranges = len(re.findall(r"(?:pack|Plan) of range \d+", log))
        assert n_ranges == {"whole": 1, "ranges": 7, "single": 46}[name]
    assert outs["whole"] == outs["ranges"] == outs["single"]
    assert len(outs["whole"][0]) > 500


def test_ingest_auto_falls_back_and_device_insists(tmp_path):
    """--ingest auto (the default): an input the device path cannot take — here a BAM whose .bai carries no linear index (the record
    starts the device needs come from it) — is packed on the host, says so, and gives the same bytes; --ingest device refuses it.
    A BAM with a damaged BGZF block: the device's inflater declines the block, the host packer takes over and zlib rejects it too —
    the run fails with the packer's message whichever path was asked for."""
    import shutil
    from oracle import bamwriter
    from tests.test_host import make_dna_bam
    bam = make_dna_bam(tmp_path, seed=6, n_reads=1200)
    vcfp, fap, bcp = (os.path.join(G, n) for n in ("test_dna.vcf", "test_dna.fa", "dna_barcodes.tsv"))
    base = ["-v", vcfp, "-f", fap, "-c", bcp, "--log-level", "info", "-s", "coverage"]
    want = str(tmp_path / "want.mtx")
    r = run_cli(base + ["-b", bam, "-o", want, "--ref-matrix", str(tmp_path / "want_ref.mtx"), "--ingest", "device"], tmp_path)
    assert "ingest on the device" in r.stderr
    # (a) an index without a linear part
    noidx = str(tmp_path / "noidx.bam")
    shutil.copy(bam, noidx)
    open(noidx + ".bai", "wb").write(b"BAI\x01" + (0).to_bytes(4, "little"))
    out = str(tmp_path / "a.mtx")
    r = run_cli(base + ["-b", noidx, "-o", out, "--ref-matrix", str(tmp_path / "a_ref.mtx")], tmp_path)
    assert "packing on the host" in r.stderr and "ingest on the device" not in r.stderr
    assert open(out).read() == open(want).read() and open(tmp_path / "a_ref.mtx").read() == open(tmp_path / "want_ref.mtx").read()
    r = subprocess.run([hostlib.CLI_PATH] + base + ["-b", noidx, "-o", str(tmp_path / "b.mtx"), "--ref-matrix", str(tmp_path / "b_ref.mtx"), "--ingest", "device"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and ".bai" in r.stdout + r.stderr and not os.path.exists(tmp_path / "b.mtx")
    # (b) a damaged block in the middle of the file
    data = bytearray(open(bam, "rb").read())
    o, offs = 0, []
    while o + 18 <= len(data):
        offs.append(o)
        o += int.from_bytes(data[o + 16:o + 18], "little") + 1
    mid = offs[len(offs) // 2]
    for k in range(40, 60):
        data[mid + k] ^= 0x5a
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(data))
    shutil.copy(bam + ".bai", bad + ".bai")
    for flags in ([], ["--ingest", "host"]):
        r = subprocess.run([hostlib.CLI_PATH] + base + ["-b", bad, "-o", str(tmp_path / "c.mtx"), "--ref-matrix", str(tmp_path / "c_ref.mtx")] + flags,
                           cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "does not inflate" in r.stdout + r.stderr, r.stdout + r.stderr
        assert not os.path.exists(tmp_path / "c.mtx")
        if not flags:
            assert "the device declined" in r.stderr
