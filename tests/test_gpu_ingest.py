"""Device-side BAM ingest (vtx_submit_bam, vartrix_amd/csrc/vtx_ingest.hip) against the host packer and zlib.

What it replaces in the reference: `bam.fetch(..)` + `bam.records()` per locus with rust-htslib / htslib / zlib below them
(src/main.rs:822-830), the read filters and their Metrics counters (:831-864), useful_alignment (:790-806), the tag lookups
(:737-757) and rec.seq() (:896).  The checker is the host packer (libvtxhost: `vtxh_pack_files_raw`, itself pinned against the
Python restatement oracle/refpipe.py in tests/test_host.py) and, for the inflater, zlib:

  * bgzf_inflate_kernel on stored / fixed / dynamic / multi-block streams, every size, truncated and corrupted streams, the
    reference's own BAM: what it accepts is byte-identical to zlib's output, it never accepts what zlib rejects;
  * the raw records, their loci, the tag arena and the read arena the device builds are THE SAME BYTES the host packer builds
    (after a stable sort by locus — the device keeps BAM order, the host groups by locus; the preparation's sort key does not care);
  * the Metrics counters, the resolved records and the triplets after vtx_run equal the host-packed path's.
"""
import os
import random
import struct
import sys
import zlib

import numpy as np
import pytest

from vartrix_amd import abi, hostlib, lib
from vartrix_amd.abi import default_config

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
sys.path.insert(0, HERE)


def raw_deflate(data, level=6, strat=zlib.Z_DEFAULT_STRATEGY, split=False):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strat)
    if split and len(data) > 10:
        return co.compress(data[:len(data) // 3]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(data[len(data) // 3:]) + co.flush()
    return co.compress(data) + co.flush()


@pytest.fixture(scope="module")
def ctx():
    with lib.Context(default_config(n_barcodes=4)) as c:
        yield c


def run_blocks(ctx, streams):
    """streams: [(raw bytes, expected size)] -> (status, outputs) through ONE launch of bgzf_inflate_kernel."""
    blob = b"".join(r for r, _ in streams)
    blocks, o = [], 0
    for r, n in streams:
        blocks.append((o, len(r), n))
        o += len(r)
    return ctx.debug_inflate(blob, blocks)


def test_inflate_kernel_equals_zlib_on_every_block_kind(ctx):
    """Stored (level 0), fixed (Z_FIXED), dynamic, Huffman-only, RLE; random bytes, ACGT text, period-8 and distance-1 runs, BAM-like
    records; sizes 0 .. 65280; several deflate blocks per stream.  ~700 streams in one launch: each lane its own kind of stream —
    the divergence the kernel's state machine is built for."""
    rng = random.Random(1)
    streams, want = [], []
    for trial in range(48):
        n = rng.choice([0, 1, 5, 100, 1000, 20000, 65280])
        kind = trial % 6
        if kind == 0:
            data = bytes(rng.getrandbits(8) for _ in range(n))
        elif kind == 1:
            data = bytes(rng.choice(b"ACGT") for _ in range(n))
        elif kind == 2:
            data = (b"ACGTTGCA" * (n // 8 + 1))[:n]
        elif kind == 3:
            data = bytes(rng.choice(b"AB") for _ in range(n))
        elif kind == 4:
            data = bytes([rng.randrange(4)]) * n
        else:
            rec = bytes(rng.getrandbits(8) for _ in range(40))
            out = bytearray()
            while len(out) < n:
                rec = bytes(b if rng.random() < 0.9 else rng.getrandbits(8) for b in rec)
                out += rec
            data = bytes(out[:n])
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                streams.append((raw_deflate(data, level, strat, split=trial % 3 == 0), len(data)))
                want.append(data)
    status, outs = run_blocks(ctx, streams)
    assert not status.any(), np.nonzero(status)[0][:10]
    for k, (o, w) in enumerate(zip(outs, want)):
        assert o == w, k
    # wrong sizes and truncated streams are declined — and a declined lane writes nothing beyond its own output
    bad = []
    for (raw, n), w in zip(streams[::7], want[::7]):
        if n > 1:                                  # (a block that claims ISIZE 0 is not decoded at all: BGZF's EOF marker)
            bad.append((raw, n - 1))
        bad.append((raw, n + 1))
        if len(raw) > 2 and n:
            bad.append((raw[:len(raw) // 2], n))
    status, _ = run_blocks(ctx, bad)
    assert status.all()


def test_inflate_kernel_never_accepts_what_zlib_rejects(ctx):
    """Bit flips: declined, or exactly zlib's bytes.  One launch of 1 500 corrupted streams next to each other."""
    rng = random.Random(7)
    streams, verdicts = [], []
    for trial in range(1500):
        n = rng.choice([50, 500, 5000, 30000])
        data = bytes(rng.choice(b"ACGTN") for _ in range(n)) if trial % 2 else bytes(rng.getrandbits(8) & 0x3f for _ in range(n))
        raw = bytearray(raw_deflate(data, rng.choice([1, 6, 9])))
        for _ in range(rng.randint(1, 4)):
            raw[rng.randrange(len(raw))] ^= 1 << rng.randrange(8)
        raw = bytes(raw)
        try:
            d = zlib.decompressobj(-15)
            z = d.decompress(raw) + d.flush()
            zok = d.eof and len(z) == n and not d.unused_data
        except zlib.error:
            zok, z = False, None
        streams.append((raw, n))
        verdicts.append((zok, z))
    status, outs = run_blocks(ctx, streams)
    accepted = 0
    for st, o, (zok, z) in zip(status, outs, verdicts):
        if st == 0:
            accepted += 1
            assert zok and o == z
    assert 100 < accepted < 1400


def test_inflate_kernel_on_the_reference_bam(ctx):
    """Every BGZF block of the reference's test.bam (multi-member gzip: one member per block), CRC32 of each trailer."""
    f = open(os.path.join(G, "test.bam"), "rb").read()
    o, blocks, crcs = 0, [], []
    while o + 18 <= len(f):
        xlen = struct.unpack_from("<H", f, o + 10)[0]
        bsize = struct.unpack_from("<H", f, o + 16)[0] + 1
        blocks.append((o + 12 + xlen, bsize - 12 - xlen - 8, struct.unpack_from("<I", f, o + bsize - 4)[0]))
        crcs.append(struct.unpack_from("<I", f, o + bsize - 8)[0])
        o += bsize
    status, outs = ctx.debug_inflate(f, blocks)
    assert not status.any()
    for (off, clen, isize), out, crc in zip(blocks, outs, crcs):
        assert out == zlib.decompress(f[off:off + clen], -15) and zlib.crc32(out) == crc


def stable_by_locus(raw, locus):
    order = np.argsort(locus, kind="stable")
    return raw[order], locus[order]


def ingest_and_compare(inputs, cfg_kw=None, pack_kw=None):
    """Device ingest of `inputs` against the host's raw pack: byte-level equality of everything the device builds, then equality of
    the resolved records and of the triplets after vtx_run.  Returns the ingest statistics."""
    pack_kw = dict(pack_kw or {})
    cfg_kw = dict(cfg_kw or {})
    use_umi = bool(pack_kw.get("use_umi", False))
    want, wmetrics, nv, barcodes, variants = hostlib.pack_files(raw=True, nibbles=True, threads=3, **inputs, **pack_kw)
    with hostlib.plan_ingest(**inputs, **pack_kw) as plan:
        assert plan.reason is None, plan.reason
        assert plan.n_variants == nv and plan.barcodes == barcodes and plan.variants == variants
        cfg = default_config(n_barcodes=len(barcodes), use_umi=int(use_umi), **cfg_kw)
        with lib.Context(cfg) as ctx:
            ctx.set_barcodes(barcodes)
            st = ctx.submit_bam(plan.ingest, plan.n_loci)
            raw = ctx.debug_ingest(abi.INGEST_RAW_RECORDS, abi.RAW_RECORD_DTYPE)
            locus = ctx.debug_ingest(abi.INGEST_RAW_LOCUS, np.uint32)
            tags = ctx.debug_ingest(abi.INGEST_TAGS)
            reads = ctx.debug_ingest(abi.INGEST_READS_PACKED)
            recs, begin, count = ctx.fetch_records()
            ctx.run()
            coo = ctx.fetch_coo()
            sc = ctx.fetch_scores()
        # the host's raw pack: records grouped by locus, BAM order inside; same arenas
        raw_s, locus_s = stable_by_locus(raw, locus)
        wl = np.repeat(np.arange(want.n_loci, dtype=np.uint32), want.loci["rec_count"])
        assert np.array_equal(locus_s, wl)
        assert np.array_equal(raw_s, want.records), np.nonzero(raw_s != want.records)[0][:5]
        assert np.array_equal(tags, want.tag_arena)
        assert np.array_equal(reads, want.read_arena)
        # metrics: the filters' counters from the device + the VCF-level ones from the plan
        got = dict(plan.metrics)
        got.update(num_reads=int(st.num_reads), num_low_mapq=int(st.num_low_mapq), num_non_primary=int(st.num_non_primary),
                   num_duplicates=int(st.num_duplicates), num_not_useful=int(st.num_not_useful),
                   num_not_cell_bc=int(st.num_no_barcode_tag), num_non_umi=0)
        assert got == wmetrics, (got, wmetrics)
        # ... and the same state after the preparation as the host-packed raw path
        with lib.Context(cfg) as c2:
            c2.set_barcodes(barcodes)
            rs = c2.submit_raw(want)
            recs2, begin2, count2 = c2.fetch_records()
            c2.run()
            coo2 = c2.fetch_coo()
            sc2 = c2.fetch_scores()
        assert (int(st.raw.num_not_cell_bc), int(st.raw.num_non_umi), int(st.raw.kept)) == (int(rs.num_not_cell_bc), int(rs.num_non_umi), int(rs.kept))
        assert np.array_equal(recs, recs2) and np.array_equal(begin, begin2) and np.array_equal(count, count2)
        assert np.array_equal(sc[0], sc2[0]) and np.array_equal(sc[1], sc2[1])
        for k in coo:
            a, b = np.asarray(coo[k]), np.asarray(coo2[k])
            assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), k
        return st


def ref_inputs():
    return dict(vcf=os.path.join(G, "test.vcf"), bam=os.path.join(G, "test.bam"), fasta=os.path.join(G, "test.fa"),
                cell_barcodes=os.path.join(G, "barcodes.tsv"))


@pytest.mark.parametrize("umi", [False, True])
def test_reference_fixture_through_the_device_ingest(umi):
    """test.vcf + test.bam (BASELINE configs[0]'s inputs): 576 reads over four contigs, lower-case FASTA, soft clips."""
    st = ingest_and_compare(ref_inputs(), pack_kw=dict(use_umi=umi))
    assert st.bam_records > 500 and st.inflated_bytes > 600000


@pytest.mark.parametrize("opts", [dict(), dict(mapq=30), dict(primary_only=True, no_duplicates=True), dict(use_umi=True, mapq=10),
                                  dict(bam_tag="CR"), dict(padding=30)])
def test_authored_bam_with_every_filter(tmp_path, opts):
    """tests/test_host.py's authored BAM over test_dna.fa — indels, soft / hard clips, N skips, secondary / supplementary / duplicate
    flags, missing and non-Z tags, reads that overlap several loci — with each filter option: the device's pairs ARE the host's."""
    from test_host import make_dna_bam
    bam = make_dna_bam(tmp_path, seed=3, n_reads=2500)
    inputs = dict(vcf=os.path.join(G, "test_dna.vcf"), bam=bam, fasta=os.path.join(G, "test_dna.fa"),
                  cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    st = ingest_and_compare(inputs, pack_kw=opts)
    assert (st.raw_records > 0) == ("bam_tag" not in opts)          # (no read carries a CR tag: every pair ends at num_not_cell_bc)


@pytest.mark.parametrize("index", ["linear", "csi"])
def test_many_blocks_many_seeds(tmp_path, index):
    """Small BGZF blocks (records cross block boundaries all the time) and a seed per 16 kb window — from a .bai's linear index, or
    from a .csi's leaf bins (round 6: src/main.rs:520-529 accepts either index)."""
    from test_host import make_dna_bam
    bam = make_dna_bam(tmp_path, seed=5, n_reads=6000, block=3000, index=index)
    inputs = dict(vcf=os.path.join(G, "test_dna.vcf"), bam=bam, fasta=os.path.join(G, "test_dna.fa"),
                  cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    ingest_and_compare(inputs, pack_kw=dict(use_umi=True))


def test_ranges_of_rows_add_up(tmp_path):
    """Streamed ranges (vtxh_plan_ingest of VCF rows [a, b)): the device ingest of each range equals the host pack of that range."""
    from test_host import make_dna_bam
    bam = make_dna_bam(tmp_path, seed=9, n_reads=3000)
    inputs = dict(vcf=os.path.join(G, "test_dna.vcf"), bam=bam, fasta=os.path.join(G, "test_dna.fa"),
                  cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    nv = hostlib.plan_ingest(**inputs).n_variants
    cuts = [0, nv // 3, 2 * nv // 3, nv]
    for a, b in zip(cuts[:-1], cuts[1:]):
        ingest_and_compare(inputs, pack_kw=dict(rows=(a, b)))


def test_declines_loudly():
    """No barcode list: VTX_E_STATE.  A seed that is not a record start: VTX_E_UNSUPPORTED (the caller packs on the host)."""
    with hostlib.plan_ingest(**ref_inputs()) as plan:
        with lib.Context(default_config(n_barcodes=len(plan.barcodes))) as ctx:
            with pytest.raises(lib.VtxError) as ei:
                ctx.submit_bam(plan.ingest, plan.n_loci)
            assert ei.value.status == abi.VTX_E_STATE
            ctx.set_barcodes(plan.barcodes)
            a = plan.arrays()
            seeds = a["seeds"].copy()
            seeds[1] += 1
            g = abi.VtxBamIngest.from_buffer_copy(plan.ingest)
            g.seeds = seeds.ctypes.data
            with pytest.raises(lib.VtxError) as ei:
                ctx.submit_bam(g, plan.n_loci)
            assert ei.value.status == abi.VTX_E_UNSUPPORTED
            st = ctx.submit_bam(plan.ingest, plan.n_loci)          # the context is fine afterwards
            assert st.bam_records > 500


def test_prefetched_bytes_give_the_same_ingest():
    """vtx_prefetch_file: the BAM's bytes travel before the plan exists; vtx_submit_bam then uses them (whole file, a sub-range that
    does not cover the plan's blocks — ignored — and a context created with n_barcodes 0)."""
    inputs = ref_inputs()
    f = np.fromfile(inputs["bam"], np.uint8)
    with hostlib.plan_ingest(**inputs) as plan:
        outs = []
        for pf in (None, (0, f.size), (f.size // 2, f.size - f.size // 2)):
            with lib.Context(default_config(n_barcodes=0 if pf else len(plan.barcodes))) as ctx:
                if pf:
                    ctx.prefetch_file(inputs["bam"], pf[0], pf[1])
                ctx.set_barcodes(plan.barcodes)
                st = ctx.submit_bam(plan.ingest, plan.n_loci)
                ctx.run()
                coo = ctx.fetch_coo()
                outs.append((int(st.raw_records), ctx.debug_ingest(abi.INGEST_RAW_RECORDS).tobytes(), coo["row"].tobytes(), coo["value"].tobytes()))
        assert outs[0] == outs[1] == outs[2] and outs[0][0] > 0


@pytest.mark.parametrize("mode", ["consensus", "coverage", "alt_frac"])
def test_matrix_market_text_from_the_device(tmp_path, mode):
    """vtx_write_mtx: the triplets formatted as Matrix-Market text on the device and streamed into the file — the bytes
    sprs::io::write_matrix_market writes (src/main.rs:381-389), i.e. what vtxh_write_mtx writes from the fetched triplets.  alt_frac's
    fractions need shortest round-trip digits: declined, nothing left behind."""
    from vartrix_amd import synth
    spec = synth.SynthSpec(n_loci=700, n_barcodes=900, reads_per_locus=40, indel_frac=0.2, use_umi=True, seed=5)
    batch = synth.make_batch(spec)
    with lib.Context(default_config(scoring_mode=mode, use_umi=1, n_barcodes=spec.n_barcodes)) as ctx:
        ctx.submit(batch)
        ctx.run()
        coo = ctx.fetch_coo()
        for which, key in ((0, "value"), (1, "ref_value")):
            p, q = str(tmp_path / ("dev%d.mtx" % which)), str(tmp_path / ("host%d.mtx" % which))
            if mode == "alt_frac":
                with pytest.raises(lib.VtxError) as ei:
                    ctx.write_mtx(p, spec.n_loci, spec.n_barcodes, which) if which == 0 else (_ for _ in ()).throw(lib.VtxError(abi.VTX_E_UNSUPPORTED, "n/a"))
                assert ei.value.status == abi.VTX_E_UNSUPPORTED and not os.path.exists(p)
                continue
            s = ctx.write_mtx(p, spec.n_loci, spec.n_barcodes, which)
            hostlib.write_mtx(q, spec.n_loci, spec.n_barcodes, coo["row"], coo["col"], coo[key])
            assert open(p, "rb").read() == open(q, "rb").read()
            assert s == float(np.asarray(coo[key]).sum()) and len(coo["row"]) > 5000
