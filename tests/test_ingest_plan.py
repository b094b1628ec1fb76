"""The plan of a device-side ingest (libvtxhost: vtxh_plan_ingest -> the struct vtx_submit_bam takes) without a GPU.

The device cuts the BAM's serial record chain at the record starts the .bai's linear index names and looks at nothing outside
[first seed, end_upos).  So, with the blocks inflated by zlib here: every seed IS a record start and every chain lands exactly on
the next seed; the stream covered holds EVERY record that overlaps a planned locus (what `bam.fetch(..)` would return,
src/main.rs:822-826); blocks are consecutive file ranges; intervals are sorted per contig.  The kernels themselves are checked
against the host packer on the device (tests/test_gpu_ingest.py)."""
import os
import struct
import sys
import zlib

import numpy as np
import pytest

from oracle import refpipe
from vartrix_amd import hostlib

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
sys.path.insert(0, HERE)


def check_plan(inputs, **kw):
    f = open(inputs["bam"], "rb").read()
    with hostlib.plan_ingest(**inputs, **kw) as plan:
        assert plan.reason is None, plan.reason
        a = plan.arrays()
    blocks, seeds, end = a["blocks"], a["seeds"], a["end_upos"]
    # consecutive blocks of the file; inflate them
    data = bytearray()
    for k, b in enumerate(blocks):
        if k:
            prev = blocks[k - 1]
            assert int(b["coff"]) >= int(prev["coff"]) + int(prev["clen"]) + 8
        out = zlib.decompress(f[int(b["coff"]):int(b["coff"]) + int(b["clen"])], -15)
        assert len(out) == int(b["isize"])
        data += out
    data = bytes(data)
    assert end <= len(data)
    # the chains
    assert np.all(np.diff(seeds.astype(np.int64)) > 0) if len(seeds) > 1 else True
    starts = []
    for i, s in enumerate(seeds):
        p, stop = int(s), int(seeds[i + 1]) if i + 1 < len(seeds) else end
        while p < stop:
            bs, = struct.unpack_from("<I", data, p)
            assert bs >= 32
            starts.append(p)
            p += 4 + bs
        assert p == stop, "chain %d does not land on the next seed" % i
    # every record that overlaps a locus is among them
    bam = refpipe.read_bam(inputs["bam"])
    iv, tb = a["intervals"], a["tid_begin"]
    for t in range(len(tb) - 1):
        assert np.all(np.diff(iv["start"][tb[t]:tb[t + 1]]) >= 0)
        if tb[t + 1] > tb[t]:
            assert (iv["end"][tb[t]:tb[t + 1]] - iv["start"][tb[t]:tb[t + 1]]).max() <= a["tid_max_span"][t]
    covered = set()
    for p in starts:
        tid, pos = struct.unpack_from("<ii", data, p + 4)
        covered.add((tid, pos, data[p + 36:p + 36 + data[p + 12]]))
    need = 0
    for r in bam.recs:
        if r.tid < 0 or r.tid >= len(tb) - 1:
            continue
        s = iv[tb[r.tid]:tb[r.tid + 1]]
        if np.any((s["start"] < r.end) & (s["end"] > r.pos)):
            need += 1
            assert (r.tid, r.pos, r.qname + b"\x00") in covered, (r.tid, r.pos, r.qname)
    return len(starts), need, len(blocks), plan.blocks_total


def test_plan_of_the_reference_fixture():
    inputs = dict(vcf=os.path.join(G, "test.vcf"), bam=os.path.join(G, "test.bam"), fasta=os.path.join(G, "test.fa"),
                  cell_barcodes=os.path.join(G, "barcodes.tsv"))
    n, need, nb, total = check_plan(inputs)
    assert need > 500 and n >= need


@pytest.mark.parametrize("block", [20000, 3000, 700])
def test_plan_of_authored_bams(tmp_path, block):
    from test_host import make_dna_bam
    bam = make_dna_bam(tmp_path, seed=4, n_reads=3000, block=block)
    inputs = dict(vcf=os.path.join(G, "test_dna.vcf"), bam=bam, fasta=os.path.join(G, "test_dna.fa"),
                  cell_barcodes=os.path.join(G, "dna_barcodes.tsv"))
    n, need, nb, total = check_plan(inputs)
    assert need > 1000
    # ranges of rows: each plan covers its own loci's reads and stops early
    nv = 46
    sizes = []
    for a, b in ((0, 10), (10, 30), (30, nv)):
        n2, need2, nb2, _ = check_plan(inputs, rows=(a, b))
        sizes.append(nb2)
    assert min(sizes) <= nb


def test_no_index_no_plan(tmp_path):
    """Without a usable .bai there is no plan (the record starts come from it): the caller packs on the host."""
    import shutil
    bam = str(tmp_path / "t.bam")
    shutil.copy(os.path.join(G, "test.bam"), bam)
    open(bam + ".bai", "wb").write(b"BAI\x01" + struct.pack("<i", 0))
    with hostlib.plan_ingest(os.path.join(G, "test.vcf"), bam, os.path.join(G, "test.fa"), os.path.join(G, "barcodes.tsv")) as plan:
        assert plan.ingest is None and ".bai" in plan.reason
        assert plan.n_variants == 4 and len(plan.barcodes) > 0
