"""Python restatement of the reference's ingest + filter + pack front half.

TEST INFRASTRUCTURE ONLY (same rules as oracle.py).  It exists so the oracle
can be pinned, end to end, against the .mtx fixtures the reference's own tests
hold (tests/golden/, reference src/main.rs:1208-1390): it decodes the BAM with
gzip+struct (no htslib here), applies the reference's read filters and builds
the packed batch; the C oracle then scores and reduces it.

Reference lines restated (src/main.rs): load_barcodes 697-718, VCF loop
221-234, evaluate_rec 610-695, construct_haplotypes 958-994, evaluate_alns
809-934 (filters 833-895), useful_alignment 790-806, merge loop 320-348,
write_matrix_market call 381-389.
"""
from __future__ import annotations

import gzip
import struct
from dataclasses import dataclass, field

import numpy as np

from oracle import oracle
from vartrix_amd.abi import LOCUS_DTYPE, RECORD_DTYPE, PackedBatch

SEQ_NT16 = b"=ACMGRSVTWYHKDBN"
_NT16_PAIR = np.array([[SEQ_NT16[i >> 4], SEQ_NT16[i & 15]] for i in range(256)], dtype=np.uint8)

FLAG_UNMAP, FLAG_SECONDARY, FLAG_DUP, FLAG_SUPP = 0x4, 0x100, 0x400, 0x800


def _lines(data: bytes):
    # BufRead::lines(): split on \n, strip one trailing \r? (Rust strips "\n" and "\r\n")
    parts = data.split(b"\n")
    if parts and parts[-1] == b"":
        parts.pop()
    return [p[:-1] if p.endswith(b"\r") else p for p in parts]


def load_barcodes(path: str) -> dict:
    """load_barcodes, src/main.rs:697-718: line -> first-occurrence index; .gz by extension."""
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as fh:
        data = fh.read()
    bcs: dict = {}
    for line in _lines(data):
        if line not in bcs:
            bcs[line] = len(bcs)
    return bcs


@dataclass
class VcfRec:
    chrom: str
    pos: int          # 0-based, rec.pos()
    alleles: list     # [REF, ALT...] as bytes; ALT "." => only REF (src/main.rs:654-659)


def read_vcf(path: str) -> list:
    op = gzip.open if path.endswith(".gz") else open
    recs = []
    with op(path, "rb") as fh:
        for line in _lines(fh.read()):
            if not line or line.startswith(b"#"):
                continue
            f = line.split(b"\t")
            alts = [] if f[4] == b"." else f[4].split(b",")
            recs.append(VcfRec(f[0].decode(), int(f[1]) - 1, [f[3]] + alts))
    return recs


def read_fasta(path: str) -> dict:
    seqs, name, chunks = {}, None, []
    with open(path, "rb") as fh:
        for line in fh.read().split(b"\n"):
            if line.startswith(b">"):
                if name is not None:
                    seqs[name] = b"".join(chunks)
                name, chunks = line[1:].split()[0].decode(), []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        seqs[name] = b"".join(chunks)
    return seqs


@dataclass
class BamRec:
    tid: int
    pos: int
    mapq: int
    flag: int
    cigar: np.ndarray     # uint32 BAM ops
    seq: bytes
    qname: bytes
    aux: bytes
    end: int = 0


@dataclass
class Bam:
    refs: list = field(default_factory=list)    # [(name, len)]
    recs: list = field(default_factory=list)


_REF_CONSUME = {0, 2, 3, 7, 8}


def read_bam(path: str) -> Bam:
    with gzip.open(path, "rb") as fh:   # BGZF = concatenated gzip members
        d = fh.read()
    assert d[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", d, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, o)
    o += 4
    bam = Bam()
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", d, o)
        name = d[o + 4:o + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<i", d, o + 4 + l_name)
        bam.refs.append((name, l_ref))
        o += 8 + l_name
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o)
        tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, _nt, _np, _tl = struct.unpack_from("<iiBBHHHiiii", d, o + 4)
        p = o + 36
        qname = d[p:p + l_rn - 1]
        p += l_rn
        cigar = np.frombuffer(d, dtype="<u4", count=n_cig, offset=p).copy()
        p += 4 * n_cig
        packed = np.frombuffer(d, dtype=np.uint8, count=(l_seq + 1) // 2, offset=p)
        seq = _NT16_PAIR[packed].reshape(-1)[:l_seq].tobytes()
        p += (l_seq + 1) // 2 + l_seq
        aux = d[p:o + 4 + bs]
        rlen = sum(int(c >> 4) for c in cigar if int(c & 15) in _REF_CONSUME)
        end = pos + (rlen if rlen > 0 else 1)
        bam.recs.append(BamRec(tid, pos, mapq, flag, cigar, seq, qname, aux, end))
        o += 4 + bs
    return bam


_AUX_FIXED = {b"A": 1, b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4, b"d": 8}


def aux_string(aux: bytes, tag: bytes):
    """rec.aux(tag) matched against Aux::String (src/main.rs:742-748): Z only."""
    o = 0
    while o + 3 <= len(aux):
        t, ty = aux[o:o + 2], aux[o + 2:o + 3]
        o += 3
        if ty in _AUX_FIXED:
            size = _AUX_FIXED[ty]
            val = None
        elif ty in (b"Z", b"H"):
            e = aux.index(b"\x00", o)
            val = aux[o:e] if ty == b"Z" else None
            size = e - o + 1
        elif ty == b"B":
            sub = aux[o:o + 1]
            cnt, = struct.unpack_from("<i", aux, o + 1)
            size = 5 + cnt * _AUX_FIXED[sub]
            val = None
        else:
            raise ValueError("bad aux type %r" % ty)
        if t == tag:
            return val
        o += size
    return None


def fetch(bam: Bam, chrom: str, start: int, end: int):
    """bam.fetch((chrom, start, end)) + records(): records of `chrom` whose
    [pos, endpos) overlaps [start, end), in file order (htslib semantics)."""
    tid = [n for n, _ in bam.refs].index(chrom)
    for r in bam.recs:
        if r.tid == tid and r.pos < end and r.end > start:
            yield r


@dataclass
class Args:
    """Arguments struct src/main.rs:420-427 + --padding/--scoring flags."""
    mapq: int = 0
    primary: bool = False
    duplicates: bool = False
    use_umi: bool = False
    bam_tag: bytes = b"CB"
    valid_chars: bytes = b"ATGCatgc"
    padding: int = 100


METRIC_NAMES = ("num_reads", "num_low_mapq", "num_non_primary", "num_duplicates", "num_not_cell_bc",
                "num_not_useful", "num_non_umi", "num_invalid_recs", "num_multiallelic_recs")


def pack(vcf: list, fasta: dict, bam: Bam, barcodes: dict, args: Args):
    """evaluate_rec + the filter half of evaluate_alns for every VCF record ->
    (PackedBatch, metrics dict).  Row i = i-th VCF record (src/main.rs:224-228)."""
    metrics = dict.fromkeys(METRIC_NAMES, 0)
    loci, recs, haps, reads = [], [], bytearray(), bytearray()
    for i, v in enumerate(vcf):
        if len(v.alleles) > 2:                              # :646-653
            metrics["num_multiallelic_recs"] += 1
            continue
        alt = v.alleles[1] if len(v.alleles) == 2 else b""  # :656-659
        start, end = v.pos, v.pos + len(v.alleles[0])       # :619-623
        ref_hap, alt_hap = oracle.construct_haplotypes(fasta[v.chrom], start, end, alt, args.padding)
        if any(c not in args.valid_chars for c in alt_hap):  # :675-684 (whole ALT haplotype)
            metrics["num_invalid_recs"] += 1
            continue
        scored = []
        umi_ids: dict = {}
        for r in fetch(bam, v.chrom, start, end):           # :822-830
            metrics["num_reads"] += 1
            if r.mapq < args.mapq:                           # :833
                metrics["num_low_mapq"] += 1
                continue
            if args.primary and (r.flag & (FLAG_SECONDARY | FLAG_SUPP)):   # :841
                metrics["num_non_primary"] += 1
                continue
            if args.duplicates and (r.flag & FLAG_DUP):      # :849
                metrics["num_duplicates"] += 1
                continue
            if not oracle.useful_alignment(r.cigar, r.pos, start, end):    # :857
                metrics["num_not_useful"] += 1
                continue
            cb = aux_string(r.aux, args.bam_tag)             # :867
            cell = barcodes.get(cb) if cb is not None else None
            if cell is None:
                metrics["num_not_cell_bc"] += 1
                continue
            umi = aux_string(r.aux, b"UB")                   # :879
            if args.use_umi and umi is None:
                metrics["num_non_umi"] += 1
                continue
            if not args.use_umi:
                umi = b"\x01"                                # :890-894 dummy UMI
            uid = umi_ids.setdefault(umi, len(umi_ids))
            scored.append((cell, uid, r.seq))
        scored.sort(key=lambda t: (t[0], t[1]))              # :932 (stable by cell) + UMI grouping
        rec_begin = len(recs)
        for cell, uid, seq in scored:
            recs.append((len(reads), len(seq), cell, uid))
            reads += seq
        loci.append((i, rec_begin, len(scored), len(haps), len(ref_hap), len(haps) + len(ref_hap), len(alt_hap), 0))
        haps += ref_hap + alt_hap
    batch = PackedBatch(np.array(loci, dtype=LOCUS_DTYPE).reshape(-1), np.array(recs, dtype=RECORD_DTYPE).reshape(-1),
                        np.frombuffer(bytes(haps), np.uint8), np.frombuffer(bytes(reads), np.uint8))
    return batch, metrics


def mtx_text(n_rows: int, n_cols: int, rows, cols, vals) -> str:
    """sprs 0.7.1 write_matrix_market of a TriMat<f64> (src/main.rs:381): header,
    comment, dims, then 1-based `row col val` in insertion order."""
    out = ["%%MatrixMarket matrix coordinate real general", "% written by sprs",
           "%d %d %d" % (n_rows, n_cols, len(rows))]
    for r, c, v in zip(rows, cols, vals):
        out.append("%d %d %s" % (int(r) + 1, int(c) + 1, oracle.format_f64(float(v))))
    return "\n".join(out) + "\n"


def read_mtx(path_or_text):
    """sprs read_matrix_market -> dict {(row0, col0): value} (the reference's tests
    compare CSR, i.e. entry order is not significant, src/main.rs:1230-1232)."""
    text = open(path_or_text).read() if "\n" not in path_or_text else path_or_text
    lines = [l for l in text.split("\n") if l and not l.startswith("%")]
    nr, nc, nnz = (int(t) for t in lines[0].split())
    ent = {}
    for l in lines[1:1 + nnz]:
        r, c, v = l.split()
        ent[(int(r) - 1, int(c) - 1)] = ent.get((int(r) - 1, int(c) - 1), 0.0) + float(v)
    return (nr, nc), ent
