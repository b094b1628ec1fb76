"""Minimal BAM writer (BGZF via zlib) — TEST INFRASTRUCTURE ONLY.

Used to author coordinate-sorted BAMs over the reference's test_dna.fa /
test_dna.vcf (whose own BAM is missing upstream) so that ingest, the read
filters and indel loci can be exercised end to end.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def _bgzf_block(data: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return hdr + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def parse_cigar(s: str) -> list:
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | _OPS[ch])
            num = ""
    return out


def record(tid, pos, qname, seq, cigar, flag=0, mapq=60, tags=()):
    """tags: iterable of (tag, type, value) with type 'Z' (bytes/str), 'i' (int), 'A' (char), 'd' / 'f' (float) or
    'B' ((subtype, [ints]))."""
    qn = qname.encode() + b"\x00"
    cig = parse_cigar(cigar) if isinstance(cigar, str) else list(cigar)
    l_seq = len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(seq):
        code = _NT16.get(ch, 15)
        packed[i >> 1] |= code << (4 if i % 2 == 0 else 0)
    aux = b""
    for tag, ty, val in tags:
        if ty == "Z":
            v = val.encode() if isinstance(val, str) else val
            aux += tag.encode() + b"Z" + v + b"\x00"
        elif ty == "i":
            aux += tag.encode() + b"i" + struct.pack("<i", val)
        elif ty == "A":
            aux += tag.encode() + b"A" + val.encode()
        elif ty == "d":
            aux += tag.encode() + b"d" + struct.pack("<d", val)
        elif ty == "f":
            aux += tag.encode() + b"f" + struct.pack("<f", val)
        elif ty == "B":         # val = (subtype char, list of ints)
            sub, vals = val
            aux += tag.encode() + b"B" + sub.encode() + struct.pack("<i", len(vals)) + b"".join(
                struct.pack({"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[sub], v) for v in vals)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(qn), mapq, 4680, len(cig), flag, l_seq, -1, -1, 0)
    body += qn + b"".join(struct.pack("<I", c) for c in cig) + bytes(packed) + b"\xff" * l_seq + aux
    return struct.pack("<i", len(body)) + body


def _ref_span(rec: bytes):
    """(tid, pos, end) of one encoded record (end like bam_endpos: pos + 1 without a reference-consuming op)."""
    tid, pos, l_rn, _mq, _bin, n_cig, flag = struct.unpack_from("<iiBBHHH", rec, 4)
    o = 36 + l_rn
    rlen = 0
    if not flag & 4:
        for k in range(n_cig):
            c, = struct.unpack_from("<I", rec, o + 4 * k)
            if (c & 15) in (0, 2, 3, 7, 8):
                rlen += c >> 4
    return tid, pos, pos + (rlen if rlen > 0 else 1)


def write_bam(path: str, refs: list, records: list, block: int = 60000, index: str = "linear"):
    """refs = [(name, length)], records = output of record() in coordinate order.
    index: "csi" writes a .csi instead (CSI v1, min_shift 14, depth 5: every record filed under its smallest containing bin, every
    bin with its chunks and its loffset — no linear index, as htslib writes it); "linear" writes a .bai whose LINEAR index is real (smallest virtual offset of an alignment overlapping each
    16 kb window, gaps filled with the previous value like htslib does) and whose bin index is empty — enough for the
    packer's index-guided skipping; "fake" writes the magic with zero references (the packer then sweeps everything)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for name, ln in refs:
        n = name.encode() + b"\x00"
        hdr += struct.pack("<i", len(n)) + n + struct.pack("<i", ln)
    data = hdr + b"".join(records)
    coffs = []
    with open(path, "wb") as fh:
        for o in range(0, len(data), block):
            coffs.append(fh.tell())
            fh.write(_bgzf_block(data[o:o + block]))
        fh.write(_bgzf_block(b""))          # EOF marker
    if index == "csi":
        _write_csi(path + ".csi", refs, records, hdr, coffs, block)
        return
    with open(path + ".bai", "wb") as fh:
        if index != "linear":
            fh.write(b"BAI\x01" + struct.pack("<i", 0))
            return
        lin = [[0] * ((ln >> 14) + 1) for _, ln in refs]
        u = len(hdr)
        for rec in records:
            tid, pos, end = _ref_span(rec)
            if 0 <= tid < len(refs):
                voff = (coffs[u // block] << 16) | (u % block)
                for w in range(max(pos, 0) >> 14, min(((end - 1) >> 14) + 1, len(lin[tid]))):
                    if lin[tid][w] == 0:
                        lin[tid][w] = voff
            u += len(rec)
        out = b"BAI\x01" + struct.pack("<i", len(refs))
        for l in lin:
            n_intv = max((i + 1 for i, v in enumerate(l) if v), default=0)
            l = l[:n_intv]
            for i in range(1, len(l)):
                if l[i] == 0:
                    l[i] = l[i - 1]
            out += struct.pack("<i", 0) + struct.pack("<i", len(l)) + b"".join(struct.pack("<Q", v) for v in l)
        fh.write(out)


def _reg2bin(beg, end, min_shift=14, depth=5):
    """CSI spec 5.1.1: the smallest bin that contains [beg, end)."""
    end -= 1
    s, t = min_shift, ((1 << depth * 3) - 1) // 7
    level = depth
    while level > 0:
        if beg >> s == end >> s:
            return t + (beg >> s)
        s += 3
        t -= 1 << ((level - 1) * 3)
        level -= 1
    return 0


def _write_csi(path, refs, records, hdr, coffs, block, min_shift=14, depth=5):
    """A real CSI: per reference the bins that hold records, each with loffset (smallest virtual offset of a record that overlaps
    the bin's interval) and one chunk [first record's offset, end of its last record)."""
    bins = [dict() for _ in refs]          # bin -> [chunk_beg, chunk_end]
    spans = [[] for _ in refs]             # (pos, end, voff) for the loffsets
    u = len(hdr)
    for rec in records:
        tid, pos, end = _ref_span(rec)
        voff = (coffs[u // block] << 16) | (u % block)
        u2 = u + len(rec)
        vend = (coffs[u2 // block] << 16) | (u2 % block) if u2 // block < len(coffs) else ((coffs[-1] + 1) << 16)
        if 0 <= tid < len(refs):
            b = _reg2bin(max(pos, 0), max(end, pos + 1), min_shift, depth)
            c = bins[tid].setdefault(b, [voff, vend])
            c[1] = vend
            spans[tid].append((max(pos, 0), max(end, pos + 1), voff))
        u = u2
    out = b"CSI\x01" + struct.pack("<iii", min_shift, depth, 0) + struct.pack("<i", len(refs))
    for t in range(len(refs)):
        out += struct.pack("<i", len(bins[t]))
        for b in sorted(bins[t]):
            # the bin's interval
            level, first = depth, ((1 << depth * 3) - 1) // 7
            while level > 0 and b < first:
                level -= 1
                first = ((1 << level * 3) - 1) // 7
            span = 1 << (min_shift + 3 * (depth - level))
            lo, hi = (b - first) * span, (b - first + 1) * span
            loff = min(v for p0, p1, v in spans[t] if p0 < hi and p1 > lo)
            out += struct.pack("<IQi", b, loff, 1) + struct.pack("<QQ", *bins[t][b])
    with open(path, "wb") as fh:
        for o in range(0, len(out), 60000):
            fh.write(_bgzf_block(out[o:o + 60000]))
        fh.write(_bgzf_block(b""))


def vcf_to_bcf(vcf_path, bcf_path):
    """A text VCF as BCF2 (BGZF): the header text verbatim, every record with CHROM as the index of its ##contig line, 0-based POS,
    the ID and the alleles as typed strings; no INFO, no FORMAT.  Enough for what the reference reads (src/main.rs:220-234)."""
    lines = open(vcf_path).read().splitlines()
    header = [ln for ln in lines if ln.startswith("#")]
    contigs = []
    for ln in header:
        if ln.startswith("##contig=<"):
            contigs.append(ln.split("ID=", 1)[1].split(",")[0].split(">")[0])
    text = ("\n".join(header) + "\n").encode() + b"\x00"
    out = bytearray(b"BCF\x02\x02" + struct.pack("<I", len(text)) + text)

    def typed_str(b):
        if len(b) < 15:
            return bytes([(len(b) << 4) | 7]) + b
        if len(b) < 128:
            return bytes([0xF7, 0x11, len(b)]) + b
        return bytes([0xF7, 0x12]) + struct.pack("<H", len(b)) + b
    for ln in lines:
        if not ln or ln.startswith("#"):
            continue
        f = ln.split("\t")
        chrom, pos, vid, ref, alt = f[0], int(f[1]) - 1, f[2], f[3], f[4]
        if chrom not in contigs:
            contigs.append(chrom)
        alleles = [ref] + ([] if alt == "." else alt.split(","))
        shared = struct.pack("<iiif", contigs.index(chrom), pos, len(ref), float("nan"))
        shared += struct.pack("<II", (len(alleles) << 16) | 0, 0)
        shared += typed_str(b"" if vid == "." else vid.encode())
        for a in alleles:
            shared += typed_str(a.encode())
        shared += bytes([0x00])                                   # FILTER: an empty typed vector
        out += struct.pack("<II", len(shared), 0) + shared
    with open(bcf_path, "wb") as fh:
        for o in range(0, len(out), 60000):
            fh.write(_bgzf_block(bytes(out[o:o + 60000])))
        fh.write(_bgzf_block(b""))
