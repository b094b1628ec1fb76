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
    index: "linear" writes a .bai whose LINEAR index is real (smallest virtual offset of an alignment overlapping each
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
