"""CPU unit tests of vartrix_amd/csrc/vtx_inflate_core.h — the per-lane raw-DEFLATE decoder bgzf_inflate_kernel is compiled from
(device-side BGZF inflate of the BAM, the bytes the reference gets through rust-htslib -> htslib bgzf_read -> zlib behind
src/main.rs:822-830) — built for the host by tests/inflatecore/Makefile, against zlib: every accepted stream equals zlib's output
byte for byte, what zlib rejects is never accepted, nothing is written beyond the block's output.  The device runs the same cases
through the kernel in tests/test_gpu_ingest.py."""
import ctypes as C
import os
import random
import struct
import subprocess
import zlib

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def core():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "inflatecore"), "-s"])
    L = C.CDLL(os.path.join(HERE, "inflatecore", "libinflate_host.so"))
    L.vtxt_inflate.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    L.vtxt_inflate.restype = C.c_uint32
    return L


def own(L, raw, n, stride=1):
    out = (C.c_uint8 * (n + 64))()
    C.memset(C.addressof(out), 0xCD, n + 64)
    trips = C.c_uint32(0)
    st = L.vtxt_inflate(raw, len(raw), C.addressof(out), n, stride, C.addressof(trips))
    assert bytes(out[n:n + 64]) == b"\xcd" * 64, "the decoder wrote beyond its output"
    return st, bytes(out[:n]), trips.value


def corpus(rng, trial, n):
    kind = trial % 6
    if kind == 0:
        return bytes(rng.getrandbits(8) for _ in range(n))
    if kind == 1:
        return bytes(rng.choice(b"ACGT") for _ in range(n))
    if kind == 2:
        return (b"ACGTTGCA" * (n // 8 + 1))[:n]
    if kind == 3:
        return bytes(rng.choice(b"AB") for _ in range(n))
    if kind == 4:
        return bytes([rng.randrange(4)]) * n
    rec = bytes(rng.getrandbits(8) for _ in range(40))            # BAM-like: records that repeat most of the previous one
    out = bytearray()
    while len(out) < n:
        rec = bytes(b if rng.random() < 0.9 else rng.getrandbits(8) for b in rec)
        out += rec
    return bytes(out[:n])


def test_equals_zlib_on_every_block_kind(core):
    """Stored, fixed and dynamic blocks, literal-only and match-heavy data (distance 1 runs, distance 8 periods), every size from
    empty to a full BGZF block, several deflate blocks per stream (Z_FULL_FLUSH): accepted and byte-identical; a wrong output size or
    a truncated stream is declined."""
    rng = random.Random(1)
    accepted = 0
    for trial in range(48):
        n = rng.choice([0, 1, 5, 100, 1000, 20000, 65280])
        data = corpus(rng, trial, n)
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strat)
                if trial % 3 == 0 and n > 10:      # several deflate blocks, a stored empty block between them
                    raw = co.compress(data[:n // 3]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(data[n // 3:]) + co.flush()
                else:
                    raw = co.compress(data) + co.flush()
                st, out, trips = own(core, raw, len(data), stride=1 + trial % 3 * 31)
                assert st == 0 and out == data, (trial, level, strat, n, st)
                accepted += 1
                if data:
                    assert own(core, raw, len(data) - 1)[0] != 0
                assert own(core, raw, len(data) + 1)[0] != 0
                if len(raw) > 2 and data:
                    assert own(core, raw[:len(raw) // 2], len(data))[0] != 0
    assert accepted == 48 * 16


def test_never_accepts_what_zlib_rejects(core):
    """Bit flips in valid streams: the decoder either declines (the ingest then falls back to the host packer, where zlib decides) or
    returns exactly what zlib returns."""
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
        st, out, _ = own(core, raw, n)
        try:
            d = zlib.decompressobj(-15)
            z = d.decompress(raw) + d.flush()
            zok = d.eof and len(z) == n and not d.unused_data
        except zlib.error:
            zok, z = False, None
        if st == 0:
            accepted += 1
            assert zok and out == z, trial
        else:
            declined += 1
    assert accepted > 100 and declined > 100


def test_long_huffman_codes(core):
    """Geometrically distributed symbols: zlib's length-limited codes reach 15 bits — every word of the register code is used."""
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
            st, out, _ = own(core, raw, len(data))
            assert st == 0 and out == data, (trial, strat)


def test_hand_made_streams(core):
    """Corner cases no compressor emits: a dynamic block whose distance table has ONE code of one bit (RFC 1951 3.2.7: accepted, the
    other bit pattern invalid), no distance code at all (literals only), an incomplete literal code (declined), block type 3."""
    def bits_to_bytes(bits):
        out = bytearray((len(bits) + 7) // 8)
        for i, b in enumerate(bits):
            out[i >> 3] |= b << (i & 7)
        return bytes(out)

    def lsb(v, n):
        return [(v >> i) & 1 for i in range(n)]

    def msb(v, n):
        return [(v >> (n - 1 - i)) & 1 for i in range(n)]

    assert own(core, bits_to_bytes([1, 1, 1]), 0)[0] != 0                       # BTYPE 3
    # dynamic block: literals 'a' (97) and end-of-block with 1-bit codes; distance table: one code of length 1 (symbol 0)
    # code-length code: symbols 0 and 1 used -> lengths 1 and 1 ("0" -> len 0, "1" -> len 1)
    def dyn(dist_lens, body):
        order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
        cl = {0: 1, 1: 2, 18: 2}          # complete: 0 -> '0', 1 -> '10', 18 -> '11'
        code = {0: msb(0, 1), 1: msb(2, 2), 18: msb(3, 2)}
        hclen = 19
        bits = [1] + lsb(2, 2) + lsb(0, 5) + lsb(len(dist_lens) - 1, 5) + lsb(hclen - 4, 4)
        for s in order:
            bits += lsb(cl.get(s, 0), 3)
        lens = [0] * 257
        lens[97] = 1
        lens[256] = 1
        i = 0
        seq = lens + dist_lens
        while i < len(seq):
            if seq[i] == 0:
                j = i
                while j < len(seq) and seq[j] == 0 and j - i < 138:
                    j += 1
                if j - i >= 11:
                    bits += code[18] + lsb(j - i - 11, 7)
                    i = j
                    continue
                bits += code[0]
                i += 1
            else:
                bits += code[1]
                i += 1
        return bits_to_bytes(bits + body)
    # 'a' = code 0, EOB = code 1 (canonical: 97 < 256)
    st, out, _ = own(core, dyn([1], [0, 0, 0, 1]), 3)
    assert st == 0 and out == b"aaa"
    st, out, _ = own(core, dyn([0], [0, 0, 1]), 2)                              # no distance code at all
    assert st == 0 and out == b"aa"
    for raw, n in ((dyn([1], [0, 0, 0, 1]), 3), (dyn([0], [0, 0, 1]), 2)):
        assert zlib.decompress(raw, -15) == b"a" * n


def test_reference_bam_blocks(core):
    """Every BGZF block of the reference's test BAM (tests/golden/test.bam = /root/reference/test/test.bam, data fixture): accepted,
    byte-identical to zlib, CRC32 of the block's trailer."""
    f = open(os.path.join(G, "test.bam"), "rb").read()
    o = blocks = 0
    while o + 18 <= len(f):
        xlen = struct.unpack_from("<H", f, o + 10)[0]
        bsize = struct.unpack_from("<H", f, o + 16)[0] + 1
        raw = f[o + 12 + xlen:o + bsize - 8]
        isize = struct.unpack_from("<I", f, o + bsize - 4)[0]
        st, out, trips = own(core, raw, isize, stride=64)
        assert st == 0 and out == zlib.decompress(raw, -15) and zlib.crc32(out) == struct.unpack_from("<I", f, o + bsize - 8)[0]
        blocks += 1
        o += bsize
    assert blocks >= 2
