"""ctypes wrapper of the CPU oracle (oracle/libvtx_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke() — never from vartrix_amd/ (the product).  See
oracle/vtx_oracle.h for what is restated and what is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from vartrix_amd.abi import PackedBatch, VtxBatch, VtxConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvtx_oracle.so")
MIN_SCORE = -858993459

_lib = None


def build(force: bool = False) -> str:
    # make decides (the Makefile lists every source and header the library depends on); -B only when forced
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        try:
            build()
        except (OSError, subprocess.CalledProcessError):
            if not os.path.exists(_SO):                 # (a box without make / gcc uses the prebuilt library)
                raise
        L = C.CDLL(_SO)
        u8p = C.c_char_p
        i32p = C.POINTER(C.c_int32)
        L.vtxo_sw_full.restype = C.c_int32
        L.vtxo_sw_full.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.vtxo_sw_banded.restype = C.c_int32
        L.vtxo_sw_banded.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.vtxo_sw_ranges.restype = C.c_int32
        L.vtxo_sw_ranges.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.vtxo_band_create.restype = C.c_int64
        L.vtxo_band_create.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.vtxo_find_kmer_matches.restype = C.c_int64
        L.vtxo_find_kmer_matches.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint32))]
        L.vtxo_sdpkpp.restype = C.c_int64
        L.vtxo_sdpkpp.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]
        L.vtxo_evaluate_scores.restype = C.c_int
        L.vtxo_evaluate_scores.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        L.vtxo_batch_scores.restype = C.c_int
        L.vtxo_batch_scores.argtypes = [C.POINTER(VtxBatch), C.POINTER(VtxConfig), C.c_void_p, C.c_void_p, C.c_int]
        L.vtxo_batch_cells.restype = C.c_uint64
        L.vtxo_batch_cells.argtypes = [C.POINTER(VtxBatch), C.POINTER(VtxConfig), C.c_int]
        L.vtxo_batch_reduce.restype = C.c_int64
        L.vtxo_batch_reduce.argtypes = [C.POINTER(VtxBatch), C.POINTER(VtxConfig), C.c_void_p, C.c_void_p] + [C.c_void_p] * 7
        L.vtxo_cigar_read_pos.restype = C.c_int
        L.vtxo_cigar_read_pos.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.vtxo_useful_alignment.restype = C.c_int
        L.vtxo_useful_alignment.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64]
        L.vtxo_construct_haplotypes.restype = None
        L.vtxo_construct_haplotypes.argtypes = [u8p, C.c_int64, C.c_int64, C.c_int64, u8p, C.c_int64, C.c_int64,
                                                C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64)]
        L.vtxo_format_f64.restype = C.c_int
        L.vtxo_format_f64.argtypes = [C.c_double, C.c_char_p]
        _lib = L
    return _lib


def _b(x) -> bytes:
    return bytes(x) if not isinstance(x, bytes) else x


def sw_full(read, hap, match=1, mismatch=-5, gap_open=-5, gap_extend=-1) -> int:
    read, hap = _b(read), _b(hap)
    return lib().vtxo_sw_full(read, len(read), hap, len(hap), match, mismatch, gap_open, gap_extend)


def sw_banded(read, hap, match=1, mismatch=-5, gap_open=-5, gap_extend=-1, k=6, w=20) -> int:
    read, hap = _b(read), _b(hap)
    return lib().vtxo_sw_banded(read, len(read), hap, len(hap), match, mismatch, gap_open, gap_extend, k, w)


def band_create(read, hap, k=6, w=20):
    read, hap = _b(read), _b(hap)
    lo = np.zeros(len(hap) + 1, np.int32)
    hi = np.zeros(len(hap) + 1, np.int32)
    cells = lib().vtxo_band_create(read, len(read), hap, len(hap), k, w, lo.ctypes.data, hi.ctypes.data)
    return lo, hi, int(cells)


def sw_ranges(read, hap, lo, hi, match=1, mismatch=-5, gap_open=-5, gap_extend=-1) -> int:
    read, hap = _b(read), _b(hap)
    lo = np.ascontiguousarray(lo, np.int32)
    hi = np.ascontiguousarray(hi, np.int32)
    return lib().vtxo_sw_ranges(read, len(read), hap, len(hap), match, mismatch, gap_open, gap_extend,
                                lo.ctypes.data, hi.ctypes.data)


def kmer_matches(x, y, k=6) -> np.ndarray:
    x, y = _b(x), _b(y)
    out = C.POINTER(C.c_uint32)()
    n = lib().vtxo_find_kmer_matches(x, len(x), y, len(y), k, C.byref(out))
    if n == 0:
        return np.zeros((0, 2), np.uint32)
    arr = np.ctypeslib.as_array(out, shape=(n * 2,)).copy().reshape(n, 2)
    C.CDLL(None).free(out)
    return arr


def sdpkpp(matches: np.ndarray, k=6, match_score=1, gap_open=-5, gap_extend=-1):
    m = np.ascontiguousarray(matches, np.uint32)
    path = np.zeros(max(len(m), 1), np.int64)
    score = C.c_int64(0)
    n = lib().vtxo_sdpkpp(m.ctypes.data, len(m), k, match_score, gap_open, gap_extend, path.ctypes.data, C.byref(score))
    return path[:n].copy(), int(score.value)


def evaluate_scores(ref_score: int, alt_score: int, min_score: int = 25) -> int:
    return lib().vtxo_evaluate_scores(ref_score, alt_score, min_score)


def batch_scores(batch: PackedBatch, cfg: VtxConfig, threads: int = 1):
    ref = np.zeros(batch.n_records, np.int32)
    alt = np.zeros(batch.n_records, np.int32)
    st = batch.as_struct()
    rc = lib().vtxo_batch_scores(C.byref(st), C.byref(cfg), ref.ctypes.data, alt.ctypes.data, threads)
    assert rc == 0
    return ref, alt


def batch_cells(batch: PackedBatch, cfg: VtxConfig, threads: int = 1) -> int:
    st = batch.as_struct()
    return int(lib().vtxo_batch_cells(C.byref(st), C.byref(cfg), threads))


def batch_reduce(batch: PackedBatch, cfg: VtxConfig, ref: np.ndarray, alt: np.ndarray) -> dict:
    n = max(batch.n_records, 1)
    out = {k: np.zeros(n, np.uint32) for k in ("row", "col", "alt", "ref", "unk")}
    out["value"] = np.zeros(n, np.float64)
    out["ref_value"] = np.zeros(n, np.float64)
    ref = np.ascontiguousarray(ref, np.int32)
    alt = np.ascontiguousarray(alt, np.int32)
    st = batch.as_struct()
    nnz = lib().vtxo_batch_reduce(C.byref(st), C.byref(cfg), ref.ctypes.data, alt.ctypes.data,
                                  *[out[k].ctypes.data for k in ("row", "col", "alt", "ref", "unk", "value", "ref_value")])
    return {k: v[:nnz].copy() for k, v in out.items()}


def cigar_read_pos(cigar, pos, ref_pos, include_softclips=False, include_dels=True):
    c = np.ascontiguousarray(cigar, np.uint32)
    q = C.c_int64(-1)
    r = lib().vtxo_cigar_read_pos(c.ctypes.data, len(c), pos, ref_pos, int(include_softclips), int(include_dels), C.byref(q))
    if r < 0:
        raise ValueError("invalid CIGAR")
    return int(q.value) if r == 1 else None


def useful_alignment(cigar, pos, start, end) -> bool:
    c = np.ascontiguousarray(cigar, np.uint32)
    return bool(lib().vtxo_useful_alignment(c.ctypes.data, len(c), pos, start, end))


def construct_haplotypes(contig: bytes, start: int, end: int, alt: bytes, padding: int = 100):
    cap = (end - start) + len(alt) + 2 * padding + 8
    rb = C.create_string_buffer(cap)
    ab = C.create_string_buffer(cap)
    rl, al = C.c_int64(0), C.c_int64(0)
    lib().vtxo_construct_haplotypes(contig, len(contig), start, end, alt, len(alt), padding,
                                    rb, C.byref(rl), ab, C.byref(al))
    return rb.raw[:rl.value], ab.raw[:al.value]


def format_f64(v: float) -> str:
    buf = C.create_string_buffer(64)
    n = lib().vtxo_format_f64(v, buf)
    return buf.raw[:n].decode()
