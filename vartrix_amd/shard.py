"""Sharding of variant loci over the GPUs of a node + the row gather.

Loci (matrix rows) are independent in the reference — the rayon map over
chunks has no shared mutable state (``src/main.rs:284-291``) and the merge loop
is a concatenation in chunk order (``src/main.rs:320-348``).  So the multi-GPU
path is: contiguous row ranges balanced by record count, one process per GPU,
no collective on the data path, and ONE exchange at the end — every rank's COO
block (row, col and the three counts; the f64 values are recomputed from the
counts on rank 0) goes to rank 0 over its direct xGMI link (a gather = grouped
point-to-point send/recv, the pattern xGMI is built for; counts first with one
all_gather), asynchronously so it overlaps with the next batch's compute.
Rank 0 concatenates in rank order, which is already (row asc, col asc).

Works on any ``torch.distributed`` backend: ``nccl`` (= RCCL on ROCm) with GPU
tensors, ``gloo`` with CPU tensors (the world_size-2 tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .abi import PackedBatch

COO_FIELDS = (("row", np.uint32), ("col", np.uint32), ("alt", np.uint32), ("ref", np.uint32),
              ("unk", np.uint32), ("value", np.float64), ("ref_value", np.float64))
_TORCH_DT = {np.uint32: torch.int32, np.float64: torch.float64}


def partition_loci(batch: PackedBatch, world: int) -> list:
    """Contiguous locus ranges [(lo, hi)] x world with ~equal record counts.

    The reference cuts loci into equal-count chunks (``src/main.rs:250-254``);
    here the cut points follow the prefix sum of records per locus so each GPU
    gets the same number of alignments.
    """
    n = batch.n_loci
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (world - 1)
    csum = np.cumsum(batch.loci["rec_count"].astype(np.int64))
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        cut = int(np.searchsorted(csum, target, side="left")) + 1 if total else n * r // world
        cuts.append(min(max(cut, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


class _DevArray:
    """Zero-copy view of a raw device pointer for ``torch.as_tensor`` (HIP exposes
    the CUDA array interface under the same name)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_coo_tensors(ctx, device, d: dict = None) -> dict:
    """Wrap device-resident triplets as torch tensors: the context's own (vtx_device_coo), or the address dict
    ``d`` of a native gather (``Context.gather_coo``)."""
    if d is None:
        d = ctx.device_coo()
    n = d["nnz"]
    out = {}
    for k, dt in COO_FIELDS:
        if n == 0:
            out[k] = torch.zeros(0, dtype=_TORCH_DT[dt], device=device)
        else:
            ts = "<i4" if dt is np.uint32 else "<f8"
            out[k] = torch.as_tensor(_DevArray(d[k], n, ts), device=device)
    return out


def coo_to_tensors(coo: dict, device="cpu") -> dict:
    out = {}
    for k, dt in COO_FIELDS:
        a = np.ascontiguousarray(coo[k], dtype=dt)
        if dt is np.uint32:
            a = a.view(np.int32)
        out[k] = torch.from_numpy(a.copy()).to(device)
    return out


def tensors_to_coo(t: dict) -> dict:
    out = {}
    for k, dt in COO_FIELDS:
        a = t[k].detach().cpu().numpy()
        out[k] = a.view(np.uint32).copy() if dt is np.uint32 else a.copy()
    return out


def values_from_counts(alt: torch.Tensor, ref: torch.Tensor, unk: torch.Tensor, mode: int):
    """(value, ref_value) of the matrix modes from the per-group counts — the same arithmetic as the
    emit kernel / reference ``src/main.rs:1120-1126, 1140-1142, 1160-1161`` (f64, IEEE division)."""
    a, r, u = alt.to(torch.float64), ref.to(torch.float64), unk.to(torch.float64)
    if mode == 0:      # consensus
        v = torch.where((ref > 0) & (alt > 0), 3.0, torch.where(alt > 0, 2.0, 1.0)).to(torch.float64)
        return v, torch.zeros_like(v)
    if mode == 1:      # alt_frac (0/0 -> NaN like the reference)
        return a / (r + a + u), torch.zeros_like(a)
    return a, r        # coverage


def _pack(local: dict, rows: int) -> torch.Tensor:
    """Triplet arrays -> one int32 payload [rows, 5] (row, col, alt, ref, unk), zero padded.  The f64 values
    are a function of the counts and the mode, so they are recomputed on the destination instead of sent."""
    dev = local["row"].device
    n = local["row"].shape[0]
    buf = torch.zeros((rows, 5), dtype=torch.int32, device=dev)
    if n:
        for c, k in enumerate(("row", "col", "alt", "ref", "unk")):
            buf[:n, c] = local[k]
    return buf


def _unpack(buf: torch.Tensor, mode: int) -> dict:
    out = {k: buf[:, c].contiguous() for c, k in enumerate(("row", "col", "alt", "ref", "unk"))}
    out["value"], out["ref_value"] = values_from_counts(out["alt"], out["ref"], out["unk"], mode)
    return out


class GatherHandle:
    """An in-flight gather_coo: ``wait()`` returns the concatenated dict on dst, None elsewhere."""

    def __init__(self, work, bufs, counts, payload, mode, is_dst):
        self._work, self._bufs, self._counts, self._payload, self._mode, self._is_dst = work, bufs, counts, payload, mode, is_dst

    def wait(self):
        if self._work is not None:
            self._work.wait()
        if not self._is_dst:
            return None
        return _unpack(torch.cat([b[:c] for b, c in zip(self._bufs, self._counts)], dim=0), self._mode)


def gather_coo_async(local: dict, mode: int, group=None, dst: int = 0) -> GatherHandle:
    """Start gathering every rank's triplets to ``dst`` (rank order = row order).

    One tiny all_gather of the counts, then ONE asynchronous gather of a packed, padded int32 payload
    per rank (RCCL lowers gather to grouped send/recv: each rank's block travels once over its direct
    xGMI link to ``dst``).  The payload is a copy, so the caller may overwrite the source arrays (run
    the next step) as soon as this returns — the exchange overlaps with compute.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local["row"].device
    n_local = torch.tensor([local["row"].shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    rows = max(max(counts), 1)
    payload = _pack(local, rows)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()      # the copy out of the source arrays is complete
    if rank == dst:
        bufs = [torch.empty((rows, 5), dtype=torch.int32, device=dev) for _ in range(world)]
        work = dist.gather(payload, gather_list=bufs, dst=dst, group=group, async_op=True)
        return GatherHandle(work, bufs, counts, payload, mode, True)
    work = dist.gather(payload, gather_list=None, dst=dst, group=group, async_op=True)
    return GatherHandle(work, None, counts, payload, mode, False)


def gather_coo(local: dict, mode: int = 2, group=None, dst: int = 0):
    """Blocking form of gather_coo_async."""
    return gather_coo_async(local, mode, group, dst).wait()


class GatherPipeline:
    """The per-step row exchange of a sharded run, one step deep: ``push`` starts the gather of this step's
    triplets and completes the previous one, so the exchange of step k overlaps with the kernels of step k+1;
    ``drain`` completes the last one.  ``last`` holds the most recent gathered matrix on ``dst`` (None elsewhere).
    bench.py's timed loop and the CPU (gloo) tests drive this same object."""

    def __init__(self, mode: int, group=None, dst: int = 0):
        self.mode, self.group, self.dst = mode, group, dst
        self._pending = None
        self.last = None
        self.completed = 0

    def push(self, local: dict):
        handle = gather_coo_async(local, self.mode, self.group, self.dst)
        if self._pending is not None:
            self.last = self._pending.wait()
            self.completed += 1
        self._pending = handle

    def drain(self):
        if self._pending is not None:
            self.last = self._pending.wait()
            self.completed += 1
            self._pending = None
        return self.last
