"""Sharding of variant loci over the GPUs of a node + the row gather.

Loci (matrix rows) are independent in the reference — the rayon map over
chunks has no shared mutable state (``src/main.rs:284-291``) and the merge loop
is a concatenation in chunk order (``src/main.rs:320-348``).  So the multi-GPU
path is: contiguous row ranges balanced by record count, one process per GPU,
no collective on the data path, and ONE exchange at the end — every rank's COO
block goes to rank 0 over its direct xGMI link (point-to-point send/recv, the
pattern xGMI is built for; counts first with one all_gather).  Rank 0
concatenates in rank order, which is already (row asc, col asc).

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


def device_coo_tensors(ctx, device) -> dict:
    """Wrap the context's device-resident triplets (vtx_device_coo) as torch tensors."""
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


def _pack(local: dict, rows: int) -> torch.Tensor:
    """Triplet arrays -> one int32 payload [rows, 9] (5 x u32, 2 x f64 as int32 pairs), zero padded."""
    dev = local["row"].device
    n = local["row"].shape[0]
    buf = torch.zeros((rows, 9), dtype=torch.int32, device=dev)
    if n:
        for c, k in enumerate(("row", "col", "alt", "ref", "unk")):
            buf[:n, c] = local[k]
        buf[:n, 5:7] = local["value"].contiguous().view(torch.int32).view(n, 2)
        buf[:n, 7:9] = local["ref_value"].contiguous().view(torch.int32).view(n, 2)
    return buf


def _unpack(buf: torch.Tensor) -> dict:
    out = {k: buf[:, c].contiguous() for c, k in enumerate(("row", "col", "alt", "ref", "unk"))}
    out["value"] = buf[:, 5:7].contiguous().view(torch.float64).view(-1)
    out["ref_value"] = buf[:, 7:9].contiguous().view(torch.float64).view(-1)
    return out


def gather_coo(local: dict, group=None, dst: int = 0):
    """Gather every rank's triplets to ``dst`` (rank order = row order).

    One tiny all_gather of the counts, then ONE gather of a packed, padded int32
    payload per rank (RCCL lowers gather to grouped send/recv: each rank's block
    travels once over its direct xGMI link to ``dst``).  Returns the
    concatenated dict on ``dst`` and ``None`` elsewhere.
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
    if rank == dst:
        bufs = [torch.empty((rows, 9), dtype=torch.int32, device=dev) for _ in range(world)]
        dist.gather(payload, gather_list=bufs, dst=dst, group=group)
        return _unpack(torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0))
    dist.gather(payload, gather_list=None, dst=dst, group=group)
    return None
