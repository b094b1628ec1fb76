"""Multi-GPU path on CPU: loci shard across ranks, one COO gather to rank 0.

world_size-2 `gloo` processes stand in for two GPUs: each rank reduces its own
shard (with the CPU oracle as the stand-in for the device path — this test is
about partitioning + the exchange, not about kernels) and the gathered matrix
must be byte-identical to the unsharded one.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vartrix_amd import shard, synth
from vartrix_amd.abi import default_config


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, spec_kwargs, mode, umi, out_path):
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = synth.SynthSpec(**spec_kwargs)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="full", scoring_mode=mode, use_umi=umi, n_barcodes=spec.n_barcodes)
    parts = shard.partition_loci(batch, world)
    lo, hi = parts[rank]
    mine = batch.slice_loci(lo, hi)
    ref, alt = oracle.batch_scores(mine, cfg)
    coo = oracle.batch_reduce(mine, cfg, ref, alt)
    from vartrix_amd.abi import MODES
    h = shard.gather_coo_async(shard.coo_to_tensors(coo), MODES[mode])     # overlappable form
    got = h.wait()
    if rank == 0:
        full_ref, full_alt = oracle.batch_scores(batch, cfg, threads=4)
        want = oracle.batch_reduce(batch, cfg, full_ref, full_alt)
        got = shard.tensors_to_coo(got)
        ok = all(np.array_equal(got[k].view(np.uint8), want[k].view(np.uint8)) for k in want)
        with open(out_path, "w") as fh:
            fh.write("ok" if ok else "mismatch")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,umi", [("consensus", 0), ("alt_frac", 1), ("coverage", 1)])
def test_two_rank_gather_equals_unsharded(tmp_path, mode, umi):
    out = str(tmp_path / "result.txt")
    spec = dict(n_loci=24, n_barcodes=30, reads_per_locus=24, use_umi=bool(umi), indel_frac=0.3, read_len=60, padding=40)
    mp.spawn(_worker, args=(2, _free_port(), spec, mode, umi, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_partition_balances_records():
    spec = synth.SynthSpec(n_loci=200, n_barcodes=50, reads_per_locus=16)
    batch = synth.make_batch(spec)
    batch.loci["rec_count"][:50]  # uneven: zero out the reads of some loci by slicing later
    for world in (1, 2, 4, 8):
        parts = shard.partition_loci(batch, world)
        assert parts[0][0] == 0 and parts[-1][1] == batch.n_loci
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        counts = [int(batch.loci["rec_count"][lo:hi].sum()) for lo, hi in parts]
        assert sum(counts) == batch.n_records
        assert max(counts) - min(counts) <= 2 * int(batch.loci["rec_count"].max())


def test_slice_loci_is_self_contained():
    from oracle import oracle
    spec = synth.SynthSpec(n_loci=20, n_barcodes=10, reads_per_locus=8, indel_frac=0.5, read_len=50, padding=30)
    batch = synth.make_batch(spec)
    cfg = default_config(aligner="full", n_barcodes=10)
    ref, alt = oracle.batch_scores(batch, cfg)
    sub = batch.slice_loci(5, 13)
    r0 = int(batch.loci["rec_begin"][5])
    sref, salt = oracle.batch_scores(sub, cfg)
    assert np.array_equal(sref, ref[r0:r0 + sub.n_records]) and np.array_equal(salt, alt[r0:r0 + sub.n_records])
    assert list(sub.loci["row"]) == list(range(5, 13))
    empty = batch.slice_loci(7, 7)
    assert empty.n_loci == 0 and empty.n_records == 0


def _pipeline_worker(rank, world, port, out_path):
    """bench.py's per-step exchange (shard.GatherPipeline) on CPU tensors: three steps with different triplets per
    step; after drain rank 0 must hold the LAST step's rows of both ranks, in rank order."""
    from oracle import oracle
    from vartrix_amd.abi import MODES
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe = shard.GatherPipeline(MODES["coverage"])
    wants = []
    for stepno in range(3):
        spec = synth.SynthSpec(n_loci=16 + 4 * stepno, n_barcodes=20, reads_per_locus=12, read_len=50, padding=30, seed=100 + stepno)
        batch = synth.make_batch(spec)
        cfg = default_config(aligner="full", scoring_mode="coverage", n_barcodes=20)
        lo, hi = shard.partition_loci(batch, world)[rank]
        mine = batch.slice_loci(lo, hi)
        ref, alt = oracle.batch_scores(mine, cfg)
        pipe.push(shard.coo_to_tensors(oracle.batch_reduce(mine, cfg, ref, alt)))
        if rank == 0:
            fr, fa = oracle.batch_scores(batch, cfg)
            wants.append(oracle.batch_reduce(batch, cfg, fr, fa))
        assert pipe.completed == stepno          # the gather of step k completes while step k + 1 is pushed
    got = pipe.drain()
    assert pipe.completed == 3
    if rank == 0:
        got = shard.tensors_to_coo(got)
        ok = all(np.array_equal(got[k].view(np.uint8), wants[-1][k].view(np.uint8)) for k in wants[-1])
        with open(out_path, "w") as fh:
            fh.write("ok" if ok else "mismatch")
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_pipeline_two_ranks(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_partition_covers_config4_shape():
    """Strong-scaling cut of bench.py --gpus N (BASELINE configs[3]): 1 / 2 / 4 / 8 ranges that tile the loci, and
    slices whose union is the unsharded batch."""
    spec = synth.SynthSpec(n_loci=400, n_barcodes=2000, reads_per_locus=24)
    batch = synth.make_batch(spec)
    for world in (1, 2, 4, 8):
        parts = shard.partition_loci(batch, world)
        pieces = [batch.slice_loci(lo, hi) for lo, hi in parts]
        assert sum(p.n_records for p in pieces) == batch.n_records
        rows = np.concatenate([p.loci["row"] for p in pieces])
        assert np.array_equal(rows, batch.loci["row"])
        reads = np.concatenate([p.read_arena[:int(p.records["read_len"].astype(np.int64).sum())] if p.n_records else np.zeros(0, np.uint8)
                                for p in pieces])
        assert reads.shape[0] == int(batch.records["read_len"].astype(np.int64).sum())


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_gather_plan_is_the_concatenation_in_rank_order(world):
    """vtx_gather_plan — the count -> offset layout vtx_gather_coo follows on every rank (src/main.rs:320-348: the chunks'
    results concatenated in chunk order) — as a pure function: empty ranks, one rank holding everything, the 2^32 limit."""
    from vartrix_amd import abi, lib
    rng = np.random.default_rng(world)
    cases = [[0] * world, [5] * world, [0] * (world - 1) + [7], [7] + [0] * (world - 1),
             [int(v) for v in rng.integers(0, 1 << 20, world)], [int(v) if i % 2 else 0 for i, v in enumerate(rng.integers(1, 99, world))]]
    for counts in cases:
        rc, off, total = lib.gather_plan(counts)
        assert rc == abi.VTX_OK
        assert total == sum(counts)
        assert off == [sum(counts[:r]) for r in range(world)]
        # blocks tile [0, total) without overlap, in rank order
        ends = [o + c for o, c in zip(off, counts)]
        assert all(ends[r] == (off[r + 1] if r + 1 < world else total) for r in range(world))
    # more than 2^32 - 1 gathered triplets: every rank gets the same refusal (before any point-to-point call)
    big = [(1 << 32) // world + 1] * world
    rc, off, total = lib.gather_plan(big)
    assert rc == abi.VTX_E_UNSUPPORTED and total == sum(big)
    rc, _, total = lib.gather_plan([(1 << 32) - 1] + [0] * (world - 1))
    assert rc == abi.VTX_OK and total == (1 << 32) - 1
